/* Slot map of the reference's `EncoderPrimitives` function table, as index arithmetic over a flat array of
 * function pointers (every member of the reference struct is a function pointer, so its layout is fully described
 * by the member ORDER: reference source/common/primitives.h:239-433; 2281 slots, 18248 bytes on LP64).
 *
 * Slot index = base of the sub-table + element index * slots per element + field ordinal.  The ordinals below
 * follow the member order of the reference's nested structs PU (:247-267), CU (:275-316), the loose members
 * (:320-386) and Chroma::PUChroma / Chroma::CUChroma (:400-429).  tests/test_primitive_table_layout.py compiles a
 * checker against the reference header (offsetof) to pin every ordinal used by x265amd_setup_primitives().
 */
#ifndef X265AMD_PRIMITIVE_TABLE_H
#define X265AMD_PRIMITIVE_TABLE_H

#include <stddef.h>

namespace x265amd {

enum { NUM_PU_SIZES = 25, NUM_CU_SIZES = 5, NUM_CSP = 4, INTRA_MODES = 35 };
enum { NONALIGNED = 0, ALIGNED = 1 };

/* field ordinals inside pu[part] */
enum PUField
{
    PU_sad, PU_sad_x3, PU_sad_x4, PU_ads, PU_satd,
    PU_luma_hpp, PU_luma_hps, PU_luma_vpp, PU_luma_vps, PU_luma_vsp, PU_luma_vss, PU_luma_hvpp,
    PU_pixelavg_pp, /* [2] */ PU_addAvg = PU_pixelavg_pp + 2, /* [2] */ PU_copy_pp = PU_addAvg + 2,
    PU_convert_p2s, /* [2] */ PU_FIELDS = PU_convert_p2s + 2
};

/* field ordinals inside cu[cu] */
enum CUField
{
    CU_dct, CU_idct, CU_standard_dct, CU_lowpass_dct,
    CU_calcresidual, /* [2] */ CU_sub_ps = CU_calcresidual + 2, CU_add_ps, /* [2] */ CU_blockfill_s = CU_add_ps + 2, /* [2] */
    CU_copy_cnt = CU_blockfill_s + 2, CU_count_nonzero, CU_cpy2Dto1D_shl, CU_cpy2Dto1D_shr,
    CU_cpy1Dto2D_shl, /* [2] */ CU_cpy1Dto2D_shr = CU_cpy1Dto2D_shl + 2,
    CU_copy_sp, CU_copy_ps, CU_copy_ss, CU_copy_pp, CU_var, CU_sse_pp, CU_sse_ss, CU_psy_cost_pp,
    CU_ssd_s, /* [2] */ CU_sa8d = CU_ssd_s + 2, CU_transpose, CU_intra_pred_allangs, CU_intra_filter,
    CU_intra_pred, /* [35] */ CU_nonPsyRdoQuant = CU_intra_pred + INTRA_MODES, CU_psyRdoQuant, CU_psyRdoQuant_1p,
    CU_psyRdoQuant_2p, CU_ssimDist, CU_normFact, CU_FIELDS
};

/* loose members after cu[] */
enum MiscField
{
    M_dst4x4, M_idst4x4, M_quant, M_nquant, M_dequant_scaling, M_dequant_normal, M_denoiseDct,
    M_scale1D_128to64, /* [2] */ M_scale2D_64to32 = M_scale1D_128to64 + 2,
    M_ssim_4x4x2_core, M_ssim_end_4, M_sign, M_saoCuOrgE0, M_saoCuOrgE1, M_saoCuOrgE1_2Rows,
    M_saoCuOrgE2, /* [2] */ M_saoCuOrgE3 = M_saoCuOrgE2 + 2, /* [2] */ M_saoCuOrgB0 = M_saoCuOrgE3 + 2,
    M_saoCuStatsBO, M_saoCuStatsE0, M_saoCuStatsE1, M_saoCuStatsE2, M_saoCuStatsE3,
    M_frameInitLowres, M_frameInitLowerRes, M_frameSubSampleLuma, M_propagateCost, M_fix8Unpack, M_fix8Pack,
    M_extendRowBorder, M_planecopy_cp, M_planecopy_sp, M_planecopy_sp_shl, M_planecopy_pp_shr, M_planeClipAndMax,
    M_weight_sp, M_weight_pp, M_scanPosLast, M_findPosFirstLast, M_costCoeffNxN, M_costCoeffRemain, M_costC1C2Flag,
    M_pelFilterLumaStrong, /* [2] */ M_pelFilterChroma = M_pelFilterLumaStrong + 2, /* [2] */
    M_integral_initv = M_pelFilterChroma + 2, /* [6] */ M_integral_inith = M_integral_initv + 6, /* [6] */
    M_FIELDS = M_integral_inith + 6
};

/* chroma[csp].pu[part] */
enum ChromaPUField
{
    CPU_satd, CPU_filter_vpp, CPU_filter_vps, CPU_filter_vsp, CPU_filter_vss, CPU_filter_hpp, CPU_filter_hps,
    CPU_addAvg, /* [2] */ CPU_copy_pp = CPU_addAvg + 2, CPU_p2s, /* [2] */ CPU_FIELDS = CPU_p2s + 2
};

/* chroma[csp].cu[cu] */
enum ChromaCUField
{
    CCU_sa8d, CCU_sse_pp, CCU_sub_ps, CCU_add_ps, /* [2] */ CCU_copy_ps = CCU_add_ps + 2, CCU_copy_sp, CCU_copy_ss, CCU_copy_pp,
    CCU_FIELDS
};

enum
{
    BASE_PU = 0,
    BASE_CU = BASE_PU + NUM_PU_SIZES * PU_FIELDS,
    BASE_MISC = BASE_CU + NUM_CU_SIZES * CU_FIELDS,
    BASE_CHROMA = BASE_MISC + M_FIELDS,
    CHROMA_CSP_SLOTS = NUM_PU_SIZES * CPU_FIELDS + NUM_CU_SIZES * CCU_FIELDS,
    TOTAL_SLOTS = BASE_CHROMA + NUM_CSP * CHROMA_CSP_SLOTS
};

constexpr int slotPU(int part, int field) { return BASE_PU + part * PU_FIELDS + field; }
constexpr int slotCU(int cu, int field) { return BASE_CU + cu * CU_FIELDS + field; }
constexpr int slotMisc(int field) { return BASE_MISC + field; }
constexpr int slotChromaPU(int csp, int part, int field) { return BASE_CHROMA + csp * CHROMA_CSP_SLOTS + part * CPU_FIELDS + field; }
constexpr int slotChromaCU(int csp, int cu, int field) { return BASE_CHROMA + csp * CHROMA_CSP_SLOTS + NUM_PU_SIZES * CPU_FIELDS + cu * CCU_FIELDS + field; }

static_assert(TOTAL_SLOTS == 2281, "EncoderPrimitives slot count (reference: 18248 bytes / 8)");

typedef void (*generic_fn)(void);

} // namespace x265amd

#endif
