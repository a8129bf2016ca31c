/* The bits of a whole intra CU by a wavefront (include/x265amd.h: x265amd_intra_cu_bits): what Search::checkIntra / encodeIntraInInter count with the entropy coder
 * in bit-counting mode once the CU's modes and levels are decided (reference: source/encoder/search.cpp:1236-1287, :1454-1507 -- codeSkipFlag / codePredMode in P and B
 * slices, codePartSize, codePredInfo (codeIntraDirLumaAng for one or four prediction units, codeIntraDirChroma), codeCoeff -> encodeTransform with the coded block
 * flags and the coefficients of every unit; source/encoder/entropy.cpp:775-1063, :1522-1664), for the cases the fused intra commands produce: one transform unit per
 * CU (8x8 .. 32x32) or the 8x8 CU coded NxN (four 4x4 luma units, one 4x4 block per chroma plane), 4:2:0, no delta QP, no transform skip.
 * It is the piece that lets a CU's decision -- and with it the contexts the next CU starts from -- be made where the transform chains run. */
#ifndef X265AMD_INTRA_CU_DEV_H
#define X265AMD_INTRA_CU_DEV_H
#include "entropy_dev.h"

struct IntraCuBitsIn
{
    uint8_t log2_cu;            /* 3 .. 5 */
    uint8_t nxn;                /* the 8x8 CU coded NxN */
    uint8_t code_part_size;     /* the CU sits at the maximum depth: the part size bin is coded (entropy.cpp:1525-1530) */
    uint8_t inter_slice;        /* P / B slice: skip flag (0) and pred mode (intra) in front */
    uint8_t skip_ctx;           /* getCtxSkipFlag: left skipped + above skipped */
    uint8_t sign_hide;
    uint8_t chroma_dir;         /* as stored: a mode number, 36 = derived from luma */
    uint8_t cbf_u, cbf_v;
    uint8_t subdiv_flag;        /* one unit where the transform tree could have gone deeper (tu-intra-depth > 1): the subdivision flag, 0, is coded (entropy.cpp:955-958) */
    uint8_t luma_dir[4], cbf_y[4];
    uint8_t preds[4][3];        /* getIntraDirLumaPredictor per prediction unit */
    const int16_t* lev_y[4]; const int16_t* lev_u; const int16_t* lev_v;        /* the units' levels (LDS or global), scan-ready as the chains leave them */
};

/* ctx: the CU's start contexts, replaced by the contexts behind the CU (X265AMD_CTX_STRIDE bytes the wavefront may write: LDS).  Returns the coder's fraction
 * behind the CU (started from frac_in & 32767, as resetBits leaves it): bits = result >> 15.  *skip_frac: the fraction behind the skip flag (0 in I slices),
 * *mv_frac: behind the prediction info -- Mode::mvBits = (mv_frac >> 15) - (skip_frac >> 15), coeffBits = total - mvBits - (skip_frac >> 15) (search.cpp:1262-1275).
 * Every lane returns the same values. */
XA_DEV uint64_t wave_intra_cu_bits(const IntraCuBitsIn& in, uint8_t* ctx, uint64_t frac_in, uint64_t* mv_frac, uint64_t* skip_frac, const uint32_t* step, const EnTabs& tabs, int lane)
{
    uint64_t frac = frac_in & 32767, skipf = 0;
    const int numPu = in.nxn ? 4 : 1;
    /* ---- flags in front of the transform tree: one lane walks them (a dozen bins; every context state lives in `ctx`) ---- */
    if (lane == 0)
    {
        if (in.inter_slice)
        {
            frac += cb_bin_t(tabs, ctx + 3 + in.skip_ctx, 0);                   /* C_SKIP + ctx: not skipped */
            skipf = frac;
            frac += cb_bin_t(tabs, ctx + 12, 1);                                /* C_PRED_MODE: intra */
        }
        if (in.code_part_size) frac += cb_bin_t(tabs, ctx + 8, in.nxn ? 0u : 1u);              /* C_PART_SIZE */
        int predIdx[4];
        for (int j = 0; j < numPu; j++)
        {
            const uint32_t d = in.luma_dir[j];
            predIdx[j] = d == in.preds[j][0] ? 0 : (d == in.preds[j][1] ? 1 : (d == in.preds[j][2] ? 2 : -1));
            frac += cb_bin_t(tabs, ctx + 13, predIdx[j] != -1 ? 1u : 0u);       /* C_ADI: prev_intra_luma_pred_flag, all units first */
        }
        for (int j = 0; j < numPu; j++) frac += (uint64_t)(predIdx[j] != -1 ? 1 + (predIdx[j] != 0) : 5) << 15;
        if (in.chroma_dir == 36) frac += cb_bin_t(tabs, ctx + 14, 0);           /* C_CHROMA_PRED */
        else { frac += cb_bin_t(tabs, ctx + 14, 1); frac += 2ull << 15; }
    }
    /* everybody learns the running fraction; the context bytes lane 0 moved are in LDS */
    xa_wave_sync();
    frac = __shfl(frac, 0, 64); skipf = __shfl(skipf, 0, 64);
    const uint64_t mvf = frac;
    /* ---- encodeTransform at depth 0 (entropy.cpp:930-1063): NxN implies the split; one unit codes the subdivision flag only where a split was allowed ---- */
    if (lane == 0)
    {
        if (in.subdiv_flag && !in.nxn) frac += cb_bin_t(tabs, ctx + 35 + 5 - in.log2_cu, 0);      /* C_TRANS_SUBDIV + 5 - log2 */
        frac += cb_bin_t(tabs, ctx + CTX_QT_CBF + 2, in.cbf_u ? 1u : 0u);       /* chroma coded block flags at depth 0: C_QT_CBF + curDepth + 2, U then V */
        frac += cb_bin_t(tabs, ctx + CTX_QT_CBF + 2, in.cbf_v ? 1u : 0u);
        if (!in.nxn) frac += cb_bin_t(tabs, ctx + CTX_QT_CBF + 1, in.cbf_y[0] ? 1u : 0u);     /* luma flag of the one unit: C_QT_CBF + !curDepth */
    }
    xa_wave_sync();
    frac = __shfl(frac, 0, 64);
    const int cLog2 = in.nxn ? 2 : in.log2_cu - 1;
    if (!in.nxn)
    {
        if (in.cbf_y[0]) frac += wave_coeff_bits(ctx, ctx, in.lev_y[0], in.log2_cu, 0, 1, in.luma_dir[0], in.sign_hide, step, lane);
        xa_wave_sync();
        if (in.cbf_y[0] || in.cbf_u || in.cbf_v)
        {
            if (in.cbf_u) frac += wave_coeff_bits(ctx, ctx, in.lev_u, cLog2, 1, 1, in.chroma_dir == 36 ? in.luma_dir[0] : in.chroma_dir, in.sign_hide, step, lane);
            xa_wave_sync();
            if (in.cbf_v) frac += wave_coeff_bits(ctx, ctx, in.lev_v, cLog2, 2, 1, in.chroma_dir == 36 ? in.luma_dir[0] : in.chroma_dir, in.sign_hide, step, lane);
            xa_wave_sync();
        }
    }
    else
    {
        /* four 4x4 luma units at depth 1: their flag (C_QT_CBF + !1), their coefficients; the chroma blocks behind the fourth */
        for (int k = 0; k < 4; k++)
        {
            if (lane == 0) frac += cb_bin_t(tabs, ctx + CTX_QT_CBF + 0, in.cbf_y[k] ? 1u : 0u);
            xa_wave_sync();
            frac = __shfl(frac, 0, 64);
            if (in.cbf_y[k]) frac += wave_coeff_bits(ctx, ctx, in.lev_y[k], 2, 0, 1, in.luma_dir[k], in.sign_hide, step, lane);
            xa_wave_sync();
        }
        const int cdir = in.chroma_dir == 36 ? in.luma_dir[0] : in.chroma_dir;
        if (in.cbf_u) frac += wave_coeff_bits(ctx, ctx, in.lev_u, 2, 1, 1, cdir, in.sign_hide, step, lane);
        xa_wave_sync();
        if (in.cbf_v) frac += wave_coeff_bits(ctx, ctx, in.lev_v, 2, 2, 1, cdir, in.sign_hide, step, lane);
        xa_wave_sync();
    }
    if (mv_frac) *mv_frac = mvf;
    if (skip_frac) *skip_frac = skipf;
    return frac;
}

#endif
