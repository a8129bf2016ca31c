/* x265amd -- MI355X (gfx950) HEVC encode hot path behind the x265 primitive-table boundary.
 *
 * C ABI of libx265amd_main.so (8-bit pixels) / libx265amd_main10.so (10-bit pixels).  Like the reference's
 * multilib scheme (reference: source/x265.h:2619-2635, encoder/api.cpp:1107-1182) the pixel type is a build-time
 * property of the library; both libraries export the same symbols.
 *
 * Three layers, all plain C (no C++ / torch / HIP types in any signature):
 *
 *  1. Per-slot entry points with HOST pointers -- one per slot family of the reference's `EncoderPrimitives`
 *     function table (reference: source/common/primitives.h:239-433), taking the slot index (LumaPU `part` or
 *     LumaCU `cu` = log2(size)-2) as first argument and then exactly the arguments of the slot's typedef
 *     (primitives.h:133-236).  `x265amd_setup_primitives()` installs size-bound thunks of these into a table
 *     with the reference's layout, the way setupAssemblyPrimitives() does (primitives.h:474).  Each call stages
 *     its operands to the GPU, runs the SAME kernels as layer 2 with a batch of one and copies the result back:
 *     they exist for slot-for-slot parity (TestBench style, reference source/test/testbench.cpp:102-261), not speed.
 *
 *  2. Batched job lists on DEVICE memory (`x265amd_run_jobs`): the host loop queues any number of independent
 *     primitive calls (all 35 intra predictions of a CU, all merge candidates, every TU of a residual quad-tree
 *     level ...) and flushes them as one launch per kernel family; one 64-lane wavefront executes one job.
 *
 *  3. Fused frame-level kernels on device-resident pictures (`x265amd_me_*`, `x265amd_intra_*`,
 *     `x265amd_tu_*`): the form the frame host loop uses; see each prototype.
 *
 * Error behaviour: the reference primitives cannot fail (void/int results, primitives.h:133-236); the per-slot
 * entry points keep those signatures and abort() with a message on a HIP runtime error (no silent CPU fallback).
 * Layer 2/3 functions return 0 on success or a negative X265AMD_E* code.
 */
#ifndef X265AMD_H
#define X265AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef X265AMD_DEPTH
#define X265AMD_DEPTH 8
#endif
#if X265AMD_DEPTH > 8
typedef uint16_t x265amd_pixel;     /* reference: common/common.h:126-139 */
#else
typedef uint8_t x265amd_pixel;
#endif

#define X265AMD_OK 0
#define X265AMD_EINVAL (-1)
#define X265AMD_EHIP (-2)
#define X265AMD_ENOMEM (-3)

/* ------------------------------------------------------------------------------------------------------- */
/* Library / device                                                                                        */
/* ------------------------------------------------------------------------------------------------------- */
int x265amd_bit_depth(void);                 /* X265_DEPTH of this build */
const char* x265amd_version(void);
int x265amd_device_count(void);              /* number of visible GPUs (0: the per-slot/batched calls will fail loudly) */
const char* x265amd_last_error(void);        /* message for the last negative return on this thread */

/* Fills the accelerated slots of a table laid out like the reference's `EncoderPrimitives`
 * (primitives.h:239-433; 2281 slots / 18248 bytes in the 8-bit build).  `table_bytes` must equal the size this
 * library was generated for (x265amd_primitives_table_bytes()); returns the number of slots written, or <0. */
int x265amd_setup_primitives(void* encoder_primitives_table, size_t table_bytes);
size_t x265amd_primitives_table_bytes(void);

/* ------------------------------------------------------------------------------------------------------- */
/* Layer 1: per-slot entry points, HOST pointers.  `part`: enum LumaPU (primitives.h:41-55);                 */
/* `cu`: enum LumaCU = log2(size)-2 (primitives.h:57-65); `csp`: X265_CSP_I420 = 1.                          */
/* ------------------------------------------------------------------------------------------------------- */
typedef x265amd_pixel pixel_t_;

/* pixelcmp_t (primitives.h:133) -- pu[part].sad, pu[part].satd, cu[cu].sa8d, cu[cu].psy_cost_pp */
int x265amd_sad(int part, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
int x265amd_satd(int part, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
int x265amd_sa8d(int cu, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
int x265amd_psy_cost_pp(int cu, const pixel_t_* source, intptr_t sstride, const pixel_t_* recon, intptr_t rstride);
int x265amd_chroma_satd(int csp, int part, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
int x265amd_chroma_sa8d(int csp, int cu, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
/* pixelcmp_x3_t / pixelcmp_x4_t (primitives.h:139-140) -- fenc stride is FENC_STRIDE (64) */
void x265amd_sad_x3(int part, const pixel_t_* fenc, const pixel_t_* fref0, const pixel_t_* fref1, const pixel_t_* fref2, intptr_t frefstride, int32_t* res);
void x265amd_sad_x4(int part, const pixel_t_* fenc, const pixel_t_* fref0, const pixel_t_* fref1, const pixel_t_* fref2, const pixel_t_* fref3, intptr_t frefstride, int32_t* res);
/* pixel_sse_t / pixel_sse_ss_t / pixel_ssd_s_t (primitives.h:135-137); sse_t widened to 64 bits in the C ABI */
uint64_t x265amd_sse_pp(int cu, const pixel_t_* fenc, intptr_t fencstride, const pixel_t_* fref, intptr_t frefstride);
uint64_t x265amd_sse_ss(int cu, const int16_t* fenc, intptr_t fencstride, const int16_t* fref, intptr_t frefstride);
uint64_t x265amd_ssd_s(int cu, const int16_t* a, intptr_t stride);
uint64_t x265amd_var(int cu, const pixel_t_* pix, intptr_t stride);                       /* var_t, primitives.h:173 */

/* pixel_sub_ps_t / pixel_add_ps_t / pixelavg_pp_t / addAvg_t (primitives.h:189-192) */
void x265amd_sub_ps(int cu, int16_t* dst, intptr_t dstride, const pixel_t_* src0, const pixel_t_* src1, intptr_t sstride0, intptr_t sstride1);
void x265amd_add_ps(int cu, pixel_t_* dst, intptr_t dstride, const pixel_t_* src0, const int16_t* src1, intptr_t sstride0, intptr_t sstride1);
void x265amd_pixelavg_pp(int part, pixel_t_* dst, intptr_t dstride, const pixel_t_* src0, intptr_t sstride0, const pixel_t_* src1, intptr_t sstride1);
void x265amd_addAvg(int part, const int16_t* src0, const int16_t* src1, pixel_t_* dst, intptr_t src0Stride, intptr_t src1Stride, intptr_t dstStride);
void x265amd_chroma_addAvg(int csp, int part, const int16_t* src0, const int16_t* src1, pixel_t_* dst, intptr_t src0Stride, intptr_t src1Stride, intptr_t dstStride);
/* weightp_pp_t / weightp_sp_t (primitives.h:164-165) */
void x265amd_weight_pp(const pixel_t_* src, pixel_t_* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset);
void x265amd_weight_sp(const int16_t* src, pixel_t_* dst, intptr_t srcStride, intptr_t dstStride, int width, int height, int w0, int round, int shift, int offset);
/* scale2D_t / scale1D_t / transpose_t (primitives.h:158,166-167) */
void x265amd_scale2D_64to32(pixel_t_* dst, const pixel_t_* src, intptr_t stride);
void x265amd_scale1D_128to64(pixel_t_* dst, const pixel_t_* src);
void x265amd_transpose(int cu, pixel_t_* dst, const pixel_t_* src, intptr_t stride);
/* cpy2Dto1D_* / cpy1Dto2D_* / copy_cnt_t / count_nonzero_t (primitives.h:147-151,163) */
void x265amd_cpy2Dto1D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t srcStride, int shift);
void x265amd_cpy2Dto1D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t srcStride, int shift);
void x265amd_cpy1Dto2D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t dstStride, int shift);
void x265amd_cpy1Dto2D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t dstStride, int shift);
uint32_t x265amd_copy_cnt(int cu, int16_t* coeff, const int16_t* residual, intptr_t resiStride);
int x265amd_count_nonzero(int cu, const int16_t* quantCoeff);

/* dct_t / idct_t (primitives.h:153-154) -- cu[cu].dct/idct, dst4x4, idst4x4 */
void x265amd_dct(int cu, const int16_t* src, int16_t* dst, intptr_t srcStride);
void x265amd_idct(int cu, const int16_t* src, int16_t* dst, intptr_t dstStride);
void x265amd_dst4x4(const int16_t* src, int16_t* dst, intptr_t srcStride);
void x265amd_idst4x4(const int16_t* src, int16_t* dst, intptr_t dstStride);
/* quant_t / nquant_t / dequant_normal_t / dequant_scaling_t (primitives.h:159-162) */
uint32_t x265amd_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff);
uint32_t x265amd_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff);
void x265amd_dequant_normal(const int16_t* quantCoef, int16_t* coef, int num, int scale, int shift);
void x265amd_dequant_scaling(const int16_t* src, const int32_t* dequantCoef, int16_t* dst, int num, int mcqp_miper, int shift);

/* intra_pred_t / intra_filter_t / intra_allangs_t (primitives.h:143-145) */
void x265amd_intra_pred(int cu, int mode, pixel_t_* dst, intptr_t dstStride, const pixel_t_* srcPix, int bFilter);
void x265amd_intra_filter(int cu, const pixel_t_* references, pixel_t_* filtered);
void x265amd_intra_allangs(int cu, pixel_t_* dst, pixel_t_* refPix, pixel_t_* filtPix, int bLuma);

/* filter_pp_t / filter_hps_t / filter_ps_t / filter_sp_t / filter_ss_t / filter_hv_pp_t / filter_p2s_t
 * (primitives.h:176-182) */
void x265amd_luma_hpp(int part, const pixel_t_* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_luma_hps(int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx, int isRowExt);
void x265amd_luma_vpp(int part, const pixel_t_* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_luma_vps(int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx);
void x265amd_luma_vsp(int part, const int16_t* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_luma_vss(int part, const int16_t* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx);
void x265amd_luma_hvpp(int part, const pixel_t_* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int idxX, int idxY);
void x265amd_luma_p2s(int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride);
void x265amd_chroma_hpp(int csp, int part, const pixel_t_* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_chroma_hps(int csp, int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx, int isRowExt);
void x265amd_chroma_vpp(int csp, int part, const pixel_t_* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_chroma_vps(int csp, int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx);
void x265amd_chroma_vsp(int csp, int part, const int16_t* src, intptr_t srcStride, pixel_t_* dst, intptr_t dstStride, int coeffIdx);
void x265amd_chroma_vss(int csp, int part, const int16_t* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride, int coeffIdx);
void x265amd_chroma_p2s(int csp, int part, const pixel_t_* src, intptr_t srcStride, int16_t* dst, intptr_t dstStride);

/* ------------------------------------------------------------------------------------------------------- */
/* Layer 2: batched job lists (DEVICE memory).                                                             */
/* ------------------------------------------------------------------------------------------------------- */
enum x265amd_op
{
    /* family 0: distortion.  a=fenc b=fref (device addresses), sa/sb strides in elements, d -> uint64 result
     * (sad_x3/x4: d -> int32[4]; the 3/4 candidate addresses are b, c, e[0], e[1]) */
    X265AMD_OP_SAD = 0, X265AMD_OP_SAD_X3, X265AMD_OP_SAD_X4, X265AMD_OP_SATD, X265AMD_OP_SA8D, X265AMD_OP_SSE_PP,
    X265AMD_OP_SSE_SS, X265AMD_OP_SSD_S, X265AMD_OP_PSY_COST_PP, X265AMD_OP_VAR, X265AMD_OP_CHROMA_SATD, X265AMD_OP_CHROMA_SA8D,
    /* family 1: pixel/residual block ops */
    X265AMD_OP_SUB_PS = 32, X265AMD_OP_ADD_PS, X265AMD_OP_PIXELAVG_PP, X265AMD_OP_ADDAVG, X265AMD_OP_WEIGHT_PP, X265AMD_OP_WEIGHT_SP,
    X265AMD_OP_SCALE2D_64TO32, X265AMD_OP_SCALE1D_128TO64, X265AMD_OP_TRANSPOSE, X265AMD_OP_CPY2DTO1D_SHL, X265AMD_OP_CPY2DTO1D_SHR,
    X265AMD_OP_CPY1DTO2D_SHL, X265AMD_OP_CPY1DTO2D_SHR, X265AMD_OP_COPY_CNT, X265AMD_OP_COUNT_NONZERO,
    /* family 2: transforms + quantisation */
    X265AMD_OP_DCT = 64, X265AMD_OP_IDCT, X265AMD_OP_DST4, X265AMD_OP_IDST4, X265AMD_OP_QUANT, X265AMD_OP_NQUANT,
    X265AMD_OP_DEQUANT_NORMAL, X265AMD_OP_DEQUANT_SCALING,
    /* family 3: intra prediction */
    X265AMD_OP_INTRA_PRED = 96, X265AMD_OP_INTRA_FILTER, X265AMD_OP_INTRA_ALLANGS,
    /* family 4: interpolation.  p[0]=taps (8 luma / 4 chroma) p[1]=width p[2]=height p[3]=coeffIdx(X) p[4]=isRowExt / coeffIdxY */
    X265AMD_OP_IP_HPP = 128, X265AMD_OP_IP_HPS, X265AMD_OP_IP_VPP, X265AMD_OP_IP_VPS, X265AMD_OP_IP_VSP, X265AMD_OP_IP_VSS,
    X265AMD_OP_IP_HVPP, X265AMD_OP_IP_P2S
};

typedef struct x265amd_job
{
    int32_t op;             /* enum x265amd_op */
    int32_t size;           /* part (LumaPU) or cu (LumaCU) index, op dependent */
    int32_t p[6];           /* op-specific integer arguments, in the order of the slot's typedef */
    uint64_t a, b, c, d;    /* device addresses of the operands (a,b,c inputs; d output) */
    uint64_t e[2];          /* extra operands (sad_x3/x4 candidates, quant deltaU ...) */
    int32_t sa, sb, sc, sd; /* strides of a,b,c,d in elements */
} x265amd_job;

/* Runs jobs[0..n) (device array) on `stream` (a hipStream_t, may be NULL).  All jobs of one call must belong to
 * the same family (op / 32).  Asynchronous: results are valid after the stream is synchronised. */
int x265amd_run_jobs(void* stream, const x265amd_job* d_jobs, int n, int family);

/* ------------------------------------------------------------------------------------------------------- */
/* Layer 3: fused frame-level kernels on device-resident, padded pictures.                                  */
/* ------------------------------------------------------------------------------------------------------- */

/* --- motion estimation: MotionEstimate::motionEstimate() (reference: source/encoder/motion.cpp:764-1594) ---
 * One job = one call of the reference's motionEstimate() after setSourcePU() (motion.cpp:193-247): a PU of the
 * source picture searched in one reference picture.  Jobs are grouped; one workgroup serves one group: it stages
 * the group's reference window and its 64x64 source tile in LDS once and its wavefronts pull the group's jobs from
 * an LDS counter, one wavefront per job.  A search that leaves the staged window keeps producing exact results
 * through a direct-from-HBM path.  Planes are padded exactly like the reference's PicYuv (common/picyuv.cpp:
 * marginX = maxCU + 32, marginY = maxCU + 16) so that, as in the reference, candidates may lie outside the picture. */
enum { X265AMD_ME_DIA = 0, X265AMD_ME_HEX = 1, X265AMD_ME_UMH = 2, X265AMD_ME_STAR = 3, X265AMD_ME_SEA = 4, X265AMD_ME_FULL = 5 };
#define X265AMD_ME_MAX_CAND 12      /* reference: MD_ABOVE_LEFT+1 spatial x 2 + 2 (search.cpp:2086-2154) */

typedef struct x265amd_me_job
{
    int16_t x, y;                       /* PU position in the picture, luma samples */
    uint8_t w, h;                       /* PU size (a LumaPU shape, primitives.h:41-55) */
    uint8_t method, subme;              /* searchMethod (X265AMD_ME_*, | X265AMD_ME_CHROMA_SATD), subpelRefine 0..7 (motion.cpp:48-58) */
    uint8_t qp, num_cand;               /* QP selecting the MV cost table (bitcost.cpp:30-58); number of mvc[] */
    int16_t merange;
    int16_t mvmin[2], mvmax[2];         /* search bounds, full-pel (search.cpp:2724-2768) */
    int16_t mvp[2];                     /* predictor, quarter-pel */
    int16_t mvc[X265AMD_ME_MAX_CAND][2];/* extra candidates, quarter-pel */
} x265amd_me_job;

typedef struct x265amd_me_group
{
    int32_t first_job, num_jobs;        /* jobs[first_job .. first_job+num_jobs) */
    int32_t ref;                        /* index into the reference plane array */
    int16_t win_x, win_y;               /* picture coordinates of the staged window's top-left sample */
    int16_t win_w, win_h;               /* staged window size in samples, win_w % 4 == 0, <= the launch maxima */
    int16_t fenc_x, fenc_y;             /* top-left of the 64x64 source tile holding every PU of the group */
} x265amd_me_group;

typedef struct x265amd_me_result { int16_t mv[2]; int32_t cost; } x265amd_me_result;

typedef struct x265amd_me_ctx x265amd_me_ctx;
/* Builds the MV cost tables (BitCost::s_costs[qp], bitcost.cpp:30-109) for qp 0..69 on the current device. */
x265amd_me_ctx* x265amd_me_open(void);
void x265amd_me_close(x265amd_me_ctx* ctx);
/* host copy of one table: 2*65536+1 uint16 values, element [65536] is MVD 0 (for parity tests of the table itself) */
const uint16_t* x265amd_me_host_mvcost(x265amd_me_ctx* ctx, int qp);
/* Host helper: groups `n` jobs (host array, all against reference `ref`) by the 64x64 source tile they lie in and
 * sizes each group's window as the bounding box of its jobs' search areas, clipped to max_win_w x max_win_h around
 * the box centre.  `order[n]` receives the job permutation (jobs must be uploaded in that order); returns the number
 * of groups written to groups[] (capacity n), or <0. */
int x265amd_me_plan(const x265amd_me_job* jobs, int n, int ref, int max_win_w, int max_win_h, x265amd_me_group* groups, int32_t* order);
/* Runs all groups.  d_fenc / d_refs[i]: device addresses of sample (0,0) of the padded source / reference luma
 * planes, all with the same `stride` (elements).  d_refs, d_groups, d_jobs, d_out: device arrays.  max_win_w/h:
 * the largest window among the groups (LDS is sized for it).  Asynchronous on `stream`.  Two launches: the
 * window-resident kernel, then a pass that redoes, reading the reference from HBM, the jobs whose search left the
 * staged window -- results never depend on the window size. */
int x265amd_me_search(x265amd_me_ctx* ctx, void* stream, const x265amd_pixel* d_fenc, const uint64_t* d_refs, intptr_t stride,
                      const x265amd_me_group* d_groups, int num_groups, const x265amd_me_job* d_jobs, x265amd_me_result* d_out,
                      int max_win_w, int max_win_h, int flags, const uint64_t* d_chroma, intptr_t cstride);
/* flags: X265AMD_ME_FLAG_STAR must be set when any job uses X265AMD_ME_STAR (selects the kernel variant that carries
 * the star search; without it such jobs still produce exact results through the slower second pass). */
#define X265AMD_ME_FLAG_STAR 1
#define X265AMD_ME_FLAG_CHROMA 2          /* some job has X265AMD_ME_CHROMA_SATD set: selects the kernel variant that carries it */
/* Chroma SATD (MotionEstimate::bChromaSATD, motion.cpp:234-237): a job whose `method` has X265AMD_ME_CHROMA_SATD set was
 * configured with bChroma = true in setSourcePU; with subme > 2 and a 4:2:0 chroma PU that is a multiple of 4x4 every
 * sub-pel comparison then adds the SATD of both chroma blocks (motion.cpp:1625-1686).  d_chroma (device array, may be NULL
 * when no job asks for it): [0],[1] = source U,V sample (0,0); [2+2r],[3+2r] = reference r U,V; all with stride `cstride`. */
#define X265AMD_ME_CHROMA_SATD 0x80

/* --- residual (transform unit) path: Quant::transformNxN / invtransformNxN (reference: source/common/quant.cpp:397-605,
 * sign-bit hiding :247-395) and the per-TU measurement of the residual quad-tree (source/encoder/search.cpp:3276-3330).
 * Flat scaling lists, no transform skip / bypass / noise reduction; plain quantisation here, RDOQ through x265amd_tu_chain_rdoq below.
 * qp_scaled = QP + 6*(depth-8) after the chroma mapping of Quant::setQPforQuant (quant.cpp:221-244);
 * slice_type: 0 B, 1 P, 2 I; ttype: 0 luma, 1 Cb, 2 Cr; dir_mode: intra direction that selects the coefficient scan
 * (CUData::getTUEntropyCodingParameters, cudata.cpp:2059-2100). */
typedef struct x265amd_tu_job
{
    uint64_t fenc, pred;            /* device addresses of the block's top-left sample (pixels) */
    uint64_t coeff;                 /* out: N*N levels (int16) */
    uint64_t resi;                  /* out: N x N reconstructed residual (int16), resi_stride */
    uint64_t recon;                 /* out: N x N reconstruction (pixels), recon_stride */
    int32_t fenc_stride, pred_stride, resi_stride, recon_stride;
    uint8_t log2_tr_size, ttype, intra, dir_mode, slice_type, qp_scaled, sign_hide, reserved;
} x265amd_tu_job;

typedef struct x265amd_tu_result
{
    uint32_t num_sig;               /* return value of transformNxN (after sign-bit hiding) */
    uint32_t zero_energy, nz_energy;/* psyCost(fenc, pred) / psyCost(fenc, recon)  (rdcost.h:114-117) */
    uint32_t reserved;
    uint64_t zero_dist, nz_dist;    /* sse_pp(fenc, pred) / sse_pp(fenc, recon); nz_* repeat zero_* when num_sig == 0 */
} x265amd_tu_result;

/* one wavefront per TU: residual, forward transform, quantisation, sign-bit hiding, and when levels remain
 * dequantisation, inverse transform (with the reference's DC shortcut) and reconstruction, plus the distortion and psy
 * energies both ways -- everything estimateResidualQT needs for a TU except the entropy bits.  Asynchronous. */
int x265amd_tu_chain(void* stream, const x265amd_tu_job* d_jobs, int n, x265amd_tu_result* d_out);

/* --- RDOQ (Quant::rdoQuant, source/common/quant.cpp:609-1424; selected by param.rdoqLevel, the slow presets) and the
 * entropy-side tables it reads.  Context states are the reference's: one byte per context, (pStateIdx << 1) | valMps,
 * X265AMD_CTX_COUNT of them in the order of source/common/contexts.h:75-106, stored X265AMD_CTX_STRIDE apart. */
#define X265AMD_CTX_COUNT 157
#define X265AMD_CTX_STRIDE 160
/* EstBitsSbac (source/encoder/entropy.h:88-97), same member order: significantCoeffGroupBits[2][2], significantBits[2][42],
 * lastBits[2][10], greaterOneBits[24][2], levelAbsBits[6][2], blockCbpBits[7][2], blockRootCbpBits[2]; FIX15 bits */
typedef struct x265amd_est_bits { int32_t v[184]; } x265amd_est_bits;
/* second record of a TU job when RDOQ is on (parallel array to x265amd_tu_job) */
typedef struct x265amd_tu_rdoq
{
    uint64_t est_bits;              /* device address of the x265amd_est_bits of the job's (size, plane) and context set */
    int64_t lambda2;                /* QpParam::lambda2 / lambda, FIX8 (quant.h:50-60): x265amd_rdoq_lambda() */
    int32_t lambda;
    int32_t psy_rdoq_scale;         /* Quant::m_psyRdoqScale = (int)(param.psyRdoq * 256) */
    uint8_t rdoq_level;             /* 0 (plain quantisation for this job), 1, 2 */
    uint8_t tu_depth;               /* cu.m_tuDepth[absPartIdx]: selects the CBF context */
    uint8_t reserved[6];
} x265amd_tu_rdoq;

/* Intra TU step of Search::codeIntraLumaQT (source/encoder/search.cpp:305-508) / codeIntraChromaQt (:819-945), fused: the
 * neighbour set of the job's one mode from the reconstructed plane (Predict::initAdiPattern(dirMode) / initAdiPatternChroma,
 * source/common/predict.cpp:600-649), the prediction (predIntraLumaAng / predIntraChromaAng 4:2:0, predict.cpp:579-598) and
 * the per-TU measurement of x265amd_tu_chain.  tu.dir_mode is the prediction mode (a chroma DM mode resolved by the caller) and
 * selects the coefficient scan; tu.intra is taken as 1; tu.pred (may be 0) receives the prediction; tu.recon may point into
 * the reconstructed plane itself so that later launches see the block as a neighbour.  avail / strong_smoothing as in
 * x265amd_intra_job.  d_rdoq: NULL, or the RDOQ records (Quant::m_rdoqLevel != 0). */
typedef struct x265amd_intra_tu_job
{
    x265amd_tu_job tu;
    uint64_t nb;                    /* device address of the block's top-left sample in the reconstructed plane (neighbour source) */
    uint64_t avail;
    int32_t nb_stride;
    uint8_t strong_smoothing;
    uint8_t reserved[11];
} x265amd_intra_tu_job;
int x265amd_intra_tu_chain(void* stream, const x265amd_intra_tu_job* d_jobs, const x265amd_tu_rdoq* d_rdoq, int n, x265amd_tu_result* d_out);

/* One prediction unit of Search::estIntraPredQT up to the point where bits have to be counted (search.cpp:1509-1650), as ONE launch: the neighbour set and the
 * 35-mode SA8D scan of the block, the mode bits (mode == an MPM: mpm_base + 1 for the first, + 2 for the others; otherwise rbits) and costs
 * (sa8d + ((bits * lambda + 128) >> 8)), the candidate list (the modes within 25 % of the best cost and the first MPM, at most max_cand of them, updateCandList
 * :3953-3972), and the transform chain of every candidate.  tmpl is the chain job of candidate 0 (its dir_mode is ignored); candidate i uses the same job with
 * dir_mode = modes[i], pred / recon moved by i * slot_pixels samples and coeff / resi by i * slot_coeffs values; d_res[i] is its result.  The serial link
 * between scan and candidates is the algorithm's, but the host round trip between them is not: this halves the waits of a prediction unit. */
typedef struct x265amd_intra_pu_job
{
    x265amd_intra_tu_job tmpl;
    uint64_t lambda;                /* RDCost::m_lambda of calcRdSADCost */
    uint32_t rbits, mpm_base;
    uint32_t slot_pixels, slot_coeffs;
    uint8_t preds[3], max_cand;     /* the three most probable modes (loadIntraDirModeLuma); max_cand <= 16 */
    uint8_t reserved[4];
} x265amd_intra_pu_job;             /* 128 bytes */
typedef struct x265amd_intra_pu_out { int32_t sa8d[35]; uint32_t num_cand; uint8_t modes[16]; } x265amd_intra_pu_out;      /* 160 bytes */
int x265amd_intra_pu(void* stream, const x265amd_intra_pu_job* d_job, x265amd_intra_pu_out* d_out, x265amd_tu_result* d_res);

/* The luma part of Search::estIntraPredQT for an 8x8 CU coded NxN (four 4x4 prediction units, search.cpp:1509-1696), decisions included, as ONE launch.  Per unit,
 * in z-order: the most probable modes from the neighbours' modes (getIntraDirLumaPredictor, cudata.cpp:910-953: the units to the left and above inside the CU are the
 * winners of this call, outside it left_mode / above_mode), the step of x265amd_intra_pu (scan, candidate list, candidate chains), then for every candidate the bits the
 * reference counts for it -- prev_intra_luma_pred_flag and the mode index (entropy.cpp:1592-1642), the coded block flag, the coefficients (lane_coeff_bits: a 4x4 unit is
 * short enough for one lane) on top of frac_start[unit] with the contexts `ctx` -- and its cost (rdcost.h:99-117: psy_scale = lambda * psyRd, 0 without psy-rd); the first
 * cheapest candidate wins (search.cpp:1680-1690), its reconstruction goes to the picture (tmpl[unit].nb) and to layer_dst, its prediction to pred_dst (tiles of stride
 * 64), and the next unit predicts from it.  A 4x4 unit has no transform split to try, so the host only repeats the winner's bookkeeping.  Four host round trips less
 * per CU, and the candidates' bits leave the host.  With all four units in place the CU's two luma measurements follow in the same launch (cu_luma: the CU's first
 * sample in the reconstructed plane is tmpl[0].nb, in the prediction tile pred_dst[0]). */
typedef struct x265amd_intra_nxn_job
{
    x265amd_intra_tu_job tmpl[4];   /* the chain job of candidate 0 of each unit; coeff / resi in DEVICE memory */
    uint64_t pred_dst[4], layer_dst[4];
    uint64_t lambda, lambda2, psy_scale;
    uint64_t frac_start[4];         /* Entropy::m_fracBits before the unit's direction bins (unit 0: after the CU's skip flag / prediction mode / partition size) */
    uint32_t scan_frac;             /* m_fracBits & 32767 of m_rqt[depth].cur: the mode bits of the scan (search.cpp:1566-1577) */
    uint32_t slot_pixels, slot_coeffs;
    uint8_t left_mode[2], above_mode[2];    /* left of unit 0 / unit 2, above of unit 0 / unit 1, as the predictor derivation sees them (DC = 1 when absent or not intra) */
    uint8_t ctx[X265AMD_CTX_STRIDE];
    uint8_t max_cand, do_chroma;
    uint8_t num_units, unit_log2;   /* 4 units of 4x4 (NxN; 0 / 0 mean that), or 1 unit of 8x8: the same CU coded 2Nx2N with its one transform unit -- then the coded block flag
                                     * is the one of transform depth 0, the 64 levels fill `levels` as one array and the luma measurements are the unit's own result */
    uint8_t no_picture;             /* 1: the winners' samples go to the tiles only, not to the reconstructed planes (tmpl[].nb / ctmpl[].nb are then only read): the caller
                                     * runs the CU's other partitioning beside this one and THAT one is the last tried, whose samples the picture keeps.  One unit only */
    uint8_t pick_sa8d;              /* 1 (one unit): no candidate list -- the unit's mode is Search::checkIntraInInter's choice (search.cpp:1291-1452): the cheapest of the 35 by
                                     * SA8D + mode bits, DC first, then planar, then the angular modes, strict improvement; its chain, its bits and the chroma decision
                                     * follow as for a list of one (encodeIntraInInter); the result record's chroma_reserved carries the mode's SA8D */
    uint8_t reserved[2];            /* a chained CU under pps.bUseDQP (role 1): [0] != 0: cu_qp_delta is counted with the coefficients of an evaluation that has any (Search::checkIntra's
                                     * codeCoeff with bCodeDQP, search.cpp:1266-1268); [1]: its value as Entropy::codeDeltaQP codes it (int8: the group's QP minus its prediction, wrapped) */
    /* do_chroma: Search::estIntraPredChromaQT for the CU's one 4x4 block per chroma plane in the same launch (search.cpp:1754-1889): the five allowed modes (planar,
     * vertical, horizontal, DC with the one equal to the first unit's luma mode replaced by 34, then the luma mode itself), each a wavefront running the U and the V
     * chain and counting the mode's bits -- intra_chroma_pred_mode, the two coded block flags, U's and V's coefficients, on the contexts `ctx` from scan_frac --; the first
     * cheapest wins.  ctmpl: the chain jobs of mode 0 of U and V (recon = slot 0 of the plane; mode m of plane p uses slot 2 m + p; no prediction output); the winner's
     * reconstruction goes to crecon_dst (stride 32), and the picture (ctmpl[p].nb) keeps the LAST tried mode's, as after the reference's loop. */
    x265amd_intra_tu_job ctmpl[2];
    uint64_t crecon_dst[2];
    uint64_t recon_dst[4];          /* optional (0: none): a third place for the winners' luma reconstruction, stride 64 (the mode's reconstruction tile) */
    uint64_t levels_dst, clevels_dst;   /* one unit larger than 8x8 (unit_log2 4: a 16x16 CU coded 2Nx2N, chroma blocks 8x8): where the winner's luma levels (N x N) and
                                         * the chroma winner's levels (U then V, N/2 x N/2 each) go instead of the result record's arrays; 0: the record */
    /* ---- the CU as a link of a chain the device runs without the host: the 8x8 CUs of a 16x16 block of an I picture (Analysis::compressIntraCU's four sub-CUs,
     * analysis.cpp:514-668).  Every CU is two commands queued in advance on two job queues: role 1, this record with four units (the NxN evaluation), and role 2,
     * one 8x8 unit (2Nx2N).  Role 1 waits for role 2's record, counts both evaluations' bits (x265amd_intra_cu_bits' walk), compares the costs as checkBestMode
     * does, puts the winner's samples into the picture, writes the CU's result for the host and hands contexts, fraction and luma modes to the next CU through the
     * chain record.  Neither waits for the host between the CUs. ---- */
    uint64_t chain;             /* 0: not chained; else the device address of the x265amd_intra_chain record the two workgroups share */
    uint64_t peer;              /* the x265amd_intra_peer record of this CU: role 2 fills it (its `out` argument is ignored), role 1 waits for it */
    uint64_t cu_out;            /* role 1: the host-visible x265amd_intra_cu8_result */
    uint64_t peer_recon[3];     /* role 1: where role 2 leaves its reconstruction (luma, stride 64; U, V, stride 32) */
    uint64_t win_dst[3];        /* role 1: where the winner's reconstruction goes beside the picture (same strides) */
    uint64_t chain_token;       /* the chain's count before this CU: the command starts when chain.seq >= chain_token; role 1 leaves chain_token + 1 */
    uint8_t chain_role;         /* 0 none, 1, 2 */
    uint8_t chain_first;        /* 1: the first CU of the block -- contexts, fraction and neighbour modes are this record's; 0: the chain's */
    uint8_t mode_src[4];        /* for left_mode[0], left_mode[1], above_mode[0], above_mode[1]: 0xFF = this record's value, else (CU << 2 | unit) of chain.mode */
    uint8_t chain_index;        /* this CU's row of chain.mode (0..3) */
    uint8_t reserved2;
    /* ---- RDOQ (Quant::m_rdoqLevel != 0, --rdoq-level 1 / 2): every transform chain of the command quantises with Quant::rdoQuant (quant.cpp:609-1424).  The bit
     * estimates Entropy::estBit (entropy.cpp:2220-2390) makes before each transformNxN are made by the command itself, in LDS, from `ctx` -- the candidates of every
     * unit and the chroma modes all start from m_rqt[depth].cur (search.cpp:356, :853, :1566-1690) -- for (unit size, luma) and (chroma block size, chroma) ---- */
    int64_t rdoq_lambda2[3];    /* per plane: QpParam::lambda2 / lambda of the plane's scaled QP (x265amd_rdoq_lambda) */
    int32_t rdoq_lambda[3];
    int32_t psy_rdoq_scale;     /* Quant::m_psyRdoqScale */
    uint8_t rdoq_level;         /* 0: plain quantisation (everything above as before) */
    uint8_t rdoq_tu_depth;      /* cu.m_tuDepth of the units (selects the CBF context in rdoQuant): 1 for the four units of an NxN CU, else 0 */
    uint8_t rdoq_general;       /* 1: an 8x8 CU coded NxN with RDOQ runs in the general form (a wavefront per candidate) instead of the sixteen-lane form: for comparisons */
    uint8_t reserved3[5];
} x265amd_intra_nxn_job;
typedef struct x265amd_intra_chain { uint64_t seq, frac; uint8_t ctx[X265AMD_CTX_STRIDE]; uint8_t mode[4][4]; } x265amd_intra_chain;
typedef struct x265amd_intra_nxn_out
{
    uint8_t mode[4], num_cand[4]; x265amd_tu_result res[4]; int16_t levels[4][16];
    uint32_t psy_energy;            /* psyCost of the CU's 8x8 luma reconstruction against the source (rdcost.h:114-117; what checkIntra measures at the end, search.cpp:1279-1283) */
    uint32_t res_energy;            /* sse of the CU's 8x8 luma prediction against the source */
    uint32_t chroma_best, chroma_reserved;      /* do_chroma: index of the winning mode in the list of five; its two results and levels */
    x265amd_tu_result cres[2];
    int16_t clevels[2][16];
} x265amd_intra_nxn_out;            /* 408 bytes */
typedef struct x265amd_intra_peer
{
    x265amd_intra_nxn_out out; uint64_t ready;      /* ready = chain_token + 1 when the record is complete */
    /* role 2 counts its own CU's bits while role 1 is still evaluating (the walk of x265amd_intra_cu_bits): the contexts and the coder's fraction behind the CU coded
     * 2Nx2N, the fraction behind its prediction info */
    uint8_t fctx[X265AMD_CTX_STRIDE]; uint64_t ffrac, fmv;
} x265amd_intra_peer;
typedef struct x265amd_intra_cu8_result
{
    uint64_t rd_cost, frac_bits;        /* of the winner: Mode::rdCost, the coder's fraction behind the CU */
    uint64_t other_cost;                /* the losing evaluation's cost */
    uint32_t total_bits, mv_bits, coeff_bits, psy_energy, res_energy, luma_dist, chroma_dist;
    uint32_t status;                    /* 1: done; 2: gave up waiting for the chain or the peer (the host must fail the picture) */
    uint8_t part_size;                  /* 0 2Nx2N, 3 NxN */
    uint8_t chroma_dir;                 /* as CUData stores it: 36 = derived from luma */
    uint8_t cbf_u, cbf_v, luma_dir[4], cbf_y[4], reserved[4];
    uint8_t ctx[X265AMD_CTX_STRIDE];    /* the contexts behind the CU */
    int16_t levels[96];                 /* luma (NxN: unit k at 16 k), then U, V */
} x265amd_intra_cu8_result;             /* 424 bytes */
int x265amd_intra_nxn(void* stream, const x265amd_intra_nxn_job* d_job, x265amd_intra_nxn_out* d_out);
/* n records stride_bytes apart run one after the other by the same workgroup as ONE launch / command (chained CUs: the records of one role; d_out is only written by
 * records that are not chained) */
int x265amd_intra_nxn_list(void* stream, const x265amd_intra_nxn_job* d_jobs, int n, size_t stride_bytes, x265amd_intra_nxn_out* d_out);

/* x265amd_tu_chain with Quant::m_rdoqLevel != 0: d_rdoq[i] belongs to d_jobs[i] */
int x265amd_tu_chain_rdoq(void* stream, const x265amd_tu_job* d_jobs, const x265amd_tu_rdoq* d_rdoq, int n, x265amd_tu_result* d_out);
void x265amd_rdoq_lambda(int qpScaled, int64_t* lambda2, int32_t* lambda);

/* Entropy::resetEntropy (source/encoder/entropy.cpp:1321-1355): context states at slice start (host arithmetic); ctx: X265AMD_CTX_STRIDE bytes */
void x265amd_entropy_reset(int sliceType, int qp, uint8_t* ctx);
/* Entropy::estBit (entropy.cpp:2220-2390), batched: job i reads the context set at `ctx` and fills the entries of `est`
 * that the reference fills for (log2_tr_size, is_luma); other entries are left as they are. */
typedef struct x265amd_est_job { uint64_t ctx, est; uint8_t log2_tr_size, is_luma, reserved[6]; } x265amd_est_job;
int x265amd_est_bit(void* stream, const x265amd_est_job* d_jobs, int n);
/* Entropy::codeCoeffNxN in bit-counting mode (entropy.cpp:1828-2200, with costCoeffNxN / costC1C2Flag / costCoeffRemain,
 * source/common/dct.cpp:838-993): FIX15 bits of the levels of one TU under the context set at ctx_in; the adapted context
 * set is written to ctx_out (may equal ctx_in).  d_bits[i] receives what the call adds to Entropy::m_fracBits. */
typedef struct x265amd_coeff_bits_job
{
    uint64_t coeff, ctx_in, ctx_out;
    uint8_t log2_tr_size, ttype, intra, dir_mode, sign_hide, reserved[3];
} x265amd_coeff_bits_job;
int x265amd_coeff_bits(void* stream, const x265amd_coeff_bits_job* d_jobs, int n, uint64_t* d_bits);
/* the same with a wavefront per unit: the contexts of a unit are independent state machines (a state moves only with the bins coded in that context), so
 * the lanes lay out from the levels which bins every context codes, then one lane per context walks its own bins -- instead of one lane coding all of them in
 * turn.  Same bits, same adapted contexts.  The form the fused intra steps use on the device (csrc/entropy_dev.h) */
int x265amd_coeff_bits_wave(void* stream, const x265amd_coeff_bits_job* d_jobs, int n, uint64_t* d_bits);
/* The bits of a whole INTRA CU, a wavefront per CU (csrc/intra_cu_dev.h): what Search::checkIntra / encodeIntraInInter count once the CU's modes and levels are
 * decided (search.cpp:1236-1287, :1454-1507: skip flag + pred mode in P / B slices, codePartSize, codePredInfo, codeCoeff with the coded block flags and the
 * coefficients of every unit), for one transform unit per CU (8x8 .. 32x32) or the 8x8 CU coded NxN; 4:2:0, no delta QP / transform skip.  The piece that lets a CU's
 * decision, and the contexts the next CU starts from, be made on the device.  lev_*: device addresses of the units' levels. */
typedef struct x265amd_intra_cu_bits_job
{
    uint8_t ctx[X265AMD_CTX_STRIDE];        /* the CU's start contexts */
    uint64_t frac_bits;                     /* the coder's fraction at the CU's start (only its low 15 bits count: resetBits) */
    uint8_t log2_cu, nxn, code_part_size, inter_slice, skip_ctx, sign_hide, chroma_dir, cbf_u, cbf_v;
    uint8_t subdiv_flag;                    /* one unit, and the tree could have split (tu-intra-depth > 1): the subdivision flag (0) is coded */
    uint8_t reserved[2];
    uint8_t luma_dir[4], cbf_y[4], preds[4][3];
    uint64_t lev_y[4], lev_u, lev_v;
} x265amd_intra_cu_bits_job;
typedef struct x265amd_intra_cu_bits_out { uint8_t ctx[X265AMD_CTX_STRIDE]; uint64_t frac_bits, mv_frac, skip_frac; } x265amd_intra_cu_bits_out;
int x265amd_intra_cu_bits(void* stream, const x265amd_intra_cu_bits_job* d_jobs, int n, x265amd_intra_cu_bits_out* d_out);
/* host-pointer forms (parity surface) */
void x265amd_est_bit_host(const uint8_t* ctx, int log2TrSize, int isLuma, int32_t* est);
uint64_t x265amd_code_coeff_bits(const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int dirMode, int signHide, uint8_t* ctx);
uint32_t x265amd_transform_tu_rdoq(const x265amd_pixel* fenc, intptr_t fencStride, const int16_t* resi, intptr_t resiStride, int16_t* coeff,
                                   int log2TrSize, int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide,
                                   int tuDepth, int rdoqLevel, int psyRdoqScale, const int32_t* est);

/* host-pointer forms of the two Quant entry points (parity surface, same staging as layer 1) */
uint32_t x265amd_transform_tu(const x265amd_pixel* fenc, intptr_t fencStride, const int16_t* resi, intptr_t resiStride, int16_t* coeff,
                              int log2TrSize, int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide);
void x265amd_invtransform_tu(int16_t* resi, intptr_t resiStride, const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int qpScaled, uint32_t numSig);

/* --- intra: neighbour set + 35-mode luma scan.  Per job: Predict::initAdiPattern with dirMode = ALL_IDX (reference:
 * source/common/predict.cpp:600-649, fillReferenceSamples :736-877: substitution of unavailable neighbours, [1 2 1] or
 * strong bilinear smoothing) on the reconstructed plane, then the sa8d of every one of the 35 luma predictions against
 * the source block -- the mode scan of Search::estIntraPredQT (source/encoder/search.cpp:1566-1613).  The host adds
 * the mode bits (entropy state) and picks the candidates. */
typedef struct x265amd_intra_job
{
    uint64_t recon;                 /* device address of the block's top-left sample inside the reconstructed plane */
    uint64_t fenc;                  /* device address of the source block's top-left sample */
    uint64_t avail;                 /* bit u: neighbour unit u (4 samples) available; order of IntraNeighbors::bNeighborFlags:
                                       below-left (bottom-most first) ... left, above-left, above ... above-right */
    int32_t recon_stride, fenc_stride;
    uint8_t log2_tr_size;           /* 2..5 */
    uint8_t strong_smoothing;       /* sps.bUseStrongIntraSmoothing */
    uint8_t reserved[6];
} x265amd_intra_job;

/* d_sa8d: n * 35 int32 (mode-major per job).  d_neighbours: optional (may be NULL) n * 2 * 129 pixels receiving the
 * unfiltered and filtered neighbour buffers ([0] above-left, [1..2N] above, [2N+1..4N] left).  Asynchronous. */
int x265amd_intra_scan(void* stream, const x265amd_intra_job* d_jobs, int n, int32_t* d_sa8d, x265amd_pixel* d_neighbours);

/* --- inter prediction: Predict::motionCompensation (reference: source/common/predict.cpp:77-243 with
 * predInterLuma/Chroma Pixel/Short :245-408, addWeightBi/Uni :411-577, Yuv::addAvg yuv.cpp:189-211 and CUData::clipMv
 * cudata.cpp:1915-1928), 4:2:0.  One job = one PU: uni- or bi-prediction from one picture per list, optional explicit
 * weights, luma and/or chroma, written to a caller-owned prediction block. */
/* WeightParam (source/common/slice.h:288-317): inputWeight, inputOffset, log2WeightDenom, wtPresent -- the layout of x265amd_mc_job.wp's entries */
typedef struct x265amd_weight { int16_t w, o; uint8_t denom, present; } x265amd_weight;
typedef struct x265amd_mc_job
{
    uint64_t dst_y, dst_u, dst_v;       /* device addresses of the PU's top-left sample in the prediction buffers */
    int32_t dst_stride, dst_cstride;
    int16_t x, y;                       /* PU position in the picture (luma samples) */
    int16_t cu_x, cu_y;                 /* position of the CU that owns the PU (clipMv is relative to the CU) */
    uint8_t w, h;                       /* PU size */
    int8_t ref0, ref1;                  /* picture index into the plane table per list, -1: list unused */
    int16_t mv0[2], mv1[2];             /* quarter-pel MVs (clipped by the kernel exactly like cu.clipMv) */
    uint8_t slice_type;                 /* 1 = P slice, 0 = B slice */
    uint8_t flags;                      /* 1 luma, 2 chroma, 4 pps.bUseWeightPred, 8 pps.bUseWeightedBiPred, 16 pixel average (see x265amd_inter_cost) */
    struct { int16_t w, o; uint8_t denom, present; } wp[2][3];    /* WeightParam inputWeight/inputOffset/log2WeightDenom/wtPresent */
    uint8_t metric;                     /* x265amd_inter_cost only: 1 SAD, 2 SATD, 3 SA8D (0: none) */
    uint8_t chroma_cost;                /* x265amd_inter_cost only: also measure U and V (SATD / SA8D) */
    uint8_t reserved[4];
} x265amd_mc_job;

/* d_planes: num_pics x 3 device addresses of sample (0,0) of the padded Y, U, V planes; all pictures share
 * `stride` / `cstride`.  pic_w / pic_h: luma picture size (clipMv).  Asynchronous. */
int x265amd_motion_compensation(void* stream, const uint64_t* d_planes, intptr_t stride, intptr_t cstride, int pic_w, int pic_h,
                                const x265amd_mc_job* d_jobs, int n);

/* Distortion of inter prediction candidates: the job's prediction is produced exactly as by x265amd_motion_compensation
 * (and written to its dst_* blocks), then measured against the source picture at the PU's position with the metric the
 * reference's decision uses at that point:
 *   metric 1, SAD : Search::selectMVP (source/encoder/search.cpp:1992-2018: bufSAD of the candidate's luma prediction);
 *   metric 2, SATD: Search::mergeEstimation (search.cpp:1891-1966) and the bi-prediction tries of predInterSearch
 *                   (search.cpp:2473-2576): bufSATD, plus bufChromaSATD when chroma_cost is set (PU chroma 4:2:0 a multiple of 4x4);
 *   metric 3, SA8D: the merge scan of Analysis::checkMerge2Nx2N_rd0_4 (source/encoder/analysis.cpp:2750-2880): cu[].sa8d of a
 *                   square PU, plus the chroma sa8d when chroma_cost is set (CU >= 16).
 * flags & 16: the prediction is pixelavg_pp of the two lists' pixel-path luma predictions (the bi-prediction try without
 * chroma SATD, search.cpp:2499-2511; luma only) instead of motionCompensation's addAvg.
 * d_fenc_planes: 3 device addresses of sample (0,0) of the source Y, U, V planes.  d_cost[2*i] = luma, d_cost[2*i+1] = U + V. */
int x265amd_inter_cost(void* stream, const uint64_t* d_planes, intptr_t stride, intptr_t cstride, int pic_w, int pic_h,
                       const x265amd_mc_job* d_jobs, int n, const uint64_t* d_fenc_planes, intptr_t fenc_stride, intptr_t fenc_cstride,
                       uint32_t* d_cost);

/* --- reference-plane production.  d_pic / d_src / d_dst: device address of sample (0,0) of a
 * padded plane with at least marginX samples left/right and marginY rows above/below.
 * x265amd_extend_pic_border = extendPicBorder (source/common/pixel.cpp:1044-1058): the margins repeat the nearest picture sample.
 * x265amd_weight_plane = MotionReference::applyWeight over all rows (source/encoder/reference.cpp:109-185): d_dst receives the
 * weighted copy (weight_pp_c, pixel.cpp:519-538, with w = inputWeight, offset = inputOffset << (depth - 8), shift =
 * log2WeightDenom) of the picture area and its margins -- the plane MotionReference::fpelPlane[] points to when wtPresent. */
int x265amd_extend_pic_border(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY);
/* the same for picture lines y_begin .. y_end - 1 only (their left / right margins; the top margin with line 0, the bottom margin with the last line):
 * the row-by-row form of FrameFilter::processPostRow (source/encoder/framefilter.cpp:592-664), used when pictures are coded in parallel */
int x265amd_extend_border_rows(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY, int y_begin, int y_end);
/* the margins of lines y_begin .. y_end - 1 whose samples are final in the columns x_begin .. x_end - 1 only (a CTU row published column by column while its
 * analysis advances): the left margin when left != 0, the right margin when right != 0, with the last picture line the bottom margin below those columns and
 * with line 0 the top margin above them (their corners under the same conditions). */
int x265amd_extend_border_band(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY, int y_begin, int y_end,
                               int x_begin, int x_end, int left, int right);
int x265amd_weight_plane(void* stream, const x265amd_pixel* d_src, x265amd_pixel* d_dst, intptr_t stride, int width, int height,
                         int marginX, int marginY, int inputWeight, int inputOffset, int log2WeightDenom);

/* --- in-loop deblocking of a picture: Deblock::deblockCTU over all CTUs (reference:
 * source/common/deblock.cpp:37-510; sample filters deblock.cpp:268-310 and source/common/loopfilter.cpp:140-180), 4:2:0.
 * The picture's coding data is one record per 4x4 unit in raster order ((width/4) x (height/4); what the per-CTU CUData arrays
 * hold in z-order).  ref[l]: identity of the picture referenced through list l (any numbering under which equal values mean the
 * same picture, as the reference compares Frame pointers), -1 when the list is unused. */
#define X265AMD_DB_INTRA 1          /* isIntra */
#define X265AMD_DB_CBF 2            /* getCbf(part, TEXT_LUMA, tuDepth): the unit's TU carries luma coefficients */
#define X265AMD_DB_BYPASS 4         /* m_tqBypass */
#define X265AMD_DB_TU_LEFT 8        /* the unit's left border is a TU or CU edge (setEdgefilterTU / bsCuEdge) */
#define X265AMD_DB_PU_LEFT 16       /* ... a PU edge inside the CU (setEdgefilterPU) */
#define X265AMD_DB_TU_TOP 32
#define X265AMD_DB_PU_TOP 64
typedef struct x265amd_slice_info x265amd_slice_info;
typedef struct x265amd_mvpred_info x265amd_mvpred_info;
typedef struct x265amd_cu_unit x265amd_cu_unit;
typedef struct x265amd_mv_unit x265amd_mv_unit;
typedef struct x265amd_deblock_unit { uint8_t flags; int8_t qp; int8_t ref[2]; int16_t mv[2][2]; } x265amd_deblock_unit;
/* d_y / d_u / d_v: sample (0,0) of the reconstructed planes (filtered in place).  width / height: multiples of 8.
 * beta / tc offsets and chroma QP offsets as in the PPS; passes: bit 0 vertical edges, bit 1 horizontal edges (3 = both, in that
 * order).  Asynchronous. */
int x265amd_deblock_picture(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                            int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                            int cbQpOffset, int crQpOffset, int bypassEnabled, int passes);

/* the edges of unit rows y4_begin .. y4_end - 1 only (even numbers; 16 per CTU row): one CTU row of FrameFilter::processRow (framefilter.cpp:559-590).
 * Bands filtered in row order give the picture-wide result. */
int x265amd_deblock_rows(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                         int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                         int cbQpOffset, int crQpOffset, int bypassEnabled, int passes, int y4_begin, int y4_end);
/* ... restricted to the CTU columns ctu_col_begin .. ctu_col_end - 1: the vertical edges right of the first column's left boundary up to and including the right
 * boundary of the last column (the CTU right of it must be analysed), then the horizontal edges inside the columns.  Chunks in column order give the band's
 * result: the last CTU rows of a picture are filtered while their analysis advances, so that pictures referencing it follow a few CTUs behind. */
int x265amd_deblock_rows_cols(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                              int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                              int cbQpOffset, int crQpOffset, int bypassEnabled, int passes, int y4_begin, int y4_end, int ctu_col_begin, int ctu_col_end);

/* the deblocking records of a picture from the maps the analysis fills in (host): edge marks from the CU / PU / TU structure as
 * Deblock::deblockCU sets them (deblock.cpp:70-185), picture identities from info->ref_poc.  out: (width/4) x (height/4) records. */
int x265amd_deblock_units(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                          x265amd_deblock_unit* out);
/* the records of unit rows y4_begin .. y4_end - 1 (same array); picture identities are numbered the same way in every call for a picture */
int x265amd_deblock_units_rows(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                               x265amd_deblock_unit* out, int y4_begin, int y4_end);
/* ... of the unit columns x4_begin .. x4_end - 1 of those rows */
int x265amd_deblock_units_rect(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                               x265amd_deblock_unit* out, int y4_begin, int y4_end, int x4_begin, int x4_end);

/* --- sample adaptive offset over a picture, the two data-parallel halves; 4:2:0, sao-non-deblock off.
 * Plane tables are HOST arrays of 3 device addresses (sample (0,0) of Y, U, V).
 * x265amd_sao_stats = SAO::calcSaoStatsCTU for every CTU and plane (reference: source/encoder/sao.cpp:735-917 with
 * saoCuStatsBO/E0..E3 :1762-1925): d_count / d_offset_org[((ctu * 3 + plane) * 5 + type) * 32 + class], type 0..3 = EO_0..3
 * (classes 0..4), 4 = BO (32 bands); rec = the deblocked planes, fenc = the source planes.
 * x265amd_sao_apply = SAO::generateLumaOffsets / generateChromaOffsets / applyPixelOffsets (sao.cpp:274-733) for given per-CTU
 * parameters (merge modes resolved by the caller): dst receives the offset picture (picture area only); every sample is
 * classified on the deblocked src planes.  The parameter choice (rdoSaoUnitCu) is the host loop's. */
typedef struct x265amd_sao_ctu
{
    int8_t type[2];                 /* SaoCtuParam::typeIdx of luma / of both chroma planes: -1 off, 0..3 EO_0..3, 4 BO */
    uint8_t band_pos[3];            /* SaoCtuParam::bandPos per plane */
    int8_t offset[3][4];            /* SaoCtuParam::offset per plane */
    uint8_t reserved[3];
} x265amd_sao_ctu;
int x265amd_sao_stats(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                      int width, int height, int32_t* d_count, int32_t* d_offset_org);
int x265amd_sao_apply(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                      int width, int height, const x265amd_sao_ctu* d_params);
/* the same for CTU rows ctu_row_begin .. ctu_row_end - 1 (arrays indexed by the CTU address in the picture): the row-by-row order of the reference's
 * frame filter (source/encoder/framefilter.cpp:559-664), used when pictures are coded in parallel.  A row's statistics need the row deblocked; its offset
 * samples need the row below deblocked as well. */
int x265amd_sao_stats_rows(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                           int width, int height, int32_t* d_count, int32_t* d_offset_org, int ctu_row_begin, int ctu_row_end);
int x265amd_sao_apply_rows(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                           int width, int height, const x265amd_sao_ctu* d_params, int ctu_row_begin, int ctu_row_end);
/* the column forms: statistics of the CTU columns ctu_col_begin .. ctu_col_end - 1 of those rows; offset samples of the luma sample columns x_begin .. x_end - 1
 * (even; the deblocked input must be final one sample beyond either end) */
int x265amd_sao_stats_rows_cols(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                                int width, int height, int32_t* d_count, int32_t* d_offset_org, int ctu_row_begin, int ctu_row_end, int ctu_col_begin, int ctu_col_end);
int x265amd_sao_apply_rows_cols(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                                int width, int height, const x265amd_sao_ctu* d_params, int ctu_row_begin, int ctu_row_end, int x_begin, int x_end);

/* --- final entropy coding of CTUs: the CABAC write pass, host code.  Entropy::encodeCTU / encodeCU /
 * encodeTransform / codePredInfo / codeCoeffNxN and the arithmetic coder (reference: source/encoder/entropy.cpp:768-1222, :1431-2200,
 * :2399-2612) with CUData's context derivations (source/common/cudata.cpp:814-1012); 4:2:0, no transform skip.
 * The decisions of a picture are one record per 4x4 unit in raster order ((pic_width/4) x (pic_height/4)); the fields are those of
 * CUData (source/common/cudata.h:190-240), per-CU / per-PU / per-TU values repeated over the units they cover. */
enum { X265AMD_MODE_NONE = 0, X265AMD_MODE_INTER = 1, X265AMD_MODE_INTRA = 2, X265AMD_MODE_SKIP = 3 };
typedef struct x265amd_cu_unit
{
    uint8_t depth;                  /* m_cuDepth */
    uint8_t pred_mode;              /* X265AMD_MODE_* (m_predMode) */
    uint8_t part_size;              /* m_partSize: 0 2Nx2N, 1 2NxN, 2 Nx2N, 3 NxN, 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N */
    uint8_t tu_depth;               /* m_tuDepth */
    uint8_t luma_dir, chroma_dir;   /* m_lumaIntraDir, m_chromaIntraDir (36 = DM_CHROMA_IDX) */
    uint8_t merge_flag, inter_dir;  /* m_mergeFlag, m_interDir (1 L0, 2 L1, 3 both) */
    uint8_t cbf[3];                 /* m_cbf[plane]: bit d = coded coefficients in the unit's TU at depth d */
    uint8_t tq_bypass;
    int8_t qp;                      /* m_qp (rewritten by the coder where no delta QP is coded, as finishCU does) */
    int8_t ref_idx[2];              /* m_refIdx */
    uint8_t mvp_idx[2];             /* m_mvpIdx ([0] carries the merge index of merged / skipped blocks) */
    uint8_t reserved;
    int16_t mvd[2][2];              /* m_mvd */
} x265amd_cu_unit;                  /* 26 bytes */
typedef struct x265amd_slice_info
{
    int32_t pic_width, pic_height;  /* multiples of 8 (minimum CU size); CTU size 64 */
    int32_t slice_type;             /* 0 B, 1 P, 2 I */
    int32_t slice_qp;
    int32_t num_ref_idx[2];
    int32_t max_num_merge_cand;
    int32_t use_dqp, max_cu_dqp_depth;      /* pps.bUseDQP, pps.maxCuDQPDepth */
    int32_t sign_hide, tq_bypass_enabled;   /* pps.bSignHideEnabled, pps.bTransquantBypassEnabled */
    int32_t wpp;                            /* pps.bEntropyCodingSyncEnabled (only read by the last-coded-QP rule) */
    int32_t max_cu_depth;                   /* param.maxCUDepth: log2(maxCUSize) - log2(minCUSize) */
    int32_t max_amp_depth;                  /* sps.maxAMPDepth */
    int32_t tu_log2_min, tu_log2_max;       /* sps.quadtreeTULog2MinSize / MaxSize */
    int32_t tu_max_depth_inter, tu_max_depth_intra;
} x265amd_slice_info;
typedef struct x265amd_cabac x265amd_cabac;
/* units: the picture's map (kept by reference; its qp fields are updated).  bits_only != 0: no bitstream, fractional bits are
 * counted instead (the reference's m_fracBits, FIX15).  Contexts start at Entropy::resetEntropy(slice). */
x265amd_cabac* x265amd_cabac_open(const x265amd_slice_info* si, x265amd_cu_unit* units, int bits_only);
void x265amd_cabac_close(x265amd_cabac*);
void x265amd_cabac_set_contexts(x265amd_cabac*, const uint8_t* ctx);
void x265amd_cabac_get_contexts(const x265amd_cabac*, uint8_t* ctx);
/* Entropy::encodeCTU for CTU ctu_addr; coeff*: the CTU's quantised levels in the reference's layout (CUData::m_trCoeff: TU blocks at
 * z-order offsets, 64*64 luma and 32*32 per chroma plane). */
int x265amd_cabac_encode_ctu(x265amd_cabac*, int ctu_addr, const int16_t* coeffY, const int16_t* coeffU, const int16_t* coeffV);
uint64_t x265amd_cabac_frac_bits(const x265amd_cabac*);      /* the reference's m_fracBits (finishCU's resetBits() keeps only the fraction at a CTU end) */
uint64_t x265amd_cabac_ctu_bits(const x265amd_cabac*);       /* bit-counting mode: FIX15 bits of the CTU coded last */
/* Entropy::finishSlice: terminating bin, flush, rbsp trailing bits; returns the slice data size in bytes, copied to out when it fits */
size_t x265amd_cabac_finish_slice(x265amd_cabac*, uint8_t* out, size_t cap);

/* --- motion vector prediction, host code: CUData::getInterMergeCandidates (reference: source/common/cudata.cpp:1458-1712)
 * and getNeighbourMV + getPMV (:1715-1875) with the temporal candidates (:1968-2045).  The motion field of a picture is one record per
 * 4x4 unit in raster order; "cur" holds what has been decided so far in coding order (including the current CU's earlier PUs), "col"
 * is the co-located picture's field (only read when temporal_mvp).  Intra / uncoded units carry ref_idx -1. */
typedef struct x265amd_mv_unit { uint8_t pred_mode, inter_dir; int8_t ref_idx[2]; int16_t mv[2][2]; } x265amd_mv_unit;      /* 12 bytes */
typedef struct x265amd_mvpred_info
{
    int32_t pic_width, pic_height;
    int32_t is_inter_b, num_ref_idx[2], max_num_merge_cand;
    int32_t temporal_mvp;           /* sps.bTemporalMVPEnabled */
    int32_t col_from_l0, check_ldc; /* slice.m_colFromL0Flag, m_bCheckLDC */
    int32_t poc, ref_poc[2][16];    /* slice.m_poc, m_refPOCList */
    int32_t col_poc, col_ref_poc[2][16];    /* the same of the co-located picture's slice */
} x265amd_mvpred_info;
typedef struct x265amd_merge_cand { int16_t mv[2][2]; int8_t ref_idx[2]; uint8_t dir; uint8_t reserved; } x265amd_merge_cand;
/* returns the number of candidates written (max_num_merge_cand); part_size as in x265amd_cu_unit */
int x265amd_merge_candidates(const x265amd_mvpred_info* info, const x265amd_mv_unit* cur, const x265amd_mv_unit* col, int cu_x, int cu_y, int log2_cu_size,
                             int part_size, int pu_idx, x265amd_merge_cand* out);
/* amvp: the two AMVP candidates of (list, ref_idx); mvc: the motion candidates for the search start (returns their number, at most 11) */
int x265amd_amvp_candidates(const x265amd_mvpred_info* info, const x265amd_mv_unit* cur, const x265amd_mv_unit* col, int cu_x, int cu_y, int log2_cu_size,
                            int part_size, int pu_idx, int list, int ref_idx, int16_t amvp[2][2], int16_t mvc[12][2]);

/* --- whole-CU inter search for a batch of CUs: Search::predInterSearch (reference: source/encoder/search.cpp:2181-2647)
 * with mergeEstimation, selectMVP, setSearchRange, checkBestMVP, getBlkBits.  Decisions are made on the host in the reference's order;
 * the block operations of each step of all CUs run as one GPU batch (x265amd_inter_cost, x265amd_me_search,
 * x265amd_motion_compensation).  No weighted prediction, HME or analysis reuse. */
typedef struct x265amd_inter_search_params
{
    int32_t search_method, subpel_refine, search_range;     /* param.searchMethod (X265AMD_ME_*), subpelRefine, searchRange */
    int32_t qp;                                             /* the CU's QP: lambda (RDCost::setQP) and the MV cost table (MotionEstimate::setQP) */
    int32_t chroma_mc;                                      /* bChromaMC (non-zero = yes): chroma in the final prediction, and chroma SATD when subpel_refine > 2 */
    int32_t ref_pic[2][16];                                 /* picture index (into the plane table) of reference r of list l */
    int32_t frame_parallel;                                 /* Search::m_bFrameParallel (param.frameNumThreads > 1, search.cpp:77): vertical search limit
                                                             * m_refLagPixels = searchRange (search.cpp:92, :2763), merge candidates (search.cpp:1934,
                                                             * analysis.cpp:2803, :2933) and AMVP candidates (search.cpp:2009) reaching below it are left out */
    int32_t lazy_sync;                                      /* device job queues only, non-zero: the call returns with the final predictions enqueued, not finished -- the
                                                             * decisions are final, and the caller's next command on the queue (the measurement of the prediction tile) is
                                                             * ordered behind them.  0 (every public caller): the call returns with everything done */
    int32_t lowres_blocks_in_row;                           /* Lowres::maxBlocksInRow of the fields below */
    int32_t me_pic[2][16];                                  /* weighted == 0: unused.  Else the plane-table index MotionEstimate searches for reference r of list l
                                                             * (MotionReference::fpelPlane: the weighted copy of the reference when its luma weight is present,
                                                             * reference.cpp:51-116; motion.cpp:642, :781, :1599); everything else -- selectMVP's candidates, merge
                                                             * candidates, final predictions -- reads ref_pic and weights in the prediction (predict.cpp:85-232) */
    int32_t weighted;                                       /* 0 none; 1: a P slice with pps.bUseWeightPred, 2: a B slice with pps.bUseWeightedBiPred: `wp` holds
                                                             * slice.m_weightPredTable and every motion compensation of the slice takes its weights from it */
    x265amd_weight wp[2][16][3];
    uint64_t lowres_mvs[2][16];                             /* HOST address of the lookahead's motion field of the current picture towards reference r of list l
                                                             * (Lowres::lowresMvs[l][|poc - refPoc|], int16_t[blocks][2], lowres full-pel x 4 as the lookahead keeps
                                                             * them), or 0 where the lookahead has not searched that distance: Search::getLowresMV
                                                             * (search.cpp:1968-1989), one more search candidate per PU when it is not zero (:2102-2107, :2413-2418) */
} x265amd_inter_search_params;
typedef struct x265amd_inter_cu { int16_t x, y; uint8_t log2_size, part_size; uint8_t reserved[2]; } x265amd_inter_cu;
typedef struct x265amd_pu_result
{
    uint8_t merge_flag, inter_dir;  /* m_mergeFlag, m_interDir */
    int8_t ref_idx[2];
    uint8_t mvp_idx[2];             /* m_mvpIdx ([0] = merge index for merged PUs) */
    uint8_t reserved[2];
    int16_t mv[2][2], mvd[2][2];
} x265amd_pu_result;
/* cur: the picture's motion field (host; patched and restored during the call); col: co-located field or NULL.  h_planes: HOST array of
 * num_pics x 3 device addresses (sample (0,0) of Y, U, V; the last picture is the source).  out: n x 2 PU results; bits_out[i]: what
 * predInterSearch adds to sa8dBits.  d_pred: device buffer of n tiles of pred_bytes_per_cu bytes, each 64x64 luma (stride 64) followed by
 * 32x32 U and V (stride 32), receiving interMode.predYuv.  Synchronous (it reads costs back between its steps). */
int x265amd_pred_inter_search(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* info, const x265amd_inter_search_params* sp,
                              x265amd_mv_unit* cur, const x265amd_mv_unit* col, const uint64_t* h_planes, int num_pics, intptr_t stride, intptr_t cstride,
                              const x265amd_inter_cu* cus, int n, x265amd_pu_result* out, int32_t* bits_out, uint64_t d_pred, size_t pred_bytes_per_cu);

/* x265amd_pred_inter_search that also reports what the reference keeps in Mode::bestME[0][list] / amvpCand of the CU's first PU (search.h:79-99),
 * which Analysis::checkBidir2Nx2N reads afterwards.  cost[l] == 0xFFFFFFFF: list l was not searched.
 * ref_masks (may be NULL): n x 2 words, predInterSearch's refMasks per PU: bit r allows reference r of list 0, bit 16 + r of list 1; 0 = all. */
typedef struct x265amd_me_detail
{
    int16_t mv[2][2], mvp[2][2];
    int16_t amvp[2][2][2];          /* the two AMVP candidates of each list's best reference */
    int8_t ref[2]; uint8_t mvp_idx[2];
    uint32_t bits[2], cost[2], mv_cost[2];
    uint32_t list_sel_bits[3];      /* Search::m_listSelBits */
} x265amd_me_detail;                /* 72 bytes */
int x265amd_pred_inter_search_ex(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* info, const x265amd_inter_search_params* sp,
                                 x265amd_mv_unit* cur, const x265amd_mv_unit* col, const uint64_t* h_planes, int num_pics, intptr_t stride, intptr_t cstride,
                                 const x265amd_inter_cu* cus, int n, x265amd_pu_result* out, int32_t* bits_out, uint64_t d_pred, size_t pred_bytes_per_cu,
                                 x265amd_me_detail* detail, const uint32_t* ref_masks);

/* --- residual RD of inter CUs: Search::encodeResAndCalcRdInterCU (reference: source/encoder/search.cpp:2822-2975) with
 * estimateResidualQT (:3178-3857), splitTU (:3126-3176), estimateNullCbfCost (:3114-3124), codeInterSubdivCbfQT (:3859-3887),
 * saveResidualQTData (:3889-3972) and checkDQP (:3974-4003), for a batch of independent candidate CUs.
 * The transform chains of every node of every CU's residual quad-tree run as one x265amd_tu_chain launch (plain quantisation does not
 * depend on the entropy state); the bit counting and the decisions walk the tree on the host in the reference's order with the
 * bit-counting CABAC coder; a second launch assembles the chosen residual, reconstructs and measures the CU.
 * Supported: 4:2:0, rdoqLevel 0-2 (psy-rdoq), no transform skip / lossless / limit-tu / ssim-rd, chroma QP offsets 0. */
typedef struct x265amd_rd_params
{
    double psy_rd;                  /* param.psyRd (RDCost::setPsyRdScale, rdcost.h:43) */
    int32_t rd_level;               /* param.rdLevel: how checkDQP prices a delta QP (>= 3 codes it, 2 counts one bit) */
    int32_t strong_intra_smoothing; /* sps.bUseStrongIntraSmoothing (intra candidates only) */
    int32_t rdoq_level;             /* param.rdoqLevel 0 / 1 / 2: with RDOQ every transform unit is quantised under the entropy state the walk has reached
                                     * (Entropy::estBit before each Quant::transformNxN, search.cpp:355, :852, :3272, :3397), one launch per unit */
    int32_t psy_rdoq_scale;         /* Quant::m_psyRdoqScale = (int)(param.psyRdoq * 256) */
    int32_t fast_intra;             /* param.bEnableFastIntra: checkIntraInInter samples every fifth angle and refines (search.cpp:1401-1434) */
    int32_t limit_tu;               /* param.limitTU 0, 2, 3, 4 (1 is not built): the inter residual quadtree stops early (search.cpp:3136-3142, :3209-3216, :3713-3726) */
} x265amd_rd_params;                /* 32 bytes */
typedef struct x265amd_rd_cu
{
    int16_t x, y;                   /* luma position of the CU in the picture */
    uint8_t log2_size;              /* 3..6 */
    int8_t qp;                      /* the CU's QP (setLambdaFromQP: RD lambdas and quantiser) */
    uint8_t reserved[2];            /* [1]: with limit_tu 3 / 4, Search::m_maxTUDepth + 1 as the analysis has it when the CU's residual is coded (0: no limit).
                                     * [0]: the QP of the LAMBDAS when it is not `qp` (0: it is).  Search::setLambdaFromQP (search.cpp:177-187) takes the lambdas from the QP the rate
                                     * control asks for, which adaptive quantisation can push up to 69, and clips the quantiser's (and the coded) QP to 51: above 51 the two differ */
    uint64_t frac_bits;             /* m_rqt[depth].cur.m_fracBits on entry (only its low 15 bits matter) */
    uint8_t ctx[X265AMD_CTX_STRIDE];/* m_rqt[depth].cur context states on entry */
} x265amd_rd_cu;                    /* 176 bytes */
typedef struct x265amd_rd_result
{
    uint64_t rd_cost, distortion;   /* Mode::rdCost, Mode::distortion */
    uint64_t frac_bits;             /* Mode::contexts.m_fracBits */
    uint32_t total_bits, mv_bits, coeff_bits, psy_energy, luma_distortion, chroma_distortion, res_energy, reserved;
    uint8_t ctx[X265AMD_CTX_STRIDE];/* Mode::contexts context states */
} x265amd_rd_result;                /* 216 bytes */
/* units: the picture's unit map (what is coded so far; read for the skip-flag / QP neighbourhood, patched and restored during the call).
 * cu_units: n x 256 records, the candidate CU's units in raster order with row length size/4, carrying the prediction fields
 * (pred_mode X265AMD_MODE_INTER, part_size, merge_flag, inter_dir, ref_idx, mvp_idx, mvd, qp); on return tu_depth, cbf, pred_mode (SKIP when
 * a merged 2Nx2N CU ends without residual) and qp are filled in as the reference leaves them in Mode::cu.
 * h_src: HOST array of 3 device addresses, sample (0,0) of the source Y, U, V (strides stride / cstride).
 * d_pred / d_recon: device tiles as x265amd_pred_inter_search writes them (64x64 luma stride 64, then 32x32 U and V stride 32);
 * tile_bytes apart.  coeff_out: HOST, n x (4096 + 2 x 1024) levels in CUData::m_trCoeff layout (TU blocks at z-order offsets), only TUs with
 * a coded block flag are written (others zero); may be NULL.  Synchronous. */
int x265amd_inter_residual_rd(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                              const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                              x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes,
                              x265amd_rd_result* out, int16_t* coeff_out);

/* The host stages x265amd_inter_residual_rd runs around its two launches, callable on their own (no GPU work inside): the addresses they
 * handle are opaque, so a test can drive them with any engine that executes x265amd_tu_job records.
 * Per-CU scratch (x265amd_inter_rd_scratch_bytes() apart): E = 4 * 4096 + 6 * 1024 elements each of levels (int16), residual (int16) and a
 * reconstruction dump (pixels).  Element offsets: luma layer L (transform log2 size 2..5) at (L - 2) * 4096, chroma layer C (2..4) of plane
 * p (1, 2) at 16384 + ((C - 2) * 2 + p - 1) * 1024; levels are TU blocks in raster TU order, residual / dump blocks sit at their position in
 * a stride-64 (luma) / stride-32 (chroma) tile.  The selection map (384 bytes per CU): for each luma 4x4 unit (row length 16), then each
 * chroma 4x4 unit of U and of V (row length 8), the layer whose residual block was kept, 0xFF for none. */
typedef struct x265amd_cu_measure
{
    uint64_t sse[3]; uint32_t psy, sa8d;    /* sse_pp per plane, luma psyCost, cu[].sa8d of Y + U + V */
    uint32_t sa8d_luma;                     /* cu[].sa8d of Y alone (rd levels below 3 rank by luma) */
    uint32_t src_mean, src_homo, reserved;  /* mean and mean absolute deviation of the SOURCE luma block (Analysis::complexityCheckCU) */
} x265amd_cu_measure;                       /* 48 bytes */
/* tile-vs-source measurement of n CUs (k_cu_measure without assembly): d_tiles as d_pred above; cus: only x, y, log2_size are read.  Synchronous. */
int x265amd_measure_tiles(void* stream, const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                          uint64_t d_tiles, size_t tile_bytes, x265amd_cu_measure* out);
/* the same for tiles at arbitrary device addresses (tile_addrs[i] for cus[i]), one launch */
int x265amd_measure_tile_list(void* stream, const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                              const uint64_t* tile_addrs, x265amd_cu_measure* out);
size_t x265amd_inter_rd_scratch_bytes(void);
/* emits the transform-chain jobs of all CUs (returns their number; jobs_out may be NULL to count) */
int x265amd_inter_rd_plan(const x265amd_slice_info* si, const x265amd_rd_cu* cus, int n, const x265amd_cu_unit* cu_units, const uint64_t* src,
                          intptr_t stride, intptr_t cstride, uint64_t pred, size_t tile_bytes, uint64_t scratch, x265amd_tu_job* jobs_out, int cap);
/* the decisions: res = the results of the plan's jobs in order, levels = the level part of CU i's scratch at levels + i * levels_stride_bytes,
 * zero_meas = prediction-vs-source measurement per CU; fills cu_units, sel, the entropy side of out, coeff_out */
int x265amd_inter_rd_walk(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                          x265amd_cu_unit* cu_units, const x265amd_tu_result* res, const int16_t* levels, size_t levels_stride_bytes,
                          const x265amd_cu_measure* zero_meas, uint8_t* sel, x265amd_rd_result* out, int16_t* coeff_out);
/* final_meas = reconstruction-vs-source measurement per CU: distortion, psy energy and rd_cost of out */
void x265amd_inter_rd_finish(const x265amd_slice_info* si, const x265amd_rd_params* rp, const x265amd_rd_cu* cus, int n,
                             const x265amd_cu_measure* final_meas, x265amd_rd_result* out);

/* Search::encodeResAndCalcRdSkipCU (reference: source/encoder/search.cpp:2770-2818) for a batch of merge candidates: reconstruction =
 * prediction, distortion / psy energy from one k_cu_measure launch, bits of the skip flag and the merge index (cu_units[].mvp_idx[0]).
 * Arguments as x265amd_inter_residual_rd; cu_units return pred_mode SKIP, cbf 0, tu_depth 0.  x265amd_skip_rd_host is its host stage. */
int x265amd_skip_rd(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                    const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                    x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes, x265amd_rd_result* out);
int x265amd_skip_rd_host(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                         x265amd_cu_unit* cu_units, const x265amd_cu_measure* meas, x265amd_rd_result* out);

/* --- intra candidate of an inter-slice CU (RD side): Search::checkIntraInInter + encodeIntraInInter (reference:
 * source/encoder/search.cpp:1291-1452, :1454-1507) with codeIntraLumaQT (:305-508), estIntraPredChromaQT (:1754-1889), codeIntraChromaQt
 * (:819-945) for ONE 2Nx2N CU of 8..32 samples: the 35-mode SA8D scan + mode bits choose the luma mode, then every transform unit is one
 * x265amd_intra_tu_chain launch whose reconstruction is written into the reconstructed picture before the next one starts (intra blocks
 * predict from their neighbours' reconstruction, so the TUs of a CU are serial), the five chroma modes are tried in the reference's order.
 * h_rec: HOST array of 3 device addresses of the reconstructed picture (read for neighbours, written with this CU's reconstruction).
 * cu_units: (size/4)^2 records (raster, row length size/4) returning pred_mode INTRA, luma_dir, chroma_dir, tu_depth, cbf, qp.
 * d_pred: tile receiving intraMode.predYuv (luma only), d_recon: tile receiving intraMode.reconYuv.  info (may be NULL): luma mode, sa8dCost,
 * sa8dBits, SA8D distortion as checkIntraInInter leaves them.  Synchronous. */
int x265amd_intra_in_inter(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                           const uint64_t* h_src, const uint64_t* h_rec, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu,
                           x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out, int16_t* coeff_out, uint64_t* info);

/* Search::checkIntra (reference: source/encoder/search.cpp:1236-1287) with estIntraPredQT (:1509-1696): the intra CU of I slices and of the
 * rd 5-6 paths.  part_size 0 (2Nx2N) or 3 (NxN, 8x8 CUs): per partition the 35-mode scan, the candidate list (updateCandList, at most
 * 2 + rd_level + depth / 2 modes within 25 % of the best or MPM[0]), simple RDO of each candidate, then the winner again with transform
 * splits allowed; chroma and the CU's bits as x265amd_intra_in_inter.  Arguments as x265amd_intra_in_inter. */
int x265amd_check_intra(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                        const uint64_t* h_src, const uint64_t* h_rec, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, int part_size,
                        x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out, int16_t* coeff_out);

/* --- CTU mode decision of inter slices: Analysis::compressCTU -> compressInterCU_rd0_4 (reference:
 * source/encoder/analysis.cpp:138-317, :1146-1848) with checkMerge2Nx2N_rd0_4 (:2750-2880), checkInter_rd0_4 (:3023-3085), checkBidir2Nx2N
 * (:3145-3277), topSkipMinDepth (:3428-3476), recursionDepthCheck (:3479-3534), addSplitFlagCost (:3405-3426).  Host recursion in the
 * reference's order over the batch entry points above; one CTU per call.
 * Built subset: rd levels 2-6; I, P and B slices (b_intra 0 / 1), 2Nx2N / rect / amp partitions, limit_refs 0-3, limit_modes 0 / 1, no delta QP,
 * rd_level 3-4, rskip 0 / 1, early_skip 0 / 1.  Anything else is rejected with X265AMD_EINVAL. */
typedef struct x265amd_analysis_params
{
    double psy_rd;                  /* param.psyRd */
    int32_t rd_level, early_skip, rskip, limit_refs, b_intra, rect, amp, limit_modes;
    int32_t strong_intra_smoothing;             /* sps.bUseStrongIntraSmoothing */
    int32_t use_sao;                            /* slice.m_bUseSao: x265amd_analyse_frame only (the row coder counts bits only when SAO is on) */
    int32_t rdoq_level, psy_rdoq_scale;         /* param.rdoqLevel, (int)(param.psyRdoq * 256) */
    int32_t fast_intra, limit_tu;               /* param.bEnableFastIntra, param.limitTU (0, 2, 3, 4) */
} x265amd_analysis_params;          /* 64 bytes */
typedef struct x265amd_cu_stat { uint32_t count[4]; uint32_t pad[2]; uint64_t avg_cost[4]; } x265amd_cu_stat;     /* FrameData::RCStatCU count / avgCost per depth */
typedef struct x265amd_ctu_result { uint64_t rd_cost, distortion, frac_bits; uint32_t total_bits, reserved; uint8_t ctx[X265AMD_CTX_STRIDE]; } x265amd_ctu_result;
/* units / cur: the picture's unit map and motion field (what is coded so far); the CTU's part is reset and then filled with the decisions.
 * ref_depth: CU depths of the co-located pictures refFrameList[0][0] and [1][0], one byte per 4x4 unit, two maps; ref_qp0: their CTU QPs
 * (2 x number of CTUs).  h_planes: HOST array of num_pics x 3 device addresses; the last picture is the source, the one before it receives
 * the reconstruction.  cu_stat: per CTU, updated.  ctx_in / frac_in: the row coder's state at the CTU start.  coeff_out: 4096 + 2 x 1024
 * levels in CUData::m_trCoeff layout (may be NULL).  merge_flag / mvp_idx / mvd of a PU are uniform over its units here (the reference only
 * writes them at the PU's first unit).  Synchronous. */
int x265amd_compress_ctu_inter(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* info, const x265amd_inter_search_params* sp,
                               const x265amd_slice_info* si, const x265amd_analysis_params* ap, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                               const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                               intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int ctu_addr, const uint8_t* ctx_in, uint64_t frac_in,
                               int16_t* coeff_out, x265amd_ctu_result* out);

/* The CTU loop of one frame (FrameEncoder::processRowEncoder, reference: source/encoder/frameencoder.cpp:1399-1700, without rate control,
 * VBV, slices and in-loop filters): clears units / cur / cu_stat, then for every CTU takes the start state from the row coder (si->wpp: one
 * coder per row, rows > 0 start from the state saved after the second CTU of the row above; otherwise one coder through all rows), runs
 * x265amd_compress_ctu_inter and lets the row coder code the CTU in bit-counting mode.  coeff_out: numCtu x (4096 + 2 x 1024) levels.
 * results (may be NULL): per CTU.  slice_data / substream_sizes / num_substreams (may be NULL): the CABAC sub-streams as
 * FrameEncoder::encodeSlice writes them (:1298-1370), back to back: one per CTU row under WPP, otherwise one; x265amd_write_slice_nal packs
 * them behind the slice header.  Other arguments as x265amd_compress_ctu_inter. */
int x265amd_analyse_frame(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* info, const x265amd_inter_search_params* sp,
                          const x265amd_slice_info* si, const x265amd_analysis_params* ap, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                          const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                          intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int16_t* coeff_out, x265amd_ctu_result* results,
                          uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams);

/* --- slice NAL units (host): Entropy::codeSliceHeader + codeSliceHeaderWPPEntryPoints + the byte-stream packing of
 * NALList::serialize / serializeSubstreams (reference: source/encoder/entropy.cpp:593-766, source/encoder/nal.cpp:60-232).  One slice per picture;
 * no weighted prediction tables, long-term references or dependent slices.  slice_type as in the bitstream: 0 B, 1 P, 2 I. */
typedef struct x265amd_slice_header
{
    int32_t nal_unit_type;          /* 0 TRAIL_N, 1 TRAIL_R, 19 IDR_W_RADL, 20 IDR_N_LP, 21 CRA ... */
    int32_t temporal_id_plus1;      /* 0 is taken as 1 */
    int32_t first_in_access_unit;   /* 4-byte start code (first NAL of the access unit) instead of 3 */
    int32_t slice_type, poc, last_idr_poc, log2_max_poc_lsb;
    int32_t rps_idx, num_rps_in_sps;/* rps_idx < 0: the reference picture set is coded in the header */
    int32_t num_negative, num_positive, delta_poc[16], used[16];
    int32_t temporal_mvp_enabled, use_sao, sao_luma, sao_chroma, selective_sao;
    int32_t num_ref_idx[2], num_ref_idx_default[2], col_from_l0, col_ref_idx, max_num_merge_cand;
    int32_t slice_qp, pps_init_qp;  /* 26 + init_qp_minus26 of the PPS */
    int32_t chroma_qp_offsets_present, cb_qp_offset, cr_qp_offset;
    int32_t deblocking_disabled;    /* pps.bPicDisableDeblockingFilter */
    int32_t slfase_flag;            /* slice.m_sLFaseFlag */
    int32_t wpp;                    /* entry points are written; one sub-stream per CTU row */
    int32_t weighted_pred;          /* pps.bUseWeightPred: a P slice carries pred_weight_table() (Entropy::codePredWeightTable, entropy.cpp:1358-1429): the two denominators, a luma
                                     * and a chroma flag per reference, then the weights of the references whose flags are set (`wp`) */
    int32_t luma_log2_weight_denom, chroma_log2_weight_denom;
    int32_t weighted_bipred;        /* pps.bUseWeightedBiPred: the same for a B slice, list 0 then list 1 */
    x265amd_weight wp[2][16][3];    /* slice.m_weightPredTable[list][ref][plane] (all zero: no reference carries weights) */
} x265amd_slice_header;
/* substreams: the raw (unescaped) CABAC sub-streams back to back, sizes[i] bytes each.  Returns the NAL size in bytes (written when it fits). */
size_t x265amd_write_slice_nal(const x265amd_slice_header* h, const uint8_t* substreams, const uint32_t* sizes, int num_substreams, uint8_t* out, size_t cap);

/* Stream headers: Encoder::getStreamHeaders' VPS, SPS and PPS NAL units (reference: source/encoder/encoder.cpp:3234-3259, Entropy::codeVPS / codeSPS /
 * codeProfileTier / codeVUI / codePPS, source/encoder/entropy.cpp:233-502).  The fields are the reference's VPS / SPS / PPS / ProfileTierLevel / VUI
 * members (source/common/slice.h) as the encoder configured them; no scaling lists, SPS reference picture sets or HRD parameters. */
typedef struct x265amd_stream_params
{
    /* ProfileTierLevel */
    int32_t tier_flag, profile_idc; uint32_t profile_compatibility_flags;   /* bit j = profileCompatibilityFlag[j] */
    int32_t progressive_source, interlaced_source, non_packed_constraint, frame_only_constraint;
    int32_t bit_depth_constraint, chroma_format_constraint, intra_constraint, one_picture_only_constraint, lower_bit_rate_constraint;   /* range extension profiles */
    int32_t level_idc;
    int32_t max_temporal_sub_layers, max_dec_pic_buffering[8], num_reorder_pics[8], max_latency_increase[8];
    /* SPS */
    int32_t chroma_format_idc, pic_width, pic_height, conformance_window, conf_win_offsets[4];      /* left, right, top, bottom in luma samples */
    int32_t bit_depth, log2_max_poc_lsb, log2_min_cu_size, log2_diff_max_min_cu_size, tu_log2_min, tu_log2_max, tu_max_depth_inter, tu_max_depth_intra;
    int32_t amp, sao, temporal_mvp, strong_intra_smoothing;
    /* VUI */
    int32_t aspect_ratio_idc, sar_width, sar_height;                /* aspect_ratio_idc 0: not present */
    int32_t overscan_info_present, overscan_appropriate, video_signal_type_present, video_format, video_full_range;
    int32_t colour_description_present, colour_primaries, transfer_characteristics, matrix_coefficients;
    int32_t chroma_loc_info_present, chroma_sample_loc_top, chroma_sample_loc_bottom, field_seq, frame_field_info_present;
    int32_t default_display_window, def_disp_win_offsets[4];
    int32_t emit_timing_info; uint32_t num_units_in_tick, time_scale;
    /* PPS */
    int32_t sign_hide, num_ref_idx_default[2], init_qp_minus26, constrained_intra_pred, transform_skip, use_dqp, max_cu_dqp_depth;
    int32_t cb_qp_offset, cr_qp_offset, slice_chroma_qp_offsets_present, weighted_pred, weighted_bipred, transquant_bypass, wpp;
    int32_t loop_filter_across_slices, deblocking_filter_control_present, pic_disable_deblocking, beta_offset_div2, tc_offset_div2;
} x265amd_stream_params;
/* Returns the size of the three NAL units in bytes (written to out when it fits), 0 on bad arguments. */
size_t x265amd_write_stream_headers(const x265amd_stream_params* p, uint8_t* out, size_t cap);
/* the user-data SEI NAL unit (prefix SEI, payload type 5 with the reference's UUID) that carries the encoder's name and option string behind the parameter sets when
 * param.bEmitInfoSEI is set (reference: source/encoder/encoder.cpp:3260-3280, sei.h:89-117); returns its size behind a 4-byte start code, 0 when it does not fit */
size_t x265amd_write_info_sei(const char* text, uint8_t* out, size_t cap);
/* one SEI message as a NAL unit of its own (prefix 39 / suffix 40; SEI::writeSEImessages, sei.cpp:39-73) and the access unit delimiter (Entropy::codeAUD; slice_type 0 B, 1 P, 2 I):
 * host code, the unit's size or 0 */
size_t x265amd_write_sei(int suffix, int payload_type, const uint8_t* payload, size_t n, uint8_t* out, size_t cap);
size_t x265amd_write_aud(int slice_type, uint8_t* out, size_t cap);
/* the decoded picture hash SEI's payload for a reconstructed 4:2:0 picture in host memory (--hash: method 1 MD5, 2 CRC, 3 checksum; H.265 D.3.19 as the reference computes it,
 * frameencoder.cpp:1228-1296): hash_type + the three planes' digests; host code (host/picture_hash.cpp).  Returns the payload's size or 0 */
size_t x265amd_picture_hash(int method, const void* const planes[3], const intptr_t strides[3], int width, int height, int depth, int ctu_size, uint8_t* payload, size_t cap);

/* FrameEncoder::encodeSlice (reference: source/encoder/frameencoder.cpp:1298-1370): the final CABAC pass over a decided picture -> sub-streams
 * (one per CTU row when si->wpp, else one).  sao / sao_flags (may be NULL): the SAO parameters of every CTU (reserved[0] = merge mode: 0 none,
 * 1 left, 2 up) and slice_sao_luma_flag / slice_sao_chroma_flag; their syntax precedes each CTU. */
int x265amd_encode_slice_data(const x265amd_slice_info* si, x265amd_cu_unit* units, const int16_t* coeffs, const x265amd_sao_ctu* sao,
                              const int32_t* sao_flags, uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams);
/* SAO parameter decision of a picture (host): SAO::startSlice + rdoSaoUnitCu for every CTU + rdoSaoUnitRowEnd (reference:
 * source/encoder/sao.cpp:227-272, :1207-1761) on the statistics of x265amd_sao_stats (host copies, same layout).  referenced: IS_REFERENCED(frame);
 * frame_threads: param.frameNumThreads (the automatic switch-off of startSlice only runs when 1); qp_min / qp_max: param.rc.qpMin / qpMax.
 * depth_sao_rate: 8 doubles kept across the pictures of an encode (SAO::m_depthSaoRate), zero at the start.  params: per CTU (reserved[0] =
 * merge mode); sao_flags[2]: slice_sao_luma_flag, slice_sao_chroma_flag.  limit-sao and sao-non-deblock are not supported. */
int x265amd_sao_rdo(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                    const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags);
/* CTU rows ctu_row_begin .. ctu_row_end - 1 of the same decision (rows in order; frame_threads > 1 unless the call covers the picture) */
int x265amd_sao_rdo_rows(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                         const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags,
                         int ctu_row_begin, int ctu_row_end);
/* the CTUs ctu_col_begin .. ctu_col_end - 1 of ONE row (frame_threads > 1): `carry` (X265AMD_CTX_STRIDE + 8 bytes, the caller's) takes the row's entropy state
 * from one call to the next; a call that starts at column 0 starts the row.  Columns in order. */
int x265amd_sao_rdo_cols(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                         const int32_t* count, const int32_t* offset_org, x265amd_sao_ctu* params, int32_t* sao_flags,
                         int ctu_row, int ctu_col_begin, int ctu_col_end, uint8_t* carry);

/* --- lookahead lowres pipeline, first stage.
 * x265amd_lowres_init = Lowres::init (reference: source/common/lowres.cpp:337-403): frame_init_lowres_core (source/common/pixel.cpp:605-628) from the
 * padded full-resolution luma plane (d_src = its sample (0,0); the filter reads one sample beyond the right and bottom edge, i.e. the margin)
 * into the four lowres planes (d_planes[k] = sample (0,0) of fpel, H, V, C), then extendPicBorder of each.  width / height: of the lowres picture.
 * x265amd_lowres_intra_costs = LookaheadTLD::lowresIntraEstimate (source/encoder/slicetype.cpp:715-824), per 8x8 block of the lowres picture:
 * d_cost[cuXY] = Lowres::intraCost (best SATD + 5 * lambda + 4), d_mode[cuXY] = Lowres::intraMode; lambda = (int)x265_lambda_tab[X265_LOOKAHEAD_QP], X265_LOOKAHEAD_QP = 12 + 6 * (bit depth - 8) (common/common.h:213): 1 for 8-bit, 16 for 10-bit.
 * The frame sums (costEst, costEstAq, rowSatds, lowresCosts) are the caller's reduction over these.  Asynchronous. */
int x265amd_lowres_init(void* stream, const x265amd_pixel* d_src, intptr_t src_stride, int width, int height, x265amd_pixel* const d_planes[4],
                        intptr_t stride, int marginX, int marginY);
int x265amd_lowres_intra_costs(void* stream, const x265amd_pixel* d_plane, intptr_t stride, int width_in_cu, int height_in_cu, int lambda,
                               int32_t* d_cost, uint8_t* d_mode);

/* x265amd_lowres_frame_cost = the block loop of CostEstimateGroup::estimateFrameCost (reference: source/encoder/slicetype.cpp:4050-4059) with
 * estimateCUCost (:4077-4249) for every 8x8 block of the lowres picture against reference p0 (d_ref0: fpel, H, V, C planes) and, for a B candidate,
 * p1 (d_ref1, else NULL): neighbour MV predictors by SATD, MotionEstimate::motionEstimate in its lowres form (hexagon, merange 16, subme 1 on the four
 * half-pel planes; source/encoder/motion.cpp), the bi-prediction and co-located averages, the intra alternative for P.  do_search: the MVs of that list are
 * still unknown (Lowres::lowresMvs[..][0].x == 0x7FFF); otherwise d_mvs / d_mv_costs are read.  Outputs per block: d_mvs (quarter-pel x, y),
 * d_mv_costs, d_lowres_costs (Lowres::lowresCosts: min(cost, 0x3FFF) | lists used << 14), d_bcost (the unclipped cost); the frame sums (costEst,
 * intraMbs, rowSatds) are the caller's reduction.  d_progress: height_in_cu ints of scratch.  No AQ, weighted prediction, HME or cooperative slices.
 * Asynchronous after a short synchronous set-up. */
int x265amd_lowres_frame_cost(void* stream, x265amd_me_ctx* me, const x265amd_pixel* d_fenc, const x265amd_pixel* const d_ref0[4],
                              const x265amd_pixel* const d_ref1[4], intptr_t stride, int width_in_cu, int height_in_cu, int do_search0, int do_search1,
                              const int32_t* d_intra_cost, int16_t* d_mvs0, int32_t* d_mv_costs0, int16_t* d_mvs1, int32_t* d_mv_costs1,
                              uint16_t* d_lowres_costs, int32_t* d_bcost, int32_t* d_progress);

/* Many estimates of pictures of one size as ONE launch (the lookahead's batches, slicetype.cpp:2668-2734: every picture of the window against the pictures up to
 * bframes + 1 before and behind it): job i is what x265amd_lowres_frame_cost takes for one estimate (d_ref1[0] == NULL: a P estimate).  Estimates of one call must not
 * search the same motion field.  Returns when all of them are done. */
typedef struct x265amd_lowres_cost_job
{
    const x265amd_pixel* d_fenc; const x265amd_pixel* d_ref0[4]; const x265amd_pixel* d_ref1[4];
    const int32_t* d_intra_cost;
    int16_t* d_mvs0; int32_t* d_mv_costs0; int16_t* d_mvs1; int32_t* d_mv_costs1;
    uint16_t* d_lowres_costs; int32_t* d_bcost;
    int32_t do_search0, do_search1;
    int32_t rows_per_slice, num_slices;     /* num_slices > 1: the estimate in cooperative slices (the reference's estimates outside its batches, param.lookaheadSlices:
                                             * Lookahead::m_numRowsPerSlice / m_numCoopSlices, slicetype.cpp:1047-1054): block rows [k rows_per_slice, (k + 1) rows_per_slice), the
                                             * last slice to the bottom, are searched independently -- a slice's bottom row takes no predictors from below.  0 / 0: one chain */
    const x265amd_pixel* d_ref0w[4];        /* weighted copies of d_ref0's planes (LookaheadTLD::weightsAnalyse chose a weight): list 0's motion search reads these, everything
                                             * else d_ref0 (estimateCUCost: wfref0 for the search, fref0 for the bi-directional candidates).  [0] NULL: none */
} x265amd_lowres_cost_job;
int x265amd_lowres_frame_cost_batch(void* stream, x265amd_me_ctx* me, const x265amd_lowres_cost_job* jobs, int n, intptr_t stride, int width_in_cu, int height_in_cu);
/* The caller's sums over the blocks of finished estimates (estimateCUCost's tail and estimateFrameCost, slicetype.cpp:4220-4248, :4062-4067), made on the device so that only
 * two numbers per estimate come back: sums[2 i] = the sum of jobs[i].d_bcost over the blocks that are not on the picture's edge (every block when the picture is two blocks or
 * less wide or high), sums[2 i + 1] = how many of those blocks have the list field of d_lowres_costs (bits 14-15) zero -- the intra blocks.  Only d_bcost and d_lowres_costs
 * of a job are looked at.  sums: HOST memory, 2 n values; returns when they are there. */
int x265amd_lowres_cost_sums(void* stream, const x265amd_lowres_cost_job* jobs, int n, int width_in_cu, int height_in_cu, int64_t* sums);

/* Weighted prediction, the measurements of the analysis (reference: LookaheadTLD::weightCostLuma, source/encoder/slicetype.cpp:826-858; weightCost's luma branch and mcLuma,
 * source/encoder/weightPrediction.cpp:58-90, :172-218): costs[i] = the sum over the 8x8 blocks of the lowres picture (width x height samples, width a multiple of 8; the last block
 * row reads into the planes' margin as the reference does) of min(SATD(source block, reference block weighted by candidate i), d_intra_cost[block]) -- d_intra_cost NULL: the SATD
 * alone.  d_ref: the four lowres planes of the reference (fpel, H, V, C; only [0] is read when d_mvs is NULL); d_mvs: Lowres::lowresMvs of the pair (x, y int16 per block) -- the
 * reference block is motion compensated with the block's vector clipped to the picture + 8 samples (Lowres::lowresMC), NULL: the co-located block.  A candidate: present 0 = no
 * weighting; else weight_pp_c's arguments (pixel.cpp:519-538) as the callers pass them: w0 = inputWeight, round = (denom ? 1 << (denom - 1) : 0) << (14 - depth), shift = denom +
 * 14 - depth, offset = inputOffset << (depth - 8).  Returns when the sums are in `costs` (host memory). */
typedef struct x265amd_weight_cand { int32_t present, w0, round, shift, offset; } x265amd_weight_cand;
int x265amd_lowres_weight_costs(void* stream, const x265amd_pixel* d_fenc, const x265amd_pixel* const d_ref[4], const int16_t* d_mvs, const int32_t* d_intra_cost,
                                intptr_t stride, int width, int height, const x265amd_weight_cand* cands, int n, uint32_t* costs);
/* Many decisions' measurements as one launch (the lookahead's weightsAnalyse of every motion search of a batch, slicetype.cpp:860-990): job i = one picture pair with its two
 * candidates (cands[0]: as a rule "no weighting", cands[1]: the guess); costs[2 i], costs[2 i + 1] = what x265amd_lowres_weight_costs returns for them.  costs: HOST memory. */
typedef struct x265amd_weight_cost_job
{
    const x265amd_pixel* d_fenc; const x265amd_pixel* d_ref[4]; const int16_t* d_mvs; const int32_t* d_intra_cost;
    x265amd_weight_cand cands[2];
} x265amd_weight_cost_job;
int x265amd_lowres_weight_costs_many(void* stream, const x265amd_weight_cost_job* jobs, int n, intptr_t stride, int width, int height, uint32_t* costs);
/* weightAnalyse's chroma planes (weightPrediction.cpp:348-375 with mcChroma :93-159 and weightCost's 4:2:0 chroma branch :205-208): costs[i] = the sum over the 8x8 blocks of the
 * chroma plane (width x height: the plane clamped to whole 16x16 luma blocks) of SATD(source block, reference block weighted by candidate i); the reference block motion
 * compensated with the lookahead's field d_mvs (NULL: not) exactly as mcChroma does it.  d_fenc / d_ref: sample (0, 0) of the two pictures' SOURCE chroma planes. */
int x265amd_chroma_weight_costs(void* stream, const x265amd_pixel* d_fenc, const x265amd_pixel* d_ref, const int16_t* d_mvs, intptr_t stride, int width, int height,
                                int low_cu_w, int low_cu_h, const x265amd_weight_cand* cands, int n, uint32_t* costs);
/* weight_pp_c over `count` samples of a buffer (the weighted copies of a reference's four lowres planes, margins included: slicetype.cpp:971-975).  Asynchronous. */
int x265amd_weight_buffer(void* stream, const x265amd_pixel* d_src, x265amd_pixel* d_dst, size_t count, int w0, int round, int shift, int offset);

/* x265amd_aq_energy = LookaheadTLD::acEnergyCu for every quantisation group of a source picture (reference: source/encoder/slicetype.cpp:48-92, :264-283):
 * d_energy[group] (raster order, ceil(width / qg) groups per row; qg_size 16 or 8) = AC energy of the luma block + the two 4:2:0 chroma blocks;
 * d_wp[0..2] = Lowres::wp_sum[plane], d_wp[3..5] = wp_ssd[plane] (sums over all groups).  planes: HOST array of the device addresses of sample (0,0)
 * of Y, U, V (padded planes: groups at the right / bottom edge read into the margin, as the reference does).  The double-precision part of
 * calcAdaptiveQuantFrame (energies -> QP offsets) is x265amd_aq_offsets below.  Asynchronous. */
int x265amd_aq_energy(void* stream, const uint64_t planes[3], intptr_t stride, intptr_t cstride, int width, int height, int qg_size,
                      uint32_t* d_energy, uint64_t* d_wp);

/* x265amd_aq_offsets = the rest of LookaheadTLD::calcAdaptiveQuantFrame (reference: source/encoder/slicetype.cpp:513-640), host code: the energies of
 * x265amd_aq_energy -> Lowres::qpAqOffset, qpCuTreeOffset (doubles) and invQscaleFactor (x265_exp2fix8) per quantisation group, for aq_mode 1 (variance),
 * 2 (auto-variance), 3 (auto-variance biased); aq_strength / aq_bias_strength = param.rc.aqStrength / aqBiasStrength.  No HDR10 offsets, external quant
 * offsets, hevc-aq or edge modes.  num_blocks: the groups of x265amd_aq_energy; avg_block_count: the count the reference averages over (lowres widthInCU x
 * heightInCU, x 4 for qg 8; equal to num_blocks when the picture size is a multiple of 16).  Returns X265AMD_OK or X265AMD_EINVAL. */
int x265amd_aq_offsets(const uint32_t* energy, int num_blocks, int avg_block_count, int aq_mode, double aq_strength, double aq_bias_strength, int qg_size,
                       double* qp_aq_offset, double* qp_cutree_offset, int32_t* inv_qscale_factor);

/* returns the device scratch the host orchestrators keep between calls (a size-class pool) to the HIP runtime */
void x265amd_release_scratch(void);
/* X265AMD_HOSTPROF=1: prints (stderr) the host CPU time by named scope collected so far (development aid) */
void x265amd_hostprof_report(void);

/* Device job queues (csrc/xa_queue.h): the CTU rows of x265amd_analyse_frame run their block operations as commands to resident workgroups instead of
 * kernel launches (environment: X265AMD_QUEUES = number of queues, default 128, 0 = launches on HIP streams as before).  The self test pushes `rounds`
 * rounds of copies, fills and rectangle copies through `numQueues` queues from as many host threads and compares every byte that comes back. */
int x265amd_queue_selftest(int rounds, int numQueues);
/* ... and what happens to a queue whose XA_OP_WAIT gives up (two seconds without the queue it follows getting there): the commands behind the wait are not run and every
 * host wait on the queue fails -- until the queue is released; its next owner starts clean (XA_CMD_RESET).  Takes about two seconds.  0: both halves behaved */
int x265amd_queue_selftest_wait_fault(void);
/* A queue handle for the `stream` argument of the orchestrating entry points (x265amd_pred_inter_search, x265amd_inter_residual_rd, x265amd_skip_rd,
 * x265amd_intra_in_inter, x265amd_check_intra, x265amd_compress_ctu_inter): they return with the queue drained.  NULL when none is free.  While a
 * queue is held a workgroup is resident on the device: release it before anything that synchronises the whole device. */
void* x265amd_queue_acquire(void);
void x265amd_queue_release(void* queue);
/* prints (stderr) what the job server's workgroups have spent per command kind so far; the counters are kept when X265AMD_QUEUE_PROF is set */
void x265amd_queue_profile_report(void);
/* Counters of the resident kernel k_job_server since the last reset -- what bench.py's roofline object is made of.  Valid while no queue is held (the
 * workgroups write them when they leave).  out[0] commands run, [1] ticks of the 100 MHz clock in command bodies, [2] in fences, [3] polling for commands,
 * [4] algorithmic bytes of all commands (what each command has to read and write, from the sizes in its job records: DESIGN.md section 5), [5] resident
 * ticks summed over the workgroups, [6] launches of k_job_server, [7] their summed duration by HIP events on its stream in microseconds, [8] workgroups per
 * launch, [9] reserved; then per command kind k (XaOp, csrc/xa_queue.h): [10 + 3 k] count, [11 + 3 k] body ticks, [12 + 3 k] algorithmic bytes.
 * n = words available in out (at least 10; 106 for everything); reset != 0 clears the counters afterwards. */
int x265amd_queue_stats(uint64_t* out, int n, int reset);

/* RDCost (reference: source/encoder/rdcost.h:34-174), 4:2:0 without chroma QP offsets: host-side integer formulas.
 * out[0..5] = lambda2 (FIX8), lambda (FIX8), psyRd, calcRdCost, calcPsyRdCost (0 when psyRd == 0), calcRdSADCost */
void x265amd_rdcost(int qp, int sliceType, double psyRdScale, uint64_t dist, uint32_t bits, uint32_t psycost, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* X265AMD_H */
