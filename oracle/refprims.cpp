/* TEST INFRASTRUCTURE ONLY.
 * Thin C-ABI driver around the REFERENCE's own primitive table (reference: source/common/primitives.h:239-433,
 * filled by setupCPrimitives(), primitives.h:472 / primitives.cpp:236-246).  It is compiled by oracle/build_ref.sh
 * against the reference objects into oracle/_ref/librefprims{8,10}.so and is used to
 *   (1) validate the CPU restatement in oracle/hevc_oracle.c, and
 *   (2) generate the golden vectors committed under tests/golden/ (tests/golden/make_golden.py).
 * This file contains no reference code: it only *calls* reference function pointers.
 * The product (x265-amod_amd/) never links, loads or calls anything from here. */
#include "common.h"
#include "primitives.h"
#include "reference.h"
#include "deblock.h"
#include "sao.h"
#include "bitstream.h"
#include "search.h"
#include "analysis.h"
#include "slicetype.h"
#include <vector>
#include "frame.h"
#include "x265.h"
#include "constants.h"
#include "lowres.h"
#include "bitcost.h"
#include "motion.h"
#include "slice.h"
#include "cudata.h"
#include "scalinglist.h"
#include "entropy.h"
#include "quant.h"
#include "rdcost.h"
#include "predict.h"
#include "framedata.h"
#include "picyuv.h"
#include "yuv.h"
#include "shortyuv.h"

using namespace X265_NS;

static EncoderPrimitives g_p;
static bool g_init = false;

static void ensure()
{
    if (!g_init)
    {
        memset(&g_p, 0, sizeof(g_p));
        setupCPrimitives(g_p);      /* C references; keeps intra_pred_allangs populated */
        setupAliasPrimitives(g_p);
        if (!primitives.pu[0].sad)  /* the global table the reference's classes (MotionEstimate ...) call through */
        {
            setupCPrimitives(primitives);
            setupAliasPrimitives(primitives);
        }
        MotionEstimate::initScales();
        g_init = true;
    }
}

extern "C" {

int ref_bit_depth(void) { return X265_DEPTH; }
int ref_sizeof_pixel(void) { return (int)sizeof(pixel); }
int ref_sizeof_sse(void) { return (int)sizeof(sse_t); }
int ref_partition_from_sizes(int w, int h) { return partitionFromSizes(w, h); }

/* ---- distortion: pixel.cpp ---- */
int ref_sad(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { ensure(); return g_p.pu[part].sad(a, sa, b, sb); }
void ref_sad_x3(int part, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{ ensure(); g_p.pu[part].sad_x3(fenc, r0, r1, r2, rs, res); }
void ref_sad_x4(int part, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{ ensure(); g_p.pu[part].sad_x4(fenc, r0, r1, r2, r3, rs, res); }
int ref_satd(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { ensure(); return g_p.pu[part].satd(a, sa, b, sb); }
int ref_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { ensure(); return g_p.cu[cu].sa8d(a, sa, b, sb); }
uint64_t ref_sse_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { ensure(); return (uint64_t)g_p.cu[cu].sse_pp(a, sa, b, sb); }
uint64_t ref_sse_ss(int cu, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { ensure(); return (uint64_t)g_p.cu[cu].sse_ss(a, sa, b, sb); }
uint64_t ref_ssd_s(int cu, const int16_t* a, intptr_t sa) { ensure(); return (uint64_t)g_p.cu[cu].ssd_s[NONALIGNED](a, sa); }
int ref_psy_cost_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { ensure(); return g_p.cu[cu].psy_cost_pp(a, sa, b, sb); }
int ref_chroma_satd(int csp, int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ ensure(); return g_p.chroma[csp].pu[part].satd ? g_p.chroma[csp].pu[part].satd(a, sa, b, sb) : -1; }
int ref_chroma_sa8d(int csp, int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ ensure(); return g_p.chroma[csp].cu[cu].sa8d(a, sa, b, sb); }
uint64_t ref_var(int cu, const pixel* a, intptr_t sa) { ensure(); return g_p.cu[cu].var(a, sa); }

/* ---- residual / recon helpers: pixel.cpp ---- */
void ref_sub_ps(int cu, int16_t* dst, intptr_t ds, const pixel* s0, const pixel* s1, intptr_t ss0, intptr_t ss1)
{ ensure(); g_p.cu[cu].sub_ps(dst, ds, s0, s1, ss0, ss1); }
void ref_add_ps(int cu, pixel* dst, intptr_t ds, const pixel* b0, const int16_t* b1, intptr_t ss0, intptr_t ss1)
{ ensure(); g_p.cu[cu].add_ps[NONALIGNED](dst, ds, b0, b1, ss0, ss1); }
void ref_pixelavg_pp(int part, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0, const pixel* s1, intptr_t ss1)
{ ensure(); g_p.pu[part].pixelavg_pp[NONALIGNED](dst, ds, s0, ss0, s1, ss1, 32); }
void ref_addAvg(int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{ ensure(); g_p.pu[part].addAvg[NONALIGNED](s0, s1, dst, ss0, ss1, ds); }
void ref_chroma_addAvg(int csp, int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{ ensure(); g_p.chroma[csp].pu[part].addAvg[NONALIGNED](s0, s1, dst, ss0, ss1, ds); }
void ref_weight_pp(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset)
{ ensure(); g_p.weight_pp(src, dst, stride, width, height, w0, round, shift, offset); }
void ref_weight_sp(const int16_t* src, pixel* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset)
{ ensure(); g_p.weight_sp(src, dst, ss, ds, width, height, w0, round, shift, offset); }
void ref_scale2D_64to32(pixel* dst, const pixel* src, intptr_t stride) { ensure(); g_p.scale2D_64to32(dst, src, stride); }
void ref_scale1D_128to64(pixel* dst, const pixel* src) { ensure(); g_p.scale1D_128to64[NONALIGNED](dst, src); }
void ref_transpose(int cu, pixel* dst, const pixel* src, intptr_t stride) { ensure(); g_p.cu[cu].transpose(dst, src, stride); }
void ref_cpy2Dto1D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t ss, int shift) { ensure(); g_p.cu[cu].cpy2Dto1D_shl(dst, src, ss, shift); }
void ref_cpy2Dto1D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t ss, int shift) { ensure(); g_p.cu[cu].cpy2Dto1D_shr(dst, src, ss, shift); }
void ref_cpy1Dto2D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t ds, int shift) { ensure(); g_p.cu[cu].cpy1Dto2D_shl[NONALIGNED](dst, src, ds, shift); }
void ref_cpy1Dto2D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t ds, int shift) { ensure(); g_p.cu[cu].cpy1Dto2D_shr(dst, src, ds, shift); }

/* ---- transforms + quant: dct.cpp ---- */
void ref_dct(int cu, const int16_t* src, int16_t* dst, intptr_t stride) { ensure(); g_p.cu[cu].dct(src, dst, stride); }
void ref_idct(int cu, const int16_t* src, int16_t* dst, intptr_t stride) { ensure(); g_p.cu[cu].idct(src, dst, stride); }
void ref_dst4x4(const int16_t* src, int16_t* dst, intptr_t stride) { ensure(); g_p.dst4x4(src, dst, stride); }
void ref_idst4x4(const int16_t* src, int16_t* dst, intptr_t stride) { ensure(); g_p.idst4x4(src, dst, stride); }
uint32_t ref_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{ ensure(); return g_p.quant(coef, quantCoeff, deltaU, qCoef, qBits, add, numCoeff); }
uint32_t ref_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{ ensure(); return g_p.nquant(coef, quantCoeff, qCoef, qBits, add, numCoeff); }
void ref_dequant_normal(const int16_t* quantCoef, int16_t* coef, int num, int scale, int shift)
{ ensure(); g_p.dequant_normal(quantCoef, coef, num, scale, shift); }
void ref_dequant_scaling(const int16_t* src, const int32_t* dequantCoef, int16_t* dst, int num, int per, int shift)
{ ensure(); g_p.dequant_scaling(src, dequantCoef, dst, num, per, shift); }
int ref_count_nonzero(int cu, const int16_t* q) { ensure(); return g_p.cu[cu].count_nonzero(q); }
uint32_t ref_copy_cnt(int cu, int16_t* coeff, const int16_t* resi, intptr_t stride) { ensure(); return g_p.cu[cu].copy_cnt(coeff, resi, stride); }

/* ---- intra: intrapred.cpp ---- */
void ref_intra_pred(int cu, int mode, pixel* dst, intptr_t ds, const pixel* srcPix, int bFilter)
{ ensure(); g_p.cu[cu].intra_pred[mode](dst, ds, srcPix, mode, bFilter); }
void ref_intra_filter(int cu, const pixel* refs, pixel* filtered) { ensure(); g_p.cu[cu].intra_filter(refs, filtered); }
void ref_intra_allangs(int cu, pixel* dst, pixel* refPix, pixel* filtPix, int bLuma)
{ ensure(); g_p.cu[cu].intra_pred_allangs(dst, refPix, filtPix, bLuma); }

/* ---- interpolation: ipfilter.cpp ---- */
void ref_luma_hpp(int part, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.pu[part].luma_hpp(src, ss, dst, ds, idx); }
void ref_luma_hps(int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx, int rowExt) { ensure(); g_p.pu[part].luma_hps(src, ss, dst, ds, idx, rowExt); }
void ref_luma_vpp(int part, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.pu[part].luma_vpp(src, ss, dst, ds, idx); }
void ref_luma_vps(int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx) { ensure(); g_p.pu[part].luma_vps(src, ss, dst, ds, idx); }
void ref_luma_vsp(int part, const int16_t* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.pu[part].luma_vsp(src, ss, dst, ds, idx); }
void ref_luma_vss(int part, const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx) { ensure(); g_p.pu[part].luma_vss(src, ss, dst, ds, idx); }
void ref_luma_hvpp(int part, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int ix, int iy) { ensure(); g_p.pu[part].luma_hvpp(src, ss, dst, ds, ix, iy); }
void ref_luma_p2s(int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds) { ensure(); g_p.pu[part].convert_p2s[NONALIGNED](src, ss, dst, ds); }
void ref_chroma_hpp(int csp, int part, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.chroma[csp].pu[part].filter_hpp(src, ss, dst, ds, idx); }
void ref_chroma_hps(int csp, int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx, int rowExt) { ensure(); g_p.chroma[csp].pu[part].filter_hps(src, ss, dst, ds, idx, rowExt); }
void ref_chroma_vpp(int csp, int part, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.chroma[csp].pu[part].filter_vpp(src, ss, dst, ds, idx); }
void ref_chroma_vps(int csp, int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx) { ensure(); g_p.chroma[csp].pu[part].filter_vps(src, ss, dst, ds, idx); }
void ref_chroma_vsp(int csp, int part, const int16_t* src, intptr_t ss, pixel* dst, intptr_t ds, int idx) { ensure(); g_p.chroma[csp].pu[part].filter_vsp(src, ss, dst, ds, idx); }
void ref_chroma_vss(int csp, int part, const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int idx) { ensure(); g_p.chroma[csp].pu[part].filter_vss(src, ss, dst, ds, idx); }
void ref_chroma_p2s(int csp, int part, const pixel* src, intptr_t ss, int16_t* dst, intptr_t ds) { ensure(); g_p.chroma[csp].pu[part].p2s[NONALIGNED](src, ss, dst, ds); }

/* ---- constant tables the path reads (constants.cpp) ---- */
const int16_t* ref_tbl_t4(void) { return &g_t4[0][0]; }
const int16_t* ref_tbl_t8(void) { return &g_t8[0][0]; }
const int16_t* ref_tbl_t16(void) { return &g_t16[0][0]; }
const int16_t* ref_tbl_t32(void) { return &g_t32[0][0]; }
const int16_t* ref_tbl_lumaFilter(void) { return &g_lumaFilter[0][0]; }
const int16_t* ref_tbl_chromaFilter(void) { return &g_chromaFilter[0][0]; }
const double* ref_tbl_lambda(void) { return x265_lambda_tab; }
const double* ref_tbl_lambda2(void) { return x265_lambda2_tab; }
const uint8_t* ref_tbl_chromaScale(void) { return g_chromaScale; }
const uint8_t* ref_tbl_intraFilterFlags(void) { return g_intraFilterFlags; }

/* ---- motion estimation: the reference's own classes (encoder/bitcost.cpp, encoder/motion.cpp) ---- */
struct BitCostProbe : public BitCost { const uint16_t* table() const { return m_cost; } };
const uint16_t* ref_mvcost_table(int qp) { static BitCostProbe bc; bc.setQP(qp); return bc.table(); }

/* setSourcePU (lookahead form, motion.cpp:193-217: luma only, blockOffset = offset) + motionEstimate (motion.cpp:764) */
int ref_motion_estimate(const pixel* fencPlane, const pixel* refPlane, intptr_t stride, int puX, int puY, int w, int h,
                        int method, int subme, int qp, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                        int numCandidates, const int32_t* mvc, int merange, int32_t* outMv)
{
    ensure();
    static MotionEstimate me;           /* one searcher reused across calls (like the encoder's per-thread Search::m_me) */
    static bool meInit = false;
    if (!meInit) { me.init(X265_CSP_I420); meInit = true; }
    me.setQP(qp);
    me.setSourcePU((pixel*)fencPlane, stride, (intptr_t)puY * stride + puX, w, h, method, subme);
    ReferencePlanes ref;
    ref.fpelPlane[0] = (pixel*)refPlane;
    ref.lumaStride = stride;
    MV cand[16];
    for (int i = 0; i < numCandidates && i < 16; i++) cand[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out;
    int cost = me.motionEstimate(&ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), MV(qmvp[0], qmvp[1]), numCandidates, cand, merange, out, 1, NULL);
    outMv[0] = out.x; outMv[1] = out.y;
    return cost;
}

/* setSourcePU (encoder form, motion.cpp:219-247: PU copied from a CU Yuv, chroma SATD when subpelRefine > 2 and the
 * chroma PU is a multiple of 4x4) + motionEstimate.  planes: sample (0,0) of Y,U,V of the source and of the reference. */
int ref_motion_estimate_c(const pixel* const* fencPl, const pixel* const* refPl, intptr_t stride, intptr_t cstride, int puX, int puY, int w, int h,
                          int method, int subme, int qp, const int32_t* mvmin, const int32_t* mvmax, const int32_t* qmvp,
                          int numCandidates, const int32_t* mvc, int merange, int bChroma, int32_t* outMv)
{
    ensure();
    static MotionEstimate me;
    static bool meInit = false;
    static Yuv fencYuv;
    static PicYuv* pic = NULL;
    static intptr_t zeroCu[1] = { 0 }, buY[256], buC[256];
    if (!meInit) { me.init(X265_CSP_I420); fencYuv.create(64, X265_CSP_I420); pic = new PicYuv; meInit = true; }
    me.setQP(qp);
    for (int y = 0; y < h; y++) memcpy(fencYuv.m_buf[0] + y * fencYuv.m_size, fencPl[0] + (intptr_t)(puY + y) * stride + puX, w * sizeof(pixel));
    for (int c = 1; c < 3; c++)
        for (int y = 0; y < h / 2; y++) memcpy(fencYuv.m_buf[c] + y * fencYuv.m_csize, fencPl[c] + (intptr_t)(puY / 2 + y) * cstride + puX / 2, (w / 2) * sizeof(pixel));
    me.setSourcePU(fencYuv, 0, 0, 0, w, h, method, subme, bChroma != 0);
    memset(buY, 0, sizeof(buY)); memset(buC, 0, sizeof(buC));
    buY[0] = (intptr_t)puY * stride + puX; buC[0] = (intptr_t)(puY / 2) * cstride + puX / 2;
    pic->m_picOrg[0] = (pixel*)refPl[0]; pic->m_picOrg[1] = (pixel*)refPl[1]; pic->m_picOrg[2] = (pixel*)refPl[2];
    pic->m_stride = stride; pic->m_strideC = cstride;
    pic->m_cuOffsetY = zeroCu; pic->m_cuOffsetC = zeroCu; pic->m_buOffsetY = buY; pic->m_buOffsetC = buC;
    ReferencePlanes ref;
    ref.fpelPlane[0] = (pixel*)refPl[0]; ref.fpelPlane[1] = (pixel*)refPl[1]; ref.fpelPlane[2] = (pixel*)refPl[2];
    ref.lumaStride = stride; ref.chromaStride = cstride; ref.reconPic = pic;
    MV cand[16];
    for (int i = 0; i < numCandidates && i < 16; i++) cand[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out;
    int cost = me.motionEstimate(&ref, MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]), MV(qmvp[0], qmvp[1]), numCandidates, cand, merange, out, 1, NULL);
    outMv[0] = out.x; outMv[1] = out.y;
    return cost;
}

/* batch form over the packed job records of include/x265amd.h (struct x265amd_me_job, 72 bytes) -- used by bench.py's
 * cpu_baseline leg so that the timed loop is the reference's C code, not Python call overhead */
struct PackedMeJob { int16_t x, y; uint8_t w, h, method, subme, qp, num_cand; int16_t merange, mvmin[2], mvmax[2], mvp[2], mvc[12][2]; };
int ref_motion_estimate_batch(const pixel* fencPlane, const pixel* refPlane, intptr_t stride, const PackedMeJob* jobs, int n, int32_t* out)
{
    for (int i = 0; i < n; i++)
    {
        const PackedMeJob& j = jobs[i];
        int32_t mn[2] = { j.mvmin[0], j.mvmin[1] }, mx[2] = { j.mvmax[0], j.mvmax[1] }, mvp[2] = { j.mvp[0], j.mvp[1] }, mvc[24], mv[2];
        for (int k = 0; k < j.num_cand; k++) { mvc[2 * k] = j.mvc[k][0]; mvc[2 * k + 1] = j.mvc[k][1]; }
        out[3 * i + 2] = ref_motion_estimate(fencPlane, refPlane, stride, j.x, j.y, j.w, j.h, j.method, j.subme, j.qp, mn, mx, mvp, j.num_cand, mvc, j.merange, mv);
        out[3 * i] = mv[0]; out[3 * i + 1] = mv[1];
    }
    return n;
}

/* ---- residual path: the reference's own Quant class (common/quant.cpp:397-605) ---- */
const uint16_t* ref_tbl_scan(int scanType, int log2TrSize) { return g_scanOrder[scanType][log2TrSize - 2]; }

struct QuantProbe : public Quant
{
    void configure(int ttype, int qpScaled) { m_qpParam[ttype].setQpParam(qpScaled); m_rdoqLevel = 0; m_nr = NULL; }
    void configureRdoq(int level, int psyRdoqScale) { m_rdoqLevel = level; m_psyRdoqScale = psyRdoqScale; }
};

struct TuEnv
{
    ScalingList sl;
    Entropy entropy;
    QuantProbe quant;
    CUData cu;
    Slice slice;
    SPS sps;
    PPS pps;
    uint8_t predMode[256], lumaDir[256], chromaDir[256], tqBypass[256], tuDepth[256];
    TuEnv()
    {
        sl.init();
        sl.m_bEnabled = false;                      /* --scaling-list off, as Encoder::create leaves it (flat lists) */
        sl.m_bDataPresent = false;
        sl.setupQuantMatrices(X265_CSP_I420);
        quant.init(0.0, sl, entropy);
        memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
        sps.quadtreeTULog2MaxSize = 5;
        slice.m_sps = &sps; slice.m_pps = &pps;
        cu.m_slice = &slice;
        cu.m_chromaFormat = X265_CSP_I420; cu.m_hChromaShift = 1; cu.m_vChromaShift = 1;
        cu.m_predMode = predMode; cu.m_lumaIntraDir = lumaDir; cu.m_chromaIntraDir = chromaDir; cu.m_tqBypass = tqBypass;
        memset(tqBypass, 0, sizeof(tqBypass));
        cu.m_tuDepth = tuDepth; memset(tuDepth, 0, sizeof(tuDepth));
    }
    void set(int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide)
    {
        memset(predMode, bIntra ? MODE_INTRA : MODE_INTER, sizeof(predMode));
        memset(lumaDir, dirMode, sizeof(lumaDir)); memset(chromaDir, dirMode, sizeof(chromaDir));
        slice.m_sliceType = (SliceType)sliceType;
        pps.bSignHideEnabled = signHide != 0;
        quant.configure(ttype, qpScaled);
    }
};
static TuEnv* tuEnv() { ensure(); static TuEnv* e = new TuEnv; return e; }

/* Quant::transformNxN (quant.cpp:397-480): qpScaled = qp + QP_BD_OFFSET after the chroma mapping of setQPforQuant;
 * sliceType: 0 B, 1 P, 2 I; dirMode: intra direction used to choose the coefficient scan */
uint32_t ref_transform_tu(const pixel* fenc, intptr_t fencStride, const int16_t* resi, intptr_t resiStride, int16_t* coeff, int log2TrSize,
                          int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide)
{
    TuEnv* e = tuEnv();
    e->set(ttype, bIntra, dirMode, sliceType, qpScaled, signHide);
    return e->quant.transformNxN(e->cu, fenc, (uint32_t)fencStride, resi, (uint32_t)resiStride, coeff, log2TrSize, (TextType)ttype, 0, false);
}

/* ---- entropy-side pieces of the residual path ---- */
void ref_ctx_tables(uint32_t* bits128, uint8_t* next256)
{
    memcpy(bits128, g_entropyBits, 128 * sizeof(uint32_t));
    memcpy(next256, g_nextState, 256);
}
/* Entropy::resetEntropy (entropy.cpp:1321-1355) */
void ref_entropy_reset(int sliceType, int qp, uint8_t* ctx)
{
    TuEnv* e = tuEnv();
    e->slice.m_sliceType = (SliceType)sliceType;
    e->slice.m_sliceQp = qp;
    e->entropy.resetEntropy(e->slice);
    memcpy(ctx, e->entropy.m_contextState, MAX_OFF_CTX_MOD);
}
/* Entropy::estBit (entropy.cpp:2220-2390); est: 184 ints in EstBitsSbac member order, untouched entries stay as passed */
void ref_est_bit(const uint8_t* ctx, int log2TrSize, int isLuma, int* est)
{
    TuEnv* e = tuEnv();
    memcpy(e->entropy.m_contextState, ctx, MAX_OFF_CTX_MOD);
    EstBitsSbac eb;
    memcpy(&eb, est, sizeof(eb));
    e->entropy.estBit(eb, log2TrSize, isLuma != 0);
    memcpy(est, &eb, sizeof(eb));
}
int ref_est_bits_ints(void) { return (int)(sizeof(EstBitsSbac) / sizeof(int)); }
/* Quant::transformNxN with RDOQ (quant.cpp:397-480 -> rdoQuant :609-1424); est as for ref_est_bit; psyRdoqScale = Quant::m_psyRdoqScale */
uint32_t ref_transform_tu_rdoq(const pixel* fenc, intptr_t fencStride, const int16_t* resi, intptr_t resiStride, int16_t* coeff, int log2TrSize,
                               int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide, int tuDepth, int rdoqLevel,
                               int psyRdoqScale, const int* est)
{
    TuEnv* e = tuEnv();
    e->set(ttype, bIntra, dirMode, sliceType, qpScaled, signHide);
    e->quant.configureRdoq(rdoqLevel, psyRdoqScale);
    memset(e->tuDepth, tuDepth, sizeof(e->tuDepth));
    memcpy(&e->entropy.m_estBitsSbac, est, sizeof(EstBitsSbac));
    uint32_t r = e->quant.transformNxN(e->cu, fenc, (uint32_t)fencStride, resi, (uint32_t)resiStride, coeff, log2TrSize, (TextType)ttype, 0, false);
    e->quant.configureRdoq(0, 0);
    memset(e->tuDepth, 0, sizeof(e->tuDepth));
    return r;
}
/* Entropy::codeCoeffNxN in bit-counting mode (entropy.cpp:1828-2200): returns the FIX15 bits added, updates ctx */
uint64_t ref_code_coeff_bits(const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int dirMode, int signHide, uint8_t* ctx)
{
    TuEnv* e = tuEnv();
    e->set(ttype, bIntra, dirMode, 1, 30, signHide);
    ALIGN_VAR_32(int16_t, c[32 * 32]);
    memcpy(c, coeff, sizeof(int16_t) << (2 * log2TrSize));
    memcpy(e->entropy.m_contextState, ctx, MAX_OFF_CTX_MOD);
    e->entropy.zeroFract();
    e->entropy.resetBits();
    e->entropy.codeCoeffNxN(e->cu, c, 0, log2TrSize, (TextType)ttype);
    memcpy(ctx, e->entropy.m_contextState, MAX_OFF_CTX_MOD);
    return e->entropy.m_fracBits;
}

struct PackedCoeffBitsJob { uint64_t coeff, ctxIn, ctxOut; uint8_t log2, ttype, intra, dir, signhide, reserved[3]; };
int ref_coeff_bits_batch(const PackedCoeffBitsJob* jobs, int n, uint64_t* bits)
{
    TuEnv* e = tuEnv();
    for (int i = 0; i < n; i++)
    {
        const PackedCoeffBitsJob& j = jobs[i];
        const int16_t* c = (const int16_t*)j.coeff;
        int nz = 0;
        for (int k = 0; k < (1 << (2 * j.log2)); k++) nz |= c[k];
        memcpy((void*)j.ctxOut, (const void*)j.ctxIn, 160);
        bits[i] = 0;
        if (!nz) continue;              /* the encoder only codes TUs with cbf != 0 */
        e->set(j.ttype, j.intra, j.dir, 1, 30, j.signhide);
        memcpy(e->entropy.m_contextState, (const void*)j.ctxIn, MAX_OFF_CTX_MOD);
        e->entropy.zeroFract();
        e->entropy.resetBits();
        e->entropy.codeCoeffNxN(e->cu, c, 0, j.log2, (TextType)j.ttype);
        memcpy((void*)j.ctxOut, e->entropy.m_contextState, MAX_OFF_CTX_MOD);
        bits[i] = e->entropy.m_fracBits;
    }
    return n;
}

/* Quant::invtransformNxN (quant.cpp:543-605) */
void ref_invtransform_tu(int16_t* resi, intptr_t resiStride, const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int qpScaled, uint32_t numSig)
{
    TuEnv* e = tuEnv();
    e->set(ttype, bIntra, 0, 1, qpScaled, 0);
    e->quant.invtransformNxN(e->cu, resi, (uint32_t)resiStride, coeff, log2TrSize, (TextType)ttype, bIntra != 0, false, numSig);
}

/* RDCost (encoder/rdcost.h:34-174) for a 4:2:0 slice without chroma QP offsets */
void ref_rdcost(int qp, int sliceType, double psyRdScale, uint64_t dist, uint32_t bits, uint32_t psycost, uint64_t* out /* lambda2, lambda, psyRd, rd, psyrd, rdsad */)
{
    TuEnv* e = tuEnv();
    e->sps.chromaFormatIdc = X265_CSP_I420;
    e->slice.m_sliceType = (SliceType)sliceType;
    RDCost rd;
    rd.setPsyRdScale(psyRdScale);
    rd.setSsimRd(0);
    rd.setQP(e->slice, qp);
    out[0] = rd.m_lambda2; out[1] = rd.m_lambda; out[2] = rd.m_psyRd;
    out[3] = rd.calcRdCost((sse_t)dist, bits);
    out[4] = rd.m_psyRd ? rd.calcPsyRdCost((sse_t)dist, bits, psycost) : 0;
    out[5] = rd.calcRdSADCost((uint32_t)dist, bits);
}

/* ---- intra: the reference's own Predict::initAdiPattern (common/predict.cpp:600-649, fillReferenceSamples :736-877) ---- */
struct IntraEnv
{
    Predict pred;
    FrameData fd;
    PicYuv* pic;
    intptr_t zeroCu[1];
    intptr_t zeroBu[256];
    IntraEnv()
    {
        pic = new PicYuv;       /* never destroyed: it does not own the sample buffer */
        zeroCu[0] = 0; memset(zeroBu, 0, sizeof(zeroBu));
        pic->m_cuOffsetY = zeroCu; pic->m_buOffsetY = zeroBu; pic->m_cuOffsetC = zeroCu; pic->m_buOffsetC = zeroBu;
        fd.m_reconPic = pic;
        pred.allocBuffers(X265_CSP_I420);       /* sets Predict::m_csp (read by the chroma paths) */
    }
};

/* recon: top-left sample of the block inside a reconstructed plane; flags[totalUnits]: bNeighborFlags in the reference's
 * order (below-left ... left, above-left, above ... above-right), 4-sample units.  dirMode -1 = ALL_IDX. */
void ref_init_adi_pattern(const pixel* recon, intptr_t stride, int log2TrSize, const uint8_t* flags, int strongSmoothing, int dirMode,
                          pixel* outRef, pixel* outFlt)
{
    static IntraEnv* ie = NULL;
    TuEnv* e = tuEnv();
    if (!ie) ie = new IntraEnv;
    int units = (1 << log2TrSize) >> 2;
    Predict::IntraNeighbors nb;
    nb.aboveUnits = 2 * units; nb.leftUnits = 2 * units; nb.totalUnits = 4 * units + 1;
    nb.unitWidth = 4; nb.unitHeight = 4; nb.log2TrSize = log2TrSize;
    nb.numIntraNeighbor = 0;
    for (int i = 0; i < nb.totalUnits; i++) { nb.bNeighborFlags[i] = flags[i] != 0; nb.numIntraNeighbor += flags[i] != 0; }
    ie->pic->m_picOrg[0] = (pixel*)recon;
    ie->pic->m_stride = stride;
    e->sps.bUseStrongIntraSmoothing = strongSmoothing != 0;
    e->cu.m_encData = &ie->fd;
    e->cu.m_cuAddr = 0;
    CUGeom g; memset(&g, 0, sizeof(g));
    ie->pred.initAdiPattern(e->cu, g, 0, nb, dirMode);
    memcpy(outRef, ie->pred.intraNeighbourBuf[0], 258 * sizeof(pixel));
    memcpy(outFlt, ie->pred.intraNeighbourBuf[1], 258 * sizeof(pixel));
}

/* one intra prediction as the TU coding loops make it: Predict::initAdiPattern(dirMode) + predIntraLumaAng (predict.cpp:579-588,
 * :600-622) for luma, initAdiPatternChroma + predIntraChromaAng (predict.cpp:590-598, :624-649) for 4:2:0 chroma */
void ref_intra_predict(const pixel* recon, intptr_t stride, int log2TrSize, const uint8_t* flags, int strongSmoothing, int isChroma, int mode,
                       pixel* pred, intptr_t predStride)
{
    static IntraEnv* ie = NULL;
    TuEnv* e = tuEnv();
    if (!ie) ie = new IntraEnv;
    int units = (1 << log2TrSize) >> 2;
    Predict::IntraNeighbors nb;
    nb.aboveUnits = 2 * units; nb.leftUnits = 2 * units; nb.totalUnits = 4 * units + 1;
    nb.unitWidth = 4; nb.unitHeight = 4; nb.log2TrSize = log2TrSize;
    nb.numIntraNeighbor = 0;
    for (int i = 0; i < nb.totalUnits; i++) { nb.bNeighborFlags[i] = flags[i] != 0; nb.numIntraNeighbor += flags[i] != 0; }
    ie->pic->m_picOrg[0] = ie->pic->m_picOrg[1] = ie->pic->m_picOrg[2] = (pixel*)recon;
    ie->pic->m_stride = stride; ie->pic->m_strideC = stride;
    e->sps.bUseStrongIntraSmoothing = strongSmoothing != 0;
    e->cu.m_encData = &ie->fd;
    e->cu.m_cuAddr = 0;
    CUGeom g; memset(&g, 0, sizeof(g));
    if (isChroma)
    {
        ie->pred.initAdiPatternChroma(e->cu, g, 0, nb, 1);
        ie->pred.predIntraChromaAng(mode, pred, predStride, log2TrSize);
    }
    else
    {
        ie->pred.initAdiPattern(e->cu, g, 0, nb, mode);
        ie->pred.predIntraLumaAng(mode, pred, predStride, log2TrSize);
    }
}

/* the 35-mode luma scan of Search::estIntraPredQT (encoder/search.cpp:1566-1613, individual-angle path): sa8d of every
 * prediction against fenc, with the reference's own primitives and neighbour buffers */
void ref_intra_scan(const pixel* fenc, intptr_t fencStride, int log2TrSize, const pixel* refBuf, const pixel* fltBuf, int32_t* sa8d35)
{
    ensure();
    int sizeIdx = log2TrSize - 2, N = 1 << log2TrSize;
    ALIGN_VAR_32(pixel, predBuf[32 * 32]);
    g_p.cu[sizeIdx].intra_pred[DC_IDX](predBuf, N, refBuf, 0, N <= 16);
    sa8d35[DC_IDX] = g_p.cu[sizeIdx].sa8d(fenc, fencStride, predBuf, N);
    const pixel* planar = (N >= 8 && N <= 32) ? fltBuf : refBuf;
    g_p.cu[sizeIdx].intra_pred[PLANAR_IDX](predBuf, N, planar, 0, 0);
    sa8d35[PLANAR_IDX] = g_p.cu[sizeIdx].sa8d(fenc, fencStride, predBuf, N);
    for (int mode = 2; mode < 35; mode++)
    {
        int filter = !!(g_intraFilterFlags[mode] & N);
        g_p.cu[sizeIdx].intra_pred[mode](predBuf, N, filter ? fltBuf : refBuf, mode, N <= 16);
        sa8d35[mode] = g_p.cu[sizeIdx].sa8d(fenc, fencStride, predBuf, N);
    }
}

/* ---- inter prediction: the reference's own Predict::motionCompensation (common/predict.cpp:77-243) ---- */
struct PackedMcJob
{
    uint64_t dstY, dstU, dstV; int32_t dstStride, dstCStride;
    int16_t x, y, cuX, cuY; uint8_t w, h; int8_t ref0, ref1; int16_t mv0[2], mv1[2];
    uint8_t sliceType, flags;           /* flags: 1 luma, 2 chroma, 4 bUseWeightPred, 8 bUseWeightedBiPred */
    struct { int16_t w, o; uint8_t denom, present; } wp[2][3];
    uint8_t reserved[2];
};

struct McEnv
{
    Predict pred;
    FrameData fd;
    x265_param param;
    PicYuv* pic[2];
    Yuv predYuv;
    intptr_t zeroCu[1], zeroBu[256];
    int8_t refIdx[2][256];
    MV mv[2][256];
    McEnv()
    {
        pred.allocBuffers(X265_CSP_I420);
        memset(&param, 0, sizeof(param)); param.maxCUSize = 64;
        fd.m_param = &param;
        zeroCu[0] = 0; memset(zeroBu, 0, sizeof(zeroBu));
        for (int i = 0; i < 2; i++)
        {
            pic[i] = new PicYuv;
            pic[i]->m_cuOffsetY = zeroCu; pic[i]->m_buOffsetY = zeroBu; pic[i]->m_cuOffsetC = zeroCu; pic[i]->m_buOffsetC = zeroBu;
        }
        predYuv.create(64, X265_CSP_I420);
    }
};

/* planes: nref x 3 addresses of sample (0,0) (Y, U, V) of padded pictures with strides stride / cstride */
int ref_motion_compensation_batch(const uint64_t* planes, intptr_t stride, intptr_t cstride, int picW, int picH, const PackedMcJob* jobs, int n)
{
    static McEnv* m = NULL;
    TuEnv* e = tuEnv();
    if (!m) m = new McEnv;
    for (int i = 0; i < n; i++)
    {
        const PackedMcJob& j = jobs[i];
        e->sps.picWidthInLumaSamples = picW; e->sps.picHeightInLumaSamples = picH;
        e->slice.m_sliceType = j.sliceType ? P_SLICE : B_SLICE;
        e->pps.bUseWeightPred = !!(j.flags & 4); e->pps.bUseWeightedBiPred = !!(j.flags & 8);
        const int8_t refs[2] = { j.ref0, j.ref1 };
        for (int l = 0; l < 2; l++)
        {
            memset(m->refIdx[l], refs[l] >= 0 ? 0 : -1, 256);
            m->mv[l][0] = l ? MV(j.mv1[0], j.mv1[1]) : MV(j.mv0[0], j.mv0[1]);
            e->slice.m_numRefIdx[l] = 1;
            e->slice.m_refReconPicList[l][0] = m->pic[l];
            if (refs[l] >= 0)
            {
                m->pic[l]->m_picOrg[0] = (pixel*)planes[3 * refs[l] + 0] + (intptr_t)j.y * stride + j.x;
                m->pic[l]->m_picOrg[1] = (pixel*)planes[3 * refs[l] + 1] + (intptr_t)(j.y >> 1) * cstride + (j.x >> 1);
                m->pic[l]->m_picOrg[2] = (pixel*)planes[3 * refs[l] + 2] + (intptr_t)(j.y >> 1) * cstride + (j.x >> 1);
                m->pic[l]->m_stride = stride; m->pic[l]->m_strideC = cstride;
            }
            for (int c = 0; c < 3; c++)
            {
                WeightParam& w = e->slice.m_weightPredTable[l][0][c];
                w.inputWeight = j.wp[l][c].w; w.inputOffset = j.wp[l][c].o; w.log2WeightDenom = j.wp[l][c].denom; w.wtPresent = j.wp[l][c].present;
            }
        }
        e->cu.m_encData = &m->fd;
        e->cu.m_refIdx[0] = m->refIdx[0]; e->cu.m_refIdx[1] = m->refIdx[1];
        e->cu.m_mv[0] = m->mv[0]; e->cu.m_mv[1] = m->mv[1];
        e->cu.m_cuPelX = j.cuX; e->cu.m_cuPelY = j.cuY;
        char pubuf[sizeof(PredictionUnit)];
        PredictionUnit* pu = (PredictionUnit*)pubuf;
        pu->ctuAddr = 0; pu->cuAbsPartIdx = 0; pu->puAbsPartIdx = 0; pu->width = j.w; pu->height = j.h;
        m->pred.motionCompensation(e->cu, *pu, m->predYuv, !!(j.flags & 1), !!(j.flags & 2));
        if (j.flags & 1)
            for (int y = 0; y < j.h; y++) memcpy((pixel*)j.dstY + (intptr_t)y * j.dstStride, m->predYuv.m_buf[0] + y * m->predYuv.m_size, j.w * sizeof(pixel));
        if (j.flags & 2)
            for (int y = 0; y < j.h / 2; y++)
            {
                memcpy((pixel*)j.dstU + (intptr_t)y * j.dstCStride, m->predYuv.m_buf[1] + y * m->predYuv.m_csize, (j.w / 2) * sizeof(pixel));
                memcpy((pixel*)j.dstV + (intptr_t)y * j.dstCStride, m->predYuv.m_buf[2] + y * m->predYuv.m_csize, (j.w / 2) * sizeof(pixel));
            }
    }
    return n;
}

/* ---- reference-plane production: extendPicBorder (common/pixel.cpp:1044-1058) and the reference's own MotionReference
 * (encoder/reference.cpp:51-185) applied to all rows of a plane ---- */
void ref_extend_pic_border(pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    ensure();
    extendPicBorder(pic, stride, width, height, marginX, marginY);
}
void ref_weight_plane(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int marginX, int marginY, int inputWeight, int inputOffset, int log2Denom)
{
    ensure();
    x265_param param;
    memset(&param, 0, sizeof(param));
    param.maxCUSize = 64; param.maxSlices = 1; param.subpelRefine = 2; param.internalCsp = X265_CSP_I420;
    PicYuv pic;
    pic.m_param = &param;
    pic.m_picOrg[0] = (pixel*)src; pic.m_picBuf[0] = (pixel*)src - marginY * stride - marginX;
    pic.m_stride = stride; pic.m_picWidth = width; pic.m_picHeight = height;
    pic.m_lumaMarginX = marginX; pic.m_lumaMarginY = marginY; pic.m_picCsp = X265_CSP_I420;
    pic.m_hChromaShift = pic.m_vChromaShift = 1;
    WeightParam wp[3];
    memset(wp, 0, sizeof(wp));
    wp[0].inputWeight = inputWeight; wp[0].inputOffset = inputOffset; wp[0].log2WeightDenom = log2Denom; wp[0].wtPresent = 1;
    {
        MotionReference mref;
        mref.init(&pic, wp, param);
        uint32_t numRows = (height + 63) / 64;
        mref.applyWeight(numRows - 1, numRows, numRows, 0);
        for (int y = -marginY; y < height + marginY; y++)
            memcpy(dst + y * stride - marginX, mref.fpelPlane[0] + y * stride - marginX, (width + 2 * marginX) * sizeof(pixel));
    }
    pic.m_picOrg[0] = pic.m_picBuf[0] = NULL;     /* the PicYuv never owned the samples */
    pic.m_param = NULL;
}

/* ---- in-loop deblocking with the reference's own Deblock class (common/deblock.cpp) on a hand-assembled FrameData: one CUData
 * per CTU filled from raster records (one per 4x4 unit) describing the coding quad-tree, then deblockCTU for every CTU in the
 * two directions ---- */
struct RefDbUnit { uint8_t log2CU, partSize, tuDepth, intra, cbf, bypass; int8_t qp; int8_t ref[2]; uint8_t pad; int16_t mv[2][2]; };
void ref_deblock_picture(pixel* const* planes, intptr_t stride, intptr_t cstride, int width, int height, const RefDbUnit* units,
                         int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset, int bypassEnabled, int sliceType, int pass)
{
    ensure();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420;
    param->maxCUSize = 64; param->minCUSize = 8; param->maxLog2CUSize = 6; param->unitSizeDepth = 4; param->num4x4Partitions = 256;
    param->bLossless = 0; param->bDynamicRefine = 0; param->rc.bStatWrite = 0;
    SPS sps; PPS pps;
    memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
    sps.numCuInWidth = (width + 63) / 64; sps.numCuInHeight = (height + 63) / 64; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    sps.numPartitions = 256; sps.numPartInCUSize = 16; sps.chromaFormatIdc = X265_CSP_I420;
    sps.picWidthInLumaSamples = width; sps.picHeightInLumaSamples = height;
    sps.log2MinCodingBlockSize = 3; sps.log2DiffMaxMinCodingBlockSize = 3;
    pps.deblockingFilterBetaOffsetDiv2 = betaOffsetDiv2; pps.deblockingFilterTcOffsetDiv2 = tcOffsetDiv2;
    pps.chromaQpOffset[0] = cbQpOffset; pps.chromaQpOffset[1] = crQpOffset; pps.bTransquantBypassEnabled = bypassEnabled != 0;
    FrameData* fd = new FrameData;
    fd->create(*param, sps, X265_CSP_I420);
    Slice* slice = fd->m_slice;
    slice->m_sps = &sps; slice->m_pps = &pps; slice->m_param = param;
    slice->m_sliceType = sliceType ? P_SLICE : B_SLICE;
    static Frame fakeRefs[16];                      /* only their addresses are compared (deblock.cpp:202-235) */
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 16; i++) slice->m_refFrameList[l][i] = &fakeRefs[i];    /* refIdx i of either list = picture i */
    PicYuv* rec = new PicYuv;
    rec->m_param = param; rec->m_picCsp = X265_CSP_I420; rec->m_hChromaShift = rec->m_vChromaShift = 1;
    rec->m_picOrg[0] = planes[0]; rec->m_picOrg[1] = planes[1]; rec->m_picOrg[2] = planes[2];
    rec->m_stride = stride; rec->m_strideC = cstride; rec->m_picWidth = width; rec->m_picHeight = height;
    rec->createOffsets(sps);
    fd->m_reconPic = rec;
    Frame frame;
    frame.m_encData = fd; frame.m_param = param;
    const int w4 = width >> 2;
    for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
    {
        CUData& ctu = fd->m_picCTU[addr];
        ctu.initCTU(frame, addr, 30, 0, 0, 0);
        ctu.m_chromaFormat = X265_CSP_I420; ctu.m_hChromaShift = ctu.m_vChromaShift = 1;
        const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
        for (uint32_t z = 0; z < 256; z++)
        {
            const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
            if (x >= width || y >= height) { ctu.m_predMode[z] = MODE_NONE; ctu.m_cuDepth[z] = 3; ctu.m_log2CUSize[z] = 3; continue; }
            const RefDbUnit& u = units[(y >> 2) * w4 + (x >> 2)];
            ctu.m_log2CUSize[z] = u.log2CU; ctu.m_cuDepth[z] = (uint8_t)(6 - u.log2CU);
            ctu.m_partSize[z] = u.partSize; ctu.m_tuDepth[z] = u.tuDepth;
            ctu.m_predMode[z] = u.intra ? MODE_INTRA : MODE_INTER;
            ctu.m_cbf[0][z] = u.cbf ? 0xFF : 0;
            ctu.m_tqBypass[z] = u.bypass; ctu.m_qp[z] = u.qp;
            for (int l = 0; l < 2; l++) { ctu.m_refIdx[l][z] = u.ref[l]; ctu.m_mv[l][z] = MV(u.mv[l][0], u.mv[l][1]); }
        }
    }
    Deblock db;
    for (int dir = 0; dir < 2; dir++)
    {
        if (!((pass >> dir) & 1)) continue;
        for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
        {
            CUGeom geoms[CUGeom::MAX_GEOMS];
            const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
            CUData::calcCTUGeoms(X265_MIN(64, width - cx), X265_MIN(64, height - cy), 64, 8, geoms);
            db.deblockCTU(&fd->m_picCTU[addr], geoms[0], dir);
        }
    }
    rec->m_picOrg[0] = rec->m_picOrg[1] = rec->m_picOrg[2] = NULL;
    X265_FREE(rec->m_cuOffsetY); X265_FREE(rec->m_cuOffsetC); X265_FREE(rec->m_buOffsetY); X265_FREE(rec->m_buOffsetC);
    rec->m_cuOffsetY = rec->m_cuOffsetC = rec->m_buOffsetY = rec->m_buOffsetC = NULL;
    delete rec;
    fd->m_reconPic = NULL;
    fd->destroy();
    delete fd;
    x265_param_free(param);
}

/* ---- sample adaptive offset with the reference's own SAO class (encoder/sao.cpp): statistics of every CTU (calcSaoStatsCTU) and
 * application of given per-CTU parameters (generateLumaOffsets / generateChromaOffsets, driven row by row like
 * FrameFilter::ParallelFilter::processSaoCTU with copySaoAboveRef, encoder/framefilter.cpp:300-343) ---- */
struct SaoProbe : public SAO
{
    const int32_t* cnt(int plane) const { return &m_count[plane][0][0]; }
    const int32_t* org(int plane) const { return &m_offsetOrg[plane][0][0]; }
    void clear() { memset(m_count, 0, sizeof(m_count)); memset(m_offsetOrg, 0, sizeof(m_offsetOrg)); }
    pixel* tmpU(int plane) { return m_tmpU[plane]; }
};
struct SaoFixture
{
    x265_param* param; SPS sps; PPS pps; FrameData* fd; PicYuv* rec; PicYuv* fenc; Frame frame; SaoProbe sao;
    SaoFixture(pixel* const* recPlanes, pixel* const* fencPlanes, intptr_t stride, intptr_t cstride, int width, int height)
    {
        param = x265_param_alloc();
        x265_param_default(param);
        param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420;
        param->maxCUSize = 64; param->minCUSize = 8; param->maxLog2CUSize = 6; param->unitSizeDepth = 4; param->num4x4Partitions = 256;
        param->bSaoNonDeblocked = 0; param->bLimitSAO = 0;
        memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
        sps.numCuInWidth = (width + 63) / 64; sps.numCuInHeight = (height + 63) / 64; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
        sps.numPartitions = 256; sps.numPartInCUSize = 16; sps.chromaFormatIdc = X265_CSP_I420;
        fd = new FrameData;
        fd->create(*param, sps, X265_CSP_I420);
        fd->m_slice->m_sps = &sps; fd->m_slice->m_pps = &pps; fd->m_slice->m_param = param; fd->m_slice->m_sliceType = P_SLICE;
        rec = mk(recPlanes, stride, cstride, width, height);
        fenc = mk(fencPlanes, stride, cstride, width, height);
        fd->m_reconPic = rec;
        frame.m_encData = fd; frame.m_param = param; frame.m_reconPic = rec; frame.m_fencPic = fenc;
        for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
        {
            uint32_t row = addr / sps.numCuInWidth;
            fd->m_picCTU[addr].initCTU(frame, addr, 30, row == 0, row == sps.numCuInHeight - 1, 0);
        }
        sao.create(param, 1);
        sao.m_frame = &frame;
    }
    PicYuv* mk(pixel* const* planes, intptr_t stride, intptr_t cstride, int width, int height)
    {
        PicYuv* p = new PicYuv;
        p->m_param = param; p->m_picCsp = X265_CSP_I420; p->m_hChromaShift = p->m_vChromaShift = 1;
        p->m_picOrg[0] = planes[0]; p->m_picOrg[1] = planes[1]; p->m_picOrg[2] = planes[2];
        p->m_stride = stride; p->m_strideC = cstride; p->m_picWidth = width; p->m_picHeight = height;
        p->createOffsets(sps);
        return p;
    }
    void drop(PicYuv* p)
    {
        p->m_picOrg[0] = p->m_picOrg[1] = p->m_picOrg[2] = NULL;
        X265_FREE(p->m_cuOffsetY); X265_FREE(p->m_cuOffsetC); X265_FREE(p->m_buOffsetY); X265_FREE(p->m_buOffsetC);
        p->m_cuOffsetY = p->m_cuOffsetC = p->m_buOffsetY = p->m_buOffsetC = NULL;
        delete p;
    }
    ~SaoFixture()
    {
        sao.destroy(1);
        drop(rec); drop(fenc);
        fd->m_reconPic = NULL; frame.m_reconPic = NULL; frame.m_fencPic = NULL; frame.m_encData = NULL;
        fd->destroy(); delete fd;
        x265_param_free(param);
    }
};
void ref_sao_stats_picture(pixel* const* rec, pixel* const* fenc, intptr_t stride, intptr_t cstride, int width, int height, int32_t* count, int32_t* offsetOrg)
{
    ensure();
    SaoFixture f(rec, fenc, stride, cstride, width, height);
    for (uint32_t addr = 0; addr < f.sps.numCUsInFrame; addr++)
    {
        f.sao.clear();
        for (int plane = 0; plane < 3; plane++)
        {
            f.sao.calcSaoStatsCTU((int)addr, plane);
            memcpy(count + ((size_t)addr * 3 + plane) * 5 * 32, f.sao.cnt(plane), 5 * 32 * sizeof(int32_t));
            memcpy(offsetOrg + ((size_t)addr * 3 + plane) * 5 * 32, f.sao.org(plane), 5 * 32 * sizeof(int32_t));
        }
    }
}
struct PackedSaoCtu { int8_t type[2]; uint8_t bandPos[3]; int8_t offset[3][4]; uint8_t pad[3]; };
/* planes are filtered in place; pre: an untouched copy of the same (deblocked) planes, the source of the saved line above each row */
void ref_sao_apply_picture(pixel* const* planes, pixel* const* pre, intptr_t stride, intptr_t cstride, int width, int height, const PackedSaoCtu* params)
{
    ensure();
    SaoFixture f(planes, pre, stride, cstride, width, height);
    const int numCtu = (int)f.sps.numCUsInFrame, ctuW = (int)f.sps.numCuInWidth;
    SaoCtuParam* cp[3];
    for (int c = 0; c < 3; c++)
    {
        cp[c] = new SaoCtuParam[numCtu];
        for (int a = 0; a < numCtu; a++)
        {
            cp[c][a].reset();
            cp[c][a].typeIdx = params[a].type[c ? 1 : 0];
            cp[c][a].bandPos = params[a].bandPos[c];
            for (int i = 0; i < 4; i++) cp[c][a].offset[i] = params[a].offset[c][i];
        }
    }
    for (int row = 0; row < (int)f.sps.numCuInHeight; row++)
    {
        for (int c = 0; c < 3; c++)
        {
            const intptr_t st = c ? cstride : stride;
            const int picW = c ? width / 2 : width, y = (row * 64) >> (c ? 1 : 0);
            const pixel* line = pre[c] + (intptr_t)(row ? y - 1 : y) * st;
            for (int x = -1; x <= picW; x++) f.sao.tmpU(c)[x] = line[x];
        }
        for (int col = 0; col < ctuW; col++)
        {
            f.sao.generateLumaOffsets(cp[0], row, col);
            f.sao.generateChromaOffsets(cp, row, col);
        }
    }
    for (int c = 0; c < 3; c++) delete[] cp[c];
}

/* ---- final entropy coding with the reference's own Entropy class (encoder/entropy.cpp: encodeCTU ... finishSlice) on CUData built from
 * the raster records of include/x265amd.h (x265amd_cu_unit / x265amd_slice_info), coefficients in the reference's per-CTU layout ---- */
struct RefCuUnit { uint8_t depth, predMode, partSize, tuDepth, lumaDir, chromaDir, mergeFlag, interDir; uint8_t cbf[3]; uint8_t tqBypass; int8_t qp; int8_t refIdx[2];
                   uint8_t mvpIdx[2]; uint8_t reserved; int16_t mvd[2][2]; };
struct RefSliceInfo { int32_t picWidth, picHeight, sliceType, sliceQp, numRefIdx[2], maxNumMergeCand, useDqp, maxCuDqpDepth, signHide, tqBypassEnabled, wpp,
                      maxCuDepth, maxAmpDepth, tuLog2Min, tuLog2Max, tuMaxDepthInter, tuMaxDepthIntra; };
/* coeff: numCtu x (64*64 + 2*32*32) int16 (Y | U | V per CTU).  bitsOnly: no bitstream; ctxOut receives the final context states,
 * qpOut (optional) the qp map after coding.  Returns the number of bytes written to out. */
size_t ref_encode_ctus(const RefSliceInfo* si, const RefCuUnit* units, const int16_t* coeff, int bitsOnly, uint8_t* out, size_t cap, uint8_t* ctxOut, int8_t* qpOut)
{
    ensure();
    const int width = si->picWidth, height = si->picHeight;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420;
    param->maxCUSize = 64; param->minCUSize = 8; param->maxLog2CUSize = 6; param->unitSizeDepth = 4; param->num4x4Partitions = 256;
    param->maxCUDepth = si->maxCuDepth; param->bLossless = 0;
    SPS sps; PPS pps;
    memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
    sps.numCuInWidth = (width + 63) / 64; sps.numCuInHeight = (height + 63) / 64; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    sps.numPartitions = 256; sps.numPartInCUSize = 16; sps.chromaFormatIdc = X265_CSP_I420;
    sps.picWidthInLumaSamples = width; sps.picHeightInLumaSamples = height;
    sps.log2MinCodingBlockSize = 3; sps.log2DiffMaxMinCodingBlockSize = 3;
    sps.quadtreeTULog2MinSize = si->tuLog2Min; sps.quadtreeTULog2MaxSize = si->tuLog2Max;
    sps.quadtreeTUMaxDepthInter = si->tuMaxDepthInter; sps.quadtreeTUMaxDepthIntra = si->tuMaxDepthIntra; sps.maxAMPDepth = si->maxAmpDepth;
    pps.bUseDQP = si->useDqp != 0; pps.maxCuDQPDepth = si->maxCuDqpDepth; pps.bSignHideEnabled = si->signHide != 0;
    pps.bTransquantBypassEnabled = si->tqBypassEnabled != 0; pps.bTransformSkipEnabled = 0; pps.bEntropyCodingSyncEnabled = si->wpp != 0;
    FrameData* fd = new FrameData;
    fd->create(*param, sps, X265_CSP_I420);
    Slice* slice = fd->m_slice;
    slice->m_sps = &sps; slice->m_pps = &pps; slice->m_param = param;
    slice->m_sliceType = si->sliceType == 2 ? I_SLICE : (si->sliceType == 1 ? P_SLICE : B_SLICE);
    slice->m_sliceQp = si->sliceQp; slice->m_numRefIdx[0] = si->numRefIdx[0]; slice->m_numRefIdx[1] = si->numRefIdx[1];
    slice->m_maxNumMergeCand = si->maxNumMergeCand;
    slice->m_endCUAddr = slice->realEndAddress(sps.numCUsInFrame * 256);
    Frame frame;
    frame.m_encData = fd; frame.m_param = param;
    const int w4 = width >> 2;
    for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
    {
        CUData& ctu = fd->m_picCTU[addr];
        ctu.initCTU(frame, addr, si->sliceQp, addr < sps.numCuInWidth, addr / sps.numCuInWidth == sps.numCuInHeight - 1, 0);
        ctu.m_chromaFormat = X265_CSP_I420; ctu.m_hChromaShift = ctu.m_vChromaShift = 1;
        const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
        for (uint32_t z = 0; z < 256; z++)
        {
            const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
            if (x >= width || y >= height)
            {
                /* what the encoder leaves there when the CTU is complete: the copyToPic of the smallest CU that has the unit's block as an absent sub-CU has written that
                 * sub-CU's depth (CUData::setEmptyPart, cudata.cpp:422-427) -- the largest block around the unit whose corner lies outside the picture.  getLastCodedQP steps
                 * over such units by that block (cudata.cpp:857-869). */
                int d = 1;
                for (; d < 4; d++) { const int sz = 64 >> d; if ((x & ~(sz - 1)) >= width || (y & ~(sz - 1)) >= height) break; }
                ctu.m_predMode[z] = MODE_NONE; ctu.m_cuDepth[z] = (uint8_t)d; continue;
            }
            const RefCuUnit& u = units[(y >> 2) * w4 + (x >> 2)];
            ctu.m_cuDepth[z] = u.depth; ctu.m_log2CUSize[z] = (uint8_t)(6 - u.depth);
            ctu.m_predMode[z] = u.predMode == 1 ? MODE_INTER : (u.predMode == 2 ? MODE_INTRA : (u.predMode == 3 ? MODE_SKIP : MODE_NONE));
            ctu.m_partSize[z] = u.partSize; ctu.m_tuDepth[z] = u.tuDepth; ctu.m_lumaIntraDir[z] = u.lumaDir; ctu.m_chromaIntraDir[z] = u.chromaDir;
            ctu.m_mergeFlag[z] = u.mergeFlag; ctu.m_interDir[z] = u.interDir; ctu.m_skipFlag[0][z] = ctu.m_skipFlag[1][z] = 0;
            for (int c = 0; c < 3; c++) ctu.m_cbf[c][z] = u.cbf[c];
            ctu.m_tqBypass[z] = u.tqBypass; ctu.m_qp[z] = u.qp;
            for (int l = 0; l < 2; l++) { ctu.m_refIdx[l][z] = u.refIdx[l]; ctu.m_mvpIdx[l][z] = u.mvpIdx[l]; ctu.m_mvd[l][z] = MV(u.mvd[l][0], u.mvd[l][1]); }
            ctu.m_transformSkip[0][z] = ctu.m_transformSkip[1][z] = ctu.m_transformSkip[2][z] = 0;
        }
    }
    Entropy e;
    Bitstream bs;
    if (!bitsOnly) e.setBitstream(&bs);
    e.resetEntropy(*slice);
    e.zeroFract();
    for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
    {
        CUData& ctu = fd->m_picCTU[addr];
        const int16_t* c = coeff + (size_t)addr * (64 * 64 + 2 * 32 * 32);
        coeff_t* save[3] = { ctu.m_trCoeff[0], ctu.m_trCoeff[1], ctu.m_trCoeff[2] };
        ctu.m_trCoeff[0] = (coeff_t*)c; ctu.m_trCoeff[1] = (coeff_t*)c + 64 * 64; ctu.m_trCoeff[2] = (coeff_t*)c + 64 * 64 + 32 * 32;
        CUGeom geoms[CUGeom::MAX_GEOMS];
        const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
        CUData::calcCTUGeoms(X265_MIN(64, width - cx), X265_MIN(64, height - cy), 64, 8, geoms);
        e.encodeCTU(ctu, geoms[0]);
        if (getenv("REF_CTX_TRACE"))
        {
            FILE* f = fopen(getenv("REF_CTX_TRACE"), addr ? "ab" : "wb");
            fwrite(e.m_contextState, 1, 160, f); fclose(f);
        }
        ctu.m_trCoeff[0] = save[0]; ctu.m_trCoeff[1] = save[1]; ctu.m_trCoeff[2] = save[2];
    }
    size_t n = 0;
    if (!bitsOnly)
    {
        e.finishSlice();
        n = bs.getNumberOfWrittenBytes();
        if (n <= cap) memcpy(out, bs.getFIFO(), n);
    }
    memcpy(ctxOut, e.m_contextState, MAX_OFF_CTX_MOD);
    if (qpOut)
        for (int y4 = 0; y4 < (height >> 2); y4++)
            for (int x4 = 0; x4 < w4; x4++)
            {
                const uint32_t addr = (y4 >> 4) * sps.numCuInWidth + (x4 >> 4);
                qpOut[y4 * w4 + x4] = fd->m_picCTU[addr].m_qp[g_rasterToZscan[(y4 & 15) * 16 + (x4 & 15)]];
            }
    frame.m_encData = NULL;
    fd->destroy(); delete fd;
    x265_param_free(param);
    return n;
}

/* ---- motion vector prediction with the reference's own CUData methods (common/cudata.cpp: getInterMergeCandidates, getNeighbourMV,
 * getPMV) on fixtures built from raster motion fields (records of include/x265amd.h) ---- */
struct RefMvUnit { uint8_t predMode, interDir; int8_t refIdx[2]; int16_t mv[2][2]; };
struct RefMvInfo { int32_t picWidth, picHeight, isInterB, numRefIdx[2], maxNumMergeCand, temporalMvp, colFromL0, checkLdc, poc, refPoc[2][16], colPoc, colRefPoc[2][16]; };
struct RefMergeCand { int16_t mv[2][2]; int8_t refIdx[2]; uint8_t dir, reserved; };
struct MvFixture
{
    x265_param* param; SPS sps; PPS pps; FrameData* fd[2]; Frame frame[2];
    MvFixture(const RefMvInfo* I, const RefMvUnit* cur, const RefMvUnit* col)
    {
        const int width = I->picWidth, height = I->picHeight;
        param = x265_param_alloc();
        x265_param_default(param);
        param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420;
        param->maxCUSize = 64; param->minCUSize = 8; param->maxLog2CUSize = 6; param->unitSizeDepth = 4; param->num4x4Partitions = 256; param->maxCUDepth = 3;
        memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
        sps.numCuInWidth = (width + 63) / 64; sps.numCuInHeight = (height + 63) / 64; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
        sps.numPartitions = 256; sps.numPartInCUSize = 16; sps.chromaFormatIdc = X265_CSP_I420;
        sps.picWidthInLumaSamples = width; sps.picHeightInLumaSamples = height; sps.bTemporalMVPEnabled = I->temporalMvp != 0;
        for (int k = 0; k < 2; k++)
        {
            fd[k] = new FrameData;
            fd[k]->create(*param, sps, X265_CSP_I420);
            Slice* sl = fd[k]->m_slice;
            sl->m_sps = &sps; sl->m_pps = &pps; sl->m_param = param;
            sl->m_sliceType = I->isInterB ? B_SLICE : P_SLICE;
            sl->m_poc = k ? I->colPoc : I->poc;
            for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) sl->m_refPOCList[l][r] = k ? I->colRefPoc[l][r] : I->refPoc[l][r];
            frame[k].m_encData = fd[k]; frame[k].m_param = param;
            const RefMvUnit* map = k ? col : cur;
            const int w4 = width >> 2;
            for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
            {
                CUData& ctu = fd[k]->m_picCTU[addr];
                ctu.initCTU(frame[k], addr, 30, addr < sps.numCuInWidth, 0, 0);
                const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
                for (uint32_t z = 0; z < 256; z++)
                {
                    const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
                    if (x >= width || y >= height || !map) { ctu.m_predMode[z] = MODE_NONE; ctu.m_refIdx[0][z] = ctu.m_refIdx[1][z] = -1; continue; }
                    const RefMvUnit& u = map[(y >> 2) * w4 + (x >> 2)];
                    ctu.m_predMode[z] = u.predMode == 1 ? MODE_INTER : (u.predMode == 2 ? MODE_INTRA : (u.predMode == 3 ? MODE_SKIP : MODE_NONE));
                    ctu.m_interDir[z] = u.interDir;
                    for (int l = 0; l < 2; l++) { ctu.m_refIdx[l][z] = u.refIdx[l]; ctu.m_mv[l][z] = MV(u.mv[l][0], u.mv[l][1]); }
                }
            }
        }
        Slice* sl = fd[0]->m_slice;
        sl->m_numRefIdx[0] = I->numRefIdx[0]; sl->m_numRefIdx[1] = I->numRefIdx[1]; sl->m_maxNumMergeCand = I->maxNumMergeCand;
        sl->m_colFromL0Flag = I->colFromL0; sl->m_colRefIdx = 0; sl->m_bCheckLDC = I->checkLdc != 0;
        for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) sl->m_refFrameList[l][r] = &frame[1];
    }
    ~MvFixture()
    {
        for (int k = 0; k < 2; k++) { frame[k].m_encData = NULL; fd[k]->destroy(); delete fd[k]; }
        x265_param_free(param);
    }
};
/* out: merge candidates (count returned in *numMerge); amvp[list][ref][2][2], mvc[list][ref][12][2], numMvc[list][ref] for every reference of every list */
void ref_mv_pred(const RefMvInfo* I, const RefMvUnit* cur, const RefMvUnit* col, int cuX, int cuY, int log2CU, int partSize, int puIdx,
                 RefMergeCand* merge, int* numMerge, int16_t* amvp, int16_t* mvc, int* numMvc)
{
    ensure();
    MvFixture f(I, cur, col);
    const uint32_t addr = (cuY >> 6) * f.sps.numCuInWidth + (cuX >> 6);
    CUData& ctu = f.fd[0]->m_picCTU[addr];
    CUGeom geoms[CUGeom::MAX_GEOMS];
    CUData::calcCTUGeoms(64, 64, 64, 8, geoms);
    const uint32_t depth = 6 - log2CU;
    const uint32_t absPartIdx = g_rasterToZscan[((cuY & 63) >> 2) * 16 + ((cuX & 63) >> 2)];
    const CUGeom* g = NULL;
    for (int i = 0; i < CUGeom::MAX_GEOMS; i++)
        if (geoms[i].depth == depth && geoms[i].absPartIdx == absPartIdx) { g = &geoms[i]; break; }
    CUDataMemPool pool;
    pool.create(depth, X265_CSP_I420, 1, *f.param);
    CUData sub;
    sub.initialize(pool, depth, *f.param, 0);
    sub.initSubCU(ctu, *g, 30);
    const int w4 = I->picWidth >> 2;
    for (uint32_t i = 0; i < g->numPartitions; i++)
    {
        const int x = cuX + g_zscanToPelX[absPartIdx + i] - g_zscanToPelX[absPartIdx], y = cuY + g_zscanToPelY[absPartIdx + i] - g_zscanToPelY[absPartIdx];
        const RefMvUnit& u = cur[(y >> 2) * w4 + (x >> 2)];
        sub.m_predMode[i] = u.predMode == 1 ? MODE_INTER : (u.predMode == 2 ? MODE_INTRA : (u.predMode == 3 ? MODE_SKIP : MODE_NONE));
        sub.m_interDir[i] = u.interDir; sub.m_partSize[i] = (uint8_t)partSize; sub.m_log2CUSize[i] = (uint8_t)log2CU; sub.m_cuDepth[i] = (uint8_t)depth;
        for (int l = 0; l < 2; l++) { sub.m_refIdx[l][i] = u.refIdx[l]; sub.m_mv[l][i] = MV(u.mv[l][0], u.mv[l][1]); }
    }
    uint32_t puAddr; int pw, ph;
    sub.getPartIndexAndSize(puIdx, puAddr, pw, ph);
    MVField cand[MRG_MAX_NUM_CANDS][2];
    uint8_t dirs[MRG_MAX_NUM_CANDS];
    *numMerge = (int)sub.getInterMergeCandidates(puAddr, puIdx, cand, dirs);
    for (int i = 0; i < *numMerge; i++)
    {
        memset(&merge[i], 0, sizeof(merge[i]));
        merge[i].dir = dirs[i];
        for (int l = 0; l < 2; l++) { merge[i].mv[l][0] = cand[i][l].mv.x; merge[i].mv[l][1] = cand[i][l].mv.y; merge[i].refIdx[l] = (int8_t)cand[i][l].refIdx; }
    }
    InterNeighbourMV nb[6];
    sub.getNeighbourMV(puIdx, puAddr, nb);
    for (int l = 0; l < (I->isInterB ? 2 : 1); l++)
        for (int r = 0; r < I->numRefIdx[l]; r++)
        {
            MV a[2], m[(MD_ABOVE_LEFT + 1) * 2 + 2];
            int n = sub.getPMV(nb, l, r, a, m);
            int16_t* ao = amvp + (l * 16 + r) * 4; int16_t* mo = mvc + (l * 16 + r) * 24;
            ao[0] = a[0].x; ao[1] = a[0].y; ao[2] = a[1].x; ao[3] = a[1].y;
            for (int k = 0; k < n; k++) { mo[2 * k] = m[k].x; mo[2 * k + 1] = m[k].y; }
            numMvc[l * 16 + r] = n;
        }
    pool.destroy();
}

/* ---- Search::predInterSearch (encoder/search.cpp:2181-2647) itself, on a fixture: CUData / Slice / MotionReference / Frame objects built
 * from raster motion fields and padded planes; one CU per call ---- */
struct RefSearchParams { int32_t searchMethod, subpelRefine, searchRange, qp, bChromaMC, numPics, refPic[2][16]; };
struct RefPuResult { uint8_t mergeFlag, interDir; int8_t refIdx[2]; uint8_t mvpIdx[2]; uint8_t pad[2]; int16_t mv[2][2]; int16_t mvd[2][2]; };
static PicYuv* mkPic(x265_param* param, const SPS& sps, const uint64_t* planes, intptr_t stride, intptr_t cstride, int width, int height, int mx, int my)
{
    PicYuv* p = new PicYuv;
    p->m_param = param; p->m_picCsp = X265_CSP_I420; p->m_hChromaShift = p->m_vChromaShift = 1;
    p->m_picOrg[0] = (pixel*)planes[0]; p->m_picOrg[1] = (pixel*)planes[1]; p->m_picOrg[2] = (pixel*)planes[2];
    p->m_picBuf[0] = p->m_picOrg[0] - my * stride - mx; p->m_picBuf[1] = p->m_picOrg[1] - (my / 2) * cstride - mx / 2; p->m_picBuf[2] = p->m_picOrg[2] - (my / 2) * cstride - mx / 2;
    p->m_stride = stride; p->m_strideC = cstride; p->m_picWidth = width; p->m_picHeight = height;
    p->m_lumaMarginX = mx; p->m_lumaMarginY = my; p->m_chromaMarginX = mx / 2; p->m_chromaMarginY = my / 2;
    p->createOffsets(sps);
    return p;
}
static void dropPic(PicYuv* p)
{
    for (int c = 0; c < 3; c++) p->m_picOrg[c] = p->m_picBuf[c] = NULL;
    X265_FREE(p->m_cuOffsetY); X265_FREE(p->m_cuOffsetC); X265_FREE(p->m_buOffsetY); X265_FREE(p->m_buOffsetC);
    p->m_cuOffsetY = p->m_cuOffsetC = p->m_buOffsetY = p->m_buOffsetC = NULL;
    delete p;
}
/* planes: numPics x 3 addresses of sample (0,0); picture numPics-1 is the source.  refPic[list][ref] = picture index of that reference.
 * predY / predU / predV: 64x64 / 32x32 buffers (strides 64 / 32) receiving interMode.predYuv; returns sa8dBits */
int ref_pred_inter_search(const RefMvInfo* I, const RefSearchParams* S, const RefMvUnit* cur, const RefMvUnit* col, const uint64_t* planes, intptr_t stride,
                          intptr_t cstride, int marginX, int marginY, int cuX, int cuY, int log2CU, int partSize, RefPuResult* out, pixel* predY, pixel* predU, pixel* predV)
{
    ensure();
    MvFixture f(I, cur, col);
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 0 done\n");
    x265_param* param = f.param;
    param->searchMethod = S->searchMethod; param->subpelRefine = S->subpelRefine; param->searchRange = S->searchRange;
    param->frameNumThreads = 1; param->maxSlices = 1; param->bframes = 0; param->bEnableWeightedPred = param->bEnableWeightedBiPred = 0;
    param->bDistributeMotionEstimation = 0; param->bEnableHME = 0; param->analysisLoadReuseLevel = 0; param->analysisSave = NULL; param->analysisLoad = NULL;
    param->analysisMultiPassRefine = 0; param->bAnalysisType = 0; param->bIntraRefresh = 0; param->bSourceReferenceEstimation = 0;
    param->psyRd = 0; param->bSsimRd = 0; param->psyRdoq = 0; param->noiseReductionIntra = param->noiseReductionInter = 0; param->limitTU = 0;
    param->interRefine = 0; param->mvRefine = 1; param->rc.bStatRead = 0;
    Slice* slice = f.fd[0]->m_slice;
    f.sps.quadtreeTULog2MaxSize = 5; f.sps.quadtreeTULog2MinSize = 2;
    std::vector<PicYuv*> pics;
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 1 done\n");
    for (int i = 0; i < S->numPics; i++) pics.push_back(mkPic(param, f.sps, planes + 3 * i, stride, cstride, I->picWidth, I->picHeight, marginX, marginY));
    static Frame refFrames[2][16];
    static MV noLowres(0x7FFF, 0);
    MotionReference (*mref)[MAX_NUM_REF + 1] = new MotionReference[2][MAX_NUM_REF + 1];       /* FrameEncoder::m_mref */
    slice->m_mref = mref;
    Frame& frame = f.frame[0];
    frame.m_fencPic = pics.back();
    for (int l = 0; l < 2; l++) for (int i = 0; i < X265_BFRAME_MAX + 2; i++) frame.m_lowres.lowresMvs[l][i] = &noLowres;
    f.fd[0]->m_reconPic = pics.back();
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < I->numRefIdx[l]; r++)
        {
            PicYuv* rp = pics[S->refPic[l][r]];
            slice->m_refReconPicList[l][r] = rp;
            refFrames[l][r].m_encData = f.fd[1]; refFrames[l][r].m_reconPic = rp; refFrames[l][r].m_fencPic = rp; refFrames[l][r].m_param = param;
            slice->m_refFrameList[l][r] = &refFrames[l][r];
            slice->m_mref[l][r].init(rp, NULL, *param);
        }
    int bits = 0;
    {
        Search* search = new Search;
        ScalingList* sl = new ScalingList;
        sl->init(); sl->m_bEnabled = false; sl->m_bDataPresent = false; sl->setupQuantMatrices(X265_CSP_I420);
        search->initSearch(*param, *sl);
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 2 done\n");
        search->m_slice = slice; search->m_frame = &frame;
        const uint32_t addr = (cuY >> 6) * f.sps.numCuInWidth + (cuX >> 6);
        CUData& ctu = f.fd[0]->m_picCTU[addr];
        search->setLambdaFromQP(ctu, S->qp);
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 3 done\n");
        CUGeom geoms[CUGeom::MAX_GEOMS];
        CUData::calcCTUGeoms(64, 64, 64, 8, geoms);
        const uint32_t depth = 6 - log2CU;
        const uint32_t absPartIdx = g_rasterToZscan[((cuY & 63) >> 2) * 16 + ((cuX & 63) >> 2)];
        const CUGeom* g = NULL;
        for (int i = 0; i < CUGeom::MAX_GEOMS; i++)
            if (geoms[i].depth == depth && geoms[i].absPartIdx == absPartIdx) { g = &geoms[i]; break; }
        CUDataMemPool pool;
        pool.create(depth, X265_CSP_I420, 1, *param);
        Mode* mode = new Mode;
        mode->cu.initialize(pool, depth, *param, 0);
        mode->predYuv.create(1 << log2CU, X265_CSP_I420);
        Yuv fenc;
        fenc.create(1 << log2CU, X265_CSP_I420);
        fenc.copyFromPicYuv(*frame.m_fencPic, addr, absPartIdx);
        mode->fencYuv = &fenc;
        mode->cu.initSubCU(ctu, *g, S->qp);
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 4 done\n");
        mode->cu.setPartSizeSubParts((PartSize)partSize);
        mode->cu.setPredModeSubParts(MODE_INTER);
        mode->initCosts();
        uint32_t masks[2] = { 0, 0 };
        search->predInterSearch(*mode, *g, S->bChromaMC != 0, masks);
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 5 done\n");
        bits = (int)mode->sa8dBits;
        const int numPart = mode->cu.getNumPartInter(0);
        for (int p = 0; p < numPart; p++)
        {
            uint32_t pa; int pw, ph;
            mode->cu.getPartIndexAndSize(p, pa, pw, ph);
            RefPuResult& r = out[p];
            memset(&r, 0, sizeof(r));
            r.mergeFlag = mode->cu.m_mergeFlag[pa]; r.interDir = mode->cu.m_interDir[pa];
            for (int l = 0; l < 2; l++)
            {
                r.refIdx[l] = mode->cu.m_refIdx[l][pa]; r.mvpIdx[l] = mode->cu.m_mvpIdx[l][pa];
                r.mv[l][0] = mode->cu.m_mv[l][pa].x; r.mv[l][1] = mode->cu.m_mv[l][pa].y;
                r.mvd[l][0] = mode->cu.m_mvd[l][pa].x; r.mvd[l][1] = mode->cu.m_mvd[l][pa].y;
            }
        }
        const int n = 1 << log2CU;
        for (int y = 0; y < n; y++) memcpy(predY + y * 64, mode->predYuv.m_buf[0] + y * mode->predYuv.m_size, n * sizeof(pixel));
        if (S->bChromaMC)
            for (int y = 0; y < n / 2; y++)
            {
                memcpy(predU + y * 32, mode->predYuv.m_buf[1] + y * mode->predYuv.m_csize, (n / 2) * sizeof(pixel));
                memcpy(predV + y * 32, mode->predYuv.m_buf[2] + y * mode->predYuv.m_csize, (n / 2) * sizeof(pixel));
            }
        mode->predYuv.destroy(); fenc.destroy();
        delete mode;
        pool.destroy();
        delete search;
    if (getenv("REF_TRACE")) fprintf(stderr, "stage 6 done\n");
        delete sl;
    }
    f.fd[0]->m_reconPic = NULL; frame.m_fencPic = NULL;
    slice->m_mref = NULL;
    delete[] mref;
    for (size_t i = 0; i < pics.size(); i++) dropPic(pics[i]);
    return bits;
}

/* ---- Search::encodeResAndCalcRdInterCU (encoder/search.cpp:2822-2975: estimateResidualQT, the no-residual alternative, the CU's
 * syntax bits, reconstruction, updateModeCost, checkDQP) itself, on a fixture: picture CUData built from the raster unit map (which already
 * carries the candidate CU's prediction fields), the source picture, and a given prediction block; one CU per call ---- */
struct RefRdParams { double psyRd; int32_t rdLevel, reserved, rdoqLevel, psyRdoqScale, fastIntra, reserved2; };
struct RefRdResult { uint64_t rdCost, distortion, fracBits; uint32_t totalBits, mvBits, coeffBits, psyEnergy, lumaDist, chromaDist, resEnergy, reserved; uint8_t ctx[160]; };
/* srcPlanes: addresses of sample (0,0) of the source Y, U, V.  predY/U/V: strides 64 / 32.  cuUnitsOut: the CU's units after the call, raster within the
 * CU (row length size/4).  coeffOut: 4096 + 2 * 1024 levels in CUData::m_trCoeff layout.  reconY/U/V: strides 64 / 32. */
static const uint64_t* g_rdReconPlanes = NULL; static int g_rdStrongSmoothing = 0; static uint64_t* g_rdIntraInfo = NULL; static pixel* g_rdPredOut = NULL; static int g_rdPartSize = 0;
static void rd_fixture_run(int skipCU, const RefSliceInfo* si, const RefRdParams* rp, const RefCuUnit* units, const uint64_t* srcPlanes, intptr_t stride, intptr_t cstride,
                           int cuX, int cuY, int log2CU, int qp, const uint8_t* ctxIn, uint64_t fracIn, const pixel* predY, const pixel* predU, const pixel* predV,
                           RefCuUnit* cuUnitsOut, int16_t* coeffOut, pixel* reconY, pixel* reconU, pixel* reconV, RefRdResult* out)
{
    ensure();
    const int width = si->picWidth, height = si->picHeight;
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420;
    param->maxCUSize = 64; param->minCUSize = 8; param->maxLog2CUSize = 6; param->unitSizeDepth = 4; param->num4x4Partitions = 256;
    param->maxCUDepth = si->maxCuDepth; param->bLossless = 0;
    param->frameNumThreads = 1; param->maxSlices = 1; param->bEnableWeightedPred = param->bEnableWeightedBiPred = 0;
    param->bDistributeMotionEstimation = 0; param->bEnableHME = 0; param->analysisLoadReuseLevel = 0; param->analysisSave = NULL; param->analysisLoad = NULL;
    param->psyRd = rp->psyRd; param->bSsimRd = 0; param->rdoqLevel = rp->rdoqLevel; param->psyRdoq = rp->rdoqLevel ? rp->psyRdoqScale / 256.0 : 0; param->bEnableFastIntra = rp->fastIntra; param->noiseReductionIntra = param->noiseReductionInter = 0;
    param->limitTU = 0; param->rdLevel = rp->rdLevel; param->bEnableTransformSkip = 0; param->bEnableTSkipFast = 0;
    param->bEnableSignHiding = si->signHide; param->maxTUSize = 1 << si->tuLog2Max;
    param->tuQTMaxInterDepth = si->tuMaxDepthInter; param->tuQTMaxIntraDepth = si->tuMaxDepthIntra;
    SPS sps; PPS pps;
    memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
    sps.numCuInWidth = (width + 63) / 64; sps.numCuInHeight = (height + 63) / 64; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    sps.numPartitions = 256; sps.numPartInCUSize = 16; sps.chromaFormatIdc = X265_CSP_I420;
    sps.picWidthInLumaSamples = width; sps.picHeightInLumaSamples = height;
    sps.log2MinCodingBlockSize = 3; sps.log2DiffMaxMinCodingBlockSize = 3;
    sps.quadtreeTULog2MinSize = si->tuLog2Min; sps.quadtreeTULog2MaxSize = si->tuLog2Max;
    sps.quadtreeTUMaxDepthInter = si->tuMaxDepthInter; sps.quadtreeTUMaxDepthIntra = si->tuMaxDepthIntra; sps.maxAMPDepth = si->maxAmpDepth;
    pps.bUseDQP = si->useDqp != 0; pps.maxCuDQPDepth = si->maxCuDqpDepth; pps.bSignHideEnabled = si->signHide != 0;
    pps.bTransquantBypassEnabled = si->tqBypassEnabled != 0; pps.bTransformSkipEnabled = 0; pps.bEntropyCodingSyncEnabled = si->wpp != 0;
    FrameData* fd = new FrameData;
    fd->create(*param, sps, X265_CSP_I420);
    Slice* slice = fd->m_slice;
    slice->m_sps = &sps; slice->m_pps = &pps; slice->m_param = param;
    slice->m_sliceType = si->sliceType == 2 ? I_SLICE : (si->sliceType == 1 ? P_SLICE : B_SLICE);
    slice->m_sliceQp = si->sliceQp; slice->m_numRefIdx[0] = si->numRefIdx[0]; slice->m_numRefIdx[1] = si->numRefIdx[1];
    slice->m_maxNumMergeCand = si->maxNumMergeCand;
    slice->m_endCUAddr = slice->realEndAddress(sps.numCUsInFrame * 256);
    Frame frame;
    frame.m_encData = fd; frame.m_param = param;
    PicYuv* src = mkPic(param, sps, srcPlanes, stride, cstride, width, height, 0, 0);
    frame.m_fencPic = src;
    PicYuv* rec = NULL;
    if (skipCU >= 2)
    {
        rec = mkPic(param, sps, g_rdReconPlanes, stride, cstride, width, height, 0, 0);
        frame.m_reconPic = rec; fd->m_reconPic = rec;
        sps.bUseStrongIntraSmoothing = g_rdStrongSmoothing != 0;
        param->bEnableStrongIntraSmoothing = g_rdStrongSmoothing; param->bEnableFastIntra = 0; param->bEnableConstrainedIntra = 0; param->rdPenalty = 0;
    }
    const int w4 = width >> 2;
    for (uint32_t addr = 0; addr < sps.numCUsInFrame; addr++)
    {
        CUData& ctu = fd->m_picCTU[addr];
        ctu.initCTU(frame, addr, si->sliceQp, addr < sps.numCuInWidth, addr / sps.numCuInWidth == sps.numCuInHeight - 1, 0);
        ctu.m_chromaFormat = X265_CSP_I420; ctu.m_hChromaShift = ctu.m_vChromaShift = 1;
        const int cx = (addr % sps.numCuInWidth) * 64, cy = (addr / sps.numCuInWidth) * 64;
        for (uint32_t z = 0; z < 256; z++)
        {
            const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
            if (x >= width || y >= height) { ctu.m_predMode[z] = MODE_NONE; ctu.m_cuDepth[z] = 0; continue; }
            const RefCuUnit& u = units[(y >> 2) * w4 + (x >> 2)];
            ctu.m_cuDepth[z] = u.depth; ctu.m_log2CUSize[z] = (uint8_t)(6 - u.depth);
            ctu.m_predMode[z] = u.predMode == 1 ? MODE_INTER : (u.predMode == 2 ? MODE_INTRA : (u.predMode == 3 ? MODE_SKIP : MODE_NONE));
            ctu.m_partSize[z] = u.partSize; ctu.m_tuDepth[z] = u.tuDepth; ctu.m_lumaIntraDir[z] = u.lumaDir; ctu.m_chromaIntraDir[z] = u.chromaDir;
            ctu.m_mergeFlag[z] = u.mergeFlag; ctu.m_interDir[z] = u.interDir; ctu.m_skipFlag[0][z] = ctu.m_skipFlag[1][z] = 0;
            for (int c = 0; c < 3; c++) ctu.m_cbf[c][z] = u.cbf[c];
            ctu.m_tqBypass[z] = u.tqBypass; ctu.m_qp[z] = u.qp;
            for (int l = 0; l < 2; l++) { ctu.m_refIdx[l][z] = u.refIdx[l]; ctu.m_mvpIdx[l][z] = u.mvpIdx[l]; ctu.m_mvd[l][z] = MV(u.mvd[l][0], u.mvd[l][1]); }
            ctu.m_transformSkip[0][z] = ctu.m_transformSkip[1][z] = ctu.m_transformSkip[2][z] = 0;
        }
    }
    {
        Search* search = new Search;
        ScalingList* sl = new ScalingList;
        sl->init(); sl->m_bEnabled = false; sl->m_bDataPresent = false; sl->setupQuantMatrices(X265_CSP_I420);
        search->initSearch(*param, *sl);
        search->m_slice = slice; search->m_frame = &frame;
        const uint32_t addr = (cuY >> 6) * sps.numCuInWidth + (cuX >> 6);
        CUData& ctu = fd->m_picCTU[addr];
        CUGeom geoms[CUGeom::MAX_GEOMS];
        CUData::calcCTUGeoms(64, 64, 64, 8, geoms);
        const uint32_t depth = 6 - log2CU;
        const uint32_t absPartIdx = g_rasterToZscan[((cuY & 63) >> 2) * 16 + ((cuX & 63) >> 2)];
        const CUGeom* g = NULL;
        for (int i = 0; i < CUGeom::MAX_GEOMS; i++)
            if (geoms[i].depth == depth && geoms[i].absPartIdx == absPartIdx) { g = &geoms[i]; break; }
        CUDataMemPool pool;
        pool.create(depth, X265_CSP_I420, 1, *param);
        Mode* mode = new Mode;
        mode->cu.initialize(pool, depth, *param, 0);
        const int n = 1 << log2CU;
        mode->predYuv.create(n, X265_CSP_I420);
        mode->reconYuv.create(n, X265_CSP_I420);
        Yuv fenc;
        fenc.create(n, X265_CSP_I420);
        fenc.copyFromPicYuv(*src, addr, absPartIdx);
        mode->fencYuv = &fenc;
        mode->cu.initSubCU(ctu, *g, qp);
        /* the candidate's fields, copied from the picture CTU (initSubCU resets them) */
        for (uint32_t i = 0; skipCU < 2 && i < g->numPartitions; i++)
        {
            const uint32_t z = absPartIdx + i;
            mode->cu.m_predMode[i] = ctu.m_predMode[z]; mode->cu.m_partSize[i] = ctu.m_partSize[z]; mode->cu.m_mergeFlag[i] = ctu.m_mergeFlag[z];
            mode->cu.m_interDir[i] = ctu.m_interDir[z]; mode->cu.m_qp[i] = ctu.m_qp[z]; mode->cu.m_tqBypass[i] = 0;
            for (int l = 0; l < 2; l++) { mode->cu.m_refIdx[l][i] = ctu.m_refIdx[l][z]; mode->cu.m_mvpIdx[l][i] = ctu.m_mvpIdx[l][z]; mode->cu.m_mvd[l][i] = ctu.m_mvd[l][z]; }
        }
        for (int y = 0; y < n; y++) memcpy(mode->predYuv.m_buf[0] + y * mode->predYuv.m_size, predY + y * 64, n * sizeof(pixel));
        for (int y = 0; y < n / 2; y++)
        {
            memcpy(mode->predYuv.m_buf[1] + y * mode->predYuv.m_csize, predU + y * 32, (n / 2) * sizeof(pixel));
            memcpy(mode->predYuv.m_buf[2] + y * mode->predYuv.m_csize, predV + y * 32, (n / 2) * sizeof(pixel));
        }
        mode->initCosts();
        search->setLambdaFromQP(mode->cu, qp);
        Entropy start;
        start.resetEntropy(*slice);
        memcpy(start.m_contextState, ctxIn, MAX_OFF_CTX_MOD);
        start.m_fracBits = fracIn;
        search->m_rqt[depth].cur.load(start);
        if (skipCU == 2)
        {
            search->checkIntraInInter(*mode, *g);
            g_rdIntraInfo[0] = mode->cu.m_lumaIntraDir[0]; g_rdIntraInfo[1] = mode->sa8dCost; g_rdIntraInfo[2] = mode->sa8dBits; g_rdIntraInfo[3] = mode->distortion;
            search->encodeIntraInInter(*mode, *g);
            for (int y = 0; y < (1 << log2CU); y++) memcpy(g_rdPredOut + y * 64, mode->predYuv.m_buf[0] + y * mode->predYuv.m_size, (1 << log2CU) * sizeof(pixel));
        }
        else if (skipCU == 3)
        {
            search->checkIntra(*mode, *g, (PartSize)g_rdPartSize);
            for (int y = 0; y < (1 << log2CU); y++) memcpy(g_rdPredOut + y * 64, mode->predYuv.m_buf[0] + y * mode->predYuv.m_size, (1 << log2CU) * sizeof(pixel));
        }
        else if (skipCU) search->encodeResAndCalcRdSkipCU(*mode);
        else search->encodeResAndCalcRdInterCU(*mode, *g);
        memset(out, 0, sizeof(*out));
        out->rdCost = mode->rdCost; out->distortion = mode->distortion; out->fracBits = mode->contexts.m_fracBits;
        out->totalBits = mode->totalBits; out->mvBits = mode->mvBits; out->coeffBits = mode->coeffBits; out->psyEnergy = mode->psyEnergy;
        out->lumaDist = (uint32_t)mode->lumaDistortion; out->chromaDist = (uint32_t)mode->chromaDistortion; out->resEnergy = (uint32_t)mode->resEnergy;
        memcpy(out->ctx, mode->contexts.m_contextState, MAX_OFF_CTX_MOD);
        const int u4 = n >> 2;
        for (uint32_t i = 0; i < g->numPartitions; i++)
        {
            const int ux = (g_zscanToPelX[absPartIdx + i] - g_zscanToPelX[absPartIdx]) >> 2, uy = (g_zscanToPelY[absPartIdx + i] - g_zscanToPelY[absPartIdx]) >> 2;
            RefCuUnit& u = cuUnitsOut[uy * u4 + ux];
            u = units[((cuY >> 2) + uy) * w4 + (cuX >> 2) + ux];
            const int pm = mode->cu.m_predMode[i];
            u.predMode = pm == MODE_SKIP ? 3 : (pm == MODE_INTRA ? 2 : (pm == MODE_INTER ? 1 : 0));
            u.tuDepth = mode->cu.m_tuDepth[i]; u.qp = mode->cu.m_qp[i];
            if (skipCU >= 2) { u.partSize = mode->cu.m_partSize[i]; u.lumaDir = mode->cu.m_lumaIntraDir[i]; u.chromaDir = mode->cu.m_chromaIntraDir[i]; u.depth = (uint8_t)depth; u.mergeFlag = 0; u.interDir = 0; }
            for (int c = 0; c < 3; c++) u.cbf[c] = mode->cu.m_cbf[c][i];
        }
        memcpy(coeffOut, mode->cu.m_trCoeff[0], n * n * sizeof(int16_t));
        memcpy(coeffOut + 4096, mode->cu.m_trCoeff[1], (n * n / 4) * sizeof(int16_t));
        memcpy(coeffOut + 4096 + 1024, mode->cu.m_trCoeff[2], (n * n / 4) * sizeof(int16_t));
        for (int y = 0; y < n; y++) memcpy(reconY + y * 64, mode->reconYuv.m_buf[0] + y * mode->reconYuv.m_size, n * sizeof(pixel));
        for (int y = 0; y < n / 2; y++)
        {
            memcpy(reconU + y * 32, mode->reconYuv.m_buf[1] + y * mode->reconYuv.m_csize, (n / 2) * sizeof(pixel));
            memcpy(reconV + y * 32, mode->reconYuv.m_buf[2] + y * mode->reconYuv.m_csize, (n / 2) * sizeof(pixel));
        }
        mode->predYuv.destroy(); mode->reconYuv.destroy(); fenc.destroy();
        delete mode;
        pool.destroy();
        delete search;
        delete sl;
    }
    frame.m_fencPic = NULL;
    dropPic(src);
    if (rec) { frame.m_reconPic = NULL; fd->m_reconPic = NULL; dropPic(rec); }
    frame.m_encData = NULL;
    fd->destroy(); delete fd;
    x265_param_free(param);
}

/* Search::checkIntraInInter + encodeIntraInInter (encoder/search.cpp:1291-1452, :1454-1507) on the same fixture, with a reconstructed picture
 * (read for the neighbours, written with the CU's reconstruction as the reference does).  info: luma mode, sa8dCost, sa8dBits, sa8d distortion
 * as checkIntraInInter leaves them; predY: intraMode.predYuv luma (stride 64) */
void ref_intra_in_inter(const RefSliceInfo* si, const RefRdParams* rp, const RefCuUnit* units, const uint64_t* srcPlanes, const uint64_t* reconPlanes, intptr_t stride,
                        intptr_t cstride, int cuX, int cuY, int log2CU, int qp, const uint8_t* ctxIn, uint64_t fracIn, int strongSmoothing,
                        RefCuUnit* cuUnitsOut, int16_t* coeffOut, pixel* predY, pixel* reconY, pixel* reconU, pixel* reconV, RefRdResult* out, uint64_t* info)
{
    static pixel dummy[64 * 64];
    g_rdReconPlanes = reconPlanes; g_rdStrongSmoothing = strongSmoothing; g_rdIntraInfo = info; g_rdPredOut = predY;
    rd_fixture_run(2, si, rp, units, srcPlanes, stride, cstride, cuX, cuY, log2CU, qp, ctxIn, fracIn, dummy, dummy, dummy, cuUnitsOut, coeffOut, reconY, reconU, reconV, out);
}

/* Search::checkIntra (encoder/search.cpp:1236-1287: estIntraPredQT, estIntraPredChromaQT, the CU's bits) on the same fixture; partSize 0 / 3 */
void ref_check_intra(const RefSliceInfo* si, const RefRdParams* rp, const RefCuUnit* units, const uint64_t* srcPlanes, const uint64_t* reconPlanes, intptr_t stride,
                     intptr_t cstride, int cuX, int cuY, int log2CU, int qp, const uint8_t* ctxIn, uint64_t fracIn, int strongSmoothing, int partSize,
                     RefCuUnit* cuUnitsOut, int16_t* coeffOut, pixel* predY, pixel* reconY, pixel* reconU, pixel* reconV, RefRdResult* out)
{
    static pixel dummy[64 * 64];
    static uint64_t info[4];
    g_rdReconPlanes = reconPlanes; g_rdStrongSmoothing = strongSmoothing; g_rdIntraInfo = info; g_rdPredOut = predY; g_rdPartSize = partSize;
    rd_fixture_run(3, si, rp, units, srcPlanes, stride, cstride, cuX, cuY, log2CU, qp, ctxIn, fracIn, dummy, dummy, dummy, cuUnitsOut, coeffOut, reconY, reconU, reconV, out);
}

void ref_inter_residual_rd(const RefSliceInfo* si, const RefRdParams* rp, const RefCuUnit* units, const uint64_t* srcPlanes, intptr_t stride, intptr_t cstride,
                           int cuX, int cuY, int log2CU, int qp, const uint8_t* ctxIn, uint64_t fracIn, const pixel* predY, const pixel* predU, const pixel* predV,
                           RefCuUnit* cuUnitsOut, int16_t* coeffOut, pixel* reconY, pixel* reconU, pixel* reconV, RefRdResult* out)
{
    rd_fixture_run(0, si, rp, units, srcPlanes, stride, cstride, cuX, cuY, log2CU, qp, ctxIn, fracIn, predY, predU, predV, cuUnitsOut, coeffOut, reconY, reconU, reconV, out);
}
/* Search::encodeResAndCalcRdSkipCU (encoder/search.cpp:2770-2818) on the same fixture */
void ref_skip_rd(const RefSliceInfo* si, const RefRdParams* rp, const RefCuUnit* units, const uint64_t* srcPlanes, intptr_t stride, intptr_t cstride,
                 int cuX, int cuY, int log2CU, int qp, const uint8_t* ctxIn, uint64_t fracIn, const pixel* predY, const pixel* predU, const pixel* predV,
                 RefCuUnit* cuUnitsOut, int16_t* coeffOut, pixel* reconY, pixel* reconU, pixel* reconV, RefRdResult* out)
{
    rd_fixture_run(1, si, rp, units, srcPlanes, stride, cstride, cuX, cuY, log2CU, qp, ctxIn, fracIn, predY, predU, predV, cuUnitsOut, coeffOut, reconY, reconU, reconV, out);
}

/* ---- Analysis::compressCTU (encoder/analysis.cpp:138-317 -> compressInterCU_rd0_4 :1146-1848 with checkMerge2Nx2N_rd0_4, checkInter_rd0_4,
 * checkBidir2Nx2N and the Search methods below them) itself, for one CTU of an inter slice, on a fixture: picture CUData from the raster
 * unit + motion maps (what is coded so far), reference pictures, their motion / depth maps, the source picture ---- */
struct RefAnalysisParams { double psyRd; int32_t rdLevel, earlySkip, rskip, limitRefs, bIntraInB, rect, amp, limitModes, strongIntraSmoothing, reserved, rdoqLevel, psyRdoqScale, fastIntra, reserved2; };
struct RefCuStat { uint32_t count[4]; uint32_t pad; uint64_t avgCost[4]; };
struct RefCtuResult { uint64_t rdCost, distortion, fracBits; uint32_t totalBits, reserved; uint8_t ctx[160]; };
/* planes: numPics x 3 addresses of sample (0,0); picture numPics-1 is the source, numPics-2 the reconstruction being written.
 * refDepth: the co-located CU depths of reference [list][0] per 4x4 unit (two maps, list 0 then list 1); refQp0[list]: their CTU QPs (per CTU).
 * unitsOut / mvOut: the CTU's 16x16 units after analysis (raster). */
void ref_compress_ctu(const RefMvInfo* I, const RefSearchParams* S, const RefSliceInfo* si, const RefAnalysisParams* A, const RefCuUnit* units,
                      const RefMvUnit* cur, const RefMvUnit* col, const uint8_t* refDepth, const int8_t* refQp0, const uint64_t* planes, intptr_t stride,
                      intptr_t cstride, int marginX, int marginY, RefCuStat* cuStat, int ctuAddr, const uint8_t* ctxIn, uint64_t fracIn,
                      RefCuUnit* unitsOut, RefMvUnit* mvOut, int16_t* coeffOut, RefCtuResult* out)
{
    ensure();
    MvFixture f(I, cur, col);
    const int width = I->picWidth, height = I->picHeight, w4 = width >> 2;
    x265_param* param = f.param;
    param->searchMethod = S->searchMethod; param->subpelRefine = S->subpelRefine; param->searchRange = S->searchRange;
    param->frameNumThreads = 1; param->maxSlices = 1; param->bframes = 0; param->bEnableWeightedPred = param->bEnableWeightedBiPred = 0;
    param->bDistributeMotionEstimation = 0; param->bDistributeModeAnalysis = 0; param->bEnableHME = 0; param->analysisLoadReuseLevel = 0; param->analysisSaveReuseLevel = 0;
    param->analysisSave = NULL; param->analysisLoad = NULL;
    param->analysisMultiPassRefine = 0; param->bAnalysisType = 0; param->bIntraRefresh = 0; param->bSourceReferenceEstimation = 0;
    param->psyRd = A->psyRd; param->bSsimRd = 0; param->rdoqLevel = A->rdoqLevel; param->psyRdoq = A->rdoqLevel ? A->psyRdoqScale / 256.0 : 0; param->bEnableFastIntra = A->fastIntra; param->noiseReductionIntra = param->noiseReductionInter = 0; param->limitTU = 0;
    param->interRefine = 0; param->mvRefine = 1; param->rc.bStatRead = 0; param->rdLevel = A->rdLevel; param->bEnableEarlySkip = A->earlySkip;
    param->recursionSkipMode = A->rskip; param->limitReferences = A->limitRefs; param->bIntraInBFrames = A->bIntraInB; param->bEnableRectInter = A->rect;
    param->bEnableAMP = A->amp; param->limitModes = A->limitModes; param->bCTUInfo = 0; param->bEnableRdRefine = 0; param->bOptCUDeltaQP = 0; param->csvLogLevel = 0;
    param->bEnableTransformSkip = 0; param->bLossless = 0; param->bCULossless = 0; param->bEnableSignHiding = si->signHide; param->maxTUSize = 1 << si->tuLog2Max;
    param->tuQTMaxInterDepth = si->tuMaxDepthInter; param->tuQTMaxIntraDepth = si->tuMaxDepthIntra; param->maxNumMergeCand = I->maxNumMergeCand;
    param->rc.aqMode = 0; param->rc.cuTree = 0; param->rc.qgSize = 64; param->maxCUDepth = si->maxCuDepth;
    f.sps.quadtreeTULog2MaxSize = si->tuLog2Max; f.sps.quadtreeTULog2MinSize = si->tuLog2Min;
    f.sps.quadtreeTUMaxDepthInter = si->tuMaxDepthInter; f.sps.quadtreeTUMaxDepthIntra = si->tuMaxDepthIntra; f.sps.maxAMPDepth = si->maxAmpDepth;
    f.sps.log2MinCodingBlockSize = 3; f.sps.log2DiffMaxMinCodingBlockSize = 3;
    f.sps.bUseStrongIntraSmoothing = A->strongIntraSmoothing != 0; param->bEnableStrongIntraSmoothing = A->strongIntraSmoothing; param->bEnableFastIntra = 0;
    param->bEnableConstrainedIntra = 0; param->rdPenalty = 0;
    f.pps.bUseDQP = si->useDqp != 0; f.pps.maxCuDQPDepth = si->maxCuDqpDepth; f.pps.bSignHideEnabled = si->signHide != 0;
    f.pps.bTransquantBypassEnabled = 0; f.pps.bTransformSkipEnabled = 0; f.pps.bEntropyCodingSyncEnabled = si->wpp != 0;
    Slice* slice = f.fd[0]->m_slice;
    slice->m_sliceQp = si->sliceQp;
    if (si->sliceType == 2) slice->m_sliceType = I_SLICE;
    param->bEnableSplitRdSkip = 0; param->intraRefine = 0;
    slice->m_endCUAddr = slice->realEndAddress(f.sps.numCUsInFrame * 256);
    /* the remaining CUData fields of the current picture, and the reference pictures' depth maps */
    for (uint32_t addr = 0; addr < f.sps.numCUsInFrame; addr++)
    {
        CUData& ctu = f.fd[0]->m_picCTU[addr];
        CUData& rctu = f.fd[1]->m_picCTU[addr];
        ctu.m_chromaFormat = X265_CSP_I420; ctu.m_hChromaShift = ctu.m_vChromaShift = 1;
        const int cx = (addr % f.sps.numCuInWidth) * 64, cy = (addr / f.sps.numCuInWidth) * 64;
        for (uint32_t z = 0; z < 256; z++)
        {
            const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
            if (x >= width || y >= height) { ctu.m_cuDepth[z] = 0; rctu.m_cuDepth[z] = 0; continue; }
            const RefCuUnit& u = units[(y >> 2) * w4 + (x >> 2)];
            ctu.m_cuDepth[z] = u.depth; ctu.m_log2CUSize[z] = (uint8_t)(6 - u.depth);
            ctu.m_partSize[z] = u.partSize; ctu.m_tuDepth[z] = u.tuDepth; ctu.m_lumaIntraDir[z] = u.lumaDir; ctu.m_chromaIntraDir[z] = u.chromaDir;
            ctu.m_mergeFlag[z] = u.mergeFlag; ctu.m_skipFlag[0][z] = ctu.m_skipFlag[1][z] = 0;
            for (int c = 0; c < 3; c++) ctu.m_cbf[c][z] = u.cbf[c];
            ctu.m_tqBypass[z] = 0; ctu.m_qp[z] = u.qp;
            for (int l = 0; l < 2; l++) { ctu.m_mvpIdx[l][z] = u.mvpIdx[l]; ctu.m_mvd[l][z] = MV(u.mvd[l][0], u.mvd[l][1]); }
            ctu.m_transformSkip[0][z] = ctu.m_transformSkip[1][z] = ctu.m_transformSkip[2][z] = 0;
            rctu.m_cuDepth[z] = refDepth[(y >> 2) * w4 + (x >> 2)];
        }
        rctu.m_qp[0] = refQp0[addr];
        for (int d = 0; d < 4; d++) { f.fd[0]->m_cuStat[addr].count[d] = cuStat[addr].count[d]; f.fd[0]->m_cuStat[addr].avgCost[d] = cuStat[addr].avgCost[d]; }
    }
    /* a second depth map for list 1: its own FrameData so that topSkipMinDepth sees different co-located CTUs per list */
    FrameData* fdL1 = new FrameData;
    fdL1->create(*param, f.sps, X265_CSP_I420);
    Frame frameL1;
    frameL1.m_encData = fdL1; frameL1.m_param = param;
    fdL1->m_slice->m_sps = &f.sps; fdL1->m_slice->m_pps = &f.pps; fdL1->m_slice->m_param = param;
    fdL1->m_slice->m_sliceType = f.fd[1]->m_slice->m_sliceType; fdL1->m_slice->m_poc = f.fd[1]->m_slice->m_poc;
    for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) fdL1->m_slice->m_refPOCList[l][r] = f.fd[1]->m_slice->m_refPOCList[l][r];
    for (uint32_t addr = 0; addr < f.sps.numCUsInFrame; addr++)
    {
        CUData& dst = fdL1->m_picCTU[addr];
        const CUData& src = f.fd[1]->m_picCTU[addr];
        dst.initCTU(frameL1, addr, 30, addr < f.sps.numCuInWidth, 0, 0);
        const int cx = (addr % f.sps.numCuInWidth) * 64, cy = (addr / f.sps.numCuInWidth) * 64;
        for (uint32_t z = 0; z < 256; z++)
        {
            dst.m_predMode[z] = src.m_predMode[z]; dst.m_interDir[z] = src.m_interDir[z];
            for (int l = 0; l < 2; l++) { dst.m_refIdx[l][z] = src.m_refIdx[l][z]; dst.m_mv[l][z] = src.m_mv[l][z]; }
            const int x = cx + g_zscanToPelX[z], y = cy + g_zscanToPelY[z];
            dst.m_cuDepth[z] = (x >= width || y >= height) ? 0 : refDepth[(size_t)(height >> 2) * w4 + (y >> 2) * w4 + (x >> 2)];
        }
        dst.m_qp[0] = refQp0[f.sps.numCUsInFrame + addr];
    }
    std::vector<PicYuv*> pics;
    for (int i = 0; i < S->numPics; i++) pics.push_back(mkPic(param, f.sps, planes + 3 * i, stride, cstride, width, height, marginX, marginY));
    static Frame refFrames[2][16];
    static MV noLowres(0x7FFF, 0);
    MotionReference (*mref)[MAX_NUM_REF + 1] = new MotionReference[2][MAX_NUM_REF + 1];
    slice->m_mref = mref;
    Frame& frame = f.frame[0];
    frame.m_fencPic = pics.back();
    for (int l = 0; l < 2; l++) for (int i = 0; i < X265_BFRAME_MAX + 2; i++) frame.m_lowres.lowresMvs[l][i] = &noLowres;
    f.fd[0]->m_reconPic = pics[S->numPics - 2];
    frame.m_reconPic = pics[S->numPics - 2];
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < I->numRefIdx[l]; r++)
        {
            PicYuv* rp = pics[S->refPic[l][r]];
            slice->m_refReconPicList[l][r] = rp;
            refFrames[l][r].m_encData = (l == 1 && r == 0) ? fdL1 : f.fd[1]; refFrames[l][r].m_reconPic = rp; refFrames[l][r].m_fencPic = rp; refFrames[l][r].m_param = param;
            slice->m_refFrameList[l][r] = &refFrames[l][r];
            slice->m_mref[l][r].init(rp, NULL, *param);
        }
    {
        Analysis* an = new Analysis;
        ScalingList* sl = new ScalingList;
        sl->init(); sl->m_bEnabled = false; sl->m_bDataPresent = false; sl->setupQuantMatrices(X265_CSP_I420);
        an->initSearch(*param, *sl);
        an->create(NULL);
        CUData& ctu = f.fd[0]->m_picCTU[ctuAddr];
        CUGeom geoms[CUGeom::MAX_GEOMS];
        const int cx = (ctuAddr % f.sps.numCuInWidth) * 64, cy = (ctuAddr / f.sps.numCuInWidth) * 64;
        CUData::calcCTUGeoms(X265_MIN(64, width - cx), X265_MIN(64, height - cy), 64, 8, geoms);
        Entropy start;
        start.resetEntropy(*slice);
        memcpy(start.m_contextState, ctxIn, MAX_OFF_CTX_MOD);
        start.m_fracBits = fracIn;
        ctu.initCTU(frame, ctuAddr, si->sliceQp, ctuAddr < (int)f.sps.numCuInWidth, ctuAddr / f.sps.numCuInWidth == f.sps.numCuInHeight - 1, 0);
        ctu.m_chromaFormat = X265_CSP_I420; ctu.m_hChromaShift = ctu.m_vChromaShift = 1;
        Mode& best = an->compressCTU(ctu, frame, geoms[0], start);
        memset(out, 0, sizeof(*out));
        out->rdCost = best.rdCost; out->distortion = best.distortion; out->totalBits = best.totalBits; out->fracBits = best.contexts.m_fracBits;
        memcpy(out->ctx, best.contexts.m_contextState, MAX_OFF_CTX_MOD);
        for (uint32_t z = 0; z < 256; z++)
        {
            const int ux = g_zscanToPelX[z] >> 2, uy = g_zscanToPelY[z] >> 2;
            RefCuUnit& u = unitsOut[uy * 16 + ux];
            RefMvUnit& m = mvOut[uy * 16 + ux];
            memset(&u, 0, sizeof(u)); memset(&m, 0, sizeof(m));
            if (cx + ux * 4 >= width || cy + uy * 4 >= height) continue;
            const int pm = ctu.m_predMode[z];
            u.depth = ctu.m_cuDepth[z]; u.predMode = pm == MODE_SKIP ? 3 : (pm == MODE_INTRA ? 2 : (pm == MODE_INTER ? 1 : 0));
            u.partSize = ctu.m_partSize[z]; u.tuDepth = ctu.m_tuDepth[z]; u.lumaDir = ctu.m_lumaIntraDir[z]; u.chromaDir = ctu.m_chromaIntraDir[z];
            u.mergeFlag = ctu.m_mergeFlag[z]; u.interDir = ctu.m_interDir[z];
            for (int c = 0; c < 3; c++) u.cbf[c] = ctu.m_cbf[c][z];
            u.qp = ctu.m_qp[z];
            m.predMode = u.predMode; m.interDir = u.interDir;
            for (int l = 0; l < 2; l++)
            {
                u.refIdx[l] = ctu.m_refIdx[l][z]; u.mvpIdx[l] = ctu.m_mvpIdx[l][z]; u.mvd[l][0] = ctu.m_mvd[l][z].x; u.mvd[l][1] = ctu.m_mvd[l][z].y;
                m.refIdx[l] = ctu.m_refIdx[l][z]; m.mv[l][0] = ctu.m_mv[l][z].x; m.mv[l][1] = ctu.m_mv[l][z].y;
            }
        }
        memcpy(coeffOut, ctu.m_trCoeff[0], 4096 * sizeof(int16_t));
        memcpy(coeffOut + 4096, ctu.m_trCoeff[1], 1024 * sizeof(int16_t));
        memcpy(coeffOut + 4096 + 1024, ctu.m_trCoeff[2], 1024 * sizeof(int16_t));
        for (uint32_t addr = 0; addr < f.sps.numCUsInFrame; addr++)
            for (int d = 0; d < 4; d++) { cuStat[addr].count[d] = f.fd[0]->m_cuStat[addr].count[d]; cuStat[addr].avgCost[d] = f.fd[0]->m_cuStat[addr].avgCost[d]; }
        an->destroy();
        delete an;
        delete sl;
    }
    f.fd[0]->m_reconPic = NULL; frame.m_fencPic = NULL; frame.m_reconPic = NULL;
    slice->m_mref = NULL;
    delete[] mref;
    for (size_t i = 0; i < pics.size(); i++) dropPic(pics[i]);
    frameL1.m_encData = NULL;
    fdL1->destroy(); delete fdL1;
}

/* distortion of inter prediction candidates with the reference's own classes and primitives: Predict::motionCompensation (or,
 * flags & 16, two Predict::predInterLumaPixel + pixelavg_pp as search.cpp:2499-2511) and then pu[].sad / pu[].satd /
 * cu[].sa8d (+ the 4:2:0 chroma satd / sa8d) against the source picture.  reserved[0] metric 1 SAD 2 SATD 3 SA8D,
 * reserved[1] add chroma.  out[i] = { luma, chroma } */
int ref_inter_cost_batch(const uint64_t* planes, intptr_t stride, intptr_t cstride, int picW, int picH, const PackedMcJob* jobs, int n,
                         const uint64_t* fencPlanes, intptr_t fstride, intptr_t fcstride, uint32_t* out)
{
    static McEnv* m = NULL;
    static Yuv* tmpYuv = NULL;
    TuEnv* e = tuEnv();
    if (!m) { m = new McEnv; tmpYuv = new Yuv[2]; tmpYuv[0].create(64, X265_CSP_I420); tmpYuv[1].create(64, X265_CSP_I420); }
    for (int i = 0; i < n; i++)
    {
        const PackedMcJob& j = jobs[i];
        const int part = partitionFromSizes(j.w, j.h);
        const bool luma = !!(j.flags & 1), chroma = (j.flags & 2) && !(j.flags & 16);
        if (j.flags & 16)
        {
            e->sps.picWidthInLumaSamples = picW; e->sps.picHeightInLumaSamples = picH;
            e->cu.m_encData = &m->fd;
            e->cu.m_cuPelX = j.cuX; e->cu.m_cuPelY = j.cuY;
            char pubuf[sizeof(PredictionUnit)];
            PredictionUnit* pu = (PredictionUnit*)pubuf;
            pu->ctuAddr = 0; pu->cuAbsPartIdx = 0; pu->puAbsPartIdx = 0; pu->width = j.w; pu->height = j.h;
            const int8_t refs[2] = { j.ref0, j.ref1 };
            for (int l = 0; l < 2; l++)
            {
                m->pic[l]->m_picOrg[0] = (pixel*)planes[3 * refs[l] + 0] + (intptr_t)j.y * stride + j.x;
                m->pic[l]->m_stride = stride; m->pic[l]->m_strideC = cstride;
                MV mv = l ? MV(j.mv1[0], j.mv1[1]) : MV(j.mv0[0], j.mv0[1]);
                e->cu.clipMv(mv);
                m->pred.predInterLumaPixel(*pu, tmpYuv[l], *m->pic[l], mv);
            }
            if (luma)
                g_p.pu[part].pixelavg_pp[NONALIGNED]((pixel*)j.dstY, j.dstStride, tmpYuv[0].m_buf[0], tmpYuv[0].m_size, tmpYuv[1].m_buf[0], tmpYuv[1].m_size, 32);
        }
        else
            ref_motion_compensation_batch(planes, stride, cstride, picW, picH, &j, 1);
        const pixel* fY = (const pixel*)fencPlanes[0] + (intptr_t)j.y * fstride + j.x;
        const pixel* fU = (const pixel*)fencPlanes[1] + (intptr_t)(j.y >> 1) * fcstride + (j.x >> 1);
        const pixel* fV = (const pixel*)fencPlanes[2] + (intptr_t)(j.y >> 1) * fcstride + (j.x >> 1);
        const int metric = j.reserved[0], addChroma = j.reserved[1];
        int cu = 0;
        while ((4 << cu) < j.w) cu++;
        uint32_t l = 0, c = 0;
        if (luma)
            l = metric == 1 ? g_p.pu[part].sad(fY, fstride, (pixel*)j.dstY, j.dstStride) : metric == 2 ? g_p.pu[part].satd(fY, fstride, (pixel*)j.dstY, j.dstStride)
              : metric == 3 ? g_p.cu[cu].sa8d(fY, fstride, (pixel*)j.dstY, j.dstStride) : 0;
        if (chroma && addChroma)
        {
            if (metric == 2)
                c = g_p.chroma[X265_CSP_I420].pu[part].satd(fU, fcstride, (pixel*)j.dstU, j.dstCStride) + g_p.chroma[X265_CSP_I420].pu[part].satd(fV, fcstride, (pixel*)j.dstV, j.dstCStride);
            else if (metric == 3)
                c = g_p.chroma[X265_CSP_I420].cu[cu].sa8d(fU, fcstride, (pixel*)j.dstU, j.dstCStride) + g_p.chroma[X265_CSP_I420].cu[cu].sa8d(fV, fcstride, (pixel*)j.dstV, j.dstCStride);
        }
        out[2 * i] = l; out[2 * i + 1] = c;
    }
    return n;
}

/* ---- batch forms for bench.py's cpu_baseline leg (records of include/x265amd.h; addresses are HOST addresses here) ---- */
struct PackedTuJob { uint64_t fenc, pred, coeff, resi, recon; int32_t fencStride, predStride, resiStride, reconStride;
                     uint8_t log2, ttype, intra, dir, slice, qp, signhide, reserved; };
struct PackedTuResult { uint32_t numSig, zeroEnergy, nzEnergy, reserved; uint64_t zeroDist, nzDist; };

/* the per-TU measurement of estimateResidualQT (search.cpp:3276-3330) with the reference's own functions */
int ref_tu_chain_batch(const PackedTuJob* jobs, int n, PackedTuResult* out)
{
    TuEnv* e = tuEnv();
    ALIGN_VAR_32(int16_t, resi[32 * 32]);
    for (int i = 0; i < n; i++)
    {
        const PackedTuJob& j = jobs[i];
        const pixel* fenc = (const pixel*)j.fenc; const pixel* pred = (const pixel*)j.pred;
        int sizeIdx = j.log2 - 2, N = 1 << j.log2;
        g_p.cu[sizeIdx].sub_ps(resi, N, fenc, pred, j.fencStride, j.predStride);
        e->set(j.ttype, j.intra, j.dir, j.slice, j.qp, j.signhide);
        uint32_t ns = e->quant.transformNxN(e->cu, fenc, j.fencStride, resi, N, (coeff_t*)j.coeff, j.log2, (TextType)j.ttype, 0, false);
        PackedTuResult& r = out[i];
        r.numSig = ns; r.reserved = 0;
        r.zeroDist = g_p.cu[sizeIdx].sse_pp(fenc, j.fencStride, pred, j.predStride);
        r.zeroEnergy = g_p.cu[sizeIdx].psy_cost_pp(fenc, j.fencStride, pred, j.predStride);
        r.nzDist = r.zeroDist; r.nzEnergy = r.zeroEnergy;
        if (ns)
        {
            e->quant.invtransformNxN(e->cu, (int16_t*)j.resi, j.resiStride, (coeff_t*)j.coeff, j.log2, (TextType)j.ttype, j.intra != 0, false, ns);
            g_p.cu[sizeIdx].add_ps[NONALIGNED]((pixel*)j.recon, j.reconStride, pred, (int16_t*)j.resi, j.predStride, j.resiStride);
            r.nzDist = g_p.cu[sizeIdx].sse_pp(fenc, j.fencStride, (pixel*)j.recon, j.reconStride);
            r.nzEnergy = g_p.cu[sizeIdx].psy_cost_pp(fenc, j.fencStride, (pixel*)j.recon, j.reconStride);
        }
    }
    return n;
}

/* fused intra TU job with the reference's own classes: prediction (ref_intra_predict), then the per-TU measurement with Quant
 * (optionally RDOQ) -- record layouts of include/x265amd.h with host addresses */
struct PackedIntraTuJob { PackedTuJob tu; uint64_t nb, avail; int32_t nbStride; uint8_t strong, reserved[11]; };
struct PackedTuRdoq { uint64_t est; int64_t lambda2; int32_t lambda, psyRdoqScale; uint8_t rdoqLevel, tuDepth, reserved[6]; };
int ref_intra_tu_chain_batch(const PackedIntraTuJob* jobs, const PackedTuRdoq* rq, int n, PackedTuResult* out)
{
    TuEnv* e = tuEnv();
    ALIGN_VAR_32(int16_t, resi[32 * 32]);
    ALIGN_VAR_32(pixel, predTmp[32 * 32]);
    uint8_t flags[33];
    for (int i = 0; i < n; i++)
    {
        const PackedIntraTuJob& J = jobs[i];
        const PackedTuJob& j = J.tu;
        int total = (1 << j.log2) + 1;
        for (int u = 0; u < total; u++) flags[u] = (uint8_t)((J.avail >> u) & 1);
        pixel* pred = j.pred ? (pixel*)j.pred : predTmp;
        intptr_t ps = j.pred ? j.predStride : 32;
        ref_intra_predict((const pixel*)J.nb, J.nbStride, j.log2, flags, J.strong, j.ttype != 0, j.dir, pred, ps);
        const pixel* fenc = (const pixel*)j.fenc;
        int sizeIdx = j.log2 - 2, N = 1 << j.log2;
        g_p.cu[sizeIdx].sub_ps(resi, N, fenc, pred, j.fencStride, ps);
        e->set(j.ttype, 1, j.dir, j.slice, j.qp, j.signhide);
        if (rq && rq[i].rdoqLevel)
        {
            e->quant.configureRdoq(rq[i].rdoqLevel, rq[i].psyRdoqScale);
            memset(e->tuDepth, rq[i].tuDepth, sizeof(e->tuDepth));
            memcpy(&e->entropy.m_estBitsSbac, (const void*)rq[i].est, sizeof(EstBitsSbac));
        }
        uint32_t ns = e->quant.transformNxN(e->cu, fenc, j.fencStride, resi, N, (coeff_t*)j.coeff, j.log2, (TextType)j.ttype, 0, false);
        e->quant.configureRdoq(0, 0);
        memset(e->tuDepth, 0, sizeof(e->tuDepth));
        PackedTuResult& r = out[i];
        r.numSig = ns; r.reserved = 0;
        r.zeroDist = g_p.cu[sizeIdx].sse_pp(fenc, j.fencStride, pred, ps);
        r.zeroEnergy = g_p.cu[sizeIdx].psy_cost_pp(fenc, j.fencStride, pred, ps);
        r.nzDist = r.zeroDist; r.nzEnergy = r.zeroEnergy;
        if (ns)
        {
            e->quant.invtransformNxN(e->cu, (int16_t*)j.resi, j.resiStride, (coeff_t*)j.coeff, j.log2, (TextType)j.ttype, true, false, ns);
            g_p.cu[sizeIdx].add_ps[NONALIGNED]((pixel*)j.recon, j.reconStride, pred, (int16_t*)j.resi, ps, j.resiStride);
            r.nzDist = g_p.cu[sizeIdx].sse_pp(fenc, j.fencStride, (pixel*)j.recon, j.reconStride);
            r.nzEnergy = g_p.cu[sizeIdx].psy_cost_pp(fenc, j.fencStride, (pixel*)j.recon, j.reconStride);
        }
        else
        {
            for (int y = 0; y < N; y++)
                for (int x = 0; x < N; x++) { ((pixel*)j.recon)[y * j.reconStride + x] = pred[y * ps + x]; ((int16_t*)j.resi)[y * j.resiStride + x] = 0; }
        }
    }
    return n;
}

struct PackedIntraJob { uint64_t recon, fenc, avail; int32_t reconStride, fencStride; uint8_t log2, strong, reserved[6]; };
int ref_intra_scan_batch(const PackedIntraJob* jobs, int n, int32_t* sa8d)
{
    pixel rb[258], fb[258];
    uint8_t flags[33];
    for (int i = 0; i < n; i++)
    {
        const PackedIntraJob& j = jobs[i];
        int total = (1 << j.log2) + 1;
        for (int u = 0; u < total; u++) flags[u] = (j.avail >> u) & 1;
        ref_init_adi_pattern((const pixel*)j.recon, j.reconStride, j.log2, flags, j.strong, -1, rb, fb);
        ref_intra_scan((const pixel*)j.fenc, j.fencStride, j.log2, rb, j.log2 >= 3 ? fb : rb, sa8d + 35 * i);
    }
    return n;
}

/* ---- lookahead lowres pipeline: the reference's own Lowres::init steps and LookaheadTLD::lowresIntraEstimate ---- */
/* src / planes[k]: sample (0,0) of padded planes */
void ref_lowres_init(const pixel* src, intptr_t srcStride, int width, int height, pixel* p0, pixel* ph, pixel* pv, pixel* pc, intptr_t stride, int marginX, int marginY)
{
    ensure();
    g_p.frameInitLowres(src, p0, ph, pv, pc, srcStride, stride, width, height);      /* lowres.cpp:368-376 */
    extendPicBorder(p0, stride, width, height, marginX, marginY);
    extendPicBorder(ph, stride, width, height, marginX, marginY);
    extendPicBorder(pv, stride, width, height, marginX, marginY);
    extendPicBorder(pc, stride, width, height, marginX, marginY);
}

/* out: intraCost[ncu], intraMode[ncu], rowSatds[heightInCU], lowresCosts[ncu]; sums[0] = costEst[0][0], sums[1] = costEstAq[0][0].  No AQ (invQscaleFactor NULL). */
void ref_lowres_intra(pixel* plane0, intptr_t stride, int widthInCU, int heightInCU, int32_t* intraCost, uint8_t* intraMode, int32_t* rowSatds, uint16_t* lowresCosts,
                      int64_t* sums)
{
    ensure();
    static LookaheadTLD* tld = new LookaheadTLD;
    tld->init(widthInCU, heightInCU, widthInCU * heightInCU);
    Lowres* fenc = (Lowres*)calloc(1, sizeof(Lowres));
    fenc->lowresPlane[0] = plane0; fenc->lumaStride = stride;
    fenc->intraCost = intraCost; fenc->intraMode = intraMode;
    fenc->rowSatds[0][0] = rowSatds; fenc->lowresCosts[0][0] = lowresCosts;
    fenc->invQscaleFactor = NULL;
    tld->lowresIntraEstimate(*fenc, 16);
    sums[0] = fenc->costEst[0][0]; sums[1] = fenc->costEstAq[0][0];
    free(fenc);
}

/* ---- lookahead frame cost: the reference's own CostEstimateGroup::estimateFrameCost (slicetype.cpp:3976-4075) on Lowres objects made by
 * Lowres::create / init from three padded luma planes (frames[0..2] in display order).  No AQ, weighted prediction, HME or cooperative slices. ---- */
namespace {
struct CostGroup : public CostEstimateGroup
{
    CostGroup(Lookahead& l, Lowres** f) : CostEstimateGroup(l, f) {}
    int64_t run(LookaheadTLD& tld, int p0, int p1, int b) { return estimateFrameCost(tld, p0, p1, b, false); }
};
}
/* luma[k]: sample (0,0) of frame k's padded luma plane.  out arrays sized by the 8x8 grid of the half-resolution picture (ncu = wcu * hcu):
 * lowresCosts u16[ncu], mvs i16[2][ncu][2], mvCosts i32[2][ncu], intraCost i32[ncu], rowSatds i32[hcu]; sums: score, costEst before scaling is not kept by the
 * reference, so sums[0] = returned score, sums[1] = costEstAq, sums[2] = intraMbs[b - p0] */
int ref_lowres_frame_cost(const pixel* const* luma, intptr_t stride, int width, int height, int marginX, int marginY, int p0, int b, int p1, int bframes,
                          uint16_t* lowresCosts, int16_t* mvs, int32_t* mvCosts, int32_t* intraCost, int32_t* rowSatds, int64_t* sums)
{
    ensure();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420; param->bframes = bframes;
    param->bEnableWeightedPred = 0; param->bEnableWeightedBiPred = 0; param->rc.aqMode = 0; param->rc.cuTree = 0; param->lookaheadSlices = 0; param->bEnableHME = 0;
    param->maxSlices = 1; param->bFrameBias = 0; param->rc.qgSize = 32; param->bEnableTemporalFilter = 0;
    Lookahead* la = new Lookahead(param, NULL);
    Lowres* frames[3];
    PicYuv* pics[3];
    for (int k = 0; k < 3; k++)
    {
        pics[k] = new PicYuv;
        pics[k]->m_param = param; pics[k]->m_picWidth = width; pics[k]->m_picHeight = height; pics[k]->m_lumaMarginX = marginX; pics[k]->m_lumaMarginY = marginY;
        pics[k]->m_stride = stride; pics[k]->m_picOrg[0] = const_cast<pixel*>(luma[k]);
        frames[k] = new Lowres();
        if (!frames[k]->create(param, pics[k], param->rc.qgSize)) return -1;
        frames[k]->init(pics[k], k);
    }
    const int wcu = la->m_8x8Width, hcu = la->m_8x8Height, ncu = wcu * hcu;
    LookaheadTLD* tld = new LookaheadTLD;
    tld->init(wcu, hcu, ncu);
    for (int k = 0; k < 3; k++) tld->lowresIntraEstimate(*frames[k], param->rc.qgSize);       /* PreLookaheadGroup::processTasks (slicetype.cpp:1745-1762) */
    CostGroup g(*la, frames);
    const int64_t score = g.run(*tld, p0, p1, b);
    Lowres* f = frames[b];
    memcpy(lowresCosts, f->lowresCosts[b - p0][p1 - b], sizeof(uint16_t) * ncu);
    for (int l = 0; l < 2; l++)
    {
        const int dist = l ? p1 - b : b - p0;
        for (int i = 0; i < ncu; i++)
        {
            const bool have = l ? p1 > b : b > p0;
            mvs[(l * ncu + i) * 2] = have ? f->lowresMvs[l][dist][i].x : 0; mvs[(l * ncu + i) * 2 + 1] = have ? f->lowresMvs[l][dist][i].y : 0;
            mvCosts[l * ncu + i] = have ? f->lowresMvCosts[l][dist][i] : 0;
        }
    }
    memcpy(intraCost, f->intraCost, sizeof(int32_t) * ncu);
    memcpy(rowSatds, f->rowSatds[b - p0][p1 - b], sizeof(int32_t) * hcu);
    sums[0] = score; sums[1] = f->costEstAq[b - p0][p1 - b]; sums[2] = f->intraMbs[b - p0];
    for (int k = 0; k < 3; k++) { frames[k]->destroy(param); delete frames[k]; pics[k]->m_picOrg[0] = NULL; }
    delete tld;
    return ncu;
}

/* ---- adaptive quantisation, block energies: the reference's own LookaheadTLD::acEnergyCu over a picture (slicetype.cpp:264-283) ---- */
int ref_aq_energy(const pixel* y, const pixel* u, const pixel* v, intptr_t stride, intptr_t cstride, int width, int height, int qgSize, uint32_t* energy, uint64_t* wp)
{
    ensure();
    struct Tld : public LookaheadTLD { uint32_t energy(Frame* f, uint32_t x, uint32_t y, uint32_t qg) { return acEnergyCu(f, x, y, X265_CSP_I420, qg); } };
    static Tld* tld = new Tld;
    Frame* frame = new Frame;
    PicYuv* pic = new PicYuv;
    pic->m_picWidth = width; pic->m_picHeight = height; pic->m_stride = stride; pic->m_strideC = cstride; pic->m_picCsp = X265_CSP_I420;
    pic->m_picOrg[0] = const_cast<pixel*>(y); pic->m_picOrg[1] = const_cast<pixel*>(u); pic->m_picOrg[2] = const_cast<pixel*>(v);
    frame->m_fencPic = pic;
    for (int p = 0; p < 3; p++) { frame->m_lowres.wp_ssd[p] = 0; frame->m_lowres.wp_sum[p] = 0; }
    int n = 0;
    for (int by = 0; by < height; by += qgSize)
        for (int bx = 0; bx < width; bx += qgSize)
            energy[n++] = tld->energy(frame, bx, by, qgSize);
    for (int p = 0; p < 3; p++) { wp[p] = frame->m_lowres.wp_sum[p]; wp[3 + p] = frame->m_lowres.wp_ssd[p]; }
    frame->m_fencPic = NULL;
    pic->m_picOrg[0] = pic->m_picOrg[1] = pic->m_picOrg[2] = NULL;
    delete pic;
    return n;
}

/* ---- adaptive quantisation, whole function: the reference's own LookaheadTLD::calcAdaptiveQuantFrame (slicetype.cpp:452-713) on a Frame fixture ---- */
int ref_aq_frame(const pixel* y, const pixel* u, const pixel* v, intptr_t stride, intptr_t cstride, int width, int height, int marginX, int marginY, int aqMode,
                 double aqStrength, double aqBiasStrength, int qgSize, double* qpAqOffset, double* qpCuTreeOffset, int32_t* invQscaleFactor)
{
    ensure();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420; param->bframes = 0;
    param->rc.aqMode = aqMode; param->rc.aqStrength = aqStrength; param->rc.aqBiasStrength = aqBiasStrength; param->rc.qgSize = qgSize; param->rc.hevcAq = 0;
    param->bEnableWeightedPred = 0; param->bEnableWeightedBiPred = 0; param->bHDR10Opt = 0; param->bDynamicRefine = 0; param->bEnableFades = 0; param->rc.bStatRead = 0;
    param->bAQMotion = 0; param->bEnableHME = 0; param->recursionSkipMode = 1;
    PicYuv* pic = new PicYuv;
    pic->m_param = param; pic->m_picWidth = width; pic->m_picHeight = height; pic->m_stride = stride; pic->m_strideC = cstride; pic->m_picCsp = X265_CSP_I420;
    pic->m_lumaMarginX = marginX; pic->m_lumaMarginY = marginY;
    pic->m_picOrg[0] = const_cast<pixel*>(y); pic->m_picOrg[1] = const_cast<pixel*>(u); pic->m_picOrg[2] = const_cast<pixel*>(v);
    Frame* frame = new Frame;
    frame->m_fencPic = pic; frame->m_param = param; frame->m_quantOffsets = NULL;
    if (!frame->m_lowres.create(param, pic, qgSize)) return -1;
    LookaheadTLD* tld = new LookaheadTLD;
    const int wcu = ((width / 2) + 7) >> 3, hcu = ((height / 2) + 7) >> 3;
    tld->init(wcu, hcu, wcu * hcu);
    tld->calcAdaptiveQuantFrame(frame, param);
    const int n = qgSize == 8 ? frame->m_lowres.maxBlocksInRowFullRes * frame->m_lowres.maxBlocksInColFullRes : wcu * hcu;
    memcpy(qpAqOffset, frame->m_lowres.qpAqOffset, sizeof(double) * n);
    memcpy(qpCuTreeOffset, frame->m_lowres.qpCuTreeOffset, sizeof(double) * n);
    memcpy(invQscaleFactor, frame->m_lowres.invQscaleFactor, sizeof(int32_t) * n);
    frame->m_lowres.destroy(param);
    frame->m_fencPic = NULL;
    pic->m_picOrg[0] = pic->m_picOrg[1] = pic->m_picOrg[2] = NULL;
    delete pic; delete tld;
    return n;
}

/* ---- cuTree: the reference's own Lookahead::cuTree (slicetype.cpp:3399-3500, with estimateCUPropagate, cuTreeFinish and primitives.propagateCost) on Lowres objects whose
 * cost estimates are handed in: n = numframes + 1 pictures (frames[0] = the last non-B picture), per picture the slice type, Lowres::intraCost, invQscaleFactor, qpAqOffset,
 * qpCuTreeOffset (in / out), propagateCost (in / out), weightedCostDelta[18]; numEst estimates (p0, p1, b) with lowresCosts and the two motion fields, entered as if
 * estimateFrameCost had made them (costEst >= 0, rowSatds set: singleCost finds them and does nothing).  Arrays per 8x8 block of the half-resolution picture. ---- */
namespace {
struct TreeLookahead : public Lookahead
{
    TreeLookahead(x265_param* p) : Lookahead(p, NULL) {}
    void tree(Lowres** frames, int numframes, bool bIntra) { cuTree(frames, numframes, bIntra); }
    int64_t recalc(Lowres** frames, int p0, int p1, int b) { return frameCostRecalculate(frames, p0, p1, b); }
};
}
int ref_cutree(int width, int height, int bframes, int bpyramid, int weightb, int lookaheadDepth, double qCompress, int fpsNum, int fpsDenom, int numframes, int bIntra,
               const int32_t* sliceTypes, const int32_t* intraCost, const int32_t* invQscale, const double* qpAq, double* qpCuTree, uint16_t* propagate, const double* weightedCostDelta,
               int numEst, const int32_t* estIdx, const uint16_t* lowresCosts, const int16_t* mvs0, const int16_t* mvs1, int64_t* recalc)
{
    ensure();
    x265_param* param = x265_param_alloc();
    x265_param_default(param);
    param->sourceWidth = width; param->sourceHeight = height; param->internalCsp = X265_CSP_I420; param->bframes = bframes; param->bBPyramid = bpyramid;
    param->bEnableWeightedPred = 0; param->bEnableWeightedBiPred = weightb; param->rc.aqMode = 2; param->rc.cuTree = 1; param->lookaheadSlices = 0; param->bEnableHME = 0;
    param->rc.qCompress = qCompress; param->fpsNum = fpsNum; param->fpsDenom = fpsDenom; param->lookaheadDepth = lookaheadDepth; param->rc.hevcAq = 0;
    param->maxSlices = 1; param->bFrameBias = 0; param->rc.qgSize = 32; param->bEnableTemporalFilter = 0; param->rc.vbvBufferSize = 0;
    TreeLookahead* la = new TreeLookahead(param);
    if (!la->create()) return -1;
    const int n = numframes + 1;
    const int wcu = la->m_8x8Width, hcu = la->m_8x8Height, ncu = wcu * hcu;
    std::vector<Lowres*> frames(n + 2, (Lowres*)NULL);
    std::vector<PicYuv*> pics(n);
    for (int k = 0; k < n; k++)
    {
        pics[k] = new PicYuv;
        pics[k]->m_param = param; pics[k]->m_picWidth = width; pics[k]->m_picHeight = height; pics[k]->m_lumaMarginX = 96; pics[k]->m_lumaMarginY = 80;
        pics[k]->m_stride = width + 192; pics[k]->m_picOrg[0] = NULL;
        Lowres* f = new Lowres();
        if (!f->create(param, pics[k], param->rc.qgSize)) return -1;
        f->frameNum = k; f->sliceType = sliceTypes[k];
        memset(f->costEst, -1, sizeof(f->costEst)); memset(f->costEstAq, -1, sizeof(f->costEstAq));
        for (int y = 0; y < bframes + 2; y++) for (int x = 0; x < bframes + 2; x++) f->rowSatds[y][x][0] = -1;
        for (int i = 0; i < bframes + 2; i++) { f->lowresMvs[0][i][0].x = 0x7FFF; f->lowresMvs[1][i][0].x = 0x7FFF; }
        memcpy(f->intraCost, intraCost + (size_t)k * ncu, sizeof(int32_t) * ncu);
        memcpy(f->invQscaleFactor, invQscale + (size_t)k * ncu, sizeof(int32_t) * ncu);
        memcpy(f->qpAqOffset, qpAq + (size_t)k * ncu, sizeof(double) * ncu);
        memcpy(f->qpCuTreeOffset, qpCuTree + (size_t)k * ncu, sizeof(double) * ncu);
        memcpy(f->propagateCost, propagate + (size_t)k * ncu, sizeof(uint16_t) * ncu);
        for (int i = 0; i < bframes + 2 && i < 18; i++) f->weightedCostDelta[i] = weightedCostDelta[(size_t)k * 18 + i];
        frames[k] = f;
    }
    for (int e = 0; e < numEst; e++)
    {
        const int p0 = estIdx[3 * e], p1 = estIdx[3 * e + 1], b = estIdx[3 * e + 2];
        Lowres* f = frames[b];
        f->costEst[b - p0][p1 - b] = 1; f->costEstAq[b - p0][p1 - b] = 1; f->rowSatds[b - p0][p1 - b][0] = 0;
        memcpy(f->lowresCosts[b - p0][p1 - b], lowresCosts + (size_t)e * ncu, sizeof(uint16_t) * ncu);
        if (b > p0) for (int i = 0; i < ncu; i++) { f->lowresMvs[0][b - p0][i].x = mvs0[((size_t)e * ncu + i) * 2]; f->lowresMvs[0][b - p0][i].y = mvs0[((size_t)e * ncu + i) * 2 + 1]; }
        if (p1 > b) for (int i = 0; i < ncu; i++) { f->lowresMvs[1][p1 - b][i].x = mvs1[((size_t)e * ncu + i) * 2]; f->lowresMvs[1][p1 - b][i].y = mvs1[((size_t)e * ncu + i) * 2 + 1]; }
    }
    la->tree(frames.data(), numframes, bIntra != 0);
    for (int k = 0; k < n; k++)
    {
        memcpy(qpCuTree + (size_t)k * ncu, frames[k]->qpCuTreeOffset, sizeof(double) * ncu);
        memcpy(propagate + (size_t)k * ncu, frames[k]->propagateCost, sizeof(uint16_t) * ncu);
    }
    /* frameCostRecalculate of every estimate handed in whose picture is not a B picture */
    for (int e = 0; e < numEst && recalc; e++)
    {
        const int p0 = estIdx[3 * e], p1 = estIdx[3 * e + 1], b = estIdx[3 * e + 2];
        recalc[e] = frames[b]->sliceType == X265_TYPE_B ? -1 : la->recalc(frames.data(), p0, p1, b);
    }
    for (int k = 0; k < n; k++) { frames[k]->destroy(param); delete frames[k]; delete pics[k]; }
    return ncu;
}

} /* extern "C" */
