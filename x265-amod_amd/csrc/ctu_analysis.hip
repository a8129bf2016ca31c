/* CTU mode decision of inter slices (include/x265amd.h: x265amd_compress_ctu_inter): SURVEY rows a1 / a2.
 *
 * Restatement of Analysis::compressCTU (reference: source/encoder/analysis.cpp:138-317) -> compressInterCU_rd0_4 (:1146-1848) at RD
 * levels 3-4 with checkMerge2Nx2N_rd0_4 (:2750-2880), checkInter_rd0_4 (:3023-3085), checkBidir2Nx2N (:3145-3277), topSkipMinDepth
 * (:3428-3476), recursionDepthCheck (:3479-3534), addSplitFlagCost (:3405-3426), checkBestMode (analysis.h:211-221), Mode::addSubCosts
 * (search.h:145-159) and CUData::copyPartFrom / copyToPic.
 *
 * The recursion, the candidate order and every comparison are the reference's and run on the host; every block operation is one of
 * the batch entry points of this library: x265amd_merge_candidates (host), x265amd_motion_compensation + x265amd_measure_tiles (merge
 * candidates, bi-prediction tries, SA8D of finished predictions), x265amd_pred_inter_search_ex (the 2Nx2N search),
 * x265amd_skip_rd / x265amd_inter_residual_rd / x265amd_intra_in_inter (RD of the chosen candidates).  Each mode of each depth owns a prediction and a
 * reconstruction tile in device memory; the best mode's units, motion and reconstruction are written into the picture maps / the
 * reconstructed picture at the end of every CU, exactly when the reference's copyToPic does, so later neighbours see what the
 * reference's would.
 *
 * I slices take compressIntraCU (:514-668): checkIntra 2Nx2N (+ NxN at 8x8) through x265amd_check_intra, then the four sub-CUs.
 *
 * Scope of this entry point: I, P and B slices (intra candidates through x265amd_intra_in_inter; --b-intra on / off), 2Nx2N, rectangular
 * (--rect) and asymmetric (--amp) partitions with --limit-modes, --limit-refs 0-3, no delta QP (aq-mode 0, no cutree), rd 3-4, rskip 0/1, early skip on/off.
 */
#include "x265amd_dev.h"
#include "inter_common.h"
#include "xa_queue.h"
#include "../host/cabac_coder.h"
#include "inter_chain_dev.h"
#include "inter_search_dev.h"
#include <string.h>
#include <immintrin.h>
#include <vector>

using namespace xa_inter;

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include "xa_fiber.h"
#include <mutex>
#include <thread>
/* X265AMD_TIMING=1: wall time per analysis stage, printed per frame by x265amd_analyse_frame */
static double g_stageMs[8];
static const char* const g_stageName[8] = { "merge", "search", "rdInter", "rdIntra", "bidir", "copies", "intraSlice", "other" };
static bool g_timing = getenv("X265AMD_TIMING") != nullptr;
static std::atomic<uint64_t> g_aheadStat[4];      /* X265AMD_TIMING: searches started ahead of a leaf's merge check, searches collected, started behind the merge check of a CU with sub-CUs, results the sub-CUs' restriction ruled out */
static std::atomic<uint64_t> g_chainStat[4], g_chainTicks[8], g_cuStat[2][4][4];      /* [B / P][depth][skipped on the device, merge check on the host, search, intra try] */     /* X265AMD_TIMING: skip chains run, CUs they skipped, stops (not a skip / vector beyond what is published); the device's stage clock */
struct StageTimer
{
    int k; std::chrono::steady_clock::time_point t0;
    explicit StageTimer(int k_) : k(k_) { if (g_timing) t0 = std::chrono::steady_clock::now(); }
    ~StageTimer() { if (g_timing) g_stageMs[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

/* ---- reference-picture guards (x265amd_host.h): the row task's hooks live in its task slot ---- */
static int xa_ref_guard_wait(int pic, int yMin, int yMax, int xMax)
{
    void** slot = xa_task_slot();
    const XaRowHooks* h = slot ? (const XaRowHooks*)*slot : nullptr;
    if (!h || !h->ref_wait) return 0;
    return h->ref_wait(h->ctx, pic, yMin, yMax, xMax);
}
int xa_ref_guard_mc(const x265amd_mc_job* jobs, int n)
{
    void** slot = xa_task_slot();
    if (!slot || !*slot) return 0;
    int yMin[32], yMax[32], xMax[32];
    for (int k = 0; k < 32; k++) { yMin[k] = 1 << 30; yMax[k] = -(1 << 30); xMax[k] = -(1 << 30); }
    for (int i = 0; i < n; i++)
    {
        const x265amd_mc_job& j = jobs[i];
        const int refs[2] = { j.ref0, j.ref1 };
        const int16_t* mv[2] = { j.mv0, j.mv1 };
        for (int l = 0; l < 2; l++)
        {
            if (refs[l] < 0 || refs[l] >= 32) continue;
            /* the block moved by the vector's full-sample part (floor) with the taps of the 8-tap luma filter around it (chroma reaches no further in luma terms) */
            const int y0 = j.y + (mv[l][1] >> 2) - 4, y1 = j.y + j.h - 1 + (mv[l][1] >> 2) + 5, x1 = j.x + j.w - 1 + (mv[l][0] >> 2) + 5;
            if (y0 < yMin[refs[l]]) yMin[refs[l]] = y0;
            if (y1 > yMax[refs[l]]) yMax[refs[l]] = y1;
            if (x1 > xMax[refs[l]]) xMax[refs[l]] = x1;
        }
    }
    for (int k = 0; k < 32; k++) if (yMax[k] > -(1 << 30)) if (xa_ref_guard_wait(k, yMin[k], yMax[k], xMax[k])) return -1;
    return 0;
}
int xa_ref_guard_me(const x265amd_me_job* jobs, const int* pics, int n)
{
    void** slot = xa_task_slot();
    if (!slot || !*slot) return 0;
    for (int i = 0; i < n; i++)
    {
        const x265amd_me_job& j = jobs[i];
        /* the search area with the interpolation margins and what the pattern / sub-sample refinement may step outside (x265amd_me_plan's bounds) */
        if (xa_ref_guard_wait(pics[i], j.y + j.mvmin[1] - 8, j.y + j.h + j.mvmax[1] + 8, j.x + j.w + j.mvmax[0] + 10)) return -1;
    }
    return 0;
}

/* ---- device-resident motion maps (inter_chain_dev.h): a host motion field (x265amd_mv_unit array) may have a mirror in device memory the host writes through the BAR;
 * the encoder object registers one per picture, frame-level callers without one get a temporary mirror (xa_analyse_frame) ---- */
namespace {
/* blocks are never given back to the runtime while the process lives (hipFree waits for the device, and the job server's kernel is resident): an entry
 * without a host field is a free block of `units` units */
struct DevMapEntry { std::atomic<const void*> host{ nullptr }; void* dev = nullptr; size_t units = 0; };
DevMapEntry g_devMaps[512];
std::mutex g_devMapM;
}
void* xa_devmap_register(const x265amd_mv_unit* host, size_t units)
{
    if (!host || !units || !xa_queues_enabled()) return nullptr;
    std::lock_guard<std::mutex> g(g_devMapM);
    DevMapEntry* blank = nullptr;
    for (DevMapEntry& e : g_devMaps)
    {
        if (e.host.load(std::memory_order_relaxed)) continue;
        if (e.dev && e.units == units) { e.host.store(host, std::memory_order_release); return e.dev; }
        if (!e.dev && !blank) blank = &e;
    }
    if (!blank) return nullptr;
    void* d = nullptr;
    if (hipExtMallocWithFlags(&d, units * sizeof(XaMapUnit), hipDeviceMallocUncached) != hipSuccess) return nullptr;
    blank->dev = d; blank->units = units;
    blank->host.store(host, std::memory_order_release);
    return d;
}
void xa_devmap_unregister(const x265amd_mv_unit* host)
{
    if (!host) return;
    std::lock_guard<std::mutex> g(g_devMapM);
    for (DevMapEntry& e : g_devMaps)
        if (e.host.load(std::memory_order_relaxed) == host) { e.host.store(nullptr, std::memory_order_release); return; }
}
void* xa_devmap_find(const x265amd_mv_unit* host)
{
    if (!host) return nullptr;
    for (DevMapEntry& e : g_devMaps) if (e.host.load(std::memory_order_acquire) == host) return e.dev;
    return nullptr;
}
static inline void devmap_store(XaMapUnit* d, const x265amd_mv_unit& v, int depth)
{
    union { XaMapUnit u; uint64_t w[2]; } t;
    t.w[0] = t.w[1] = 0;
    t.u.pred_mode = v.pred_mode; t.u.inter_dir = v.inter_dir; t.u.ref_idx[0] = v.ref_idx[0]; t.u.ref_idx[1] = v.ref_idx[1];
    t.u.mv[0][0] = v.mv[0][0]; t.u.mv[0][1] = v.mv[0][1]; t.u.mv[1][0] = v.mv[1][0]; t.u.mv[1][1] = v.mv[1][1]; t.u.depth = (uint8_t)depth;
    volatile uint64_t* q = reinterpret_cast<volatile uint64_t*>(d);
    q[0] = t.w[0]; q[1] = t.w[1];
}
/* rows [y4a, y4b) of a host field into its mirror (a picture whose rows arrive from another process: x265amd_encoder_import_row) */
void xa_devmap_push_rows(const x265amd_mv_unit* host, const x265amd_cu_unit* units, int w4, int y4a, int y4b)
{
    XaMapUnit* d = (XaMapUnit*)xa_devmap_find(host);
    if (!d) return;
    for (int i = y4a * w4; i < y4b * w4; i++) devmap_store(d + i, host[i], units ? units[i].depth : 0);
    /* the mirror is written through the write-combining BAR mapping and the caller publishes the rows to OTHER threads next (a release fence drains nothing on
     * x86): the units must have left this core's buffers before a row task on another core can queue a command that reads them */
    _mm_sfence();
}

namespace {

#if X265AMD_DEPTH < 10
typedef uint32_t sse_t;
#else
typedef uint64_t sse_t;
#endif

enum { PRED_MERGE, PRED_SKIP, PRED_2Nx2N, PRED_BIDIR, PRED_INTRA, PRED_INTRA_NxN, PRED_2NxN, PRED_Nx2N, PRED_2NxnU, PRED_2NxnD, PRED_nLx2N, PRED_nRx2N, PRED_SPLIT, NUM_PRED };
struct SplitData { uint32_t splitRefs, mvCost[2]; uint64_t sa8dCost; };
const uint64_t kMaxCost = 0x7FFFFFFFFFFFFFFFULL;
const int kTileElems = 4096 + 2048;

struct Snap { uint8_t ctx[X265AMD_CTX_STRIDE]; uint64_t frac; };

struct Mode
{
    x265amd_cu_unit u[256];         /* the CU's units, raster with row length size/4 */
    x265amd_mv_unit m[256];
    std::vector<int16_t> coeff;     /* CUData::m_trCoeff layout */
    Snap contexts;
    uint64_t rdCost, sa8dCost;
    uint32_t sa8dBits, psyEnergy, totalBits, mvBits, coeffBits;
    sse_t resEnergy, lumaDistortion, chromaDistortion, distortion;
    int predTile, reconTile;        /* tile indices in the device arena */
    Mode() : coeff(kTileElems, 0) { initCosts(); }
    void initCosts()
    {
        rdCost = 0; sa8dCost = 0; sa8dBits = 0; psyEnergy = 0; resEnergy = 0; lumaDistortion = 0; chromaDistortion = 0; distortion = 0;
        totalBits = 0; mvBits = 0; coeffBits = 0;
    }
    void addSubCosts(const Mode& s)
    {
        rdCost += s.rdCost; sa8dCost += s.sa8dCost; sa8dBits += s.sa8dBits; psyEnergy += s.psyEnergy; resEnergy += s.resEnergy;
        lumaDistortion += s.lumaDistortion; chromaDistortion += s.chromaDistortion; distortion += s.distortion;
        totalBits += s.totalBits; mvBits += s.mvBits; coeffBits += s.coeffBits;
    }
    bool isSkipped() const { return u[0].pred_mode == X265AMD_MODE_SKIP; }
};

struct ModeDepth { Mode pred[NUM_PRED]; Mode* best; Snap cur; uint32_t mvCost2Nx2N[2]; x265amd_me_detail det; uint32_t srcMean, srcHomo; };

struct DevBuf
{
    void* p = nullptr;
    ~DevBuf() { xa_scratch_free(p); }
    hipError_t alloc(size_t bytes) { return xa_scratch_alloc(&p, bytes ? bytes : 16); }
};

struct Analyzer
{
    x265amd_me_ctx* me; hipStream_t st;
    const x265amd_mvpred_info* I; const x265amd_inter_search_params* S; const x265amd_slice_info* si; const x265amd_analysis_params* A;
    x265amd_cu_unit* units; x265amd_mv_unit* cur; const x265amd_mv_unit* col;
    const uint8_t* refDepth; const int8_t* refQp0;
    const uint64_t* planes; int numPics; intptr_t stride, cstride;
    x265amd_cu_stat* cuStat; int ctuAddr, ctuX, ctuY, ctuW, w4, h4, qp;
    ModeDepth md[4];
    DevBuf dTiles, dPlanes;
    XaMapped dJobs;                         /* motion compensation jobs: host memory the kernel reads in place */
    size_t tileBytes;
    uint64_t lambda2, lambda; uint32_t psyRd;
    x265amd_rd_params rp;
    int err;
    /* ---- delta QP (pps.bUseDQP): `qp` above is the QP in force -- what Search::setLambdaFromQP was last called with (search.cpp:177-187): the lambdas, the quantiser and the
     * motion costs follow it.  A CU's own QP is the value in force when compress() is entered: set by its parent for a CU at the quantisation groups' depth or above, inherited
     * from its group below that. ---- */
    const int8_t* cuQp = nullptr;           /* this CTU's group QPs (xa_analyse_frame's cu_qp): [0] the 64x64 CU, [1 + q] its 32x32 CUs */
    int ctuQp = 0;                          /* CUData::m_qp[0] of the CTU as compressCTU sets it (topSkipMinDepth's currentQP) */
    x265amd_cabac* qpCoder = nullptr;       /* a bit-counting coder on the picture map: getRefQP and the price of cu_qp_delta (checkDQPForSplitPred) */
    int lambdaQp = 0;                       /* the QP setLambdaFromQP was given (up to 69): the lambdas' and the motion costs'; `qp` is the quantiser's and the coded one, clipped to 51 */
    int setLambdaFromQP(int q)
    {
        if (q < 0 || q > 69) return fail("a CU's QP outside 0..69");
        lambdaQp = q;
        qp = q > 51 ? 51 : q;
        /* (the device-run inter chains carry one QP: none of them is on under delta QP, and without it no QP passes 51 -- the slice QP is clipped) */
        if (lambdaQp != qp && chain.on) return fail("a QP above 51 on a device-run chain");
        uint64_t rd[6];
        x265amd_rdcost(lambdaQp, si->slice_type, A->psy_rd, 0, 0, 0, rd);
        lambda2 = rd[0]; lambda = rd[1]; psyRd = (uint32_t)rd[2];
        return 0;
    }
    void cuQp2(x265amd_rd_cu& c) const { c.qp = (int8_t)qp; c.reserved[0] = (uint8_t)(lambdaQp != qp ? lambdaQp : 0); c.reserved[1] = (uint8_t)(maxTUDepth + 1); }
    /* --limit-tu 3 / 4: Search::m_maxTUDepth as the recursion leaves it (loaded for every CU of 16x16 and up, never restored: a CU's own modes behind its sub-CUs see the
     * last sub-CU's value), and the per-CTU records it is loaded from (CUData::m_refTuDepth[geomRecurId]: the deepest transform unit of the CU decided at that place) */
    int maxTUDepth = -1;
    const XaTuRecs* tuRecs = nullptr;
    static int geomId(int lx, int ly, int depth) { return depth == 0 ? 0 : depth == 1 ? 1 + (ly >> 5) * 2 + (lx >> 5) : 5 + (ly >> 4) * 4 + (lx >> 4); }        /* calcCTUGeoms: raster inside a depth */
    /* Analysis::loadTUDepth (analysis.cpp:375-424) */
    void loadTUDepth(int x, int y, int depth)
    {
        if (A->limit_tu < 3 || depth > 2 || !tuRecs) return;
        const int id = geomId(x - ctuX, y - ctuY, depth);
        float predDepth = 0;
        int count = 0;
        auto add = [&](const int8_t* recs, int addr) { predDepth += recs[(size_t)addr * 21 + id]; count++; };
        add(tuRecs->ref[0], ctuAddr);
        if (I->is_inter_b) add(tuRecs->ref[1], ctuAddr);
        const int cx = ctuAddr % ctuW;
        if (ctuAddr >= ctuW)
        {
            add(tuRecs->cur, ctuAddr - ctuW);
            if (cx > 0) add(tuRecs->cur, ctuAddr - ctuW - 1);
            if (cx < ctuW - 1) add(tuRecs->cur, ctuAddr - ctuW + 1);
        }
        if (cx > 0) add(tuRecs->cur, ctuAddr - 1);
        predDepth /= count;
        if (predDepth == 0) maxTUDepth = 0;
        else if (predDepth < 1) maxTUDepth = 1;
        else if (predDepth >= 1 && predDepth <= 1.5) maxTUDepth = 2;
        else if (predDepth > 1.5 && predDepth <= 2.5) maxTUDepth = 3;
        else maxTUDepth = -1;
    }
    void saveTUDepth(const Mode& m, int x, int y, int depth)
    {
        if (A->limit_tu < 3 || depth > 2 || !tuRecs) return;
        int8_t v = -1;
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++) v = std::max<int8_t>(v, (int8_t)m.u[i].tu_depth);
        tuRecs->cur[(size_t)ctuAddr * 21 + geomId(x - ctuX, y - ctuY, depth)] = v;
    }
    /* Analysis::calculateQpforCuSize of the sub-CU q of the CU at `depth` -- when that is a quantisation group's CU (analysis.cpp:1363-1364); else the QP in force stays */
    int childQp(int depth, int q)
    {
        if (!si->use_dqp || depth + 1 > si->max_cu_dqp_depth) return 0;
        return setLambdaFromQP(cuQp[1 + q]);          /* (max_cu_dqp_depth 1: the children of the CTU) */
    }
    /* the price of cu_qp_delta for the mode's first unit (Entropy::codeDeltaQP(cu, 0) in bit-counting mode: mode.contexts.resetBits(); codeDeltaQP; getNumberOfWrittenBits) */
    int addDeltaQpBits(Mode& m, int x, int y)
    {
        if (!qpCoder)
        {
            if (!(qpCoder = x265amd_cabac_open(si, units, 1))) return fail("delta QP: coder");
            qpCoder->ctuInProgress = true;          /* the analysis of a CTU asks (cabac_coder.h: lastQP) */
        }
        if (A->rd_level >= 3)
        {
            x265amd_cu_unit& at = units[(y >> 2) * w4 + (x >> 2)];
            const int8_t saved = at.qp;
            at.qp = m.u[0].qp;
            memcpy(qpCoder->ctx, m.contexts.ctx, X265AMD_CTX_STRIDE);
            qpCoder->fracBits = m.contexts.frac & 32767;    /* resetBits() */
            qpCoder->deltaQP(x, y);
            at.qp = saved;
            m.totalBits += (uint32_t)(qpCoder->fracBits >> 15);
            memcpy(m.contexts.ctx, qpCoder->ctx, X265AMD_CTX_STRIDE);
            m.contexts.frac = qpCoder->fracBits;
        }
        else m.totalBits++;
        updateModeCost(m);
        return 0;
    }
    /* Search::checkDQP (search.cpp:3974-4003) as the merge checks call it on their winner (analysis.cpp:2879, :3019) -- for the mode with a residual a second time: the walk
     * of encodeResAndCalcRdInterCU has priced cu_qp_delta already (inter_rd.hip), and the reference adds it again here */
    int checkDQP(Mode& m, int x, int y, int depth)
    {
        if (!si->use_dqp || depth > si->max_cu_dqp_depth) return 0;
        if (m.u[0].cbf[0] || m.u[0].cbf[1] || m.u[0].cbf[2]) return addDeltaQpBits(m, x, y);
        if (!qpCoder)
        {
            if (!(qpCoder = x265amd_cabac_open(si, units, 1))) return fail("delta QP: coder");
            qpCoder->ctuInProgress = true;
        }
        const int8_t refQp = (int8_t)qpCoder->refQP(x, y);
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++) m.u[i].qp = refQp;
        return 0;
    }
    /* Search::checkDQPForSplitPred (search.cpp:4005-4050) on the mode at (x, y): at the quantisation groups' depth the price of cu_qp_delta when anything in the CU carries a
     * residual -- added whatever the mode is, as the reference does -- and the QP of the CUs in front of the first one with a residual (all of them if there is none) set to
     * the predicted QP, which is what a decoder takes for them */
    int checkDQPForSplitPred(Mode& m, int x, int y, int depth)
    {
        if (!si->use_dqp || depth != si->max_cu_dqp_depth) return 0;
        if (!qpCoder)
        {
            if (!(qpCoder = x265amd_cabac_open(si, units, 1))) return fail("delta QP: coder");
            qpCoder->ctuInProgress = true;
        }
        const int n4 = 16 >> depth;
        bool hasResidual = false;
        for (int i = 0; i < n4 * n4 && !hasResidual; i++) hasResidual = m.u[i].cbf[0] || m.u[i].cbf[1] || m.u[i].cbf[2];
        const int8_t refQp = (int8_t)qpCoder->refQP(x, y);
        if (hasResidual)
        {
            if (addDeltaQpBits(m, x, y)) return err;
            /* CUData::setQPSubCUs (cudata.cpp:1012-1032): CUs in coding order until the first with a residual */
            bool done = false;
            std::function<void(int, int, int)> walk = [&](int ux, int uy, int d) {
                if (done) return;
                const int s4 = 16 >> d;
                if (x + ux * 4 >= I->pic_width || y + uy * 4 >= I->pic_height) return;
                const x265amd_cu_unit& u0 = m.u[uy * n4 + ux];
                if (u0.depth > d) { for (int q = 0; q < 4; q++) walk(ux + (q & 1) * (s4 >> 1), uy + (q >> 1) * (s4 >> 1), d + 1); return; }
                if (u0.cbf[0] || u0.cbf[1] || u0.cbf[2]) { done = true; return; }
                for (int yy = 0; yy < s4; yy++) for (int xx = 0; xx < s4; xx++) m.u[(uy + yy) * n4 + ux + xx].qp = refQp;
            };
            walk(0, 0, depth);
        }
        else
            for (int i = 0; i < n4 * n4; i++) m.u[i].qp = refQp;
        return 0;
    }
    void* intraWs = nullptr;                /* the intra RD's working set, kept for the CUs of this CTU (intra_rd.hip) */
    ~Analyzer() { for (auto& a : chain.ahead) if (a.on && a.q) (void)xa_stream_sync(a.q); xa_intra_ws_free(intraWs); if (qpCoder) x265amd_cabac_close(qpCoder); }       /* (a search nobody collected still writes to this CTU's buffers) */

    /* ---- the device-resident motion map and the skip chain (inter_chain_dev.h) ---- */
    XaMapUnit* dCur = nullptr; const XaMapUnit* dCol = nullptr;
    struct Chain
    {
        bool on = false;
        XaChainNode nodes[XA_CHAIN_MAX_NODES + 3]; int numNodes = 0;
        uint8_t status[XA_CHAIN_MAX_NODES + 3] = {};    /* 0 not run, 1 skipped on the device, 2 the chain stopped here */
        XaChainCuOut res[XA_CHAIN_MAX_NODES + 3];
        XaMapped mJob, mNodes; XaMappedOut mOut; DevBuf dScratch;
        bool nodesPushed = false;
        uint64_t stopFrac = 0; uint8_t stopCtx[X265AMD_CTX_STRIDE];     /* the device's coder state where the last chain stopped (X265AMD_CHAIN_VERIFY) */
        int stopNode = -1;
        XaChainStop stop;                               /* the merge check of the CU the last chain stopped at, when the device made it */
        uint8_t usedFlags[4] = { 0, 0, 0, 0 };          /* per depth of the host's recursion: the node flags (bits 0-1) it went by */
        int frNode[4]; bool frDirty[4];                 /* the host's recursion: node and "something below it was decided on the host" per depth */
        bool lastDevComplete = false;                   /* of the compress() call that has just returned: everything in its area is the device's */
        uint64_t runs = 0, skipped = 0;
        XaMapped mSearch[5], mLuma; XaMappedOut mSearchOut[5]; DevBuf dSearchScratch[5];     /* the fused search command (inter_search_dev.h): [0] the one the CU waits for, [1 + depth] one started ahead at that depth */
        bool lumaPushed = false;
        /* A CU that cannot split has nothing between its merge check and its search: when the chain starts AT such a CU (the CU before it was not skipped, so this
         * one probably is not either), its search runs beside the chain's merge check on a second queue (xa_queue_aux) and checkInterFused collects it.  A search
         * nobody asks for -- the CU was skipped after all -- is waited for before its records are used again.
         * A CU that CAN split runs its sub-CUs between the merge check and the search: its search starts behind the merge check on a third queue (every reference picture: the
         * sub-CUs' restriction is not known yet) and is collected after them -- valid when the winner is a picture the restriction allows (the cheapest of all, first of
         * equals, is then the cheapest of the allowed ones), else the search runs again. */
        struct Ahead { bool on = false; int x = 0, y = 0; void* q = nullptr; XaSearchJob J; } ahead[4];
    } chain;
    bool fusedRd[4] = { false, false, false, false };      /* per depth: the 2Nx2N mode's rate-distortion came with its search (checkInterFused) */
    int buildNodes(int x, int y, int depth, int parent)
    {
        const int idx = chain.numNodes++;
        const int size = 64 >> depth;
        XaChainNode& n = chain.nodes[idx];
        n.x = (int16_t)x; n.y = (int16_t)y; n.log2 = (uint8_t)(6 - depth); n.parent = (uint8_t)(parent < 0 ? 255 : parent);
        const bool mightNotSplit = x + size <= I->pic_width && y + size <= I->pic_height;
        n.flags = 0;
        for (int ge = 1; ge >= 0; ge--)
        {
            /* (the QP test of topSkipMinDepth follows the CTU's first unit, which the CUs coded on the way change under delta QP: both answers, the device picks) */
            const bool checked = mightNotSplit && (uint32_t)depth >= topSkipMinDepth(x, y, depth, ge);
            n.flags |= (uint8_t)(((checked ? 1 : 0) | (mightNotSplit && !checked && depth < si->max_cu_depth ? 2 : 0)) << (ge ? 0 : 2));
        }
        if (depth < si->max_cu_depth)
            for (int q = 0; q < 4; q++)
            {
                const int cx = x + (q & 1) * (size >> 1), cy = y + (q >> 1) * (size >> 1);
                if (cx < I->pic_width && cy < I->pic_height) buildNodes(cx, cy, depth + 1, idx);
            }
        chain.nodes[idx].next = (uint8_t)chain.numNodes;
        return idx;
    }
    /* the chain from `node` to the end of the innermost enclosing CU that has something decided on the host (its reconstruction tiles are the host's to keep) */
    int runChain(int node, int depth)
    {
        XA_HOSTPROF("an.runChain");
        int end = chain.numNodes;
        for (int k = depth - 1; k >= 0; k--) if (chain.frDirty[k]) { end = chain.nodes[chain.frNode[k]].next; break; }
        if (!chain.mJob.p && (chain.mJob.alloc(sizeof(XaChainJob)) != hipSuccess || chain.mNodes.alloc(sizeof(chain.nodes)) != hipSuccess || chain.mOut.alloc(sizeof(XaChainOut)) != hipSuccess))
            return fail("chain records");
        if (!chain.nodesPushed)
        {
            volatile uint64_t* d = (volatile uint64_t*)chain.mNodes.p; const uint64_t* s = (const uint64_t*)chain.nodes;
            for (int i = 0; i < chain.numNodes; i++) d[i] = s[i];
            chain.nodesPushed = true;
        }
        XaChainJob J;
        memset(&J, 0, sizeof(J));
        J.info = *I;
        memcpy(J.ref_pic, S->ref_pic, sizeof(J.ref_pic));
        J.planes = (uint64_t)(uintptr_t)dPlanes.p; J.cur = (uint64_t)(uintptr_t)dCur; J.col = (uint64_t)(uintptr_t)dCol;
        J.tiles = (uint64_t)(uintptr_t)dTiles.p; J.tile_bytes = tileBytes; J.out = (uint64_t)(uintptr_t)chain.mOut.p; J.nodes = (uint64_t)(uintptr_t)chain.mNodes.p;
        J.stride = (int32_t)stride; J.cstride = (int32_t)cstride; J.num_pics = numPics; J.w4 = w4;
        J.start = node; J.end = end; J.num_nodes = chain.numNodes;
        J.tiles_per_depth = 2 * NUM_PRED + 6; J.cand_tile0 = 2 * NUM_PRED; J.split_recon_tile = NUM_PRED + PRED_SPLIT;
        J.skip_recon_tile = NUM_PRED + PRED_SKIP; J.merge_recon_tile = NUM_PRED + PRED_MERGE;
        J.frame_parallel = S->frame_parallel; J.search_range = S->search_range; J.chroma_sa8d = A->rd_level >= 3; J.slice_type = si->slice_type;
        {
            /* Quant::setQPforQuant (quant.cpp:221-244), chroma QP offsets 0: as make_plan (inter_rd.hip) */
            static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                                     32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
            const int qpQuant = qp < 0 ? 0 : (qp > 51 ? 51 : qp), bd = 6 * (X265AMD_DEPTH - 8);
            int qpC = qpQuant < -bd ? -bd : (qpQuant > 57 ? 57 : qpQuant);
            if (qpC >= 30) qpC = chromaScale[qpC];
            J.qp_luma = qpQuant + bd; J.qp_chroma = qpC + bd;
        }
        J.tu_log2_max = si->tu_log2_max;
        J.ctu_x = ctuX; J.ctu_y = ctuY; J.lambda = lambda; J.lambda2 = lambda2; J.psy_rd = psyRd;
        { static const int dbg = getenv("X265AMD_CHAIN_DBG") ? atoi(getenv("X265AMD_CHAIN_DBG")) : 0; J.dbg = dbg; }
        /* the 64x64 CU with levels decided on the device (chain_merge_rd64): built, verified candidate by candidate (X265AMD_CHAIN_VERIFY=2) and measured as no gain -- the
         * twelve units' chains take on the device what the host's two round trips took, and the pictures' links are the last column's searched CTUs either way -- so it
         * stays off unless asked for (X265AMD_CHAIN_64=1); the host's merge check goes on from the device's candidate instead (chainMergeFrom) */
        { static const bool on64 = getenv("X265AMD_CHAIN_64") && atoi(getenv("X265AMD_CHAIN_64")) != 0; J.chain64_off = !on64; }
        J.rd_level = A->rd_level; J.sign_hide = si->sign_hide != 0; J.max_cu_depth = si->max_cu_depth;
        if (!chain.dScratch.p && chain.dScratch.alloc(x265amd_inter_rd_scratch_bytes()) != hipSuccess) return fail("chain scratch");
        J.scratch = (uint64_t)(uintptr_t)chain.dScratch.p;
        J.frac = md[depth].cur.frac; memcpy(J.ctx, md[depth].cur.ctx, X265AMD_CTX_STRIDE);
        J.previous_qp = refQp0[ctuAddr];
        J.first_qp = si->use_dqp ? units[(ctuY >> 2) * w4 + (ctuX >> 2)].qp : ctuQp;
        memcpy(J.anc_flags, chain.usedFlags, 4);
        if (si->use_dqp)
        {
            if (!qpCoder)
            {
                if (!(qpCoder = x265amd_cabac_open(si, units, 1))) return fail("delta QP: coder");
                qpCoder->ctuInProgress = true;
            }
            J.use_dqp = 1; J.max_dqp_depth = si->max_cu_dqp_depth;
            J.prev_qp = qpCoder->lastQP(ctuX, ctuY);
            static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                                     32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
            const int bd = 6 * (X265AMD_DEPTH - 8);
            for (int k = 0; k < (si->max_cu_dqp_depth ? 5 : 1); k++)
            {
                XaChainJob::QpSet& q = J.qps[k];
                const int qv = cuQp[k];              /* (not above 51 on a CTU the chain runs on: compress_ctu_impl) */
                uint64_t rd[6];
                x265amd_rdcost(qv, si->slice_type, A->psy_rd, 0, 0, 0, rd);
                q.lambda2 = rd[0]; q.lambda = rd[1]; q.psy_rd = (uint32_t)rd[2]; q.qp = qv;
                int qpC = qv < -bd ? -bd : (qv > 57 ? 57 : qv);
                if (qpC >= 30) qpC = chromaScale[qpC];
                q.qp_luma = qv + bd; q.qp_chroma = qpC + bd;
            }
            J.last_src[0] = -1;
            for (int k = 1; k < 4 && si->max_cu_dqp_depth; k++)
            {
                const int gx = ctuX + (k & 1) * 32, gy = ctuY + (k >> 1) * 32;
                J.last_src[k] = -1;
                if (gx >= I->pic_width || gy >= I->pic_height) continue;
                int px = 0, py = 0;
                if (qpCoder->lastQPUnitInCtu(gx, gy, px, py))
                {
                    J.last_src[k] = (int8_t)(((py - ctuY) >> 5) * 2 + ((px - ctuX) >> 5));
                    J.last_val[k] = units[(py >> 2) * w4 + (px >> 2)].qp;
                }
                if (k & 1) J.left_val[k] = units[(gy >> 2) * w4 + (gx >> 2) - 1].qp;
                if (k & 2) J.above_val[k] = units[((gy >> 2) - 1) * w4 + (gx >> 2)].qp;
            }
            /* a chain that starts inside a quantisation group ends with the group: what the group's CUs leave in the QP records is then the host's to say (checkDQPForSplitPred) */
            int a = node;
            while (chain.nodes[a].parent != 255 && 6 - chain.nodes[a].log2 > si->max_cu_dqp_depth) a = chain.nodes[a].parent;
            if (a != node && (chain.nodes[a].x != chain.nodes[node].x || chain.nodes[a].y != chain.nodes[node].y) && chain.nodes[a].next < end) { end = chain.nodes[a].next; J.end = end; }
        }
        {
            void** slot = xa_task_slot();
            const XaRowHooks* h = slot ? (const XaRowHooks*)*slot : nullptr;
            if (h && h->ref_wait)
            {
                J.guard_on = 1; J.guard_r0 = 1; J.guard_r1 = 0; J.guard_need = 0;           /* nothing, unless the gate says what it has waited for */
                if (h->ctu_reach) h->ctu_reach(h->ctx, ctuY >> 6, ctuX >> 6, &J.guard_r0, &J.guard_r1, &J.guard_need);
            }
        }
        {
            volatile uint64_t* d = (volatile uint64_t*)chain.mJob.p; const uint64_t* s = (const uint64_t*)&J;
            for (size_t i = 0; i < sizeof(J) / 8; i++) d[i] = s[i];
        }
        const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)chain.mJob.p, 0, 0, 0, 1 };
        if (xa_q_enqueue(st, XA_OP_INTER_CHAIN, &qa, sizeof(qa), 1, 0) != hipSuccess || xa_stream_sync(st) != hipSuccess) return fail("chain command");
        const XaChainOut* o = (const XaChainOut*)chain.mOut.p;
        const uint32_t count = o->count, stopNode = o->stop_node, reason = o->reason;
        if (count > (uint32_t)chain.numNodes || stopNode > (uint32_t)chain.numNodes) return fail("chain result");
        for (uint32_t k = 0; k < count; k++)
        {
            const XaChainCuOut& c = o->cu[k];
            if (c.node >= (uint32_t)chain.numNodes) return fail("chain result");
            chain.res[c.node] = c; chain.status[c.node] = 1;
        }
        if (reason != XA_CHAIN_END && stopNode < (uint32_t)chain.numNodes)
        {
            chain.status[stopNode] = 2;
            chain.stopNode = (int)stopNode; chain.stopFrac = o->frac; memcpy(chain.stopCtx, (const void*)o->ctx, X265AMD_CTX_STRIDE);
            chain.stop.valid = 0;
            if (o->stop.valid == 1 && o->stop.node == stopNode) memcpy(&chain.stop, (const void*)&o->stop, sizeof(XaChainStop));
            else if (o->stop.valid == 2 && o->stop.node == stopNode) memcpy(&chain.stop, (const void*)&o->stop, offsetof(XaChainStop, ctx));     /* (the head alone: chainMergeFrom) */
        }
        chain.runs++; chain.skipped += count;
        if (g_timing)
        {
            g_chainStat[0]++; g_chainStat[1] += count; g_chainStat[2] += reason == XA_CHAIN_NOTSKIP; g_chainStat[3] += reason == XA_CHAIN_GUARD;
            for (int k = 0; k < 8; k++) g_chainTicks[k] += o->ticks[k];
        }
        return 0;
    }
    /* the merge check of the CU the chain stopped at, made on the device (the residual mode beat the skip mode): both modes as checkMerge2Nx2N_rd0_4 leaves them */
    int chainMerge(int node, int x, int y, int depth, bool& taken)
    {
        taken = false;
        static const bool on = !(getenv("X265AMD_CHAIN_MERGE") && atoi(getenv("X265AMD_CHAIN_MERGE")) == 0);
        /* (rd 2 reads the source block's mean and deviation from the merge check's measurement -- complexityCheckCU -- which the device's record does not carry) */
        if (!on || A->rd_level < 3 || chain.status[node] != 2 || chain.stopNode != node || chain.stop.valid != 1 || chain.stop.node != (uint32_t)node) return 0;
        XA_HOSTPROF("an.chainMerge");
        const XaChainStop& c = chain.stop;
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2, n4 = 16 >> depth;
        Mode* skip = &d.pred[PRED_SKIP];
        Mode* merge = &d.pred[PRED_MERGE];
        skip->initCosts(); merge->initCosts();
        const int16_t zero[2][2] = { { 0, 0 }, { 0, 0 } };
        const uint8_t noIdx[2] = { 0, 0 };
        const int numCand = I->max_num_merge_cand;
        const uint32_t bits = (uint32_t)(c.cand + (c.cand < numCand - 1));
        x265amd_cu_measure ms = c.meas; ms.sa8d = c.sa8d; ms.sa8d_luma = c.sa8d_luma;
        for (Mode* m : { skip, merge })
        {
            setInter(*m, depth, 1, c.cand, c.dir, c.ref_idx, c.mv, zero, noIdx);
            m->sa8dCost = calcRdSADCost(sa8dOf(ms), bits); m->sa8dBits = bits;
            m->predTile = candTile(depth, c.cand);
        }
        skip->reconTile = reconTile(depth, PRED_SKIP); merge->reconTile = reconTile(depth, PRED_MERGE);
        d.srcMean = 0; d.srcHomo = 0;
        x265amd_rd_cu rc;
        memset(&rc, 0, sizeof(rc));
        rc.x = (int16_t)x; rc.y = (int16_t)y; rc.log2_size = (uint8_t)log2; cuQp2(rc);
        memcpy(rc.ctx, d.cur.ctx, X265AMD_CTX_COUNT);
        rc.frac_bits = d.cur.frac;
        x265amd_rd_result r;
        const int rcode = x265amd_skip_rd_host(si, &rp, units, &rc, 1, skip->u, &c.meas, &r);
        if (rcode != X265AMD_OK) return err = rcode;
        std::fill(skip->coeff.begin(), skip->coeff.end(), 0);
        skip->rdCost = r.rd_cost; skip->distortion = (sse_t)r.distortion; skip->totalBits = r.total_bits; skip->mvBits = r.mv_bits; skip->coeffBits = r.coeff_bits;
        skip->psyEnergy = r.psy_energy; skip->lumaDistortion = r.luma_distortion; skip->chromaDistortion = r.chroma_distortion; skip->resEnergy = r.res_energy;
        memcpy(skip->contexts.ctx, r.ctx, X265AMD_CTX_STRIDE);
        skip->contexts.frac = r.frac_bits;
        for (int i = 0; i < n4 * n4; i++) skip->m[i].pred_mode = skip->u[i].pred_mode;
        /* the residual mode: units as the walk leaves them (one transform unit per plane: tu_depth 0, the coded block flags at depth 0), levels in CUData::m_trCoeff order */
        for (int i = 0; i < n4 * n4; i++)
        {
            x265amd_cu_unit& u = merge->u[i];
            u.depth = (uint8_t)depth; u.tu_depth = 0; u.cbf[0] = c.cbf[0]; u.cbf[1] = c.cbf[1]; u.cbf[2] = c.cbf[2]; u.pred_mode = X265AMD_MODE_INTER;
            merge->m[i].pred_mode = X265AMD_MODE_INTER;
        }
        std::fill(merge->coeff.begin(), merge->coeff.end(), 0);
        memcpy(&merge->coeff[0], c.levels, sizeof(int16_t) * size * size);
        memcpy(&merge->coeff[4096], c.levels + 1024, sizeof(int16_t) * (size >> 1) * (size >> 1));
        memcpy(&merge->coeff[5120], c.levels + 1280, sizeof(int16_t) * (size >> 1) * (size >> 1));
        merge->rdCost = c.rd_cost; merge->lumaDistortion = (sse_t)c.luma_dist; merge->chromaDistortion = (sse_t)c.chroma_dist; merge->distortion = (sse_t)(c.luma_dist + c.chroma_dist);
        merge->totalBits = c.total_bits; merge->mvBits = c.mv_bits; merge->coeffBits = c.coeff_bits; merge->psyEnergy = c.psy_energy; merge->resEnergy = (sse_t)c.meas.sse[0];
        memset(merge->contexts.ctx, 0, X265AMD_CTX_STRIDE);
        memcpy(merge->contexts.ctx, c.ctx, X265AMD_CTX_COUNT);
        merge->contexts.frac = c.frac;
        d.best = merge->rdCost < skip->rdCost ? merge : skip;
        /* the winner keeps the candidate's prediction: out of the candidate tiles, which the next merge scan reuses */
        const int keep = predTile(depth, d.best == merge ? PRED_MERGE : PRED_SKIP);
        copyTile(keep, candTile(depth, c.cand), 0, 0, size);
        d.best->predTile = keep;
        taken = true;
        return checkDQP(*d.best, x, y, depth);              /* analysis.cpp:2879 */
    }
    /* the merge check of a 64x64 CU the chain stopped at with a level somewhere in its residual (XaChainStop::valid 2): the device has predicted and ranked the candidates
     * as checkMerge2Nx2N_rd0_4 does (the chain's own choice of every skipped CU), so the check goes on from the chosen candidate -- its prediction lies in its tile --
     * with the two modes' rate-distortion (rdMergePair); one wait less than checkMerge's own predictions and measurements */
    int chainMergeFrom(int node, int x, int y, int depth, bool& taken)
    {
        taken = false;
        static const bool on = !(getenv("X265AMD_CHAIN_MERGE_FROM") && atoi(getenv("X265AMD_CHAIN_MERGE_FROM")) == 0);
        if (!on || A->rd_level < 3 || rp.rdoq_level || chain.status[node] != 2 || chain.stopNode != node || chain.stop.valid != 2 || chain.stop.node != (uint32_t)node) return 0;
        XA_HOSTPROF("an.chainMergeFrom");
        const XaChainStop& c = chain.stop;
        ModeDepth& d = md[depth];
        const int size = 64 >> depth;
        Mode* bestPred = &d.pred[PRED_SKIP];
        Mode* tempPred = &d.pred[PRED_MERGE];
        tempPred->initCosts(); bestPred->initCosts();
        const int16_t zero[2][2] = { { 0, 0 }, { 0, 0 } };
        const uint8_t noIdx[2] = { 0, 0 };
        const int numCand = I->max_num_merge_cand;
        const uint32_t bits = (uint32_t)(c.cand + (c.cand < numCand - 1));
        x265amd_cu_measure ms;
        memset(&ms, 0, sizeof(ms));
        ms.sa8d = c.sa8d; ms.sa8d_luma = c.sa8d_luma;
        for (Mode* m : { bestPred, tempPred })
        {
            setInter(*m, depth, 1, c.cand, c.dir, c.ref_idx, c.mv, zero, noIdx);
            m->sa8dCost = calcRdSADCost(sa8dOf(ms), bits); m->sa8dBits = bits;
            m->predTile = candTile(depth, c.cand);
        }
        bestPred->reconTile = reconTile(depth, PRED_SKIP); tempPred->reconTile = reconTile(depth, PRED_MERGE);
        d.srcMean = 0; d.srcHomo = 0;          /* (read at rd 2 only) */
        if (rdMergePair(*bestPred, *tempPred, x, y, depth)) return err;
        d.best = tempPred->rdCost < bestPred->rdCost ? tempPred : bestPred;
        const int keep = predTile(depth, d.best == tempPred ? PRED_MERGE : PRED_SKIP);
        copyTile(keep, candTile(depth, c.cand), 0, 0, size);
        d.best->predTile = keep;
        taken = true;
        return checkDQP(*d.best, x, y, depth);
    }
    /* the merge check of a CU decided on the device as a skip: Mode PRED_SKIP as checkMerge leaves it (candidate, costs, contexts) */
    int chainSkip(int node, int x, int y, int depth, bool& skipped)
    {
        skipped = false;
        if (chain.status[node] == 0) { if (runChain(node, depth)) return err; }
        static const bool verify = getenv("X265AMD_CHAIN_VERIFY") != nullptr;
        if (verify && chain.status[node] == 2 && chain.stopNode == node)
        {
            /* the device carried the entropy coder's state from CU to CU on its own: where it stopped it must be what the host arrives with */
            const Snap& h = md[depth].cur;
            if ((h.frac & 32767) != (chain.stopFrac & 32767) || memcmp(h.ctx, chain.stopCtx, X265AMD_CTX_COUNT))
            {
                int first = -1;
                for (int i = 0; i < X265AMD_CTX_COUNT && first < 0; i++) if (h.ctx[i] != chain.stopCtx[i]) first = i;
                fprintf(stderr, "x265amd chain verify: poc %d CU (%d,%d) size %d: coder state differs: fraction host %llu device %llu, first context %d (host %d device %d)\n", I->poc, x, y, 64 >> depth,
                        (unsigned long long)(h.frac & 32767), (unsigned long long)(chain.stopFrac & 32767), first, first >= 0 ? h.ctx[first] : 0, first >= 0 ? chain.stopCtx[first] : 0);
                return fail("chain verify: the device's entropy state is not the host's");
            }
        }
        if (chain.status[node] != 1) return 0;
        XA_HOSTPROF("an.chainSkip");
        const XaChainCuOut& c = chain.res[node];
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth;
        Mode* bestPred = &d.pred[PRED_SKIP];
        bestPred->initCosts();
        if (verify)
        {
            x265amd_merge_cand cand[5];
            const int numCand = x265amd_merge_candidates(I, cur, col, x, y, log2, 0, 0, cand);
            const x265amd_merge_cand& b = cand[c.cand < 5 ? c.cand : 0];
            bool same = c.cand < numCand && b.dir == c.dir;
            for (int l = 0; l < 2 && same; l++)
                if ((c.dir >> l) & 1) same = b.ref_idx[l] == c.ref_idx[l] && b.mv[l][0] == c.mv[l][0] && b.mv[l][1] == c.mv[l][1];
            if (!same)
            {
                fprintf(stderr, "x265amd chain verify: poc %d CU (%d,%d) size %d: device candidate %d dir %d refs %d %d mv (%d,%d) (%d,%d); host candidate dir %d refs %d %d mv (%d,%d) (%d,%d) of %d\n", I->poc, x, y,
                        1 << log2, c.cand, c.dir, c.ref_idx[0], c.ref_idx[1], c.mv[0][0], c.mv[0][1], c.mv[1][0], c.mv[1][1], b.dir, b.ref_idx[0], b.ref_idx[1], b.mv[0][0], b.mv[0][1],
                        b.mv[1][0], b.mv[1][1], numCand);
                return fail("chain verify: the device's merge candidate is not the host's");
            }
        }
        const int16_t zero[2][2] = { { 0, 0 }, { 0, 0 } };
        const uint8_t noIdx[2] = { 0, 0 };
        const int numCand = I->max_num_merge_cand;
        const uint32_t bits = (uint32_t)(c.cand + (c.cand < numCand - 1));
        setInter(*bestPred, depth, 1, c.cand, c.dir, c.ref_idx, c.mv, zero, noIdx);
        bestPred->sa8dCost = calcRdSADCost(sa8dOf(c.meas), bits); bestPred->sa8dBits = bits;
        bestPred->predTile = candTile(depth, c.cand); bestPred->reconTile = reconTile(depth, PRED_SKIP);
        d.srcMean = c.meas.src_mean; d.srcHomo = c.meas.src_homo;
        x265amd_rd_cu rc;
        memset(&rc, 0, sizeof(rc));
        rc.x = (int16_t)x; rc.y = (int16_t)y; rc.log2_size = (uint8_t)log2; cuQp2(rc);
        memcpy(rc.ctx, d.cur.ctx, X265AMD_CTX_COUNT);
        rc.frac_bits = d.cur.frac;
        x265amd_rd_result r;
        const int rcode = x265amd_skip_rd_host(si, &rp, units, &rc, 1, bestPred->u, &c.meas, &r);
        if (rcode != X265AMD_OK) return err = rcode;
        Mode& m = *bestPred;
        std::fill(m.coeff.begin(), m.coeff.end(), 0);
        m.rdCost = r.rd_cost; m.distortion = (sse_t)r.distortion; m.totalBits = r.total_bits; m.mvBits = r.mv_bits; m.coeffBits = r.coeff_bits;
        m.psyEnergy = r.psy_energy; m.lumaDistortion = r.luma_distortion; m.chromaDistortion = r.chroma_distortion; m.resEnergy = r.res_energy;
        memcpy(m.contexts.ctx, r.ctx, X265AMD_CTX_STRIDE);
        m.contexts.frac = r.frac_bits;
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++) m.m[i].pred_mode = m.u[i].pred_mode;
        d.best = bestPred;
        skipped = true;
        return checkDQP(*bestPred, x, y, depth);           /* analysis.cpp:2879 */
    }

    uint64_t tileAddr(int t) const { return (uint64_t)(uintptr_t)dTiles.p + (size_t)t * tileBytes; }
    /* tiles: per depth NUM_PRED prediction + NUM_PRED reconstruction tiles, then 5 merge-candidate tiles + 1 scratch tile per depth */
    int predTile(int depth, int k) const { return depth * (2 * NUM_PRED + 6) + k; }
    int reconTile(int depth, int k) const { return depth * (2 * NUM_PRED + 6) + NUM_PRED + k; }
    int candTile(int depth, int k) const { return depth * (2 * NUM_PRED + 6) + 2 * NUM_PRED + k; }

    uint64_t calcRdCost(sse_t d, uint32_t b) const { return d + (((uint64_t)b * lambda2 + 128) >> 8); }
    uint64_t calcPsyRdCost(sse_t d, uint32_t b, uint32_t e) const { return d + ((lambda * psyRd * e) >> 24) + (((uint64_t)b * lambda2) >> 8); }
    uint64_t calcRdSADCost(uint32_t d, uint32_t b) const { return d + (((uint64_t)b * lambda + 128) >> 8); }
    uint32_t getCost(uint32_t b) const { return (uint32_t)(((uint64_t)b * lambda + 128) >> 8); }
    void updateModeCost(Mode& m) const { m.rdCost = psyRd ? calcPsyRdCost(m.distortion, m.totalBits, m.psyEnergy) : calcRdCost(m.distortion, m.totalBits); }

    /* m_bChromaSa8d (analysis.cpp:707): SA8D ranking includes chroma from rd level 3 */
    uint32_t sa8dOf(const x265amd_cu_measure& ms) const { return A->rd_level >= 3 ? ms.sa8d : ms.sa8d_luma; }

    int fail(const char* msg) { if (!err) err = xa_fail(X265AMD_EHIP, msg); return err; }

    /* ---- block operations ---- */
    x265amd_mc_job mcJob(int x, int y, int size, int tile, int dir, const int8_t ref[2], const int16_t mv[2][2])
    {
        x265amd_mc_job j;
        memset(&j, 0, sizeof(j));
        const size_t isz = sizeof(pixel);
        j.dst_y = tileAddr(tile); j.dst_u = j.dst_y + 4096 * isz; j.dst_v = j.dst_u + 1024 * isz;
        j.dst_stride = 64; j.dst_cstride = 32;
        j.x = (int16_t)x; j.y = (int16_t)y; j.cu_x = (int16_t)x; j.cu_y = (int16_t)y; j.w = (uint8_t)size; j.h = (uint8_t)size;
        j.ref0 = (dir & 1) && ref[0] >= 0 ? (int8_t)S->ref_pic[0][ref[0]] : -1;
        j.ref1 = (dir & 2) && ref[1] >= 0 ? (int8_t)S->ref_pic[1][ref[1]] : -1;
        j.mv0[0] = mv[0][0]; j.mv0[1] = mv[0][1]; j.mv1[0] = mv[1][0]; j.mv1[1] = mv[1][1];
        j.slice_type = (uint8_t)!I->is_inter_b; j.flags = 3;
        if (S->weighted)
        {
            /* Predict::motionCompensation in a slice with weights (predict.cpp:85-232): pps.bUseWeightPred / bUseWeightedBiPred and the references' table entries */
            j.flags |= S->weighted == 1 ? 4 : 8;
            for (int l = 0; l < 2; l++)
                if (((dir >> l) & 1) && ref[l] >= 0) memcpy(&j.wp[l][0], &S->wp[l][ref[l]][0], sizeof(j.wp[l]));
        }
        return j;
    }
    /* motionCompensation(luma + chroma) of every PU of an inter mode into its prediction tile (rd 2: the chosen mode's chroma was not predicted yet) */
    int mcMode(Mode& m, int x, int y, int depth)
    {
        const int size = 64 >> depth, n4 = size >> 2, part = m.u[0].part_size;
        static const uint8_t nb[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
        std::vector<x265amd_mc_job> jobs;
        for (int k = 0; k < nb[part]; k++)
        {
            const Geo g = pu_geo(0, 0, size, part, k);
            const x265amd_mv_unit& v = m.m[(g.y >> 2) * n4 + (g.x >> 2)];
            x265amd_mc_job j = mcJob(x, y, size, m.predTile, v.inter_dir, v.ref_idx, v.mv);
            j.x = (int16_t)(x + g.x); j.y = (int16_t)(y + g.y); j.w = (uint8_t)g.w; j.h = (uint8_t)g.h;
            j.dst_y += (size_t)(g.y * 64 + g.x) * sizeof(pixel);
            j.dst_u += (size_t)((g.y >> 1) * 32 + (g.x >> 1)) * sizeof(pixel); j.dst_v += (size_t)((g.y >> 1) * 32 + (g.x >> 1)) * sizeof(pixel);
            jobs.push_back(j);
        }
        if (xa_ref_guard_mc(jobs.data(), (int)jobs.size())) return fail("a reference picture failed");
        memcpy(dJobs.p, jobs.data(), sizeof(x265amd_mc_job) * jobs.size());
        if (x265amd_motion_compensation(st, (const uint64_t*)dPlanes.p, stride, cstride, I->pic_width, I->pic_height, (const x265amd_mc_job*)dJobs.p, (int)jobs.size()) != X265AMD_OK)
            return err = X265AMD_EHIP;
        return 0;
    }
    /* motionCompensation(luma + chroma) of the jobs into their tiles, then sa8d / sse / psy of each tile against the source */
    int predictAndMeasure(std::vector<x265amd_mc_job>& jobs, int x, int y, int log2, const int* tiles, x265amd_cu_measure* meas)
    {
        XA_HOSTPROF("an.predictAndMeasure");
        const int n = (int)jobs.size();
        if (xa_ref_guard_mc(jobs.data(), n)) return fail("a reference picture failed");
        memcpy(dJobs.p, jobs.data(), sizeof(x265amd_mc_job) * n);
        if (x265amd_motion_compensation(st, (const uint64_t*)dPlanes.p, stride, cstride, I->pic_width, I->pic_height, (const x265amd_mc_job*)dJobs.p, n) != X265AMD_OK)
            return err = X265AMD_EHIP;
        x265amd_rd_cu c[8];
        uint64_t addr[8];
        memset(c, 0, sizeof(c));
        for (int k = 0; k < n; k++) { c[k].x = (int16_t)x; c[k].y = (int16_t)y; c[k].log2_size = (uint8_t)log2; addr[k] = tileAddr(tiles[k]); }
        if (x265amd_measure_tile_list(st, planes + 3 * (numPics - 1), stride, cstride, c, n, addr, meas) != X265AMD_OK) return err = X265AMD_EHIP;      /* one launch */
        return 0;
    }
    void copyTile(int dst, int src, int dx, int dy, int size)       /* src tile (size x size at its origin) -> dst tile at (dx, dy) */
    {
        const size_t isz = sizeof(pixel);
        const uint64_t d = tileAddr(dst), s = tileAddr(src);
        XaRects r;
        r.n = 3;
        r.dst[0] = d + ((size_t)dy * 64 + dx) * isz; r.src[0] = s; r.dst_stride[0] = r.src_stride[0] = 64; r.w[0] = r.h[0] = size;
        for (int p = 0; p < 2; p++)
        {
            r.dst[1 + p] = d + (4096 + p * 1024 + (size_t)(dy / 2) * 32 + dx / 2) * isz; r.src[1 + p] = s + (4096 + p * 1024) * isz;
            r.dst_stride[1 + p] = r.src_stride[1 + p] = 32; r.w[1 + p] = r.h[1 + p] = size / 2;
        }
        xa_copy_rects(st, r);
    }
    void tileToPicture(int tile, int x, int y, int size)
    {
        const size_t isz = sizeof(pixel);
        const int w = size < I->pic_width - x ? size : I->pic_width - x, h = size < I->pic_height - y ? size : I->pic_height - y;
        const uint64_t* rec = planes + 3 * (numPics - 2);
        const uint64_t s = tileAddr(tile);
        XaRects r;
        r.n = 3;
        r.dst[0] = rec[0] + ((size_t)y * stride + x) * isz; r.src[0] = s; r.dst_stride[0] = (int16_t)stride; r.src_stride[0] = 64; r.w[0] = w; r.h[0] = h;
        for (int p = 0; p < 2; p++)
        {
            r.dst[1 + p] = rec[1 + p] + ((size_t)(y / 2) * cstride + x / 2) * isz; r.src[1 + p] = s + (4096 + p * 1024) * isz;
            r.dst_stride[1 + p] = (int16_t)cstride; r.src_stride[1 + p] = 32; r.w[1 + p] = w / 2; r.h[1 + p] = h / 2;
        }
        xa_copy_rects(st, r);
    }

    /* ---- CU bookkeeping ---- */
    void initSubCU(Mode& m, int depth)          /* CUData::initSubCU: a clean inter-less CU of this depth and QP */
    {
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++)
        {
            memset(&m.u[i], 0, sizeof(x265amd_cu_unit));
            memset(&m.m[i], 0, sizeof(x265amd_mv_unit));
            m.u[i].depth = (uint8_t)depth; m.u[i].qp = (int8_t)qp; m.u[i].ref_idx[0] = m.u[i].ref_idx[1] = -1;
            m.m[i].ref_idx[0] = m.m[i].ref_idx[1] = -1;
        }
    }
    void setInter(Mode& m, int depth, int mergeFlag, int mergeIdx, int dir, const int8_t ref[2], const int16_t mv[2][2], const int16_t mvd[2][2], const uint8_t mvpIdx[2])
    {
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++)
        {
            x265amd_cu_unit& u = m.u[i];
            u.pred_mode = X265AMD_MODE_INTER; u.part_size = 0; u.merge_flag = (uint8_t)mergeFlag; u.inter_dir = (uint8_t)dir;
            for (int l = 0; l < 2; l++)
            {
                u.ref_idx[l] = (dir & (1 << l)) ? ref[l] : -1;
                u.mvp_idx[l] = mergeFlag ? (l ? 0 : (uint8_t)mergeIdx) : mvpIdx[l];
                u.mvd[l][0] = mergeFlag ? 0 : mvd[l][0]; u.mvd[l][1] = mergeFlag ? 0 : mvd[l][1];
            }
            x265amd_mv_unit& v = m.m[i];
            v.pred_mode = X265AMD_MODE_INTER; v.inter_dir = (uint8_t)dir;
            for (int l = 0; l < 2; l++)
            {
                v.ref_idx[l] = (dir & (1 << l)) ? ref[l] : -1;
                v.mv[l][0] = (dir & (1 << l)) ? mv[l][0] : 0; v.mv[l][1] = (dir & (1 << l)) ? mv[l][1] : 0;
            }
        }
    }
    void toPicture(const Mode& m, int x, int y, int depth, bool mirror = true)      /* CUData::copyToPic; mirror: also into the device's motion map (not what the device itself decided) */
    {
        const int n4 = 16 >> depth;
        for (int yy = 0; yy < n4; yy++)
            for (int xx = 0; xx < n4; xx++)
            {
                if (x + xx * 4 >= I->pic_width || y + yy * 4 >= I->pic_height) continue;
                units[((y >> 2) + yy) * w4 + (x >> 2) + xx] = m.u[yy * n4 + xx];
                x265amd_mv_unit v = m.m[yy * n4 + xx];
                v.pred_mode = m.u[yy * n4 + xx].pred_mode;
                cur[((y >> 2) + yy) * w4 + (x >> 2) + xx] = v;
                if (dCur && mirror) devmap_store(dCur + ((y >> 2) + yy) * w4 + (x >> 2) + xx, v, m.u[yy * n4 + xx].depth);
            }
    }

    /* RD of one candidate through the batch entry points (n = 1) */
    int rdInter(Mode& m, int x, int y, int depth, bool skipOnly)
    {
        XA_HOSTPROF("an.rdInter (all)");
        StageTimer timer_(2);
        x265amd_rd_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)(6 - depth); cuQp2(c);
        memcpy(c.ctx, md[depth].cur.ctx, X265AMD_CTX_COUNT);
        c.frac_bits = md[depth].cur.frac;
        x265amd_rd_result r;
        /* the batch entry points take 256 records per CU in raster order with row length size/4: Mode::u is laid out like that */
        int rc;
        xa_phase(XA_PH_ANALYZER);
        struct PhEnd { ~PhEnd() { xa_phase(XA_PH_INTER_RD); } } phEnd;
        if (skipOnly)
            rc = x265amd_skip_rd(st, si, &rp, units, planes + 3 * (numPics - 1), stride, cstride, &c, 1, m.u, tileAddr(m.predTile), tileAddr(m.reconTile), tileBytes, &r);
        else
            rc = xa_inter_residual_rd_lazy(st, si, &rp, units, planes + 3 * (numPics - 1), stride, cstride, &c, m.u, tileAddr(m.predTile), tileAddr(m.reconTile), tileBytes,
                                           &r, m.coeff.data());
        if (rc != X265AMD_OK) return err = rc;
        if (skipOnly) std::fill(m.coeff.begin(), m.coeff.end(), 0);
        m.rdCost = r.rd_cost; m.distortion = (sse_t)r.distortion; m.totalBits = r.total_bits; m.mvBits = r.mv_bits; m.coeffBits = r.coeff_bits;
        m.psyEnergy = r.psy_energy; m.lumaDistortion = r.luma_distortion; m.chromaDistortion = r.chroma_distortion; m.resEnergy = r.res_energy;
        memcpy(m.contexts.ctx, r.ctx, X265AMD_CTX_STRIDE);
        m.contexts.frac = r.frac_bits;
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++) m.m[i].pred_mode = m.u[i].pred_mode;
        return 0;
    }

    /* encodeResAndCalcRdSkipCU on `skip` and encodeResAndCalcRdInterCU on `merge` (the same candidate, the same prediction tile) with the shared measurement done once */
    int rdMergePair(Mode& skip, Mode& merge, int x, int y, int depth)
    {
        XA_HOSTPROF("an.rdMergePair (all)");
        StageTimer timer_(2);
        x265amd_rd_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)(6 - depth); cuQp2(c);
        memcpy(c.ctx, md[depth].cur.ctx, X265AMD_CTX_COUNT);
        c.frac_bits = md[depth].cur.frac;
        x265amd_rd_result rs, rm;
        int same = 0;
        const int rc = xa_merge_rd(st, si, &rp, units, planes + 3 * (numPics - 1), stride, cstride, &c, skip.u, merge.u, tileAddr(skip.predTile), tileAddr(skip.reconTile),
                                   tileAddr(merge.reconTile), &rs, &rm, merge.coeff.data(), &same);
        if (rc != X265AMD_OK) return err = rc;
        std::fill(skip.coeff.begin(), skip.coeff.end(), 0);
        const int n4 = 16 >> depth;
        Mode* ms[2] = { &skip, &merge };
        const x265amd_rd_result* rr[2] = { &rs, &rm };
        for (int k = 0; k < 2; k++)
        {
            Mode& m = *ms[k]; const x265amd_rd_result& r = *rr[k];
            m.rdCost = r.rd_cost; m.distortion = (sse_t)r.distortion; m.totalBits = r.total_bits; m.mvBits = r.mv_bits; m.coeffBits = r.coeff_bits;
            m.psyEnergy = r.psy_energy; m.lumaDistortion = r.luma_distortion; m.chromaDistortion = r.chroma_distortion; m.resEnergy = r.res_energy;
            memcpy(m.contexts.ctx, r.ctx, X265AMD_CTX_STRIDE);
            m.contexts.frac = r.frac_bits;
            for (int i = 0; i < n4 * n4; i++) m.m[i].pred_mode = m.u[i].pred_mode;
        }
        return 0;
    }

    /* checkIntraInInter + encodeIntraInInter */
    int rdIntra(Mode& m, int x, int y, int depth, int slot = PRED_INTRA, bool full = false, int partSize = 0)
    {
        XA_HOSTPROF("an.rdIntra (all)");
        StageTimer timer_(3);
        x265amd_rd_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)(6 - depth); cuQp2(c);
        memcpy(c.ctx, md[depth].cur.ctx, X265AMD_CTX_COUNT);
        c.frac_bits = md[depth].cur.frac;
        x265amd_rd_result r;
        m.initCosts();
        m.predTile = predTile(depth, slot); m.reconTile = reconTile(depth, slot);
        uint64_t info[4] = { 0, 0, 0, 0 };
        const int rc = full ? xa_check_intra_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, &c, partSize, m.u,
                                                tileAddr(m.predTile), tileAddr(m.reconTile), &r, m.coeff.data(), &intraWs)
                            : xa_intra_in_inter_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, &c, m.u, tileAddr(m.predTile),
                                                   tileAddr(m.reconTile), &r, m.coeff.data(), info, &intraWs);
        if (rc != X265AMD_OK) return err = rc;
        m.sa8dCost = info[1]; m.sa8dBits = (uint32_t)info[2];
        m.rdCost = r.rd_cost; m.distortion = (sse_t)r.distortion; m.totalBits = r.total_bits; m.mvBits = r.mv_bits; m.coeffBits = r.coeff_bits;
        m.psyEnergy = r.psy_energy; m.lumaDistortion = r.luma_distortion; m.chromaDistortion = r.chroma_distortion; m.resEnergy = r.res_energy;
        memcpy(m.contexts.ctx, r.ctx, X265AMD_CTX_STRIDE);
        m.contexts.frac = r.frac_bits;
        const int n4 = 16 >> depth;
        for (int i = 0; i < n4 * n4; i++)
        {
            memset(&m.m[i], 0, sizeof(x265amd_mv_unit));
            m.m[i].pred_mode = X265AMD_MODE_INTRA; m.m[i].ref_idx[0] = m.m[i].ref_idx[1] = -1;
        }
        return 0;
    }

    /* ---- Analysis helpers ---- */
    uint32_t topSkipMinDepth(int x, int y, int depth, int ge = -1)         /* ge: the QP test's answer given (the skip chain's node table), -1: as the records say */
    {
        /* parentCTU.m_qp[0] (analysis.cpp:3432) -- and parentCTU IS the picture's CTU record (frameencoder.cpp:1490): what compressCTU set for the whole CTU until the first
         * CU of the CTU has been copied to the picture, that CU's QP afterwards */
        const int currentQP = si->use_dqp ? units[(ctuY >> 2) * w4 + (ctuX >> 2)].qp : ctuQp;
        int previousQP = currentQP;
        uint32_t minDepth0 = 4, minDepth1 = 4, sum = 0;
        int numRefs = 0;
        const int size = 64 >> depth;
        for (int l = 0; l < 2; l++)
        {
            if (!I->num_ref_idx[l]) continue;
            numRefs++;
            const uint8_t* map = refDepth + (size_t)l * w4 * h4;
            if (l == 0) previousQP = refQp0[ctuAddr];
            if (!map[(y >> 2) * w4 + (x >> 2)]) return 0;
            uint32_t& mn = l ? minDepth1 : minDepth0;
            for (int yy = y; yy < y + size; yy += 8)
                for (int xx = x; xx < x + size; xx += 8)
                {
                    const uint32_t d = (xx < I->pic_width && yy < I->pic_height) ? map[(yy >> 2) * w4 + (xx >> 2)] : 0;
                    mn = d < mn ? d : mn;
                    sum += d;
                }
        }
        if (!numRefs) return 0;
        uint32_t minDepth = minDepth0 < minDepth1 ? minDepth0 : minDepth1;
        const uint32_t thresh = minDepth * numRefs * ((uint32_t)(size >> 2) * (size >> 2) >> 2);
        if (minDepth && (ge < 0 ? currentQP >= previousQP : ge != 0) && (sum <= thresh + (thresh >> 1))) minDepth -= 1;
        return minDepth;
    }
    bool recursionDepthCheck(int depth, const Mode& best)
    {
        const x265amd_cu_stat& cs = cuStat[ctuAddr];
        const uint64_t cuCost = cs.avg_cost[depth] * cs.count[depth], cuCount = cs.count[depth];
        uint64_t neighCost = 0, neighCount = 0;
        const int cx = ctuAddr % ctuW;
        const bool above = ctuAddr >= ctuW, left = cx > 0;
        auto add = [&](int addr) { neighCost += cuStat[addr].avg_cost[depth] * cuStat[addr].count[depth]; neighCount += cuStat[addr].count[depth]; };
        if (above)
        {
            add(ctuAddr - ctuW);
            if (left) add(ctuAddr - ctuW - 1);
            if (cx < ctuW - 1) add(ctuAddr - ctuW + 1);
        }
        if (left) add(ctuAddr - 1);
        if (neighCount + cuCount)
        {
            const uint64_t avgCost = ((3 * cuCost) + (2 * neighCost)) / ((3 * cuCount) + (2 * neighCount));
            if (best.rdCost < avgCost && avgCost) return true;
        }
        return false;
    }
    void addSplitFlagCost(Mode& m, int x, int y, int depth)
    {
        const x265amd_cu_unit* l = (x >> 2) > 0 ? &units[(y >> 2) * w4 + (x >> 2) - 1] : nullptr;
        const x265amd_cu_unit* a = (y >> 2) > 0 ? &units[((y >> 2) - 1) * w4 + (x >> 2)] : nullptr;
        const int c = (l && l->pred_mode != X265AMD_MODE_NONE && l->depth > depth) + (a && a->pred_mode != X265AMD_MODE_NONE && a->depth > depth);
        if (A->rd_level == 2) { m.totalBits++; updateModeCost(m); return; }
        const uint32_t flag = m.u[0].depth > depth;
        m.contexts.frac &= 32767;
        const uint8_t state = m.contexts.ctx[C_SPLIT + c];
        m.contexts.frac += k_bits[state ^ flag];
        m.contexts.ctx[C_SPLIT + c] = ctxNext(state, flag);
        m.totalBits += (uint32_t)(m.contexts.frac >> 15);
        updateModeCost(m);
    }
    void checkBestMode(Mode& m, int depth)
    {
        if (md[depth].best) { if (m.rdCost < md[depth].best->rdCost) md[depth].best = &m; }
        else md[depth].best = &m;
    }

    /* checkMerge2Nx2N_rd0_4 */
    int checkMerge(int x, int y, int depth)
    {
        XA_HOSTPROF("an.checkMerge (all)");
        StageTimer timer_(0);
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        Mode* tempPred = &d.pred[PRED_MERGE];
        Mode* bestPred = &d.pred[PRED_SKIP];
        tempPred->initCosts(); bestPred->initCosts();
        x265amd_merge_cand cand[5];
        xa_phase(XA_PH_ANALYZER);
        const int numCand = x265amd_merge_candidates(I, cur, col, x, y, log2, 0, 0, cand);
        xa_phase(XA_PH_MERGE_CAND);
        if (numCand <= 0) return 0;
        std::vector<x265amd_mc_job> jobs;
        int tiles[5];
        for (int i = 0; i < numCand; i++)
        {
            tiles[i] = candTile(depth, i);
            jobs.push_back(mcJob(x, y, size, tiles[i], cand[i].dir, cand[i].ref_idx, cand[i].mv));
        }
        x265amd_cu_measure meas[5];
        if (predictAndMeasure(jobs, x, y, log2, tiles, meas)) return err;
        xa_phase(XA_PH_MERGE);
        uint64_t bestCost = kMaxCost; uint32_t bestBits = 0;
        int bestSadCand = -1;
        for (int i = 0; i < numCand; i++)
        {
            /* pictures coded in parallel: candidates reaching below the rows the references have finished are left out (analysis.cpp:2789-2806) */
            if (S->frame_parallel && (below_lag(cand[i].mv[0][1], S->search_range) || below_lag(cand[i].mv[1][1], S->search_range))) continue;
            const uint32_t bits = (uint32_t)(i + (i < numCand - 1));          /* getTUBits */
            const uint64_t c = calcRdSADCost(sa8dOf(meas[i]), bits);
            if (c < bestCost) { bestCost = c; bestBits = bits; bestSadCand = i; }
        }
        if (bestSadCand < 0) return 0;
        d.srcMean = meas[0].src_mean; d.srcHomo = meas[0].src_homo;
        const x265amd_merge_cand& b = cand[bestSadCand];
        const int16_t zero[2][2] = { { 0, 0 }, { 0, 0 } };
        const uint8_t noIdx[2] = { 0, 0 };
        for (Mode* m : { bestPred, tempPred })
        {
            setInter(*m, depth, 1, bestSadCand, b.dir, b.ref_idx, b.mv, zero, noIdx);
            m->sa8dCost = bestCost; m->sa8dBits = bestBits;
            m->predTile = tiles[bestSadCand];
        }
        bestPred->reconTile = reconTile(depth, PRED_SKIP);
        tempPred->reconTile = reconTile(depth, PRED_MERGE);
        if (rp.rdoq_level)
        {
            if (rdInter(*bestPred, x, y, depth, true)) return err;
            if (rdInter(*tempPred, x, y, depth, false)) return err;
        }
        else { xa_phase(XA_PH_ANALYZER); if (rdMergePair(*bestPred, *tempPred, x, y, depth)) return err; xa_phase(XA_PH_MERGE_RD); }
        d.best = tempPred->rdCost < bestPred->rdCost ? tempPred : bestPred;
        /* the winner keeps the candidate's prediction: move it out of the candidate tiles, which the next merge scan reuses */
        const int keep = predTile(depth, d.best == tempPred ? PRED_MERGE : PRED_SKIP);
        copyTile(keep, tiles[bestSadCand], 0, 0, size);
        d.best->predTile = keep;
        return checkDQP(*d.best, x, y, depth);
    }

    /* the fused search's record for a CU -- everything but the addresses of its buffers; false in `ok`: not this configuration */
    int fusedBuild(int x, int y, int depth, uint32_t refMask, XaSearchJob& J, bool& ok)
    {
        ok = false;
        static const bool on = !(getenv("X265AMD_FUSED_SEARCH") && atoi(getenv("X265AMD_FUSED_SEARCH")) == 0);
        const int log2 = 6 - depth, size = 1 << log2;
        const int method = S->search_method & 0x7f;
        if (!on || I->is_inter_b || S->weighted || !xa_is_queue(st) || S->subpel_refine > 2 || (method != X265AMD_ME_DIA && method != X265AMD_ME_HEX && method != X265AMD_ME_STAR) ||
            I->num_ref_idx[0] < 1 || I->num_ref_idx[0] > XA_SEARCH_MAX_REFS || rp.rdoq_level || si->tu_max_depth_inter != 1 || A->rd_level < 3 || log2 > 5)
            return 0;
        if (!xa_me_device_bitsize(me) || !me) return 0;
        const size_t isz = sizeof(pixel);
        (void)isz;
        if (!chain.mLuma.p && chain.mLuma.alloc((size_t)numPics * 8) != hipSuccess) return fail("search records");
        if (!chain.lumaPushed)
        {
            volatile uint64_t* t = (volatile uint64_t*)chain.mLuma.p;
            for (int i = 0; i < numPics; i++) t[i] = planes[3 * i];
            chain.lumaPushed = true;
        }
        ModeDepth& d = md[depth];
        memset(&J, 0, sizeof(J));
        const int lagPixels = S->frame_parallel ? S->search_range : I->pic_height;
        x265amd_me_job guardJobs[2 * XA_SEARCH_MAX_REFS]; int guardPics[2 * XA_SEARCH_MAX_REFS]; int ng = 0;
        for (int ref = 0; ref < I->num_ref_idx[0]; ref++)
        {
            const uint32_t m = refMask ? refMask : 0xFFFFFFFFu;
            if (!((m >> ref) & 1u)) continue;
            XaSearchRef& R = J.ref[J.num_refs++];
            int16_t mvc[12][2];
            R.num_mvc = x265amd_amvp_candidates(I, cur, col, x, y, log2, 0, 0, 0, ref, R.amvp, mvc);
            if (S->lowres_mvs[0][ref])
            {
                /* getLowresMV: the lookahead's vector of the 16x16 block under the PU's centre, scaled up */
                const int16_t (*lm)[2] = reinterpret_cast<const int16_t (*)[2]>((uintptr_t)S->lowres_mvs[0][ref]);
                const size_t idx = (size_t)((y + size / 2) >> 4) * S->lowres_blocks_in_row + ((x + size / 2) >> 4);
                const int16_t lx = (int16_t)(lm[idx][0] * 2), ly = (int16_t)(lm[idx][1] * 2);
                if (lx || ly) { mvc[R.num_mvc][0] = lx; mvc[R.num_mvc][1] = ly; R.num_mvc++; }
            }
            memcpy(R.mvc, mvc, sizeof(R.mvc));
            R.ref_pic = S->ref_pic[0][ref]; R.ref_idx = ref;
            /* what the searches may read of this reference picture, whichever predictor wins: wait for it (xa_ref_guard_me) */
            for (int k = 0; k < 2; k++)
            {
                Mv mn, mx;
                search_range(Mv{ R.amvp[k][0], R.amvp[k][1] }, S->search_range, x, y, I->pic_width, I->pic_height, lagPixels, mn, mx);
                x265amd_me_job& g = guardJobs[ng];
                memset(&g, 0, sizeof(g));
                g.x = (int16_t)x; g.y = (int16_t)y; g.w = (uint8_t)size; g.h = (uint8_t)size;
                g.mvmin[0] = (int16_t)mn.x; g.mvmin[1] = (int16_t)mn.y; g.mvmax[0] = (int16_t)mx.x; g.mvmax[1] = (int16_t)mx.y;
                guardPics[ng++] = R.ref_pic;
            }
        }
        if (!J.num_refs) return 0;
        if (xa_ref_guard_me(guardJobs, guardPics, ng)) return fail("a reference picture failed");
        J.x = x; J.y = y; J.log2 = log2;
        J.pic_w = I->pic_width; J.pic_h = I->pic_height; J.stride = (int32_t)stride; J.cstride = (int32_t)cstride; J.num_pics = numPics; J.num_ref_idx0 = I->num_ref_idx[0];
        J.search_method = S->search_method; J.subme = S->subpel_refine; J.merange = S->search_range; J.me_qp = lambdaQp; J.frame_parallel = S->frame_parallel;
        J.search_range = S->search_range; J.lag_pixels = lagPixels; J.list_sel_bits0 = 1;
        J.me_lambda = (uint64_t)floor(256.0 * is_lambda(lambdaQp));
        J.mvcost = (uint64_t)(uintptr_t)xa_me_device_mvcost(me, lambdaQp); J.bitsize = (uint64_t)(uintptr_t)xa_me_device_bitsize(me); J.me_tables = (uint64_t)(uintptr_t)xa_me_device_tables(me);
        J.planes = (uint64_t)(uintptr_t)dPlanes.p; J.luma_tab = (uint64_t)(uintptr_t)chain.mLuma.p;
        J.pred_tile = tileAddr(predTile(depth, PRED_2Nx2N)); J.recon_tile = tileAddr(reconTile(depth, PRED_2Nx2N));
        J.lambda = lambda; J.lambda2 = lambda2; J.psy_rd = psyRd;
        {
            static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                                     32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
            const int qpQuant = qp < 0 ? 0 : (qp > 51 ? 51 : qp), bd = 6 * (X265AMD_DEPTH - 8);
            int qpC = qpQuant < -bd ? -bd : (qpQuant > 57 ? 57 : qpQuant);
            if (qpC >= 30) qpC = chromaScale[qpC];
            J.qp_luma = qpQuant + bd; J.qp_chroma = qpC + bd;
        }
        J.sign_hide = si->sign_hide != 0; J.chroma_sa8d = A->rd_level >= 3; J.rd_level = A->rd_level; J.do_rd = 1; J.slice_type = si->slice_type;
        {
            const x265amd_cu_unit* l = (x >> 2) > 0 ? &units[(y >> 2) * w4 + (x >> 2) - 1] : nullptr;
            const x265amd_cu_unit* a = (y >> 2) > 0 ? &units[((y >> 2) - 1) * w4 + (x >> 2)] : nullptr;
            J.skip_ctx = (l && l->pred_mode == X265AMD_MODE_SKIP) + (a && a->pred_mode == X265AMD_MODE_SKIP);
        }
        J.frac = d.cur.frac; memcpy(J.ctx, d.cur.ctx, X265AMD_CTX_STRIDE);
        if (si->use_dqp)
        {
            /* Entropy::codeDeltaQP's value (entropy.cpp:1737-1756): the QP in force against the group's prediction -- which nothing inside the group moves, so it can be
             * taken before the CU's merge check and before its sub-CUs (searchAhead) */
            if (!qpCoder)
            {
                if (!(qpCoder = x265amd_cabac_open(si, units, 1))) return fail("delta QP: coder");
                qpCoder->ctuInProgress = true;
            }
            const int bd = 6 * (X265AMD_DEPTH - 8);
            int dqp = qp - qpCoder->refQP(x, y);
            dqp = (dqp + 78 + bd + (bd / 2)) % (52 + bd) - 26 - (bd / 2);
            J.dqp = (1 << 16) | (depth <= si->max_cu_dqp_depth ? 1 << 17 : 0) | (uint8_t)(int8_t)dqp;
        }
        ok = true;
        return 0;
    }
    /* the record goes to buffer set `k` and the command to `q`; nobody waits here */
    int fusedSubmit(XaSearchJob& J, int k, void* q)
    {
        if (!chain.mSearch[k].p && (chain.mSearch[k].alloc(sizeof(XaSearchJob)) != hipSuccess || chain.mSearchOut[k].alloc(sizeof(XaSearchOut)) != hipSuccess ||
                                    chain.dSearchScratch[k].alloc(8192 + (size_t)2 * XA_SEARCH_MAX_REFS * 4096 * sizeof(pixel) + 1536 * (4 + sizeof(pixel)) + 256) != hipSuccess))
            return fail("search records");
        J.scratch = (uint64_t)(uintptr_t)chain.dSearchScratch[k].p; J.out = (uint64_t)(uintptr_t)chain.mSearchOut[k].p;
        {
            volatile uint64_t* dd = (volatile uint64_t*)chain.mSearch[k].p; const uint64_t* ss = (const uint64_t*)&J;
            for (size_t i = 0; i < sizeof(J) / 8; i++) dd[i] = ss[i];
        }
        XaSearchOut* o = (XaSearchOut*)chain.mSearchOut[k].p;
        *(volatile uint32_t*)&o->valid = 0;
        const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)chain.mSearch[k].p, 0, 0, 0, 1 };
        if (xa_q_enqueue(q, XA_OP_INTER_SEARCH, &qa, sizeof(qa), 1, 0) != hipSuccess) return fail("search command");
        return 0;
    }
    /* a search started ahead of the point where the reference runs it (struct Chain::Ahead): beside the merge check of a CU that cannot split (leaf: the row's second
     * queue), or behind the merge check of a CU whose sub-CUs come first (the third queue) */
    int searchAhead(int x, int y, int depth, bool leaf)
    {
        static const int on = getenv("X265AMD_SEARCH_AHEAD") ? atoi(getenv("X265AMD_SEARCH_AHEAD")) : 3;      /* bit 0: leaves, bit 1: CUs with sub-CUs */
        if (!(on & (leaf ? 1 : 2))) return 0;
        void* aux = xa_queue_aux(st);
        void* q = leaf ? aux : (aux && xa_queue_aux(aux) ? xa_queue_aux(aux) : aux);
        if (!q) return 0;
        Chain::Ahead& a = chain.ahead[depth];
        if (a.on) { a.on = false; if (xa_stream_sync(a.q) != hipSuccess) return fail("search ahead"); }
        bool ok = false;
        if (fusedBuild(x, y, depth, 0, a.J, ok)) return err;
        if (!ok) return 0;
        /* the other queue's workgroup reads what this CTU's queue has been sent so far (the plane table) */
        if (xa_queue_follow(q, st) != hipSuccess || fusedSubmit(a.J, 1 + depth, q)) return fail("search ahead");
        a.on = true; a.x = x; a.y = y; a.q = q;
        if (g_timing) g_aheadStat[leaf ? 0 : 2]++;
        return 0;
    }

    /* checkInter_rd0_4(2Nx2N) of a CU of a P picture as ONE device command: the predictors' costs, the searches in every allowed reference picture, the choice, the
     * prediction with its SA8D and (rd 3+, one transform unit per plane) encodeResAndCalcRdInterCU -- inter_search_dev.h.  The host derives what depends on the maps (the
     * AMVP candidates, the search's extra candidates) and waits once.  false in `used`: not this configuration, the ordinary path runs */
    int checkInterFused(int x, int y, int depth, uint32_t refMask, bool& used)
    {
        used = false;
        XA_HOSTPROF("an.checkInterFused (all)");
        StageTimer timer_(1);
        ModeDepth& d = md[depth];
        Mode& inter = d.pred[PRED_2Nx2N];
        const int log2 = 6 - depth, size = 1 << log2;
        XaSearchJob local;
        const XaSearchJob* Jp = nullptr;
        int set = 0;
        Chain::Ahead& ah = chain.ahead[depth];
        if (ah.on)
        {
            /* a search started ahead at this depth: this CU's, or one that was never asked for (a leaf the chain skipped) */
            ah.on = false;
            xa_phase(XA_PH_ANALYZER);
            if (xa_stream_sync(ah.q) != hipSuccess || xa_stream_fence(st, XA_CMD_ACQUIRE) != hipSuccess) return fail("search ahead");
            xa_phase(XA_PH_INTER_SEARCH);
            if (ah.x == x && ah.y == y)
            {
                const XaSearchOut* ao = (const XaSearchOut*)chain.mSearchOut[1 + depth].p;
                const bool winnerAllowed = ao->valid == 1 && ao->best >= 0 && ao->best < ah.J.num_refs && (!refMask || ((refMask >> ah.J.ref[ao->best].ref_idx) & 1u));
                if (winnerAllowed) { Jp = &ah.J; set = 1 + depth; if (g_timing) g_aheadStat[1]++; }
                else if (g_timing) g_aheadStat[3]++;
            }
        }
        if (!Jp)
        {
            bool ok = false;
            if (fusedBuild(x, y, depth, refMask, local, ok)) return err;
            if (!ok) return 0;
            xa_phase(XA_PH_ANALYZER);
            if (fusedSubmit(local, 0, st) || xa_stream_sync(st) != hipSuccess) return fail("search command");
            xa_phase(XA_PH_INTER_SEARCH);
            Jp = &local;
        }
        const XaSearchJob& J = *Jp;
        inter.initCosts();
        inter.predTile = predTile(depth, PRED_2Nx2N); inter.reconTile = reconTile(depth, PRED_2Nx2N);
        XaSearchOut* o = (XaSearchOut*)chain.mSearchOut[set].p;
        if (o->valid != 1 || o->best < 0 || o->best >= J.num_refs) return fail("search result");
        const XaSearchRef& R = J.ref[o->best];
        int8_t refs[2] = { (int8_t)R.ref_idx, -1 };
        int16_t mv[2][2] = { { o->mv[0], o->mv[1] }, { 0, 0 } }, mvd[2][2] = { { (int16_t)(o->mv[0] - o->mvp[0]), (int16_t)(o->mv[1] - o->mvp[1]) }, { 0, 0 } };
        uint8_t mvpIdx[2] = { (uint8_t)o->mvp_idx, 0 };
        setInter(inter, depth, 0, o->mvp_idx, 1, refs, mv, mvd, mvpIdx);
        d.mvCost2Nx2N[0] = o->mv_cost; d.mvCost2Nx2N[1] = 0;
        memset(&d.det, 0, sizeof(d.det));
        d.det.cost[0] = o->cost; d.det.cost[1] = 0xFFFFFFFFu; d.det.ref[0] = (int8_t)R.ref_idx; d.det.ref[1] = -1;
        inter.sa8dBits = o->bits;
        const uint32_t sa8d = A->rd_level >= 3 ? o->sa8d : o->sa8d_luma;
        inter.distortion = sa8d;
        inter.sa8dCost = calcRdSADCost(sa8d, inter.sa8dBits);
        fusedRd[depth] = false;
        d.pred[PRED_BIDIR].initCosts(); d.pred[PRED_BIDIR].sa8dCost = kMaxCost; d.pred[PRED_BIDIR].rdCost = kMaxCost;       /* checkBidir2Nx2N in a P slice: nothing to try */
        if (o->rd_done)
        {
            const int n4 = 16 >> depth;
            for (int i = 0; i < n4 * n4; i++)
            {
                x265amd_cu_unit& u = inter.u[i];
                u.depth = (uint8_t)depth; u.tu_depth = 0; u.cbf[0] = o->cbf[0]; u.cbf[1] = o->cbf[1]; u.cbf[2] = o->cbf[2]; u.pred_mode = X265AMD_MODE_INTER;
                inter.m[i].pred_mode = X265AMD_MODE_INTER;
            }
            std::fill(inter.coeff.begin(), inter.coeff.end(), 0);
            if (o->cbf[0] || o->cbf[1] || o->cbf[2])
            {
                memcpy(&inter.coeff[0], o->levels, sizeof(int16_t) * size * size);
                memcpy(&inter.coeff[4096], o->levels + 1024, sizeof(int16_t) * (size >> 1) * (size >> 1));
                memcpy(&inter.coeff[5120], o->levels + 1280, sizeof(int16_t) * (size >> 1) * (size >> 1));
            }
            inter.rdCost = o->rd_cost; inter.lumaDistortion = (sse_t)o->luma_dist; inter.chromaDistortion = (sse_t)o->chroma_dist; inter.distortion = (sse_t)(o->luma_dist + o->chroma_dist);
            inter.totalBits = o->total_bits; inter.mvBits = o->mv_bits; inter.coeffBits = o->coeff_bits; inter.psyEnergy = o->psy_energy; inter.resEnergy = (sse_t)o->res_energy;
            memset(inter.contexts.ctx, 0, X265AMD_CTX_STRIDE);
            memcpy(inter.contexts.ctx, (const void*)o->ctx, X265AMD_CTX_COUNT);
            inter.contexts.frac = o->frac;
            if (si->use_dqp && depth <= si->max_cu_dqp_depth && !(o->cbf[0] || o->cbf[1] || o->cbf[2]))
            {
                /* checkDQP of a CU without a residual: setQPSubParts(getRefQP) (search.cpp:3996-3999) */
                const int8_t refQp = (int8_t)qpCoder->refQP(x, y);
                for (int i = 0; i < n4 * n4; i++) inter.u[i].qp = refQp;
            }
            fusedRd[depth] = true;
        }
        used = true;
        return 0;
    }

    /* checkInter_rd0_4(2Nx2N) + checkBidir2Nx2N; searchOnly: predInterSearch alone (checkInter_rd5_6 runs the RD itself, the bi-prediction try comes later) */
    int checkInter(int x, int y, int depth, uint32_t refMask, bool searchOnly = false)
    {
        XA_HOSTPROF("an.checkInter (all)");
        StageTimer timer_(1);
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        Mode& inter = d.pred[PRED_2Nx2N];
        inter.initCosts();
        inter.predTile = predTile(depth, PRED_2Nx2N); inter.reconTile = reconTile(depth, PRED_2Nx2N);
        x265amd_inter_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)log2; c.part_size = 0;
        x265amd_pu_result pu[2];
        int32_t bits = 0;
        x265amd_me_detail& det = d.det;
        x265amd_inter_search_params sp = *S;
        sp.qp = lambdaQp; sp.chroma_mc = A->rd_level >= 3;        /* bChromaMC = m_bChromaSa8d: below rd 3 the search is luma only (no chroma SATD either) */
        static const bool lazyMc = !(getenv("X265AMD_LAZY_MC") && atoi(getenv("X265AMD_LAZY_MC")) == 0);
        sp.lazy_sync = !searchOnly && lazyMc;       /* the measurement below waits for the final prediction */
        const uint32_t masks[2] = { refMask, 0 };
        xa_phase(XA_PH_ANALYZER);
        struct PhEnd { ~PhEnd() { xa_phase(XA_PH_INTER_SEARCH); } } phEnd;
        int rc = x265amd_pred_inter_search_ex(me, st, I, &sp, cur, col, planes, numPics, stride, cstride, &c, 1, pu, &bits, tileAddr(inter.predTile), tileBytes, &det, masks);
        if (rc != X265AMD_OK) return err = rc;
        setInter(inter, depth, pu[0].merge_flag, pu[0].mvp_idx[0], pu[0].inter_dir, pu[0].ref_idx, pu[0].mv, pu[0].mvd, pu[0].mvp_idx);
        d.mvCost2Nx2N[0] = det.mv_cost[0]; d.mvCost2Nx2N[1] = det.mv_cost[1];       /* bestME[0][list].mvCost (zero where the list was not searched) */
        inter.sa8dBits = (uint32_t)bits;
        if (searchOnly) return 0;
        x265amd_rd_cu rc1;
        memset(&rc1, 0, sizeof(rc1));
        rc1.x = (int16_t)x; rc1.y = (int16_t)y; rc1.log2_size = (uint8_t)log2;
        x265amd_cu_measure ms;
        if (x265amd_measure_tiles(st, planes + 3 * (numPics - 1), stride, cstride, &rc1, 1, tileAddr(inter.predTile), tileBytes, &ms) != X265AMD_OK) return err = X265AMD_EHIP;
        inter.sa8dBits = (uint32_t)bits;
        inter.distortion = sa8dOf(ms);
        inter.sa8dCost = calcRdSADCost(sa8dOf(ms), inter.sa8dBits);
        return checkBidir(x, y, depth);
    }
    /* checkBidir2Nx2N (analysis.cpp:3145-3277) on what the 2Nx2N search left in Mode::bestME */
    int checkBidir(int x, int y, int depth)
    {
        StageTimer timer_(4);
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        Mode& bidir = d.pred[PRED_BIDIR];
        const x265amd_me_detail& det = d.det;
        bidir.initCosts();
        bidir.sa8dCost = kMaxCost; bidir.rdCost = kMaxCost;
        if (!I->is_inter_b || det.cost[0] == 0xFFFFFFFFu || det.cost[1] == 0xFFFFFFFFu) return 0;
        bidir.predTile = predTile(depth, PRED_BIDIR); bidir.reconTile = reconTile(depth, PRED_BIDIR);
        const int8_t refs[2] = { det.ref[0], det.ref[1] };
        int16_t mv[2][2] = { { det.mv[0][0], det.mv[0][1] }, { det.mv[1][0], det.mv[1][1] } };
        Mv mvp[2] = { Mv{ det.mvp[0][0], det.mvp[0][1] }, Mv{ det.mvp[1][0], det.mvp[1][1] } };
        int mvpIdx[2] = { det.mvp_idx[0], det.mvp_idx[1] };
        const Mv best[2] = { Mv{ mv[0][0], mv[0][1] }, Mv{ mv[1][0], mv[1][1] } };
        bool bTryZero = best[0].x || best[0].y || best[1].x || best[1].y;
        if (bTryZero)
        {
            Mv mn, mx;
            search_range(Mv{ 0, 0 }, I->pic_width > I->pic_height ? I->pic_width : I->pic_height, x, y, I->pic_width, I->pic_height,
                         S->frame_parallel ? S->search_range : I->pic_height, mn, mx);
            mx.y += 2;
            mn.x <<= 2; mn.y <<= 2; mx.x <<= 2; mx.y <<= 2;
            for (int l = 0; l < 2; l++) bTryZero &= mvp[l].x >= mn.x && mvp[l].x <= mx.x && mvp[l].y >= mn.y && mvp[l].y <= mx.y;
        }
        std::vector<x265amd_mc_job> jobs;
        int tiles[2] = { bidir.predTile, candTile(depth, 5) };
        const int16_t zeroMv[2][2] = { { 0, 0 }, { 0, 0 } };
        jobs.push_back(mcJob(x, y, size, tiles[0], 3, refs, mv));
        if (bTryZero) jobs.push_back(mcJob(x, y, size, tiles[1], 3, refs, zeroMv));
        x265amd_cu_measure meas[2];
        if (predictAndMeasure(jobs, x, y, log2, tiles, meas)) return err;
        const uint32_t* ls = det.list_sel_bits;
        bidir.sa8dBits = det.bits[0] + det.bits[1] + ls[2] - (ls[0] + ls[1]);
        bidir.sa8dCost = (uint64_t)sa8dOf(meas[0]) + getCost(bidir.sa8dBits);
        bool zeroWins = false;
        if (bTryZero)
        {
            const Mv zero{ 0, 0 };
            uint32_t bits0 = det.bits[0] - is_bitcost(best[0], mvp[0]) + is_bitcost(zero, mvp[0]);
            uint32_t bits1 = det.bits[1] - is_bitcost(best[1], mvp[1]) + is_bitcost(zero, mvp[1]);
            for (int l = 0; l < 2; l++)             /* checkBestMVP (search.cpp:2702-2713): only the bits and the predictor matter here */
            {
                const Mv am[2] = { Mv{ det.amvp[l][0][0], det.amvp[l][0][1] }, Mv{ det.amvp[l][1][0], det.amvp[l][1][1] } };
                uint32_t& b = l ? bits1 : bits0;
                const int diffBits = (int)is_bitcost(zero, am[!mvpIdx[l]]) - (int)is_bitcost(zero, am[mvpIdx[l]]);
                if (diffBits < 0) { mvpIdx[l] = !mvpIdx[l]; b = b + diffBits; }
                mvp[l] = am[mvpIdx[l]];
            }
            const uint32_t zbits = bits0 + bits1 + ls[2] - (ls[0] + ls[1]);
            const uint32_t zcost = sa8dOf(meas[1]) + getCost(zbits);
            if (zcost < bidir.sa8dCost)
            {
                bidir.sa8dBits = zbits; bidir.sa8dCost = zcost;
                mv[0][0] = mv[0][1] = mv[1][0] = mv[1][1] = 0;
                copyTile(bidir.predTile, tiles[1], 0, 0, size);
                zeroWins = true;
            }
        }
        if (!zeroWins) { mvp[0] = Mv{ det.mvp[0][0], det.mvp[0][1] }; mvp[1] = Mv{ det.mvp[1][0], det.mvp[1][1] }; mvpIdx[0] = det.mvp_idx[0]; mvpIdx[1] = det.mvp_idx[1]; }
        int16_t mvd[2][2]; uint8_t idx[2];
        for (int l = 0; l < 2; l++) { mvd[l][0] = (int16_t)(mv[l][0] - mvp[l].x); mvd[l][1] = (int16_t)(mv[l][1] - mvp[l].y); idx[l] = (uint8_t)mvpIdx[l]; }
        setInter(bidir, depth, 0, 0, 3, refs, mv, mvd, idx);
        return 0;
    }

    /* compressInterCU_rd0_4 */
    static uint32_t bestRefIdx(const x265amd_cu_unit& u)        /* CUData::getBestRefIdx (cudata.h:279-280) */
    {
        return ((u.inter_dir & 1) && u.ref_idx[0] >= 0 ? 1u << u.ref_idx[0] : 0) | ((u.inter_dir & 2) && u.ref_idx[1] >= 0 ? 1u << (u.ref_idx[1] + 16) : 0);
    }

    /* checkInter_rd0_4 for a two-part CU (rect / AMP): predInterSearch of both PUs, then the SA8D of the whole CU's prediction */
    int checkInterPart(int x, int y, int depth, int part, int slot, const uint32_t refMasks[2], bool searchOnly = false)
    {
        StageTimer timer_(1);
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2, n4 = size >> 2;
        Mode& m = d.pred[slot];
        m.initCosts();
        m.predTile = predTile(depth, slot); m.reconTile = reconTile(depth, slot);
        x265amd_inter_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)log2; c.part_size = (uint8_t)part;
        x265amd_pu_result pu[2];
        int32_t bits = 0;
        x265amd_inter_search_params sp = *S;
        sp.qp = lambdaQp; sp.chroma_mc = A->rd_level >= 3;        /* bChromaMC = m_bChromaSa8d: below rd 3 the search is luma only (no chroma SATD either) */
        static const bool lazyMc = !(getenv("X265AMD_LAZY_MC") && atoi(getenv("X265AMD_LAZY_MC")) == 0);
        sp.lazy_sync = !searchOnly && lazyMc;
        int rc = x265amd_pred_inter_search_ex(me, st, I, &sp, cur, col, planes, numPics, stride, cstride, &c, 1, pu, &bits, tileAddr(m.predTile), tileBytes, nullptr, refMasks);
        if (rc != X265AMD_OK) return err = rc;
        for (int i = 0; i < n4 * n4; i++) { m.u[i].pred_mode = X265AMD_MODE_INTER; m.u[i].part_size = (uint8_t)part; }
        for (int k = 0; k < 2; k++)
        {
            const Geo g = pu_geo(0, 0, size, part, k);
            const x265amd_pu_result& r = pu[k];
            for (int yy = g.y >> 2; yy < (g.y + g.h) >> 2; yy++)
                for (int xx = g.x >> 2; xx < (g.x + g.w) >> 2; xx++)
                {
                    x265amd_cu_unit& u = m.u[yy * n4 + xx];
                    x265amd_mv_unit& v = m.m[yy * n4 + xx];
                    u.merge_flag = r.merge_flag; u.inter_dir = r.inter_dir;
                    v.pred_mode = X265AMD_MODE_INTER; v.inter_dir = r.inter_dir;
                    for (int l = 0; l < 2; l++)
                    {
                        const bool used = (r.inter_dir >> l) & 1;
                        u.ref_idx[l] = used ? r.ref_idx[l] : -1; u.mvp_idx[l] = r.mvp_idx[l];
                        u.mvd[l][0] = r.merge_flag ? 0 : r.mvd[l][0]; u.mvd[l][1] = r.merge_flag ? 0 : r.mvd[l][1];
                        v.ref_idx[l] = used ? r.ref_idx[l] : -1; v.mv[l][0] = used ? r.mv[l][0] : 0; v.mv[l][1] = used ? r.mv[l][1] : 0;
                    }
                }
        }
        m.sa8dBits = (uint32_t)bits;
        if (searchOnly) return 0;
        x265amd_rd_cu rc1;
        memset(&rc1, 0, sizeof(rc1));
        rc1.x = (int16_t)x; rc1.y = (int16_t)y; rc1.log2_size = (uint8_t)log2;
        x265amd_cu_measure ms;
        if (x265amd_measure_tiles(st, planes + 3 * (numPics - 1), stride, cstride, &rc1, 1, tileAddr(m.predTile), tileBytes, &ms) != X265AMD_OK) return err = X265AMD_EHIP;
        m.distortion = sa8dOf(ms);
        m.sa8dCost = calcRdSADCost(sa8dOf(ms), m.sa8dBits);
        return 0;
    }
    uint32_t bestRefIdxCu(const Mode& m, int depth) const       /* OR of getBestRefIdx over the CU's PUs */
    {
        const int size = 64 >> depth, n4 = size >> 2, part = m.u[0].part_size;
        static const uint8_t nb[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
        uint32_t r = 0;
        for (int k = 0; k < nb[part]; k++) { const Geo g = pu_geo(0, 0, size, part, k); r |= bestRefIdx(m.u[(g.y >> 2) * n4 + (g.x >> 2)]); }
        return r;
    }

    /* checkMerge2Nx2N_rd5_6 (analysis.cpp:2883-3019): every merge candidate by RD, with residual until one codes without, and as a skip */
    int checkMerge56(int x, int y, int depth)
    {
        StageTimer timer_(0);
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        Mode* tempPred = &d.pred[PRED_MERGE];
        Mode* bestPred = &d.pred[PRED_SKIP];
        tempPred->initCosts(); bestPred->initCosts();
        tempPred->predTile = predTile(depth, PRED_MERGE); tempPred->reconTile = reconTile(depth, PRED_MERGE);
        bestPred->predTile = predTile(depth, PRED_SKIP); bestPred->reconTile = reconTile(depth, PRED_SKIP);
        x265amd_merge_cand cand[5];
        const int numCand = x265amd_merge_candidates(I, cur, col, x, y, log2, 0, 0, cand);
        bool foundCbf0Merge = false, triedPZero = false, triedBZero = false;
        bestPred->rdCost = kMaxCost;
        const int16_t zero[2][2] = { { 0, 0 }, { 0, 0 } };
        const uint8_t noIdx[2] = { 0, 0 };
        for (int i = 0; i < numCand; i++)
        {
            const x265amd_merge_cand& c = cand[i];
            if (S->frame_parallel && (below_lag(c.mv[0][1], S->search_range) || below_lag(c.mv[1][1], S->search_range))) continue;       /* analysis.cpp:2919-2936 */
            const bool z0 = !c.mv[0][0] && !c.mv[0][1] && !c.ref_idx[0], z1 = !c.mv[1][0] && !c.mv[1][1] && !c.ref_idx[1];
            if (c.dir == 1 && z0) { if (triedPZero) continue; triedPZero = true; }
            else if (c.dir == 3 && z0 && z1) { if (triedBZero) continue; triedBZero = true; }
            setInter(*tempPred, depth, 1, i, c.dir, c.ref_idx, c.mv, zero, noIdx);
            {
                std::vector<x265amd_mc_job> jobs;
                jobs.push_back(mcJob(x, y, size, tempPred->predTile, c.dir, c.ref_idx, c.mv));
                x265amd_cu_measure meas;
                const int tiles[1] = { tempPred->predTile };
                if (predictAndMeasure(jobs, x, y, log2, tiles, &meas)) return err;
            }
            bool hasCbf = true, swapped = false;
            if (!foundCbf0Merge)
            {
                if (rdInter(*tempPred, x, y, depth, false)) return err;
                hasCbf = tempPred->u[0].cbf[0] || tempPred->u[0].cbf[1] || tempPred->u[0].cbf[2];
                foundCbf0Merge = !hasCbf;
                if (tempPred->rdCost < bestPred->rdCost) { std::swap(tempPred, bestPred); swapped = true; }
            }
            if (hasCbf)
            {
                if (swapped)
                {
                    setInter(*tempPred, depth, 1, i, c.dir, c.ref_idx, c.mv, zero, noIdx);
                    copyTile(tempPred->predTile, bestPred->predTile, 0, 0, size);
                }
                if (rdInter(*tempPred, x, y, depth, true)) return err;
                if (tempPred->rdCost < bestPred->rdCost) std::swap(tempPred, bestPred);
            }
        }
        if (bestPred->rdCost < kMaxCost) { d.best = bestPred; return checkDQP(*bestPred, x, y, depth); }
        return 0;
    }

    /* compressInterCU_rd5_6 (analysis.cpp:1850-2417) without analysis reuse / CTU info / lossless / edge-based rskip */
    int compress56(int x, int y, int depth, SplitData& splitOut)
    {
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        d.best = nullptr;
        const bool mightSplit = depth < si->max_cu_depth;
        const bool mightNotSplit = x + size <= I->pic_width && y + size <= I->pic_height;
        bool skipModes = false, skipRecursion = false, splitIntra = true;
        const int myQp = qp;                    /* the `qp` argument of compressInterCU_rd5_6 */
        SplitData splitData[4];
        memset(splitData, 0, sizeof(splitData));
        d.mvCost2Nx2N[0] = d.mvCost2Nx2N[1] = 0;
        for (int k = 0; k < NUM_PRED; k++) initSubCU(d.pred[k], depth);
        d.pred[PRED_2Nx2N].rdCost = 0;
        loadTUDepth(x, y, depth);           /* analysis.cpp:1907-1908 */
        uint32_t allSplitRefs = 0;
        auto rootCbf = [](const Mode& m) { return m.u[0].cbf[0] || m.u[0].cbf[1] || m.u[0].cbf[2]; };
        auto interRd = [&](int part, int slot, uint32_t m0, uint32_t m1) -> int {          /* checkInter_rd5_6 + checkBestMode */
            const uint32_t masks[2] = { m0, m1 };
            if (part == 0 ? checkInter(x, y, depth, m0, true) : checkInterPart(x, y, depth, part, slot, masks, true)) return err;
            if (rdInter(d.pred[slot], x, y, depth, false)) return err;
            checkBestMode(d.pred[slot], depth);
            return 0;
        };
        /* Step 1: merge / skip candidates and 2Nx2N */
        if (mightNotSplit)
        {
            if (checkMerge56(x, y, depth)) return err;
            skipModes = A->early_skip && d.best && !rootCbf(*d.best);
            if (interRd(0, PRED_2Nx2N, allSplitRefs, 0)) return err;
            if (A->rskip == 1 && depth && md[depth - 1].best) skipRecursion = d.best && !rootCbf(*d.best);
        }
        /* Step 2: the four sub-blocks in series */
        if (mightSplit && !skipRecursion)
        {
            Mode& split = d.pred[PRED_SPLIT];
            split.initCosts();
            split.predTile = predTile(depth, PRED_SPLIT); split.reconTile = reconTile(depth, PRED_SPLIT);
            const int n4 = 16 >> depth, half = size >> 1, h4n = n4 >> 1;
            const Snap* nextContext = &d.cur;
            splitIntra = false;
            for (int q = 0; q < 4; q++)
            {
                const int cx = x + (q & 1) * half, cy = y + (q >> 1) * half;
                if (cx < I->pic_width && cy < I->pic_height)
                {
                    md[depth + 1].cur = *nextContext;
                    if (childQp(depth, q)) return err;
                    if (compress56(cx, cy, depth + 1, splitData[q])) return err;
                    const Mode& nb = *md[depth + 1].best;
                    splitIntra |= nb.u[0].pred_mode == X265AMD_MODE_INTRA;
                    for (int yy = 0; yy < h4n; yy++)
                        for (int xx = 0; xx < h4n; xx++)
                        {
                            split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.u[yy * h4n + xx];
                            split.m[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.m[yy * h4n + xx];
                        }
                    split.addSubCosts(nb);
                    copyTile(split.reconTile, nb.reconTile, (q & 1) * half, (q >> 1) * half, half);
                    const int nc = half * half;
                    memcpy(&split.coeff[(size_t)q * nc], nb.coeff.data(), sizeof(int16_t) * nc);
                    memcpy(&split.coeff[4096 + (size_t)q * nc / 4], nb.coeff.data() + 4096, sizeof(int16_t) * nc / 4);
                    memcpy(&split.coeff[5120 + (size_t)q * nc / 4], nb.coeff.data() + 5120, sizeof(int16_t) * nc / 4);
                    nextContext = &nb.contexts;
                }
                else
                    for (int yy = 0; yy < h4n; yy++)            /* setEmptyPart */
                        for (int xx = 0; xx < h4n; xx++) split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx].depth = (uint8_t)(depth + 1);
            }
            split.contexts = *nextContext;
            if (mightNotSplit) addSplitFlagCost(split, x, y, depth);
            else updateModeCost(split);
            if (checkDQPForSplitPred(split, x, y, depth)) return err;          /* analysis.cpp:2082 */
        }
        allSplitRefs = splitData[0].splitRefs | splitData[1].splitRefs | splitData[2].splitRefs | splitData[3].splitRefs;
        /* Step 3: bi-prediction, rectangular / asymmetric partitions and intra at the current depth */
        if (mightNotSplit)
        {
            if (si->use_dqp && depth <= si->max_cu_dqp_depth && si->max_cu_dqp_depth != 0 && setLambdaFromQP(myQp)) return err;       /* analysis.cpp:2105-2106 */
            if (!skipModes)
            {
                if (A->limit_refs & 2)
                {
                    allSplitRefs = bestRefIdx(d.pred[PRED_2Nx2N].u[0]);
                    for (int q = 0; q < 4; q++) splitData[q].splitRefs = allSplitRefs;
                }
                if (I->is_inter_b)
                {
                    if (checkBidir(x, y, depth)) return err;
                    Mode& bidir = d.pred[PRED_BIDIR];
                    if (bidir.sa8dCost < kMaxCost)
                    {
                        if (rdInter(bidir, x, y, depth, false)) return err;
                        checkBestMode(bidir, depth);
                    }
                }
                const bool isP = !I->is_inter_b;
                auto thr = [&](int a, int b) { return isP ? splitData[a].mvCost[0] + splitData[b].mvCost[0]
                                                          : (splitData[a].mvCost[0] + splitData[b].mvCost[0] + splitData[a].mvCost[1] + splitData[b].mvCost[1] + 1) >> 1; };
                const uint64_t splitCost = splitData[0].sa8dCost + splitData[1].sa8dCost + splitData[2].sa8dCost + splitData[3].sa8dCost;
                const uint32_t top = splitData[0].splitRefs | splitData[1].splitRefs, bot = splitData[2].splitRefs | splitData[3].splitRefs;
                const uint32_t lft = splitData[0].splitRefs | splitData[2].splitRefs, rgt = splitData[1].splitRefs | splitData[3].splitRefs;
                if (A->rect)
                {
                    const uint32_t t2NxN = thr(0, 1), tNx2N = thr(0, 2);
                    const bool first2NxN = t2NxN < tNx2N;
                    if (first2NxN && splitCost < d.best->rdCost + t2NxN) { if (interRd(1, PRED_2NxN, top, bot)) return err; }
                    if (splitCost < d.best->rdCost + tNx2N) { if (interRd(2, PRED_Nx2N, lft, rgt)) return err; }
                    if (!first2NxN && splitCost < d.best->rdCost + t2NxN) { if (interRd(1, PRED_2NxN, top, bot)) return err; }
                }
                if ((A->rect || A->amp) && A->amp && si->max_amp_depth > depth)
                {
                    const uint32_t tU = thr(0, 1), tD = thr(2, 3), tL = thr(0, 2), tR = thr(1, 3);
                    bool bHor = false, bVer = false;
                    const int bp = d.best->u[0].part_size;
                    if (bp == 1) bHor = true;
                    else if (bp == 2) bVer = true;
                    else if (bp == 0 && !d.best->u[0].merge_flag) { bHor = true; bVer = true; }
                    if (bHor)
                    {
                        const bool firstD = tD < tU;
                        if (firstD && splitCost < d.best->rdCost + tD) { if (interRd(5, PRED_2NxnD, allSplitRefs, bot)) return err; }
                        if (splitCost < d.best->rdCost + tU) { if (interRd(4, PRED_2NxnU, top, allSplitRefs)) return err; }
                        if (!firstD && splitCost < d.best->rdCost + tD) { if (interRd(5, PRED_2NxnD, allSplitRefs, bot)) return err; }
                    }
                    if (bVer)
                    {
                        const bool firstR = tR < tL;
                        if (firstR && splitCost < d.best->rdCost + tR) { if (interRd(7, PRED_nRx2N, allSplitRefs, rgt)) return err; }
                        if (splitCost < d.best->rdCost + tL) { if (interRd(6, PRED_nLx2N, lft, allSplitRefs)) return err; }
                        if (!firstR && splitCost < d.best->rdCost + tR) { if (interRd(7, PRED_nRx2N, allSplitRefs, rgt)) return err; }
                    }
                }
                if ((!I->is_inter_b || A->b_intra) && log2 != 6 && (!A->limit_refs || splitIntra))
                {
                    if (rdIntra(d.pred[PRED_INTRA], x, y, depth, PRED_INTRA, true, 0)) return err;
                    checkBestMode(d.pred[PRED_INTRA], depth);
                    if (log2 == 3 && si->tu_log2_min < 3)
                    {
                        if (rdIntra(d.pred[PRED_INTRA_NxN], x, y, depth, PRED_INTRA_NxN, true, 3)) return err;
                        checkBestMode(d.pred[PRED_INTRA_NxN], depth);
                    }
                }
            }
            if (mightSplit) addSplitFlagCost(*d.best, x, y, depth);
        }
        if (mightNotSplit && d.best) saveTUDepth(*d.best, x, y, depth);          /* (before the split is compared: analysis.cpp:2329-2339) */
        if (mightSplit && !skipRecursion)
        {
            Mode& split = d.pred[PRED_SPLIT];
            if (!d.best) d.best = &split;
            else checkBestMode(split, depth);
        }
        memset(&splitOut, 0, sizeof(splitOut));
        if (A->limit_refs & 1)
            splitOut.splitRefs = d.best == &d.pred[PRED_SPLIT] ? allSplitRefs
                                                                : bestRefIdxCu(d.best->u[0].pred_mode == X265AMD_MODE_INTRA ? d.pred[PRED_2Nx2N] : *d.best, depth);
        if (A->limit_modes)
        {
            splitOut.mvCost[0] = d.mvCost2Nx2N[0]; splitOut.mvCost[1] = d.mvCost2Nx2N[1];
            splitOut.sa8dCost = d.pred[PRED_2Nx2N].rdCost;
        }
        toPicture(*d.best, x, y, depth);
        tileToPicture(d.best->reconTile, x, y, size);
        return 0;
    }

    /* compressIntraCU (analysis.cpp:514-668) without analysis reuse / split-rd-skip */
    int compressIntra(int x, int y, int depth)
    {
        XA_HOSTPROF("an.compressIntra (all, children included)");
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        d.best = nullptr;
        const bool mightSplit = depth < si->max_cu_depth;
        const bool mightNotSplit = x + size <= I->pic_width && y + size <= I->pic_height;
        for (int k = 0; k < NUM_PRED; k++) initSubCU(d.pred[k], depth);
        /* a 16x16 CU: its 2Nx2N evaluation may start now on a queue of its own and be collected after the four sub-CUs (intra_rd.hip); the comparisons keep
         * the reference's order */
        bool deferred = false, deferredDone = false;
        if ((log2 == 4 || log2 == 5) && mightNotSplit && mightSplit)
        {
            Mode& m = d.pred[PRED_INTRA];
            x265amd_rd_cu c;
            memset(&c, 0, sizeof(c));
            c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)log2; cuQp2(c);
            memcpy(c.ctx, d.cur.ctx, X265AMD_CTX_COUNT);
            c.frac_bits = d.cur.frac;
            int b;
            if (log2 == 5)
            {
                XA_HOSTPROF("an.intra begin 32");
                b = xa_check_intra_begin_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, &c,
                                            tileAddr(predTile(depth, PRED_INTRA)), tileAddr(reconTile(depth, PRED_INTRA)), &intraWs);
            }
            else
            {
                XA_HOSTPROF("an.intra begin 16");
                b = xa_check_intra_begin_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, &c,
                                            tileAddr(predTile(depth, PRED_INTRA)), tileAddr(reconTile(depth, PRED_INTRA)), &intraWs);
            }
            if (b < 0) return err = b;
            deferred = b == 1;
            (void)m;
        }
        if (log2 != 6 && mightNotSplit && !deferred)
        {
            if (log2 == 3 && si->tu_log2_min < 3 && xa_queue_helper(st))         /* the NxN try below may start beside this one */
                xa_intra_ws_hint_nxn(&intraWs, tileAddr(predTile(depth, PRED_INTRA_NxN)), tileAddr(reconTile(depth, PRED_INTRA_NxN)));
            if (rdIntra(d.pred[PRED_INTRA], x, y, depth, PRED_INTRA, true, 0)) return err;
            checkBestMode(d.pred[PRED_INTRA], depth);
            if (log2 == 3 && si->tu_log2_min < 3)
            {
                if (rdIntra(d.pred[PRED_INTRA_NxN], x, y, depth, PRED_INTRA_NxN, true, 3)) return err;
                checkBestMode(d.pred[PRED_INTRA_NxN], depth);
            }
            if (mightSplit) addSplitFlagCost(*d.best, x, y, depth);
        }
        if (mightSplit)
        {
            Mode& split = d.pred[PRED_SPLIT];
            split.initCosts();
            split.predTile = predTile(depth, PRED_SPLIT); split.reconTile = reconTile(depth, PRED_SPLIT);
            const int n4 = 16 >> depth, half = size >> 1, h4n = n4 >> 1;
            const Snap* nextContext = &d.cur;
            /* a 16x16 block's four 8x8 CUs: decided on the device one after the other, the host reads the four results (intra_rd.hip: xa_intra_quad8_ws) */
            bool chained = false;
            if (log2 == 4 && mightNotSplit && xa_is_queue(st))
            {
                x265amd_intra_cu8_result r4[4];
                const uint64_t tilesN[2] = { tileAddr(predTile(depth + 1, PRED_INTRA_NxN)), tileAddr(reconTile(depth + 1, PRED_INTRA_NxN)) };
                const uint64_t tiles2[2] = { tileAddr(predTile(depth + 1, PRED_INTRA)), tileAddr(reconTile(depth + 1, PRED_INTRA)) };
                /* while the chain runs the CU's own 2Nx2N evaluation (started on its queue above) is collected: its bits are counted on the host */
                struct Between { Analyzer* a; int x, y, depth; bool done; int rc; } bt{ this, x, y, depth, false, 0 };
                static const bool overlap = !(getenv("X265AMD_INTRA_COLLECT_EARLY") && atoi(getenv("X265AMD_INTRA_COLLECT_EARLY")) == 0);
                const int qrc = xa_intra_quad8_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, x, y, qp, d.cur.ctx, d.cur.frac,
                                                  tileAddr(split.reconTile), tilesN, tiles2, r4, &intraWs,
                                                  deferred && overlap ? [](void* c) { Between* b = (Between*)c; b->rc = b->a->rdIntra(b->a->md[b->depth].pred[PRED_INTRA], b->x, b->y, b->depth, PRED_INTRA, true, 0); b->done = true; } : (void (*)(void*))nullptr,
                                                  &bt, lambdaQp != qp ? lambdaQp : 0);
                if (bt.done) { if (bt.rc) return err; deferredDone = true; }
                if (qrc < 0) return err = qrc;
                if (qrc == 0)
                {
                    XA_HOSTPROF("an.compressIntra chain results");
                    chained = true;
                    if (const char* lg = getenv("X265AMD_CHAIN_LOG"))
                    {
                        int lx = -1, ly = -1;
                        if (sscanf(lg, "%d,%d", &lx, &ly) == 2 && (lx & ~15) == x && (ly & ~15) == y)
                            for (int q = 0; q < 4; q++)
                                fprintf(stderr, "chain (%d,%d) cu %d: part %d dirs %d %d %d %d chroma %d cbf %d%d%d%d %d %d rd %llu other %llu bits %u mv %u dist %u+%u psy %u res %u frac %llu\n", x, y, q,
                                        r4[q].part_size, r4[q].luma_dir[0], r4[q].luma_dir[1], r4[q].luma_dir[2], r4[q].luma_dir[3], r4[q].chroma_dir, r4[q].cbf_y[0], r4[q].cbf_y[1],
                                        r4[q].cbf_y[2], r4[q].cbf_y[3], r4[q].cbf_u, r4[q].cbf_v, (unsigned long long)r4[q].rd_cost, (unsigned long long)r4[q].other_cost, r4[q].total_bits,
                                        r4[q].mv_bits, r4[q].luma_dist, r4[q].chroma_dist, r4[q].psy_energy, r4[q].res_energy, (unsigned long long)r4[q].frac_bits);
                    }
                    for (int q = 0; q < 4; q++)
                    {
                        const x265amd_intra_cu8_result& r = r4[q];
                        const int cx = x + (q & 1) * 8, cy = y + (q >> 1) * 8;
                        const bool nxn = r.part_size != 0;
                        const bool anyY = r.cbf_y[0] || r.cbf_y[1] || r.cbf_y[2] || r.cbf_y[3];
                        for (int k = 0; k < 4; k++)
                        {
                            x265amd_cu_unit u;
                            memset(&u, 0, sizeof(u));
                            u.depth = (uint8_t)(depth + 1); u.pred_mode = X265AMD_MODE_INTRA; u.part_size = r.part_size; u.tu_depth = nxn ? 1 : 0;
                            u.luma_dir = r.luma_dir[k]; u.chroma_dir = r.chroma_dir; u.qp = (int8_t)qp; u.ref_idx[0] = u.ref_idx[1] = -1;
                            if (nxn)
                            {
                                /* CUData::m_cbf after checkIntra: the unit's own flag one level down, the CU's flag (depth 0) on its first unit only */
                                u.cbf[0] = (uint8_t)((r.cbf_y[k] ? 2 : 0) | (k == 0 && anyY ? 1 : 0));
                                u.cbf[1] = (uint8_t)((r.cbf_u ? 2 : 0) | (k == 0 && r.cbf_u ? 1 : 0));
                                u.cbf[2] = (uint8_t)((r.cbf_v ? 2 : 0) | (k == 0 && r.cbf_v ? 1 : 0));
                            }
                            else { u.cbf[0] = r.cbf_y[0] ? 1 : 0; u.cbf[1] = r.cbf_u ? 1 : 0; u.cbf[2] = r.cbf_v ? 1 : 0; }
                            const int idx = ((q >> 1) * h4n + (k >> 1)) * n4 + (q & 1) * h4n + (k & 1);
                            split.u[idx] = u;
                            memset(&split.m[idx], 0, sizeof(x265amd_mv_unit));
                            split.m[idx].pred_mode = X265AMD_MODE_INTRA; split.m[idx].ref_idx[0] = split.m[idx].ref_idx[1] = -1;
                            /* CUData::copyToPic of the sub-CU */
                            units[((cy >> 2) + (k >> 1)) * w4 + (cx >> 2) + (k & 1)] = u;
                            cur[((cy >> 2) + (k >> 1)) * w4 + (cx >> 2) + (k & 1)] = split.m[idx];
                            if (dCur) devmap_store(dCur + ((cy >> 2) + (k >> 1)) * w4 + (cx >> 2) + (k & 1), split.m[idx], u.depth);
                        }
                        split.rdCost += r.rd_cost; split.psyEnergy += r.psy_energy; split.resEnergy += r.res_energy;
                        split.lumaDistortion += r.luma_dist; split.chromaDistortion += r.chroma_dist; split.distortion += r.luma_dist + r.chroma_dist;
                        split.totalBits += r.total_bits; split.mvBits += r.mv_bits; split.coeffBits += r.coeff_bits;
                        memcpy(&split.coeff[(size_t)q * 64], r.levels, sizeof(int16_t) * 64);
                        memcpy(&split.coeff[4096 + (size_t)q * 16], r.levels + 64, sizeof(int16_t) * 16);
                        memcpy(&split.coeff[5120 + (size_t)q * 16], r.levels + 80, sizeof(int16_t) * 16);
                    }
                    memcpy(split.contexts.ctx, r4[3].ctx, X265AMD_CTX_STRIDE);
                    split.contexts.frac = r4[3].frac_bits;
                    nextContext = &split.contexts;
                }
            }
            for (int q = 0; q < 4 && !chained; q++)
            {
                const int cx = x + (q & 1) * half, cy = y + (q >> 1) * half;
                if (cx < I->pic_width && cy < I->pic_height)
                {
                    md[depth + 1].cur = *nextContext;
                    if (childQp(depth, q)) return err;
                    if (compressIntra(cx, cy, depth + 1)) return err;
                    const Mode& nb = *md[depth + 1].best;
                    if (const char* lg = depth == 2 ? getenv("X265AMD_CHAIN_LOG") : nullptr)
                    {
                        int lx = -1, ly = -1;
                        if (sscanf(lg, "%d,%d", &lx, &ly) == 2 && (lx & ~15) == x && (ly & ~15) == y)
                        {
                            const Mode& a = md[depth + 1].pred[PRED_INTRA]; const Mode& b = md[depth + 1].pred[PRED_INTRA_NxN];
                            fprintf(stderr, "one by one (%d,%d) cu %d: part %d dirs %d %d %d %d chroma %d cbf %d %d %d %d %d %d rd %llu (2Nx2N %llu NxN %llu) bits %u mv %u dist %u+%u psy %u res %u frac %llu\n", x, y, q,
                                    nb.u[0].part_size, nb.u[0].luma_dir, nb.u[1].luma_dir, nb.u[2].luma_dir, nb.u[3].luma_dir, nb.u[0].chroma_dir, nb.u[0].cbf[0], nb.u[1].cbf[0], nb.u[2].cbf[0],
                                    nb.u[3].cbf[0], nb.u[0].cbf[1], nb.u[0].cbf[2], (unsigned long long)nb.rdCost, (unsigned long long)a.rdCost, (unsigned long long)b.rdCost, nb.totalBits, nb.mvBits,
                                    (unsigned)nb.lumaDistortion, (unsigned)nb.chromaDistortion, nb.psyEnergy, (unsigned)nb.resEnergy, (unsigned long long)nb.contexts.frac);
                        }
                    }
                    for (int yy = 0; yy < h4n; yy++)
                        for (int xx = 0; xx < h4n; xx++)
                        {
                            split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.u[yy * h4n + xx];
                            split.m[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.m[yy * h4n + xx];
                        }
                    split.addSubCosts(nb);
                    copyTile(split.reconTile, nb.reconTile, (q & 1) * half, (q >> 1) * half, half);
                    const int nc = half * half;
                    memcpy(&split.coeff[(size_t)q * nc], nb.coeff.data(), sizeof(int16_t) * nc);
                    memcpy(&split.coeff[4096 + (size_t)q * nc / 4], nb.coeff.data() + 4096, sizeof(int16_t) * nc / 4);
                    memcpy(&split.coeff[5120 + (size_t)q * nc / 4], nb.coeff.data() + 5120, sizeof(int16_t) * nc / 4);
                    nextContext = &nb.contexts;
                }
                else
                    for (int yy = 0; yy < h4n; yy++)
                        for (int xx = 0; xx < h4n; xx++) split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx].depth = (uint8_t)(depth + 1);
            }
            split.contexts = *nextContext;
            if (mightNotSplit) addSplitFlagCost(split, x, y, depth);
            else updateModeCost(split);
            XA_HOSTPROF("an.compressIntra collect + compare");
            if (deferred)
            {
                /* now the 2Nx2N result, then the comparisons in the reference's order */
                if (!deferredDone && rdIntra(d.pred[PRED_INTRA], x, y, depth, PRED_INTRA, true, 0)) return err;
                checkBestMode(d.pred[PRED_INTRA], depth);
                addSplitFlagCost(*d.best, x, y, depth);
            }
            if (checkDQPForSplitPred(split, x, y, depth)) return err;          /* analysis.cpp:642 */
            checkBestMode(split, depth);
        }
        XA_HOSTPROF("an.compressIntra tail (toPicture)");
        saveTUDepth(*d.best, x, y, depth);          /* analysis.cpp:653-659 (I slices leave the records the P pictures behind them load) */
        toPicture(*d.best, x, y, depth);
        if (d.best != &d.pred[PRED_SPLIT]) tileToPicture(d.best->reconTile, x, y, size);
        return 0;
    }

    /* the intra try of a CU of a P / B picture starts on a queue of its own (intra_rd.hip: xa_intra_in_inter_begin_ws): 1 started, 0 not this configuration, < 0 an error */
    int intraBegin(int x, int y, int depth)
    {
        const int log2 = 6 - depth;
        ModeDepth& d = md[depth];
        x265amd_rd_cu c;
        memset(&c, 0, sizeof(c));
        c.x = (int16_t)x; c.y = (int16_t)y; c.log2_size = (uint8_t)log2; cuQp2(c);
        memcpy(c.ctx, d.cur.ctx, X265AMD_CTX_COUNT);
        c.frac_bits = d.cur.frac;
        return xa_intra_in_inter_begin_ws(st, si, &rp, units, planes + 3 * (numPics - 1), planes + 3 * (numPics - 2), stride, cstride, &c,
                                          tileAddr(predTile(depth, PRED_INTRA)), tileAddr(reconTile(depth, PRED_INTRA)), &intraWs);
    }
    int compress(int x, int y, int depth, SplitData& splitOut, int node = 0)
    {
        ModeDepth& d = md[depth];
        const int log2 = 6 - depth, size = 1 << log2;
        d.best = nullptr;
        const bool mightSplit = depth < si->max_cu_depth;
        const bool mightNotSplit = x + size <= I->pic_width && y + size <= I->pic_height;
        const uint32_t minDepth = topSkipMinDepth(x, y, depth);
        const int myQp = qp;                    /* the `qp` argument of compressInterCU_rd0_4 */
        bool skipModes = false, skipRecursion = false, splitIntra = true;
        SplitData splitData[4];
        memset(splitData, 0, sizeof(splitData));
        d.mvCost2Nx2N[0] = d.mvCost2Nx2N[1] = 0;
        { XA_HOSTPROF("an.initSubCU x13"); for (int k = 0; k < NUM_PRED; k++) initSubCU(d.pred[k], depth); }
        d.pred[PRED_2Nx2N].sa8dCost = 0;                 /* what a parent reads under --limit-modes when 2Nx2N is not searched here */
        loadTUDepth(x, y, depth);           /* analysis.cpp:1201-1202 */
        chain.frNode[depth] = node; chain.frDirty[depth] = false;
        {
            const bool checked = mightNotSplit && (uint32_t)depth >= minDepth;
            chain.usedFlags[depth] = (uint8_t)(checked ? 1 : (mightNotSplit && mightSplit ? 2 : 0));
            /* the device walks the CTU's tree by the same rule (XaChainNode::flags): where it has been, the host must arrive the same way */
            if (chain.on && ((!checked && chain.status[node] != 0) || (checked && chain.status[node] == 0 && mightSplit && node + 1 < chain.numNodes && chain.status[node + 1] != 0)))
                return fail("skip chain: the device's walk through the CTU is not the host's");
        }
        bool devSkip = false, childrenDev = true, intraBegun = false;

        /* Step 1: merge / skip candidates */
        if (mightNotSplit && (uint32_t)depth >= minDepth)
        {
            if (chain.on && !mightSplit && chain.status[node] == 0 && si->slice_type == 1 && !(A->rect || A->amp))
            {
                /* a CU that cannot split at the head of a chain: its search and its intra try start beside the merge check (the conditions of step 3's checkInterFused) */
                if (searchAhead(x, y, depth, true)) return err;
                static const bool specIntraOn = !(getenv("X265AMD_INTRA_AHEAD_LEAF") && atoi(getenv("X265AMD_INTRA_AHEAD_LEAF")) == 0);
                if (specIntraOn && log2 != 6 && xa_is_queue(st))
                {
                    const int b = intraBegin(x, y, depth);
                    if (b < 0) return err = b;
                    intraBegun = b == 1;
                }
            }
            if (chain.on && chainSkip(node, x, y, depth, devSkip)) return err;
            if (g_timing) g_cuStat[si->slice_type == 1][depth][devSkip ? 0 : 1]++;
            static const bool verify2 = getenv("X265AMD_CHAIN_VERIFY") && atoi(getenv("X265AMD_CHAIN_VERIFY")) >= 2;
            if (devSkip && verify2)
            {
                /* debugging: the host's own merge check of the CU the device skipped must arrive at the same mode, cost and coder state */
                const uint64_t devCost = d.best->rdCost; const Snap devCtx = d.best->contexts; const int devCand = d.best->u[0].mvp_idx[0];
                d.best = nullptr;
                if (checkMerge(x, y, depth)) return err;
                if (!d.best || !d.best->isSkipped() || d.best->rdCost != devCost || d.best->u[0].mvp_idx[0] != devCand || d.best->contexts.frac != devCtx.frac ||
                    memcmp(d.best->contexts.ctx, devCtx.ctx, X265AMD_CTX_COUNT))
                {
                    fprintf(stderr, "x265amd chain verify: poc %d CU (%d,%d) size %d: device skip (candidate %d, cost %llu); host %s candidate %d cost %llu\n", I->poc, x, y, size, devCand,
                            (unsigned long long)devCost, !d.best ? "nothing" : (d.best->isSkipped() ? "skip" : "merge with residual"), d.best ? d.best->u[0].mvp_idx[0] : -1,
                            (unsigned long long)(d.best ? d.best->rdCost : 0));
                    return fail("chain verify: the device skipped a CU the host does not skip the same way");
                }
            }
            if (!devSkip)
            {
                chain.frDirty[depth] = true;            /* this CU is the host's: what the chain decides below it stays inside it */
                bool devMerge = false;
                if (chain.on && chainMerge(node, x, y, depth, devMerge)) return err;
                static const bool verify3 = getenv("X265AMD_CHAIN_VERIFY") && atoi(getenv("X265AMD_CHAIN_VERIFY")) >= 2;
                if (devMerge && verify3)
                {
                    /* debugging: the host's own merge check must leave the same two modes */
                    const uint64_t c0 = d.pred[PRED_SKIP].rdCost, c1 = d.pred[PRED_MERGE].rdCost; const Snap s1 = d.pred[PRED_MERGE].contexts;
                    const uint32_t b1 = d.pred[PRED_MERGE].totalBits, mv1 = d.pred[PRED_MERGE].mvBits; const uint8_t cb[3] = { d.pred[PRED_MERGE].u[0].cbf[0], d.pred[PRED_MERGE].u[0].cbf[1], d.pred[PRED_MERGE].u[0].cbf[2] };
                    const std::vector<int16_t> lv = d.pred[PRED_MERGE].coeff;
                    const bool mergeBest = d.best == &d.pred[PRED_MERGE];
                    d.best = nullptr;
                    if (checkMerge(x, y, depth)) return err;
                    const Mode& hm = d.pred[PRED_MERGE];
                    if (d.pred[PRED_SKIP].rdCost != c0 || hm.rdCost != c1 || hm.totalBits != b1 || hm.mvBits != mv1 || hm.u[0].cbf[0] != cb[0] || hm.u[0].cbf[1] != cb[1] || hm.u[0].cbf[2] != cb[2] ||
                        hm.contexts.frac != s1.frac || memcmp(hm.contexts.ctx, s1.ctx, X265AMD_CTX_COUNT) || hm.coeff != lv || (d.best == &d.pred[PRED_MERGE]) != mergeBest)
                    {
                        fprintf(stderr, "x265amd chain verify: poc %d CU (%d,%d) size %d: merge check differs: skip cost device %llu host %llu, residual mode cost %llu / %llu bits %u / %u mv bits %u / %u "
                                "cbf %d%d%d / %d%d%d fraction %llu / %llu levels %s best %d / %d\n", I->poc, x, y, size, (unsigned long long)c0, (unsigned long long)d.pred[PRED_SKIP].rdCost,
                                (unsigned long long)c1, (unsigned long long)hm.rdCost, b1, hm.totalBits, mv1, hm.mvBits, cb[0], cb[1], cb[2], hm.u[0].cbf[0], hm.u[0].cbf[1], hm.u[0].cbf[2],
                                (unsigned long long)s1.frac, (unsigned long long)hm.contexts.frac, hm.coeff == lv ? "same" : "differ", (int)mergeBest, (int)(d.best == &d.pred[PRED_MERGE]));
                        return fail("chain verify: the device's merge check is not the host's");
                    }
                }
                bool devFrom = false;
                if (!devMerge && chain.on && chainMergeFrom(node, x, y, depth, devFrom)) return err;
                if (devFrom && verify3)
                {
                    /* debugging: the host's own merge check, candidates and all, must leave the same two modes */
                    const uint64_t c0 = d.pred[PRED_SKIP].rdCost, c1 = d.pred[PRED_MERGE].rdCost, s0 = d.pred[PRED_MERGE].sa8dCost; const Snap s1 = d.best->contexts;
                    const int cand0 = d.pred[PRED_MERGE].u[0].mvp_idx[0]; const bool mergeBest = d.best == &d.pred[PRED_MERGE];
                    d.best = nullptr;
                    if (checkMerge(x, y, depth)) return err;
                    if (!d.best || d.pred[PRED_SKIP].rdCost != c0 || d.pred[PRED_MERGE].rdCost != c1 || d.pred[PRED_MERGE].sa8dCost != s0 || d.pred[PRED_MERGE].u[0].mvp_idx[0] != cand0 ||
                        (d.best == &d.pred[PRED_MERGE]) != mergeBest || d.best->contexts.frac != s1.frac || memcmp(d.best->contexts.ctx, s1.ctx, X265AMD_CTX_COUNT))
                    {
                        fprintf(stderr, "x265amd chain verify: poc %d CU (%d,%d) size %d: merge check from the device's candidate %d differs: skip cost %llu / %llu, residual mode %llu / %llu, sa8d cost %llu / %llu, "
                                "candidate %d, best %d / %d\n", I->poc, x, y, size, cand0, (unsigned long long)c0, (unsigned long long)d.pred[PRED_SKIP].rdCost, (unsigned long long)c1,
                                (unsigned long long)d.pred[PRED_MERGE].rdCost, (unsigned long long)s0, (unsigned long long)d.pred[PRED_MERGE].sa8dCost, d.pred[PRED_MERGE].u[0].mvp_idx[0], (int)mergeBest,
                                (int)(d.best == &d.pred[PRED_MERGE]));
                        return fail("chain verify: the merge check from the device's candidate is not the host's");
                    }
                }
                if (!devMerge && !devFrom && checkMerge(x, y, depth)) return err;
            }
            skipModes = A->early_skip && d.best && d.best->isSkipped();
        }
        if (d.best && A->rskip)
        {
            skipRecursion = d.best->isSkipped();
            if (mightSplit && !skipRecursion && (uint32_t)depth >= minDepth && A->rskip == 1)
            {
                if (depth) skipRecursion = recursionDepthCheck(depth, *d.best);
                /* complexityCheckCU on HD pictures (m_bHD: sourceHeight >= 1080, analysis.cpp:116) at rd 2 (analysis.cpp:1326, :3538-3559) */
                if (I->pic_height >= 1080 && !skipRecursion && A->rd_level == 2 && size != 64)
                    skipRecursion = (double)d.srcHomo < (.1 * d.srcMean);
            }
        }
        /* the intra try of step 3 depends on nothing that happens until then: when the CU is not skipped and a queue is to spare it starts now, beside the sub-CUs
         * and the motion searches (intra_rd.hip; a try the analysis does not get to is dropped) */
        if (!intraBegun && mightNotSplit && (uint32_t)depth >= minDepth && !skipModes && (!I->is_inter_b || A->b_intra) && log2 != 6 && xa_is_queue(st))
        {
            const int b = intraBegin(x, y, depth);
            if (b < 0) return err = b;
        }
        /* the search of step 3 does not depend on the sub-CUs either, but for the reference pictures they restrict it to: it starts now on a queue of its own and is
         * collected behind them (searchAhead; valid when its winner is an allowed picture) */
        if (chain.on && mightSplit && !skipRecursion && mightNotSplit && (uint32_t)depth >= minDepth && !skipModes && si->slice_type == 1 && !(A->rect || A->amp) &&
            searchAhead(x, y, depth, false))
            return err;
        /* Step 2: the four sub-blocks in series */
        if (mightSplit && !skipRecursion)
        {
            Mode& split = d.pred[PRED_SPLIT];
            split.initCosts();
            split.predTile = predTile(depth, PRED_SPLIT); split.reconTile = reconTile(depth, PRED_SPLIT);
            const int n4 = 16 >> depth, half = size >> 1, h4n = n4 >> 1;
            const Snap* nextContext = &d.cur;
            splitIntra = false;
            int childNode = node + 1;                   /* pre-order: the first sub-CU follows its parent, the others follow their elder's subtree */
            for (int q = 0; q < 4; q++)
            {
                const int cx = x + (q & 1) * half, cy = y + (q >> 1) * half;
                if (cx < I->pic_width && cy < I->pic_height)
                {
                    md[depth + 1].cur = *nextContext;
                    if (childQp(depth, q)) return err;
                    if (compress(cx, cy, depth + 1, splitData[q], childNode)) return err;
                    childNode = chain.nodes[childNode].next;
                    const bool childDev = chain.lastDevComplete;
                    if (!childDev) { childrenDev = false; chain.frDirty[depth] = true; }
                    const Mode& nb = *md[depth + 1].best;
                    splitIntra |= nb.u[0].pred_mode == X265AMD_MODE_INTRA;
                    for (int yy = 0; yy < h4n; yy++)
                        for (int xx = 0; xx < h4n; xx++)
                        {
                            split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.u[yy * h4n + xx];
                            split.m[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx] = nb.m[yy * h4n + xx];
                        }
                    split.addSubCosts(nb);
                    if (!childDev) copyTile(split.reconTile, nb.reconTile, (q & 1) * half, (q >> 1) * half, half);     /* the device put its CUs' samples into every enclosing tile itself */
                    const int nc = half * half;
                    memcpy(&split.coeff[(size_t)q * nc], nb.coeff.data(), sizeof(int16_t) * nc);
                    memcpy(&split.coeff[4096 + (size_t)q * nc / 4], nb.coeff.data() + 4096, sizeof(int16_t) * nc / 4);
                    memcpy(&split.coeff[5120 + (size_t)q * nc / 4], nb.coeff.data() + 5120, sizeof(int16_t) * nc / 4);
                    nextContext = &nb.contexts;
                }
                else
                    for (int yy = 0; yy < h4n; yy++)            /* setEmptyPart */
                        for (int xx = 0; xx < h4n; xx++) split.u[((q >> 1) * h4n + yy) * n4 + (q & 1) * h4n + xx].depth = (uint8_t)(depth + 1);
            }
            split.contexts = *nextContext;
            if (mightNotSplit) addSplitFlagCost(split, x, y, depth);
            else updateModeCost(split);
        }
        uint32_t allSplitRefs = splitData[0].splitRefs | splitData[1].splitRefs | splitData[2].splitRefs | splitData[3].splitRefs;
        /* Step 3: ME and RD at the current depth */
        if (mightNotSplit && (uint32_t)depth >= minDepth)
        {
            /* (the sub-CUs left their own QP in force: analysis.cpp:1413-1414) */
            if (si->use_dqp && depth <= si->max_cu_dqp_depth && si->max_cu_dqp_depth != 0 && setLambdaFromQP(myQp)) return err;
            if (!skipModes)
            {
                if (g_timing) g_cuStat[si->slice_type == 1][depth][2]++;
                bool fused = false;
                fusedRd[depth] = false;
                if (!(A->rect || A->amp) && checkInterFused(x, y, depth, allSplitRefs, fused)) return err;
                if (fused)
                {
                    static const bool verifyS = getenv("X265AMD_CHAIN_VERIFY") && atoi(getenv("X265AMD_CHAIN_VERIFY")) >= 2;
                    if (verifyS)
                    {
                        /* debugging: the ordinary search and rate-distortion of the same CU must give the same mode */
                        const Mode fm = d.pred[PRED_2Nx2N];
                        if (checkInter(x, y, depth, allSplitRefs)) return err;
                        Mode& hm = d.pred[PRED_2Nx2N];
                        const uint64_t hs = hm.sa8dCost; const uint32_t hb = hm.sa8dBits;
                        if (rdInter(hm, x, y, depth, false)) return err;
                        if (memcmp(&fm.u[0], &hm.u[0], sizeof(x265amd_cu_unit)) || memcmp(&fm.m[0], &hm.m[0], sizeof(x265amd_mv_unit)) || fm.sa8dCost != hs || fm.sa8dBits != hb ||
                            (fusedRd[depth] && (fm.rdCost != hm.rdCost || fm.totalBits != hm.totalBits || fm.mvBits != hm.mvBits || fm.coeff != hm.coeff || fm.contexts.frac != hm.contexts.frac ||
                                                memcmp(fm.contexts.ctx, hm.contexts.ctx, X265AMD_CTX_COUNT) || fm.psyEnergy != hm.psyEnergy || fm.distortion != hm.distortion)))
                        {
                            fprintf(stderr, "x265amd search verify: poc %d CU (%d,%d) size %d: device ref %d mv (%d,%d) mvd (%d,%d) mvp %d sa8d cost %llu bits %u rd %llu bits %u/%u cbf %d%d%d dist %llu psy %u; "
                                    "host ref %d mv (%d,%d) mvd (%d,%d) mvp %d sa8d cost %llu bits %u rd %llu bits %u/%u cbf %d%d%d dist %llu psy %u levels %s contexts %s\n", I->poc, x, y, size,
                                    fm.u[0].ref_idx[0], fm.m[0].mv[0][0], fm.m[0].mv[0][1], fm.u[0].mvd[0][0], fm.u[0].mvd[0][1], fm.u[0].mvp_idx[0], (unsigned long long)fm.sa8dCost, fm.sa8dBits,
                                    (unsigned long long)fm.rdCost, fm.totalBits, fm.mvBits, fm.u[0].cbf[0], fm.u[0].cbf[1], fm.u[0].cbf[2], (unsigned long long)fm.distortion, fm.psyEnergy,
                                    hm.u[0].ref_idx[0], hm.m[0].mv[0][0], hm.m[0].mv[0][1], hm.u[0].mvd[0][0], hm.u[0].mvd[0][1], hm.u[0].mvp_idx[0], (unsigned long long)hs, hb,
                                    (unsigned long long)hm.rdCost, hm.totalBits, hm.mvBits, hm.u[0].cbf[0], hm.u[0].cbf[1], hm.u[0].cbf[2], (unsigned long long)hm.distortion, hm.psyEnergy,
                                    fm.coeff == hm.coeff ? "same" : "differ", memcmp(fm.contexts.ctx, hm.contexts.ctx, X265AMD_CTX_COUNT) ? "differ" : "same");
                            return fail("search verify: the fused search is not the host's");
                        }
                        hm.sa8dCost = hs; hm.sa8dBits = hb;
                        fusedRd[depth] = true;
                    }
                }
                else if (checkInter(x, y, depth, allSplitRefs)) return err;
                Mode* bestInter = &d.pred[PRED_2Nx2N];
                if (A->limit_refs & 2)                                                     /* X265_REF_LIMIT_CU */
                {
                    allSplitRefs = bestRefIdx(bestInter->u[0]);
                    for (int q = 0; q < 4; q++) splitData[q].splitRefs = allSplitRefs;
                }
                Mode& bidir = d.pred[PRED_BIDIR];
                /* rectangular and asymmetric partitions (analysis.cpp:1447-1594); with limit-modes only where the sub-CUs' motion suggests it */
                {
                    const Mode& p2N = d.pred[PRED_2Nx2N];
                    const bool isP = !I->is_inter_b;
                    auto thr = [&](int a, int b) { return isP ? splitData[a].mvCost[0] + splitData[b].mvCost[0]
                                                              : (splitData[a].mvCost[0] + splitData[b].mvCost[0] + splitData[a].mvCost[1] + splitData[b].mvCost[1] + 1) >> 1; };
                    const uint64_t splitCost = splitData[0].sa8dCost + splitData[1].sa8dCost + splitData[2].sa8dCost + splitData[3].sa8dCost;
                    auto tryPart = [&](int part, int slot, uint32_t m0, uint32_t m1) -> int {
                        const uint32_t masks[2] = { m0, m1 };
                        if (checkInterPart(x, y, depth, part, slot, masks)) return err;
                        if (d.pred[slot].sa8dCost < bestInter->sa8dCost) bestInter = &d.pred[slot];
                        return 0;
                    };
                    const uint32_t top = splitData[0].splitRefs | splitData[1].splitRefs, bot = splitData[2].splitRefs | splitData[3].splitRefs;
                    const uint32_t lft = splitData[0].splitRefs | splitData[2].splitRefs, rgt = splitData[1].splitRefs | splitData[3].splitRefs;
                    if (A->rect)
                    {
                        const uint32_t t2NxN = thr(0, 1), tNx2N = thr(0, 2);
                        const bool first2NxN = t2NxN < tNx2N;
                        if (first2NxN && splitCost < p2N.sa8dCost + t2NxN) { if (tryPart(1, PRED_2NxN, top, bot)) return err; }
                        if (splitCost < p2N.sa8dCost + tNx2N) { if (tryPart(2, PRED_Nx2N, lft, rgt)) return err; }
                        if (!first2NxN && splitCost < p2N.sa8dCost + t2NxN) { if (tryPart(1, PRED_2NxN, top, bot)) return err; }
                    }
                    if (A->amp && si->max_amp_depth > depth)
                    {
                        const uint32_t tU = thr(0, 1), tD = thr(2, 3), tL = thr(0, 2), tR = thr(1, 3);
                        bool bHor = false, bVer = false;
                        const int bp = bestInter->u[0].part_size;
                        if (bp == 1) bHor = true;
                        else if (bp == 2) bVer = true;
                        else if (bp == 0 && d.best && (d.best->u[0].cbf[0] || d.best->u[0].cbf[1] || d.best->u[0].cbf[2])) { bHor = true; bVer = true; }
                        if (bHor)
                        {
                            const bool firstD = tD < tU;
                            if (firstD && splitCost < p2N.sa8dCost + tD) { if (tryPart(5, PRED_2NxnD, allSplitRefs, bot)) return err; }
                            if (splitCost < p2N.sa8dCost + tU) { if (tryPart(4, PRED_2NxnU, top, allSplitRefs)) return err; }
                            if (!firstD && splitCost < p2N.sa8dCost + tD) { if (tryPart(5, PRED_2NxnD, allSplitRefs, bot)) return err; }
                        }
                        if (bVer)
                        {
                            const bool firstR = tR < tL;
                            if (firstR && splitCost < p2N.sa8dCost + tR) { if (tryPart(7, PRED_nRx2N, allSplitRefs, rgt)) return err; }
                            if (splitCost < p2N.sa8dCost + tL) { if (tryPart(6, PRED_nLx2N, lft, allSplitRefs)) return err; }
                            if (!firstR && splitCost < p2N.sa8dCost + tR) { if (tryPart(7, PRED_nRx2N, allSplitRefs, rgt)) return err; }
                        }
                    }
                }
                const bool bTryIntra = (!I->is_inter_b || A->b_intra) && log2 != 6;
                if (A->rd_level >= 3)
                {
                    if (!(bestInter == &d.pred[PRED_2Nx2N] && fusedRd[depth]) && rdInter(*bestInter, x, y, depth, false)) return err;
                    checkBestMode(*bestInter, depth);
                    if (I->is_inter_b && bidir.sa8dCost != kMaxCost && bidir.sa8dCost * 16 <= bestInter->sa8dCost * 17)
                    {
                        if (rdInter(bidir, x, y, depth, false)) return err;
                        checkBestMode(bidir, depth);
                    }
                    const x265amd_cu_unit& b0 = d.best->u[0];
                    if (bTryIntra && (b0.cbf[0] || b0.cbf[1] || b0.cbf[2]) && (!A->limit_refs || splitIntra))
                    {
                        if (g_timing) g_cuStat[si->slice_type == 1][depth][3]++;
                        if (rdIntra(d.pred[PRED_INTRA], x, y, depth)) return err;
                        checkBestMode(d.pred[PRED_INTRA], depth);
                    }
                }
                else
                {
                    /* rd 2 (analysis.cpp:1655-1703): SA8D choice between merge / skip, inter, bidir and intra; only the winner is coded */
                    if (!d.best || bestInter->sa8dCost < d.best->sa8dCost) d.best = bestInter;
                    if (I->is_inter_b && bidir.sa8dCost < d.best->sa8dCost) d.best = &bidir;
                    bool intraCoded = false;
                    if ((bTryIntra || d.best->sa8dCost == kMaxCost) && (!A->limit_refs || splitIntra))
                    {
                        /* checkIntraInInter ranks by SA8D; encodeIntraInInter's result is kept only if intra wins (it has no side effects) */
                        if (rdIntra(d.pred[PRED_INTRA], x, y, depth)) return err;
                        if (d.pred[PRED_INTRA].sa8dCost < d.best->sa8dCost) { d.best = &d.pred[PRED_INTRA]; intraCoded = true; }
                    }
                    const x265amd_cu_unit& b0 = d.best->u[0];
                    const bool codedMerge = (d.best == &d.pred[PRED_MERGE] || d.best == &d.pred[PRED_SKIP]) && b0.part_size == 0;
                    if (!codedMerge && !intraCoded)
                    {
                        const uint64_t sa8dCost = d.best->sa8dCost; const uint32_t sa8dBits = d.best->sa8dBits;
                        if (mcMode(*d.best, x, y, depth)) return err;
                        if (rdInter(*d.best, x, y, depth, false)) return err;
                        d.best->sa8dCost = sa8dCost; d.best->sa8dBits = sa8dBits;
                    }
                }
            }
            if (mightSplit) addSplitFlagCost(*d.best, x, y, depth);
        }
        if (mightSplit && !skipRecursion)
        {
            Mode& split = d.pred[PRED_SPLIT];
            if (!d.best) d.best = &split;
            else checkBestMode(split, depth);
            if (checkDQPForSplitPred(*d.best, x, y, depth)) return err;       /* (on the winner, whatever it is: analysis.cpp:1755) */
        }
        /* which motion references the parent CU should search (X265_REF_LIMIT_DEPTH) */
        memset(&splitOut, 0, sizeof(splitOut));
        if (A->limit_refs & 1)
            splitOut.splitRefs = d.best == &d.pred[PRED_SPLIT] ? allSplitRefs
                                                                : bestRefIdxCu(d.best->u[0].pred_mode == X265AMD_MODE_INTRA ? d.pred[PRED_2Nx2N] : *d.best, depth);
        if (A->limit_modes)
        {
            splitOut.mvCost[0] = d.mvCost2Nx2N[0]; splitOut.mvCost[1] = d.mvCost2Nx2N[1];
            splitOut.sa8dCost = d.pred[PRED_2Nx2N].sa8dCost;
        }
        if (mightNotSplit && d.best->isSkipped())
        {
            x265amd_cu_stat& cs = cuStat[ctuAddr];
            const uint64_t temp = cs.avg_cost[depth] * cs.count[depth];
            cs.count[depth] += 1;
            cs.avg_cost[depth] = (temp + d.best->rdCost) / cs.count[depth];
        }
        /* everything in this CU's area decided (and put in place) by the device: the CU itself skipped there, or a CU that is not coded at this depth whose sub-CUs all are */
        const bool devComplete = devSkip || (chain.on && !(mightNotSplit && (uint32_t)depth >= minDepth) && d.best == &d.pred[PRED_SPLIT] && childrenDev);
        toPicture(*d.best, x, y, depth, !devComplete);
        if (mightNotSplit) saveTUDepth(*d.best, x, y, depth);            /* analysis.cpp:1796-1805 */
        if (!devComplete) tileToPicture(d.best->reconTile, x, y, size);
        chain.lastDevComplete = devComplete;
        return 0;
    }
};

} // namespace

static int compress_ctu_impl(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                             const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                             const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                             intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int ctu_addr, const uint8_t* ctx_in, uint64_t frac_in,
                             int16_t* coeff_out, x265amd_ctu_result* out, XaMapUnit* dCur, const XaMapUnit* dCol, const int8_t* cu_qp = nullptr, const XaTuRecs* tu_recs = nullptr);
extern "C" int x265amd_compress_ctu_inter(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                                          const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                                          const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                                          intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int ctu_addr, const uint8_t* ctx_in, uint64_t frac_in,
                                          int16_t* coeff_out, x265amd_ctu_result* out)
{
    return compress_ctu_impl(me, stream, I, S, si, A, units, cur, col, ref_depth, ref_qp0, h_planes, num_pics, stride, cstride, cu_stat, ctu_addr, ctx_in, frac_in, coeff_out, out,
                             (XaMapUnit*)xa_devmap_find(cur), (const XaMapUnit*)xa_devmap_find(col));
}
static int compress_ctu_impl(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                             const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                             const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                             intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int ctu_addr, const uint8_t* ctx_in, uint64_t frac_in,
                             int16_t* coeff_out, x265amd_ctu_result* out, XaMapUnit* dCur, const XaMapUnit* dCol, const int8_t* cu_qp, const XaTuRecs* tu_recs)
{
    if ((!me && si && si->slice_type != 2) || !I || !S || !si || !A || !units || !cur || !ref_depth || !ref_qp0 || !h_planes || !cu_stat || !ctx_in || !out || num_pics < 2)
        return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: null argument");
    if (si->slice_type != 2 && (si->slice_type == 0) != (I->is_inter_b != 0)) return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: slice type");
    if (A->limit_tu < 0 || A->limit_tu > 4 || A->limit_tu == 1) return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: limit_tu 1 (the breadth-first form) is not built");
    if (A->limit_tu >= 3 && (!tu_recs || !tu_recs->cur || (si->slice_type != 2 && !tu_recs->ref[0]) || (si->slice_type == 0 && !tu_recs->ref[1])))
        return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: limit_tu 3 / 4 needs the pictures' transform depth records (the encoder object keeps them)");
    if (A->rdoq_level < 0 || A->rdoq_level > 2 || A->rd_level < 2 || A->rd_level > 6 || (A->rd_level > 4 && A->rskip == 2) || A->limit_refs < 0 || A->limit_refs > 3 || (si->use_dqp && (!cu_qp || si->max_cu_dqp_depth < 0 || si->max_cu_dqp_depth > 1)) || si->tq_bypass_enabled || (A->rskip != 0 && A->rskip != 1))
        return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: configuration outside the built subset (rd 2-6, delta QP with the quantisation groups' QPs handed in and groups of 64 or 32 samples, rskip 0/1)");
    if ((I->pic_width & 7) || (I->pic_height & 7) || I->pic_width != si->pic_width || I->pic_height != si->pic_height) return xa_fail(X265AMD_EINVAL, "compress_ctu_inter: picture size");
    /* debugging aid: X265AMD_DUMP_CTU=<dir> X265AMD_DUMP_POC=<poc> X265AMD_DUMP_MARGIN=<mx>,<my> writes every input of this call (before) and its outputs
     * (after) to <dir>/ctu_<addr>.bin so that the reference's compressCTU can be run on exactly the same state (dbg/ctu_replay.py) */
    FILE* dump = nullptr;
    int dumpMx = 0, dumpMy = 0;
    if (const char* dir = getenv("X265AMD_DUMP_CTU"))
    {
        const char* poc = getenv("X265AMD_DUMP_POC"); const char* mg = getenv("X265AMD_DUMP_MARGIN");
        if (poc && mg && atoi(poc) == I->poc && sscanf(mg, "%d,%d", &dumpMx, &dumpMy) == 2)
        {
            char path[512];
            snprintf(path, sizeof(path), "%s/ctu_%d.bin", dir, ctu_addr);
            dump = fopen(path, "wb");
        }
    }
    if (dump)
    {
        const int w4d = si->pic_width >> 2, h4d = si->pic_height >> 2, nctu = ((si->pic_width + 63) >> 6) * ((si->pic_height + 63) >> 6);
        const int32_t hdr[12] = { si->pic_width, si->pic_height, num_pics, (int32_t)stride, (int32_t)cstride, dumpMx, dumpMy, ctu_addr, (int32_t)sizeof(pixel), nctu, 0, 0 };
        fwrite(hdr, sizeof(hdr), 1, dump);
        fwrite(I, sizeof(*I), 1, dump); fwrite(S, sizeof(*S), 1, dump); fwrite(si, sizeof(*si), 1, dump); fwrite(A, sizeof(*A), 1, dump);
        fwrite(units, sizeof(x265amd_cu_unit), (size_t)w4d * h4d, dump); fwrite(cur, sizeof(x265amd_mv_unit), (size_t)w4d * h4d, dump);
        if (col) fwrite(col, sizeof(x265amd_mv_unit), (size_t)w4d * h4d, dump); else { std::vector<x265amd_mv_unit> z((size_t)w4d * h4d); memset(z.data(), 0, z.size() * sizeof(x265amd_mv_unit)); fwrite(z.data(), sizeof(x265amd_mv_unit), z.size(), dump); }
        fwrite(ref_depth, 1, (size_t)2 * w4d * h4d, dump); fwrite(ref_qp0, 1, (size_t)2 * nctu, dump);
        fwrite(cu_stat, sizeof(x265amd_cu_stat), (size_t)nctu + 1, dump);
        fwrite(ctx_in, 1, X265AMD_CTX_STRIDE, dump); fwrite(&frac_in, 8, 1, dump);
        (void)hipDeviceSynchronize();
        for (int k = 0; k < num_pics * 3; k++)
        {
            const bool luma = k % 3 == 0;
            const int mx = luma ? dumpMx : dumpMx / 2, my = luma ? dumpMy : dumpMy / 2, ph = luma ? si->pic_height : si->pic_height / 2;
            const intptr_t st = luma ? stride : cstride;
            std::vector<pixel> buf((size_t)(ph + 2 * my) * st);
            (void)hipMemcpy(buf.data(), (const pixel*)(uintptr_t)h_planes[k] - (intptr_t)my * st - mx, buf.size() * sizeof(pixel), hipMemcpyDeviceToHost);
            fwrite(buf.data(), sizeof(pixel), buf.size(), dump);
        }
    }
    Analyzer* an;
    { XA_HOSTPROF("ctu.new Analyzer"); an = new Analyzer; }
    Analyzer& a = *an;
    XA_HOSTPROF("ctu.all but new / delete");
    a.me = me; a.st = (hipStream_t)stream; a.I = I; a.S = S; a.si = si; a.A = A; a.units = units; a.cur = cur; a.col = col;
    a.tuRecs = tu_recs;
    if (tu_recs && tu_recs->cur) memset(tu_recs->cur + (size_t)ctu_addr * 21, -1, 21);          /* CUData::initCTU (cudata.cpp:312-313) */
    a.refDepth = ref_depth; a.refQp0 = ref_qp0; a.planes = h_planes; a.numPics = num_pics; a.stride = stride; a.cstride = cstride;
    a.cuStat = cu_stat; a.ctuAddr = ctu_addr; a.ctuW = (I->pic_width + 63) >> 6; a.w4 = I->pic_width >> 2; a.h4 = I->pic_height >> 2;
    a.ctuX = (ctu_addr % a.ctuW) * 64; a.ctuY = (ctu_addr / a.ctuW) * 64; a.qp = si->slice_qp; a.err = 0;
    if (si->use_dqp) { a.cuQp = cu_qp + (size_t)ctu_addr * (si->max_cu_dqp_depth >= 1 ? 5 : 1); a.qp = a.cuQp[0]; }           /* compressCTU: calculateQpforCuSize(ctu, cuGeom) (analysis.cpp:149) */
    a.ctuQp = a.qp;
    int rc = X265AMD_OK;
    if (a.ctuX >= I->pic_width || a.ctuY >= I->pic_height || ctu_addr < 0) rc = xa_fail(X265AMD_EINVAL, "compress_ctu_inter: CTU address");
    a.tileBytes = (size_t)kTileElems * sizeof(pixel);
    if (rc == X265AMD_OK && (a.dTiles.alloc(a.tileBytes * 4 * (2 * NUM_PRED + 6)) != hipSuccess || a.dPlanes.alloc((size_t)num_pics * 24) != hipSuccess ||
                             a.dJobs.alloc(sizeof(x265amd_mc_job) * 8) != hipSuccess))
        rc = xa_fail(X265AMD_EHIP, "compress_ctu_inter: out of device memory");
    /* a device job queue is a resident workgroup: it has to be told to look at what other rows, pictures and copies wrote (and, below, to publish) */
    if (rc == X265AMD_OK && xa_stream_fence(a.st, XA_CMD_ACQUIRE) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "compress_ctu_inter: queue");
    if (rc == X265AMD_OK && xa_copy_async(a.st, a.dPlanes.p, h_planes, (size_t)num_pics * 24, hipMemcpyHostToDevice) != hipSuccess)
        rc = xa_fail(X265AMD_EHIP, "compress_ctu_inter: plane table upload");
    if (rc == X265AMD_OK)
    {
        if (a.setLambdaFromQP(a.qp)) rc = a.err;
        a.rp.limit_tu = A->limit_tu;
        a.rp.psy_rd = A->psy_rd; a.rp.rd_level = A->rd_level; a.rp.strong_intra_smoothing = A->strong_intra_smoothing;
        a.rp.rdoq_level = A->rdoq_level; a.rp.psy_rdoq_scale = A->rdoq_level ? A->psy_rdoq_scale : 0; a.rp.fast_intra = A->fast_intra != 0;
        /* CUData::initCTU: nothing of this CTU is decided yet */
        for (int yy = a.ctuY >> 2; yy < (a.ctuY >> 2) + 16 && yy < a.h4; yy++)
            for (int xx = a.ctuX >> 2; xx < (a.ctuX >> 2) + 16 && xx < a.w4; xx++)
            {
                memset(&units[yy * a.w4 + xx], 0, sizeof(x265amd_cu_unit));
                units[yy * a.w4 + xx].qp = (int8_t)a.qp;
                memset(&cur[yy * a.w4 + xx], 0, sizeof(x265amd_mv_unit));
                cur[yy * a.w4 + xx].ref_idx[0] = cur[yy * a.w4 + xx].ref_idx[1] = -1;
            }
        memset(&a.md[0].cur, 0, sizeof(Snap));
        memcpy(a.md[0].cur.ctx, ctx_in, X265AMD_CTX_COUNT);
        a.md[0].cur.frac = frac_in;
        SplitData topSplit;
        a.dCur = dCur; a.dCol = dCol;
        {
            /* the skip chain (inter_chain_dev.h): what it assumes of the configuration -- a skipped CU ends there (early skip + recursion skip), one transform size per
             * plane, plain quantisation -- and a device job queue to run on */
            static const bool chainEnv = !(getenv("X265AMD_INTER_CHAIN") && atoi(getenv("X265AMD_INTER_CHAIN")) == 0);
            /* (a slice with weights: its predictions are weighted and its searches read weighted copies -- the host path does both; the chain does not) */
            /* under delta QP: cu_qp_delta is priced as rd 3-4 price it, and the device carries one QP for the quantiser and the lambdas (none of the CTU's above 51) */
            bool dqpOk = true;
            if (si->use_dqp)
            {
                static const bool chainDqp = !(getenv("X265AMD_CHAIN_DQP") && atoi(getenv("X265AMD_CHAIN_DQP")) == 0);
                dqpOk = chainDqp && A->rd_level >= 3;
                for (int k = 0; k < (si->max_cu_dqp_depth ? 5 : 1); k++) dqpOk = dqpOk && a.cuQp[k] <= 51;
            }
            a.chain.on = chainEnv && !S->weighted && si->slice_type != 2 && A->rd_level <= 4 && A->early_skip && A->rskip == 1 && !A->rdoq_level && si->tu_max_depth_inter == 1 && dqpOk &&
                         xa_is_queue(a.st) && dCur && (dCol || !I->temporal_mvp) && I->max_num_merge_cand >= 1 && I->max_num_merge_cand <= 5 && !dump;
            if (a.chain.on) a.buildNodes(a.ctuX, a.ctuY, 0, -1);
            else a.chain.nodes[0].next = 0;
        }
        if (rc == X265AMD_OK) rc = si->slice_type == 2 ? a.compressIntra(a.ctuX, a.ctuY, 0) : (A->rd_level > 4 ? a.compress56(a.ctuX, a.ctuY, 0, topSplit) : a.compress(a.ctuX, a.ctuY, 0, topSplit, 0));
        XA_HOSTPROF("ctu.final fence + sync");
        if (rc == X265AMD_OK && (xa_stream_fence(a.st, XA_CMD_RELEASE) != hipSuccess || xa_stream_sync(a.st) != hipSuccess)) rc = xa_fail(X265AMD_EHIP, "compress_ctu_inter: synchronize");
    }
    if (rc == X265AMD_OK)
    {
        const Mode& b = *a.md[0].best;
        memset(out, 0, sizeof(*out));
        out->rd_cost = b.rdCost; out->distortion = b.distortion; out->total_bits = b.totalBits; out->frac_bits = b.contexts.frac;
        memcpy(out->ctx, b.contexts.ctx, X265AMD_CTX_COUNT);
        if (coeff_out) memcpy(coeff_out, b.coeff.data(), sizeof(int16_t) * kTileElems);
    }
    if (dump)
    {
        const int w4d = si->pic_width >> 2, h4d = si->pic_height >> 2;
        fwrite(out, sizeof(*out), 1, dump);
        fwrite(units, sizeof(x265amd_cu_unit), (size_t)w4d * h4d, dump); fwrite(cur, sizeof(x265amd_mv_unit), (size_t)w4d * h4d, dump);
        if (rc == X265AMD_OK) fwrite(a.md[0].best->coeff.data(), sizeof(int16_t), kTileElems, dump);
        fclose(dump);
    }
    { XA_HOSTPROF("ctu.delete Analyzer"); delete an; }
    return rc;
}

/* The CTU loop of one frame: FrameEncoder::processRowEncoder (reference: source/encoder/frameencoder.cpp:1399-1700) without rate control,
 * VBV, slices or filters: every CTU in raster order (which respects the wavefront dependencies), its start state taken from the row coder
 * (one per row under WPP: row r > 0 starts from the state saved after the second CTU of row r - 1, frameencoder.cpp:1568-1573 / :1596-1598;
 * otherwise one coder runs through all rows), compressCTU, then the row coder codes the CTU in bit-counting mode to advance its state.
 * Without WPP the slice data can be written afterwards (FrameEncoder::encodeSlice, :1298-1370).
 * The reference's row coder only counts bits when SAO is on; without SAO it writes the final bitstream itself (frameencoder.cpp:683-699)
 * and its m_fracBits stays 0, so every CTU then starts from a zero bit fraction (ap->use_sao). */
extern "C" int x265amd_analyse_frame(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                                     const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                                     const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                                     intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int16_t* coeff_out, x265amd_ctu_result* results,
                                     uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams)
{
    return xa_analyse_frame(me, stream, I, S, si, A, units, cur, col, ref_depth, ref_qp0, h_planes, num_pics, stride, cstride, cu_stat, coeff_out, results, slice_data, cap,
                            substream_sizes, num_substreams, nullptr);
}

/* hooks (pictures coded in parallel, FrameEncoder::compressFrame's row loop, frameencoder.cpp:880-960): before_row blocks until the reference pictures have
 * finished the rows this CTU row may read; after_row hands the analysed row to the in-loop filters */
int xa_analyse_frame(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                     const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                     const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                     intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int16_t* coeff_out, x265amd_ctu_result* results,
                     uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams, const XaRowHooks* hooks, const int8_t* cu_qp, const XaTuRecs* tu_recs)
{
    if (!I || !si || !units || !cur || !cu_stat || !coeff_out) return xa_fail(X265AMD_EINVAL, "analyse_frame: null argument");
    const int ctuW = (si->pic_width + 63) >> 6, ctuH = (si->pic_height + 63) >> 6, numCtu = ctuW * ctuH, w4 = si->pic_width >> 2, h4 = si->pic_height >> 2;
    for (int i = 0; i < w4 * h4; i++)
    {
        memset(&units[i], 0, sizeof(x265amd_cu_unit));
        units[i].qp = (int8_t)si->slice_qp;
        memset(&cur[i], 0, sizeof(x265amd_mv_unit));
        cur[i].ref_idx[0] = cur[i].ref_idx[1] = -1;
    }
    memset(cu_stat, 0, sizeof(x265amd_cu_stat) * (numCtu + 1));          /* FrameData::reinit */
    const bool wpp = si->wpp != 0;
    /* the motion fields' mirrors in device memory (inter_chain_dev.h): the encoder object registers them with its pictures; other callers get them for the call */
    struct TmpMap { const x265amd_mv_unit* h = nullptr; ~TmpMap() { if (h) xa_devmap_unregister(h); } } tmpCur, tmpCol;
    XaMapUnit* frameDCur = (XaMapUnit*)xa_devmap_find(cur);
    const XaMapUnit* frameDCol = (const XaMapUnit*)xa_devmap_find(col);
    if (xa_queues_enabled() && !getenv("X265AMD_DUMP_CTU"))
    {
        if (!frameDCur && (frameDCur = (XaMapUnit*)xa_devmap_register(cur, (size_t)w4 * h4)) != nullptr) tmpCur.h = cur;
        if (col && !frameDCol && si->slice_type != 2 && (frameDCol = (const XaMapUnit*)xa_devmap_register(col, (size_t)w4 * h4)) != nullptr)
        {
            tmpCol.h = col;
            for (int i = 0; i < w4 * h4; i++) devmap_store((XaMapUnit*)frameDCol + i, col[i], 0);
        }
    }
    std::vector<x265amd_cabac*> rows(wpp ? ctuH : 1, nullptr);
    for (size_t r = 0; r < rows.size(); r++)
    {
        rows[r] = x265amd_cabac_open(si, units, 1);
        if (!rows[r]) { for (x265amd_cabac* c : rows) if (c) x265amd_cabac_close(c); return xa_fail(X265AMD_EINVAL, "analyse_frame: slice description"); }
    }
    std::vector<uint8_t> buffered((size_t)ctuH * X265AMD_CTX_STRIDE, 0);
    int rc = X265AMD_OK;
    /* one CTU: analysis with the row coder's state, then the row coder codes it (bits only) to carry the contexts on */
    auto doCtu = [&](int addr, void* st) -> int {
        const int row = addr / ctuW, colIdx = addr % ctuW;
        x265amd_cabac* rowCoder = rows[wpp ? row : 0];
        if (wpp && !colIdx && row)
        {
            /* copyState(m_initSliceContext) + loadContexts(bufferedEntropy of the row above) */
            memcpy(rowCoder->ctx, &buffered[(size_t)(row - 1) * X265AMD_CTX_STRIDE], X265AMD_CTX_STRIDE);
            rowCoder->fracBits = 0;
        }
        x265amd_ctu_result res;
        int16_t* coeff = coeff_out + (size_t)addr * kTileElems;
        int r = compress_ctu_impl(me, st, I, S, si, A, units, cur, col, ref_depth, ref_qp0, h_planes, num_pics, stride, cstride, cu_stat, addr,
                                  rowCoder->ctx, A->use_sao ? rowCoder->fracBits : 0, coeff, &res, frameDCur, frameDCol, cu_qp, tu_recs);
        if (r != X265AMD_OK) return r;
        if (results) results[addr] = res;
        xa_phase(XA_PH_ANALYZER);
        { XA_HOSTPROF("row.cabac_encode_ctu"); r = x265amd_cabac_encode_ctu(rowCoder, addr, coeff, coeff + 4096, coeff + 5120); }
        xa_phase(XA_PH_CABAC_CTU);
        if (wpp && colIdx == 1) memcpy(&buffered[(size_t)row * X265AMD_CTX_STRIDE], rowCoder->ctx, X265AMD_CTX_STRIDE);
        return r;
    };
    int rowThreads = 1;
    if (wpp && ctuH > 1 && ctuW > 1)
    {
        const char* e = getenv("X265AMD_ROW_THREADS");
        rowThreads = e ? atoi(e) : 64;        /* the wavefront never holds more than min(rows, columns / 2) CTUs at once */
        if (rowThreads > ctuH) rowThreads = ctuH;
        if (rowThreads > (ctuW + 1) / 2) rowThreads = (ctuW + 1) / 2;
    }
    const bool dumping = getenv("X265AMD_DUMP_CTU") != nullptr;        /* the dump synchronises the device: not behind a resident kernel */
    if (rowThreads <= 1)
    {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "analyse_frame: synchronize");
        void* q = dumping ? nullptr : xa_queue_acquire();
        for (int addr = 0; addr < numCtu && rc == X265AMD_OK; addr++)
        {
            if (hooks && addr % ctuW == 0)
            {
                struct W { const XaRowHooks* h; int row; } w{ hooks, addr / ctuW };
                xa_wait_until([](void* c) -> int { W* x = (W*)c; return x->h->row_ready(x->h->ctx, x->row) != 0; }, &w);
                if (hooks->row_ready(hooks->ctx, addr / ctuW) < 0) { rc = xa_fail(X265AMD_EHIP, "analyse_frame: a reference picture failed"); break; }
                hooks->before_row(hooks->ctx, addr / ctuW);
            }
            if (hooks && hooks->ctu_wait && hooks->ctu_wait(hooks->ctx, addr / ctuW, addr % ctuW)) { rc = xa_fail(X265AMD_EHIP, "analyse_frame: a reference picture failed"); break; }
            if (hooks && hooks->before_ctu) hooks->before_ctu(hooks->ctx, addr / ctuW, addr % ctuW);
            rc = doCtu(addr, q ? q : stream);
            if (rc == X265AMD_OK && xa_stream_sync(q ? q : stream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "analyse_frame: synchronize");
            if (hooks && rc == X265AMD_OK && hooks->after_ctu) hooks->after_ctu(hooks->ctx, addr / ctuW, addr % ctuW);
            if (hooks && rc == X265AMD_OK && addr % ctuW == ctuW - 1) hooks->after_row(hooks->ctx, addr / ctuW);
        }
        if (q) xa_queue_release(q);
    }
    else
    {
        /* Wavefront parallel processing as the reference's row jobs run it (frameencoder.cpp:1399-1968): CTU (r, c) starts when (r - 1, c + 1)
         * is finished -- its neighbours' decisions, reconstruction, cost statistics and the contexts saved after (r - 1, 1) are then final, so
         * every CTU sees exactly what the serial order shows it.  Every CTU row is a task on the worker threads (xa_fiber.h) with a device job queue
         * while it runs; the block operations of different rows overlap on the device, and a row that waits (for the device, for the row above, for
         * a reference picture) leaves its worker thread to the rows that can go on.  A row takes its queue only when the reference pictures let it
         * start and after the row above has taken one, and gives it back at the end of the row: every held queue belongs to a row that can run,
         * whatever the number of pictures in flight, so waiting for a queue always ends. */
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "analyse_frame: synchronize");
        struct Frame
        {
            std::vector<volatile uint64_t*> done;   /* CTUs finished per row: counters the parked rows below wait on (xa_fiber.h) */
            volatile uint64_t* queuedRows;          /* rows that hold (or have held) a queue */
            std::atomic<int> firstErr{ X265AMD_OK };
            const XaRowHooks* hooks; int ctuW, ctuH; bool dumping; int poc; bool intraOnly, intraTry, pSlice;
            std::function<int(int, void*)> doCtu;
            explicit Frame(int rowsN) : done(rowsN) { for (auto& d : done) d = xa_counter_alloc(); queuedRows = xa_counter_alloc(); }
            ~Frame() { for (auto& d : done) xa_counter_free(d); xa_counter_free(queuedRows); }
        } F(ctuH);
        F.hooks = hooks; F.ctuW = ctuW; F.ctuH = ctuH; F.dumping = dumping; F.poc = I->poc; F.doCtu = doCtu; F.intraOnly = si->slice_type == 2; F.pSlice = si->slice_type == 1;
        F.intraTry = si->slice_type == 1 || (si->slice_type == 0 && A->b_intra);       /* pictures whose CUs try intra beside their inter modes */
        struct Row { Frame* f; int row; };
        std::vector<Row> rowsArg((size_t)ctuH);
        std::vector<XaTask> tasks((size_t)ctuH);
        static std::atomic<uint64_t> frameSeq{ 0 };
        const uint64_t seqCall = frameSeq.fetch_add(1);     /* analyses start in coding order: older pictures' rows go first */
        const uint64_t seq = hooks && hooks->order ? hooks->order : seqCall;
        auto rowReady = [](void* c) -> int {
            Row* r = (Row*)c; Frame& f = *r->f;
            if (f.firstErr.load() != X265AMD_OK) return 1;
            return f.hooks ? f.hooks->row_ready(f.hooks->ctx, r->row) != 0 : 1;
        };
        auto rowMain = [](void* c) {
            Row* r = (Row*)c; Frame& f = *r->f;
            const int row = r->row, ctuW = f.ctuW;
            const auto tStart = std::chrono::steady_clock::now();
            if (void** slot = xa_task_slot()) *slot = (void*)f.hooks;          /* the guards of this row's reference reads (xa_ref_guard_*) */
            if (f.hooks && f.firstErr.load() == X265AMD_OK)
            {
                if (f.hooks->row_ready(f.hooks->ctx, row) < 0) { int ok = X265AMD_OK; f.firstErr.compare_exchange_strong(ok, xa_fail(X265AMD_EHIP, "analyse_frame: a reference picture failed")); }
                else f.hooks->before_row(f.hooks->ctx, row);
            }
            void* st = f.dumping ? nullptr : xa_queue_acquire();
            hipStream_t own = nullptr;
            if (!st)
            {
                if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess) { int ok = X265AMD_OK; f.firstErr.compare_exchange_strong(ok, xa_fail(X265AMD_EHIP, "analyse_frame: stream")); }
                st = own;
            }
            {
                /* the last rows of a picture are what the pictures behind it wait for (and a cut last row is a picture's slowest): their waits for the device poll a
                 * while before the task parks (X265AMD_SPIN_US: microseconds, X265AMD_SPIN_ROWS: how many rows from the bottom; 0 = park at once) */
                static const int spinUs = getenv("X265AMD_SPIN_US") ? atoi(getenv("X265AMD_SPIN_US")) : 0;
                static const int spinRows = getenv("X265AMD_SPIN_ROWS") ? atoi(getenv("X265AMD_SPIN_ROWS")) : 1;
                if (spinUs > 0 && row >= f.ctuH - spinRows) xa_task_spin_ns((uint64_t)spinUs * 1000);
            }
            {
                static const char* const lg = getenv("X265AMD_QUEUE_LOG");
                int lp = -1, lr = -1;
                if (lg && sscanf(lg, "%d,%d", &lp, &lr) == 2 && lp == f.poc && lr == row && st && !own) xa_queue_log(st, lp, lr);
            }
            /* I pictures: a second queue for the row when one is to spare -- the two partitionings of an 8x8 CU are evaluated side by side (intra_rd.hip) */
            void* helper = (f.intraOnly && st && !own) ? xa_queue_try_acquire() : nullptr;
            void* helper2 = helper ? xa_queue_try_acquire() : nullptr;         /* and a third and a fourth: the 16x16 / 32x32 CUs' 2Nx2N evaluations beside their sub-CUs */
            void* helper3 = helper2 ? xa_queue_try_acquire() : nullptr;
            /* P pictures: a second queue for the searches that start ahead of their CU's merge check (Analyzer::searchAhead) */
            static const int auxSpare = getenv("X265AMD_AUX_SPARE") ? atoi(getenv("X265AMD_AUX_SPARE")) : 112;       /* (an I picture started meanwhile needs four queues per row) */
            void* aux = (f.pSlice && st && !own) ? xa_queue_try_acquire_spare(auxSpare) : nullptr;
            void* aux2 = aux ? xa_queue_try_acquire_spare(auxSpare) : nullptr;                /* and a third for the searches of CUs whose sub-CUs come first */
            if (aux) xa_queue_set_aux(st, aux);
            if (aux2) xa_queue_set_aux(aux, aux2);
            if (helper) xa_queue_set_helper(st, helper);
            if (helper2) xa_queue_set_helper(helper, helper2);
            if (helper3) xa_queue_set_helper(helper2, helper3);
            std::atomic_thread_fence(std::memory_order_release);
            *f.queuedRows = (uint64_t)(row + 1);
            const auto tQueue = std::chrono::steady_clock::now();
            for (int c2 = 0; c2 < ctuW; c2++)
            {
                /* a row of an I picture that started while every queue was taken (the picture behind a scene cut starts beside the pictures in front of it) asks again:
                 * without its second to fourth queue its CTUs take twice as long */
                /* a row of a P picture likewise: the rows that started while an I picture held the queues (in lockstep behind it) are the ones still running when it has
                 * ended -- the tail of the clip, where every link is a last-column CTU searched CU by CU -- and the searches ahead of the merge checks need the second and third queue */
                static const bool auxRetry = !(getenv("X265AMD_AUX_RETRY") && atoi(getenv("X265AMD_AUX_RETRY")) == 0);
                if (auxRetry && f.pSlice && st && !own && !aux2)
                {
                    if (!aux) { aux = xa_queue_try_acquire_spare(auxSpare); if (aux) xa_queue_set_aux(st, aux); }
                    if (aux && !aux2) { aux2 = xa_queue_try_acquire_spare(auxSpare); if (aux2) xa_queue_set_aux(aux, aux2); }
                }
                if (f.intraOnly && st && !own && !helper3)
                {
                    if (!helper) { helper = xa_queue_try_acquire(); if (helper) xa_queue_set_helper(st, helper); }
                    if (helper && !helper2) { helper2 = xa_queue_try_acquire(); if (helper2) xa_queue_set_helper(helper, helper2); }
                    if (helper2 && !helper3) { helper3 = xa_queue_try_acquire(); if (helper3) xa_queue_set_helper(helper2, helper3); }
                }
                if (row)
                {
                    const int need = c2 + 2 < ctuW ? c2 + 2 : ctuW;
                    const auto w0 = std::chrono::steady_clock::now();
                    const int wc = xa_task_wait_class(1);
                    xa_wait_counter(f.done[row - 1], (uint64_t)need);          /* a failing row sets its counter to the end, so this always ends */
                    xa_task_wait_class(wc);
                    std::atomic_thread_fence(std::memory_order_acquire);
                    xa_prof_dependency_wait((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count());
                }
                int r2 = f.firstErr.load();
                if (r2 == X265AMD_OK && !st) r2 = X265AMD_EHIP;
                /* the reference pictures are published column by column: this CTU follows them (parked on their counters meanwhile) */
                static const bool ctuLog = getenv("X265AMD_CTU_LOG") != nullptr;
                static const int ctuLogPoc = ctuLog && getenv("X265AMD_CTU_LOG")[0] == 'p' ? atoi(getenv("X265AMD_CTU_LOG") + 1) : -1;       /* "p9": every row of picture 9; anything else: the last three rows of every picture */
                static const auto tLog0 = std::chrono::steady_clock::time_point() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double, std::milli>(
                    floor(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() / 1e6) * 1e6));       /* milliseconds modulo 10^6 of the steady clock: X265AMD_PUB_LOG's clock */
                const double tA = ctuLog ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLog0).count() : 0;
                const int wc2 = xa_task_wait_class(2);
                if (r2 == X265AMD_OK && f.hooks && f.hooks->ctu_wait && f.hooks->ctu_wait(f.hooks->ctx, row, c2)) r2 = xa_fail(X265AMD_EHIP, "analyse_frame: a reference picture failed");
                xa_task_wait_class(wc2);
                const double tB = ctuLog ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLog0).count() : 0;
                struct CtuLog { bool on; int poc, row, col; double a, b; uint64_t p0[4], run0;
                                ~CtuLog() { if (!on) return; uint64_t p1[4]; xa_task_parked_ns(p1);
                                            fprintf(stderr, "x265amd ctu: poc %d row %d col %d: row-above wait from %.2f, gate passed %.2f, done %.2f; in the CTU: running %.2f, waiting for the device %.2f, for reference samples %.2f\n", poc, row, col, a, b,
                                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tLog0).count(), (xa_task_run_ns_always() - run0) / 1e6, (p1[0] - p0[0]) / 1e6, (p1[3] - p0[3]) / 1e6); } }
                    ctuLogger{ ctuLog && (ctuLogPoc >= 0 ? f.poc == ctuLogPoc : row >= f.ctuH - 3), f.poc, row, c2, tA, tB, { 0, 0, 0, 0 }, 0 };
                if (ctuLogger.on) { xa_task_parked_ns(ctuLogger.p0); ctuLogger.run0 = xa_task_run_ns_always(); }
                if (r2 == X265AMD_OK && f.hooks && f.hooks->before_ctu) f.hooks->before_ctu(f.hooks->ctx, row, c2);
                if (r2 == X265AMD_OK) r2 = f.doCtu(row * ctuW + c2, st);
                if (r2 == X265AMD_OK && xa_stream_sync(st) != hipSuccess) r2 = xa_fail(X265AMD_EHIP, "analyse_frame: row stream");
                if (r2 != X265AMD_OK) { int ok = X265AMD_OK; f.firstErr.compare_exchange_strong(ok, r2); }
                std::atomic_thread_fence(std::memory_order_release);
                *f.done[row] = (uint64_t)(c2 + 1);
                if (r2 != X265AMD_OK) break;
                if (f.hooks && f.hooks->after_ctu) f.hooks->after_ctu(f.hooks->ctx, row, c2);
                if (f.hooks && c2 == ctuW - 1) f.hooks->after_row(f.hooks->ctx, row);
            }
            if (f.firstErr.load() != X265AMD_OK) *f.done[row] = (uint64_t)ctuW;
            if (aux2) { xa_queue_set_aux(aux, nullptr); xa_queue_release_helper(aux2); }
            if (aux) { xa_queue_set_aux(st, nullptr); xa_queue_release_helper(aux); }
            if (helper3) { xa_queue_set_helper(helper2, nullptr); xa_queue_release_helper(helper3); }
            if (helper2) { xa_queue_set_helper(helper, nullptr); xa_queue_release_helper(helper2); }
            if (helper) { xa_queue_set_helper(st, nullptr); xa_queue_release_helper(helper); }
            if (own) (void)hipStreamDestroy(own);
            else if (st) xa_queue_release(st);
            if (g_timing && f.hooks)
            {
                static const auto t00 = std::chrono::steady_clock::now();
                auto ms = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - t00).count(); };
                fprintf(stderr, "x265amd: row: poc %d row %d start %.1f queue %.1f end %.1f ran %.1f\n", f.poc, row, ms(tStart), ms(tQueue), ms(std::chrono::steady_clock::now()), xa_task_run_ns() / 1e6);
            }
        };
        for (int r = 0; r < ctuH; r++)
        {
            rowsArg[r] = Row{ &F, r };
            tasks[r].fn = rowMain; tasks[r].arg = &rowsArg[r]; tasks[r].ready = rowReady; tasks[r].readyCtx = &rowsArg[r];
            tasks[r].startCounter = F.queuedRows; tasks[r].startValue = (uint64_t)r;          /* after the row above has taken its queue */
            tasks[r].priority = (seq << 12) | (uint64_t)r;
        }
        if (rc == X265AMD_OK)
        {
            xa_tasks_run(tasks.data(), ctuH);
            rc = F.firstErr.load();
            if (rc != X265AMD_OK) xa_fail(rc, "analyse_frame: a CTU row failed");
        }
    }
    for (x265amd_cabac* c : rows) x265amd_cabac_close(c);
    if (g_timing)
    {
        uint64_t ss[3];
        xa_sched_stats(ss);
        xa_phase_report();
        fprintf(stderr, "x265amd: workers so far: %.1f ms running tasks, %.1f ms looking for one, %llu switches\n", ss[0] / 1e6, ss[1] / 1e6, (unsigned long long)ss[2]);
        fprintf(stderr, "x265amd: skip chains so far: %llu commands, %llu CUs skipped on the device, stopped %llu times at a CU that is not skipped and %llu times at a vector beyond what is published; "
                "device ms: candidates %.1f, predictions + SA8D %.1f, choice %.1f, transform units %.1f, rate-distortion %.1f, placing %.1f, coder state %.1f (%llu CUs)\n", (unsigned long long)g_chainStat[0].load(),
                (unsigned long long)g_chainStat[1].load(), (unsigned long long)g_chainStat[2].load(), (unsigned long long)g_chainStat[3].load(), g_chainTicks[0].load() / 1e5, g_chainTicks[1].load() / 1e5,
                g_chainTicks[2].load() / 1e5, g_chainTicks[3].load() / 1e5, g_chainTicks[4].load() / 1e5, g_chainTicks[5].load() / 1e5, g_chainTicks[7].load() / 1e5, (unsigned long long)g_chainTicks[6].load());
        fprintf(stderr, "x265amd: searches started ahead so far: %llu beside a leaf's merge check, %llu behind the merge check of a CU with sub-CUs; collected %llu, ruled out by the sub-CUs' reference pictures %llu\n",
                (unsigned long long)g_aheadStat[0].load(), (unsigned long long)g_aheadStat[2].load(), (unsigned long long)g_aheadStat[1].load(), (unsigned long long)g_aheadStat[3].load());
        for (int t = 0; t < 2; t++)
            fprintf(stderr, "x265amd: CUs of %s pictures so far by depth 0..3 (skipped on the device / merge check on the host / searched / intra try): %llu/%llu/%llu/%llu %llu/%llu/%llu/%llu %llu/%llu/%llu/%llu %llu/%llu/%llu/%llu\n",
                    t ? "P" : "B", (unsigned long long)g_cuStat[t][0][0].load(), (unsigned long long)g_cuStat[t][0][1].load(), (unsigned long long)g_cuStat[t][0][2].load(), (unsigned long long)g_cuStat[t][0][3].load(),
                    (unsigned long long)g_cuStat[t][1][0].load(), (unsigned long long)g_cuStat[t][1][1].load(), (unsigned long long)g_cuStat[t][1][2].load(), (unsigned long long)g_cuStat[t][1][3].load(),
                    (unsigned long long)g_cuStat[t][2][0].load(), (unsigned long long)g_cuStat[t][2][1].load(), (unsigned long long)g_cuStat[t][2][2].load(), (unsigned long long)g_cuStat[t][2][3].load(),
                    (unsigned long long)g_cuStat[t][3][0].load(), (unsigned long long)g_cuStat[t][3][1].load(), (unsigned long long)g_cuStat[t][3][2].load(), (unsigned long long)g_cuStat[t][3][3].load());
        fprintf(stderr, "x265amd: analysis stages (ms):");
        for (int k = 0; k < 5; k++) { fprintf(stderr, " %s %.1f", g_stageName[k], g_stageMs[k]); g_stageMs[k] = 0; }
        fprintf(stderr, "\n");
    }
    if (rc == X265AMD_OK && slice_data && substream_sizes && num_substreams)
    {
        if (A->use_sao) return xa_fail(X265AMD_EINVAL, "analyse_frame: with SAO the slice data follows the SAO decision: call x265amd_encode_slice_data afterwards");
        rc = x265amd_encode_slice_data(si, units, coeff_out, nullptr, nullptr, slice_data, cap, substream_sizes, num_substreams);
    }
    return rc;
}

/* FrameEncoder::encodeSlice (frameencoder.cpp:1298-1370): the final CABAC pass over the decided picture.  One sub-stream per CTU row under WPP
 * (each row starts from the state saved after the second CTU of the row above and ends with finishSlice), otherwise one for the picture; the
 * SAO syntax of a CTU precedes its coding tree when the slice uses SAO. */
extern "C" int x265amd_encode_slice_data(const x265amd_slice_info* si, x265amd_cu_unit* units, const int16_t* coeffs, const x265amd_sao_ctu* sao,
                                         const int32_t* sao_flags, uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams)
{
    if (!si || !units || !coeffs || !slice_data || !substream_sizes || !num_substreams) return xa_fail(X265AMD_EINVAL, "encode_slice_data: null argument");
    const int ctuW = (si->pic_width + 63) >> 6, ctuH = (si->pic_height + 63) >> 6, numCtu = ctuW * ctuH;
    const bool wpp = si->wpp != 0;
    std::vector<uint8_t> buffered((size_t)ctuH * X265AMD_CTX_STRIDE, 0);
    size_t total = 0;
    int count = 0, rc = X265AMD_OK;
    x265amd_cabac* w = nullptr;
    for (int addr = 0; addr < numCtu && rc == X265AMD_OK; addr++)
    {
        const int row = addr / ctuW, colIdx = addr % ctuW;
        if (!w)
        {
            w = x265amd_cabac_open(si, units, 0);
            if (!w) return xa_fail(X265AMD_EINVAL, "encode_slice_data: slice description");
            if (wpp && row) memcpy(w->ctx, &buffered[(size_t)(row - 1) * X265AMD_CTX_STRIDE], X265AMD_CTX_STRIDE);
        }
        if (sao && sao_flags) w->saoCtu(colIdx, row == 0, sao[addr], sao_flags[0] != 0, sao_flags[1] != 0);
        const int16_t* coeff = coeffs + (size_t)addr * kTileElems;
        rc = x265amd_cabac_encode_ctu(w, addr, coeff, coeff + 4096, coeff + 5120);
        if (wpp && colIdx == 1) memcpy(&buffered[(size_t)row * X265AMD_CTX_STRIDE], w->ctx, X265AMD_CTX_STRIDE);
        if ((wpp && colIdx == ctuW - 1) || addr == numCtu - 1)
        {
            const size_t n = x265amd_cabac_finish_slice(w, slice_data + total, cap > total ? cap - total : 0);
            if (total + n > cap) rc = xa_fail(X265AMD_EINVAL, "encode_slice_data: buffer too small");
            substream_sizes[count++] = (uint32_t)n;
            total += n;
            x265amd_cabac_close(w);
            w = nullptr;
        }
    }
    if (w) x265amd_cabac_close(w);
    *num_substreams = count;
    return rc;
}
