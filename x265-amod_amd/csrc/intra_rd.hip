/* Intra candidates of inter slices (include/x265amd.h: x265amd_intra_in_inter): SURVEY row a7's RD side.
 *
 * Restatement of Search::checkIntraInInter (reference: source/encoder/search.cpp:1291-1452) and encodeIntraInInter (:1454-1507) with
 * codeIntraLumaQT (:305-508), extractIntraResultQT (:763-786), estIntraPredChromaQT (:1754-1889), codeIntraChromaQt (:819-945),
 * codeSubdivCbfQTChroma / codeCoeffQTChroma (:233-303), getIntraRemModeBits (:1698-1709) and Predict::initIntraNeighbors
 * (source/common/predict.cpp:664-715 with the availability look-ups :878-976, cudata.cpp:745-811).
 *
 * Intra blocks predict from the reconstruction of their neighbours, so the transform units of one CU are inherently serial: each TU is
 * one x265amd_intra_tu_chain launch (neighbours from the reconstructed picture, prediction, residual chain, reconstruction) whose
 * reconstruction goes back into the picture before the next TU starts, exactly as the reference's loops do.  The mode scan is one
 * x265amd_intra_scan launch; both chroma planes of a TU share a launch.  Bits and decisions are the host's (host/cabac_coder.h).
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"
#include "xa_queue.h"
#include "../host/cabac_coder.h"
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace {

#if X265AMD_DEPTH < 10
typedef uint32_t sse_t;
#else
typedef uint64_t sse_t;
#endif

struct Snap { uint8_t ctx[X265AMD_CTX_STRIDE]; uint64_t frac; };
struct Cost { uint64_t rdcost; uint32_t bits; sse_t distortion; uint32_t energy; };
const uint64_t kMaxCost = 0x7FFFFFFFFFFFFFFFULL;

struct DevBuf
{
    void* p = nullptr;
    ~DevBuf() { xa_scratch_free(p); }
    hipError_t alloc(size_t bytes) { return xa_scratch_alloc(&p, bytes ? bytes : 16); }
};
struct MappedBuf : XaMapped { void free() { xa_mapped_free(p); p = nullptr; } };

inline unsigned zUnit(int ux, int uy)           /* z-order of unit (ux, uy) inside its CTU */
{
    unsigned r = 0;
    for (int b = 0; b < 4; b++) r |= (((unsigned)ux >> b) & 1u) << (2 * b) | (((unsigned)uy >> b) & 1u) << (2 * b + 1);
    return r;
}

/* The device-side records of a chain of 8x8 CUs (four x265amd_intra_peer + one x265amd_intra_chain) come from a list of their own, not from the pools: the counts in them
 * only ever grow (one counter for the process), so whatever a workgroup still holds of such a record from an earlier chain is an OLDER count and at worst makes it look
 * again -- memory that had been pixels or levels before could read as a count from the future. */
/* what the row's queue has written is out before the other queue's workgroup looks: on the device (xa_queue_follow), or -- X265AMD_QUEUE_FOLLOW=0 -- through the host
 * (the row's queue drained, a signalling command releases; the other queue acquires) */
static hipError_t xa_follow_or_sync(void* follower, void* leader)
{
    static const bool follow = !(getenv("X265AMD_QUEUE_FOLLOW") && atoi(getenv("X265AMD_QUEUE_FOLLOW")) == 0);
    if (follow) return xa_queue_follow(follower, leader);
    if (xa_stream_sync(leader) != hipSuccess) return hipErrorUnknown;
    return xa_stream_fence(follower, XA_CMD_ACQUIRE);
}
static std::mutex g_chainLock;
static std::vector<void*> g_chainFree;
static const size_t kChainBlock = 4 * sizeof(x265amd_intra_peer) + sizeof(x265amd_intra_chain);
static void* chain_block_get()
{
    {
        std::lock_guard<std::mutex> g(g_chainLock);
        if (!g_chainFree.empty()) { void* p = g_chainFree.back(); g_chainFree.pop_back(); return p; }
    }
    void* p = nullptr;
    if (hipMalloc(&p, kChainBlock) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kChainBlock) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(p); return nullptr; }
    return p;
}
static void chain_block_put(void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(g_chainLock);
    g_chainFree.push_back(p);
}

struct IntraRd
{
    hipStream_t st;
    const x265amd_slice_info* si; const x265amd_rd_params* rp;
    x265amd_cu_unit* units; int w4;
    const uint64_t* src; const uint64_t* rec; intptr_t stride, cstride;
    int cuX, cuY, log2, size, depth, qp, qpLumaScaled, qpChromaScaled;
    int range[2];
    x265amd_cabac* c;
    Snap cur, rqtRoot[6], rqtTest[6];
    uint64_t lambda2, lambda; uint32_t psyRd;
    uint64_t predTile, reconTile;
    DevBuf dResi, dLayer, dCand;
    MappedBuf dJobs; XaMapped dScanJob, dPuJob, dNxnJob; XaMappedOut dRes, dCoeff, dScan, dPuOut, dNxnOut;
    DevBuf dCoeffDev;                                   /* candidate levels / residuals of the device-decided NxN path (device memory: only the winner's levels travel) */
    /* The NxN evaluation of an 8x8 CU beside its 2Nx2N evaluation, on a second device job queue: both start from the same contexts and read the same
     * neighbours, neither needs the other's result -- only the reference's ORDER has 2Nx2N first, and the order matters for what the picture holds afterwards
     * (the last tried mode's samples), which the NxN command writes while the 2Nx2N command is told to leave the picture alone.  helper: the second queue;
     * hintPred / hintRecon: the NxN mode's tiles, announced by the caller before the 2Nx2N call; ahead: a running NxN command and the CU it belongs to.  Its
     * scratch is a set of its own (two workgroups write at the same time). */
    void* helper = nullptr;
    uint64_t hintPred = 0, hintRecon = 0;
    struct Ahead { bool on = false; int x = 0, y = 0; } ahead;
    DevBuf dCand2, dCoeffDev2; XaMapped dNxnJob2; XaMappedOut dNxnOut2;
    XaMappedOut dDevLevels, dDevCLevels;                /* the device-decided 16x16 unit: its 256 luma levels, the chroma winner's 2 x 64 */
    /* The four 8x8 CUs of a 16x16 block of an I picture as a chain the device runs by itself (xa_intra_quad8_ws; include/x265amd.h: x265amd_intra_nxn_job.chain):
     * eight job records queued in advance (per CU the NxN evaluation on this queue, the 2Nx2N evaluation on the second), the chain and peer records in device
     * memory, the four results the host reads, the count of chained CUs so far (the chain's tokens never repeat). */
    XaMapped qJobs; XaMappedOut qOut;
    void* qBlock = nullptr;                             /* four peer records and the chain record: memory that is never anything else (chain_block_get) */
    ~IntraRd();
    /* A 16x16 CU's 2Nx2N evaluation started BEFORE the recursion into its four 8x8 CUs, on a third queue, and collected after it (xa_check_intra_begin_ws):
     * it reads only what lies outside the CU and writes only its own tiles (no_picture), so it runs beside the sub-CUs, which own the picture meanwhile. */
    struct Big { void* q = nullptr; bool on = false, owned = false; int x = 0, y = 0; DevBuf cand, coeffDev; XaMapped job; XaMappedOut out, levels, clevels; } big[2];    /* [0] 16x16 (third queue), [1] 32x32 (fourth) */
    const int16_t* curLevels = nullptr; const int16_t* curCLevels = nullptr;       /* where the last device-decided large unit left its levels */
    /* The intra try of a CU of a P / B picture (checkIntraInInter + encodeIntraInInter) as ONE command with the mode picked on the device (pick_sa8d), started at
     * the CU's entry on the queue of its depth (second, third, fourth: 32x32, 16x16, 8x8) and collected when the analysis gets to it -- after the recursion into
     * the sub-CUs and the CU's own inter modes.  A command nobody collects is waited for before the slot is used again. */
    Big inter[3];      /* job / result / level records: host memory the kernels read and write in place (x265amd_host.h) */
    XaMapped mCtx, mEstJob, mRdoq;                      /* RDOQ: the contexts the bit-estimate table is made from, its job record, the per-job RDOQ records */
    DevBuf dEst;                                        /* Entropy::m_estBitsSbac */
    enum { MAX_JOBS = 16 };
    /* a luma TU whose chain already ran in a batch (the candidates of one partition share their neighbours, so they run as one launch):
     * codeIntraLumaQT takes the result instead of launching; the winner's prediction / reconstruction are copied when it is measured again */
    struct Pre { bool on; int x, y, log2; x265amd_tu_result r; const int16_t* lv; uint64_t recon, pred; bool copyBlocks; } pre;
    std::vector<int16_t> coeffL[4];             /* luma levels per transform layer (CUData offsets) */
    std::vector<int16_t> coeffC[2], coeffCBest[2];
    int err;

    /* ---- entropy / cost helpers (as in inter_rd.hip) ---- */
    uint32_t bits() const { return (uint32_t)(c->fracBits >> 15); }
    void resetBits() { c->fracBits &= 32767; }
    void store(Snap& s) const { memcpy(s.ctx, c->ctx, X265AMD_CTX_STRIDE); s.frac = c->fracBits; }
    void load(const Snap& s) { memcpy(c->ctx, s.ctx, X265AMD_CTX_STRIDE); c->fracBits = s.frac; }
    uint64_t calcRdCost(sse_t d, uint32_t b) const { return d + (((uint64_t)b * lambda2 + 128) >> 8); }
    uint64_t calcPsyRdCost(sse_t d, uint32_t b, uint32_t e) const { return d + ((lambda * psyRd * e) >> 24) + (((uint64_t)b * lambda2) >> 8); }
    uint64_t cost(sse_t d, uint32_t b, uint32_t e) const { return psyRd ? calcPsyRdCost(d, b, e) : calcRdCost(d, b); }
    uint64_t calcRdSADCost(uint32_t d, uint32_t b) const { return d + (((uint64_t)b * lambda + 128) >> 8); }

    x265amd_cu_unit& U(int x, int y) { return c->U(x >> 2, y >> 2); }
    bool cbfBit(int x, int y, int plane, int d) { return (U(x, y).cbf[plane] >> d) & 1; }
    void setTuDepth(int x, int y, int sz, int d) { for (int yy = y; yy < y + sz; yy += 4) for (int xx = x; xx < x + sz; xx += 4) U(xx, yy).tu_depth = (uint8_t)d; }
    void setCbf(int plane, int x, int y, int sz, int v) { for (int yy = y; yy < y + sz; yy += 4) for (int xx = x; xx < x + sz; xx += 4) U(xx, yy).cbf[plane] = (uint8_t)v; }
    uint32_t zInCu(int x, int y) const { return zUnit((x - cuX) >> 2, (y - cuY) >> 2); }

    /* Predict::initIntraNeighbors without constrained intra prediction: which 4-sample neighbour units exist and are already coded.
     * (x, y): luma position of the block, sz: its luma extent (a 4:2:0 chroma block covers the same units). */
    uint64_t available(int x, int y, int sz) const
    {
        const int n = sz >> 2, picW = si->pic_width, picH = si->pic_height;
        uint64_t m = 0;
        const int xRT = x + sz - 4, yLB = y + sz - 4;
        for (int i = 0; i < n; i++)             /* below-left, bottom-most first */
        {
            const int k = n - i, ny = yLB + 4 * k;
            bool a = false;
            if (ny < picH && ((yLB & 63) >> 2) < 16 - k)
                a = (x & 63) ? zUnit((x & 63) >> 2, (yLB & 63) >> 2) > zUnit(((x & 63) >> 2) - 1, (ny & 63) >> 2) : x > 0;
            if (a) m |= 1ull << i;
        }
        for (int r = 0; r < n; r++) if (x > 0) m |= 1ull << (2 * n - 1 - r);         /* left */
        if (x > 0 && y > 0) m |= 1ull << (2 * n);                                    /* above-left */
        for (int j = 0; j < n; j++) if (y > 0) m |= 1ull << (2 * n + 1 + j);         /* above */
        for (int k = 1; k <= n; k++)            /* above-right */
        {
            const int nx = xRT + 4 * k;
            bool a = false;
            if (nx < picW)
            {
                if (((xRT & 63) >> 2) < 16 - k)
                    a = (y & 63) ? zUnit((xRT & 63) >> 2, (y & 63) >> 2) > zUnit((nx & 63) >> 2, ((y & 63) >> 2) - 1) : y > 0;
                else
                    a = !(y & 63) && y > 0;
            }
            if (a) m |= 1ull << (3 * n + k);
        }
        return m;
    }

    /* The kernels take one flag per 4 samples of the block's own plane; a 4:2:0 chroma block's flags come per 2 chroma samples (= one luma
     * unit).  Chroma blocks cover 8-aligned luma areas and everything around them is coded in 8x8 luma granules, so the two luma units
     * behind one 4-sample chroma unit always agree: fold them. */
    static uint64_t foldChroma(uint64_t fine, int n)
    {
        const int m = n >> 1;
        uint64_t out = 0;
        auto both = [&](int a, int b) { return ((fine >> a) & 1) & ((fine >> b) & 1); };
        for (int i = 0; i < 2 * m; i++) out |= (uint64_t)both(2 * i, 2 * i + 1) << i;                       /* below-left and left runs */
        out |= ((fine >> (2 * n)) & 1) << (2 * m);                                                          /* above-left */
        for (int j = 0; j < 2 * m; j++) out |= (uint64_t)both(2 * n + 1 + 2 * j, 2 * n + 2 + 2 * j) << (2 * m + 1 + j);   /* above and above-right runs */
        return out;
    }

    int fail(const char* msg) { if (!err) err = xa_fail(X265AMD_EHIP, msg); return err; }

    /* one launch of up to two intra TU jobs; results and levels come back to the host */
    /* estCtx (RDOQ only): the contexts Entropy::estBit(log2 size, luma?) reads before these units' Quant::transformNxN; NULL keeps the table */
    int runJobs(x265amd_intra_tu_job* jobs, int n, x265amd_tu_result* res, int16_t* const* levelsOut, int numCoeff, const uint8_t* estCtx, int tuDepth)
    {
        memcpy(dJobs.p, jobs, sizeof(x265amd_intra_tu_job) * n);
        const x265amd_tu_rdoq* rdoq = nullptr;
        if (rp->rdoq_level)
        {
            if (estCtx)
            {
                memcpy(mCtx.p, estCtx, X265AMD_CTX_COUNT);
                x265amd_est_job* ej = (x265amd_est_job*)mEstJob.p;
                memset(ej, 0, sizeof(*ej));
                ej->ctx = (uint64_t)(uintptr_t)mCtx.p; ej->est = (uint64_t)(uintptr_t)dEst.p; ej->log2_tr_size = jobs[0].tu.log2_tr_size; ej->is_luma = jobs[0].tu.ttype == 0;
                if (x265amd_est_bit(st, ej, 1) != X265AMD_OK) return err = X265AMD_EHIP;
            }
            x265amd_tu_rdoq* rq = (x265amd_tu_rdoq*)mRdoq.p;
            for (int k = 0; k < n; k++)
            {
                memset(&rq[k], 0, sizeof(rq[k]));
                rq[k].est_bits = (uint64_t)(uintptr_t)dEst.p;
                x265amd_rdoq_lambda(jobs[k].tu.qp_scaled, &rq[k].lambda2, &rq[k].lambda);
                rq[k].psy_rdoq_scale = rp->psy_rdoq_scale; rq[k].rdoq_level = (uint8_t)rp->rdoq_level; rq[k].tu_depth = (uint8_t)tuDepth;
            }
            rdoq = rq;
        }
        if (x265amd_intra_tu_chain(st, (const x265amd_intra_tu_job*)dJobs.p, rdoq, n, (x265amd_tu_result*)dRes.p) != X265AMD_OK) return err = X265AMD_EHIP;
        if (xa_stream_sync(st) != hipSuccess) return fail("intra rd: synchronize");
        memcpy(res, dRes.p, sizeof(x265amd_tu_result) * n);
        for (int k = 0; k < n; k++) memcpy(levelsOut[k], (const int16_t*)dCoeff.p + 1024 * k, sizeof(int16_t) * numCoeff);
        return 0;
    }
    void fillJob(x265amd_intra_tu_job& j, int plane, int x, int y, int log2N, int mode, uint64_t pred, int predStride, uint64_t recon, int reconStride, int slot)
    {
        memset(&j, 0, sizeof(j));
        const size_t isz = sizeof(pixel);
        const int sh = plane ? 1 : 0;
        const intptr_t ps = plane ? cstride : stride;
        j.tu.fenc = src[plane] + ((uint64_t)(y >> sh) * ps + (x >> sh)) * isz;
        j.tu.pred = pred; j.tu.pred_stride = predStride;
        j.tu.coeff = (uint64_t)(uintptr_t)((int16_t*)dCoeff.p + 1024 * slot);
        j.tu.resi = (uint64_t)(uintptr_t)((int16_t*)dResi.p + 1024 * slot); j.tu.resi_stride = 1 << log2N;
        j.tu.recon = recon; j.tu.recon_stride = reconStride;
        j.tu.fenc_stride = (int32_t)ps;
        j.tu.log2_tr_size = (uint8_t)log2N; j.tu.ttype = (uint8_t)plane; j.tu.intra = 1; j.tu.dir_mode = (uint8_t)mode;
        j.tu.slice_type = (uint8_t)si->slice_type; j.tu.qp_scaled = (uint8_t)(plane ? qpChromaScaled : qpLumaScaled); j.tu.sign_hide = (uint8_t)(si->sign_hide != 0);
        j.nb = rec[plane] + ((uint64_t)(y >> sh) * ps + (x >> sh)) * isz;
        j.nb_stride = (int32_t)ps;
        j.strong_smoothing = (uint8_t)(rp->strong_intra_smoothing != 0);
    }
    void copy2D(uint64_t dst, size_t dstStride, uint64_t s, size_t srcStride, int w, int h)
    {
        XaRects r;
        r.n = 1; r.dst[0] = dst; r.src[0] = s; r.dst_stride[0] = (int16_t)dstStride; r.src_stride[0] = (int16_t)srcStride; r.w[0] = w; r.h[0] = h;
        xa_copy_rects(st, r);
    }
    void copy2Dx2(uint64_t dst0, size_t dstStride0, uint64_t s0, size_t srcStride0, uint64_t dst1, size_t dstStride1, uint64_t s1, size_t srcStride1, int w, int h)
    {
        XaRects r;
        r.n = 2;
        r.dst[0] = dst0; r.src[0] = s0; r.dst_stride[0] = (int16_t)dstStride0; r.src_stride[0] = (int16_t)srcStride0; r.w[0] = w; r.h[0] = h;
        r.dst[1] = dst1; r.src[1] = s1; r.dst_stride[1] = (int16_t)dstStride1; r.src_stride[1] = (int16_t)srcStride1; r.w[1] = w; r.h[1] = h;
        xa_copy_rects(st, r);
    }

    /* ---- luma ---- */
    int codeIntraLumaQT(int x, int y, int tuDepth, bool bAllowSplit, Cost& outCost)
    {
        const int log2TrSize = log2 - tuDepth, fullDepth = depth + tuDepth, trSize = 1 << log2TrSize, layer = log2TrSize - 2;
        const bool mightNotSplit = log2TrSize <= range[1];
        const bool mightSplit = (log2TrSize > range[0]) && (bAllowSplit || !mightNotSplit);
        const size_t isz = sizeof(pixel);
        Cost fullCost = { 0, 0, 0, 0 };
        uint32_t bCBF = 0;
        x265amd_tu_result rFull;
        memset(&rFull, 0, sizeof(rFull));
        if (!tuDepth) haveWhole = false;
        const uint64_t layerRecon = (uint64_t)(uintptr_t)dLayer.p + ((size_t)layer * 4096 + (size_t)(y - cuY) * 64 + (x - cuX)) * isz;
        if (mightNotSplit)
        {
            if (mightSplit) store(rqtRoot[fullDepth]);
            x265amd_cu_unit& u = U(x, y);
            x265amd_tu_result r;
            int16_t* lv = coeffL[layer].data() + ((size_t)zInCu(x, y) << 4);
            const bool hit = pre.on && pre.x == x && pre.y == y && pre.log2 == log2TrSize;
            x265amd_intra_tu_job job;
            if (!hit)
            {
                fillJob(job, 0, x, y, log2TrSize, u.luma_dir, predTile + ((size_t)(y - cuY) * 64 + (x - cuX)) * isz, 64, layerRecon, 64, 0);
                job.avail = available(x, y, trSize);
            }
            if (hit)
            {
                r = pre.r;
                memcpy(lv, pre.lv, sizeof(int16_t) * trSize * trSize);
                /* a candidate's blocks stay where the batch left them; the winner's go to the layer, the prediction tile and the picture -- in one command below
                 * when no split is tried; before the split trial otherwise (its units then predict into the same tile, as in the reference) */
                if (pre.copyBlocks && mightSplit)
                    copy2Dx2(layerRecon, 64, pre.recon, trSize, predTile + ((size_t)(y - cuY) * 64 + (x - cuX)) * isz, 64, pre.pred, trSize, trSize, trSize);
            }
            else if (runJobs(&job, 1, &r, &lv, trSize * trSize, c->ctx, tuDepth)) return err;
            rFull = r;
            setTuDepth(x, y, trSize, tuDepth);
            bCBF = (uint32_t)(r.num_sig != 0) << tuDepth;
            setCbf(0, x, y, trSize, bCBF);
            fullCost.distortion = (sse_t)r.nz_dist;

            resetBits();
            const bool firstOfCu = x == cuX && y == cuY;
            if (firstOfCu)
            {
                if (si->slice_type != 2)
                {
                    const x265amd_cu_unit* l = c->at((cuX >> 2) - 1, cuY >> 2);
                    const x265amd_cu_unit* a = c->at(cuX >> 2, (cuY >> 2) - 1);
                    const int skipCtx = (x265amd_cabac::coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (x265amd_cabac::coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
                    c->bin(0, C_SKIP + skipCtx);
                    c->bin(1, C_PRED_MODE);
                }
                c->partSize(u, depth, size);
            }
            if (U(cuX, cuY).part_size == 0)
            {
                if (firstOfCu) c->intraDirLuma(&cuX, &cuY, 1);
            }
            else
            {
                const int half = size >> 1;
                if (!tuDepth)
                    for (int q = 0; q < 4; q++) { const int px = cuX + (q & 1) * half, py = cuY + (q >> 1) * half; c->intraDirLuma(&px, &py, 1); }
                else if (!((x - cuX) & (half - 1)) && !((y - cuY) & (half - 1)))
                    c->intraDirLuma(&x, &y, 1);
            }
            if (log2TrSize != range[0]) c->bin(0, C_TRANS_SUBDIV + 5 - log2TrSize);
            c->bin(r.num_sig != 0, C_QT_CBF + !tuDepth);
            if (cbfBit(x, y, 0, tuDepth)) c->coeffNxN(lv, log2TrSize, 0, u);
            fullCost.bits = bits();
            fullCost.energy = psyRd ? r.nz_energy : 0;
            fullCost.rdcost = cost(fullCost.distortion, fullCost.bits, fullCost.energy);
        }
        else
            fullCost.rdcost = kMaxCost;

        if (mightSplit)
        {
            if (mightNotSplit)
            {
                store(rqtTest[fullDepth]);
                load(rqtRoot[fullDepth]);
            }
            Cost splitCost = { 0, 0, 0, 0 };
            uint32_t cbf = 0;
            const int half = trSize >> 1;
            for (int q = 0; q < 4; q++)
            {
                const int qx = x + (q & 1) * half, qy = y + (q >> 1) * half;
                if (codeIntraLumaQT(qx, qy, tuDepth + 1, bAllowSplit, splitCost)) return err;
                cbf |= cbfBit(qx, qy, 0, tuDepth + 1);
            }
            U(x, y).cbf[0] |= (uint8_t)(cbf << tuDepth);
            if (mightNotSplit && log2TrSize != range[0])
            {
                resetBits();
                c->bin(1, C_TRANS_SUBDIV + 5 - log2TrSize);
                splitCost.bits += bits();
                splitCost.rdcost = cost(splitCost.distortion, splitCost.bits, splitCost.energy);
            }
            if (splitCost.rdcost < fullCost.rdcost)
            {
                outCost.rdcost += splitCost.rdcost; outCost.distortion += splitCost.distortion; outCost.bits += splitCost.bits; outCost.energy += splitCost.energy;
                return 0;
            }
            load(rqtTest[fullDepth]);
            setTuDepth(x, y, trSize, tuDepth);
            setCbf(0, x, y, trSize, bCBF);
        }
        /* the reconstruction becomes the neighbourhood of the next blocks.  A candidate taken from the batch (pre.on without copyBlocks) is never looked at
         * in the picture: the same block is measured again as the winner before anything reads it, so its copy is left out. */
        const bool fromBatch = mightNotSplit && pre.on && pre.x == x && pre.y == y && pre.log2 == log2TrSize;
        if (fromBatch && pre.copyBlocks && !mightSplit)
        {
            XaRects r3;
            r3.n = 4;
            r3.dst[0] = layerRecon; r3.src[0] = pre.recon; r3.dst_stride[0] = 64; r3.src_stride[0] = (int16_t)trSize;
            r3.dst[1] = predTile + ((size_t)(y - cuY) * 64 + (x - cuX)) * isz; r3.src[1] = pre.pred; r3.dst_stride[1] = 64; r3.src_stride[1] = (int16_t)trSize;
            r3.dst[2] = rec[0] + ((uint64_t)y * stride + x) * isz; r3.src[2] = pre.recon; r3.dst_stride[2] = (int16_t)stride; r3.src_stride[2] = (int16_t)trSize;
            r3.dst[3] = reconTile + ((size_t)(y - cuY) * 64 + (x - cuX)) * isz; r3.src[3] = pre.recon; r3.dst_stride[3] = 64; r3.src_stride[3] = (int16_t)trSize;     /* the mode's reconstruction tile */
            for (int k = 0; k < 4; k++) { r3.w[k] = (int16_t)trSize; r3.h[k] = (int16_t)trSize; }
            xa_copy_rects(st, r3);
            if (!tuDepth && trSize == size) lumaTileDone = true;
        }
        else if (!fromBatch || pre.copyBlocks)
            copy2D(rec[0] + ((uint64_t)y * stride + x) * isz, stride, layerRecon, 64, trSize, trSize);
        outCost.rdcost += fullCost.rdcost; outCost.distortion += fullCost.distortion; outCost.bits += fullCost.bits; outCost.energy += fullCost.energy;
        /* (after a split trial the prediction tile holds the trial's predictions, which is what the reference measures then: no shortcut) */
        if (!tuDepth && trSize == size && U(cuX, cuY).part_size == 0 && !mightSplit) { haveWhole = true; wholeRes = rFull; }
        return 0;
    }
    void extractLuma(int x, int y, int tuDepth, int16_t* coeffCu)
    {
        const int log2TrSize = log2 - tuDepth;
        if (tuDepth == U(x, y).tu_depth)
        {
            const size_t off = (size_t)zInCu(x, y) << 4;
            memcpy(coeffCu + off, coeffL[log2TrSize - 2].data() + off, sizeof(int16_t) << (2 * log2TrSize));
            return;
        }
        const int half = 1 << (log2TrSize - 1);
        for (int q = 0; q < 4; q++) extractLuma(x + (q & 1) * half, y + (q >> 1) * half, tuDepth + 1, coeffCu);
    }

    /* the 35-mode scan of one block and the mode bits (loadIntraDirModeLuma + getIntraRemModeBits + bitsIntraModeMPM / NonMPM,
     * entropy.h:196-197, :219-224): cost / bits / sa8d per mode */
    int lumaScan(int x, int y, int log2N, uint64_t* modeCosts, uint32_t* modeBitsOut, uint32_t* sadOut)
    {
        const size_t isz = sizeof(pixel);
        x265amd_intra_job sj;
        memset(&sj, 0, sizeof(sj));
        sj.recon = rec[0] + ((uint64_t)y * stride + x) * isz; sj.fenc = src[0] + ((uint64_t)y * stride + x) * isz;
        sj.avail = available(x, y, 1 << log2N);
        sj.recon_stride = (int32_t)stride; sj.fenc_stride = (int32_t)stride; sj.log2_tr_size = (uint8_t)log2N; sj.strong_smoothing = (uint8_t)(rp->strong_intra_smoothing != 0);
        int32_t sa8d[35];
        xa_phase(XA_PH_INTRA_CAND);
        memcpy(dScanJob.p, &sj, sizeof(sj));
        if (x265amd_intra_scan(st, (const x265amd_intra_job*)dScanJob.p, 1, (int32_t*)dScan.p, nullptr) != X265AMD_OK || xa_stream_sync(st) != hipSuccess)
            return fail("intra rd: mode scan");
        memcpy(sa8d, dScan.p, sizeof(sa8d));
        xa_phase(XA_PH_INTRA_SCAN);
        uint32_t preds[3];
        c->lumaPreds(x, y, preds);
        const uint64_t frac = cur.frac & 32767;
        const uint8_t adi = cur.ctx[C_ADI];
        const uint32_t rbits = (uint32_t)((frac + k_bits[adi ^ 0]) >> 15) + 5;
        const uint32_t mpmBase = (uint32_t)((frac + k_bits[adi ^ 1]) >> 15);
        for (uint32_t mode = 0; mode < 35; mode++)
        {
            uint32_t b = rbits;
            for (int i = 0; i < 3; i++) if (preds[i] == mode) { b = mpmBase + (mode == preds[0] ? 1u : 2u); break; }
            modeBitsOut[mode] = b; sadOut[mode] = (uint32_t)sa8d[mode];
            modeCosts[mode] = calcRdSADCost((uint32_t)sa8d[mode], b);
        }
        mpm0 = preds[0];
        return 0;
    }
    uint32_t mpm0;
    /* the transform unit that covers the whole CU, when the CU ends up with that one unit: its measurements are the CU's (psy energy of the reconstruction,
     * residual energy of the prediction), so no separate measurement launch is needed */
    bool haveWhole = false;
    x265amd_tu_result wholeRes;
    bool haveNxn = false;                       /* the device-decided NxN path measured the CU's luma block itself */
    uint32_t nxnPsy = 0, nxnRes = 0;
    x265amd_intra_nxn_out nxnChroma;            /* ... and chose the chroma mode */
    bool haveDevChroma = false;
    bool lumaTileDone = false;                  /* the mode's reconstruction tile already holds the CU's luma (written with the winner's other copies) */

    /* the job record of x265amd_intra_nxn for this CU coded NxN (partSize 3: four 4x4 units) or 2Nx2N (one 8x8 unit); tiles: the mode's prediction / reconstruction
     * tiles; cand / coeffDev: the command's scratch */
    void buildDevJob(x265amd_intra_nxn_job& nj, int partSize, int rdLevel, uint64_t predTileM, uint64_t reconTileM, uint64_t cand, uint64_t coeffDev,
                     void* levelsBuf = nullptr, void* clevelsBuf = nullptr)
    {
        const int initTuDepth = partSize != 0;
        const int devUnits = partSize != 0 ? 4 : 1, devLog2 = partSize != 0 ? 2 : log2, devN = 1 << devLog2;
        const int cLog2 = log2 - 1, cN = 1 << cLog2;        /* the chroma block per plane: 4x4 for an 8x8 CU, 8x8 for a 16x16 CU */
        const size_t isz = sizeof(pixel);
            memset(&nj, 0, sizeof(nj));
            const uint64_t slot0 = cand;
            for (int k = 0; k < devUnits; k++)
            {
                const int px = cuX + (k & 1) * 4, py = cuY + (k >> 1) * 4;
                fillJob(nj.tmpl[k], 0, px, py, devLog2, 0, slot0 + 1024 * isz, devN, slot0, devN, 0);
                nj.tmpl[k].tu.coeff = coeffDev;
                nj.tmpl[k].tu.resi = coeffDev + (size_t)MAX_JOBS * 1024 * 2; nj.tmpl[k].tu.resi_stride = devN;
                nj.tmpl[k].avail = available(px, py, devN);
                nj.pred_dst[k] = predTileM + ((size_t)(py - cuY) * 64 + (px - cuX)) * isz;
                nj.layer_dst[k] = (uint64_t)(uintptr_t)dLayer.p + ((size_t)(devLog2 - 2) * 4096 + (size_t)(py - cuY) * 64 + (px - cuX)) * isz;     /* the units' layer */
                nj.recon_dst[k] = reconTileM + ((size_t)(py - cuY) * 64 + (px - cuX)) * isz;
                nj.frac_start[k] = cur.frac & 32767;
            }
            nj.num_units = (uint8_t)devUnits; nj.unit_log2 = (uint8_t)devLog2;
            {
                /* what codeIntraLumaQT codes in front of the first unit's direction: skip flag and prediction mode (P / B slices), the partition size */
                load(cur); resetBits();
                if (si->slice_type != 2)
                {
                    const x265amd_cu_unit* l = c->at((cuX >> 2) - 1, cuY >> 2);
                    const x265amd_cu_unit* a = c->at(cuX >> 2, (cuY >> 2) - 1);
                    const int skipCtx = (x265amd_cabac::coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (x265amd_cabac::coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
                    c->bin(0, C_SKIP + skipCtx);
                    c->bin(1, C_PRED_MODE);
                }
                c->partSize(U(cuX, cuY), depth, size);
                nj.frac_start[0] = c->fracBits;
                load(cur);
            }
            auto seen = [&](int x, int y, bool aboveOne) -> uint8_t {       /* getIntraDirLumaPredictor's view of a neighbour outside the CU (cudata.cpp:910-953) */
                const x265amd_cu_unit* u = aboveOne ? ((y & 63) ? c->at(x >> 2, (y >> 2) - 1) : nullptr) : c->at((x >> 2) - 1, y >> 2);
                return (uint8_t)((x265amd_cabac::coded(u) && u->pred_mode == X265AMD_MODE_INTRA) ? u->luma_dir : 1);
            };
            nj.left_mode[0] = seen(cuX, cuY, false); nj.left_mode[1] = seen(cuX, cuY + 4, false);
            nj.above_mode[0] = seen(cuX, cuY, true); nj.above_mode[1] = seen(cuX + 4, cuY, true);
            nj.lambda = lambda; nj.lambda2 = lambda2; nj.psy_scale = psyRd ? lambda * psyRd : 0;
            nj.scan_frac = (uint32_t)(cur.frac & 32767);
            nj.slot_pixels = 2048; nj.slot_coeffs = 1024;
            memcpy(nj.ctx, cur.ctx, X265AMD_CTX_STRIDE);
            nj.max_cand = (uint8_t)(2 + rdLevel + ((depth + initTuDepth) >> 1));
            {
                /* the chroma decision rides along (estIntraPredChromaQT's one 4x4 block per plane: its inputs do not depend on the luma result, its mode list does
                 * and the device knows the first unit's winner) */
                const uint64_t avail = foldChroma(available(cuX, cuY, size), size >> 2);
                for (int pl = 1; pl < 3; pl++)
                {
                    fillJob(nj.ctmpl[pl - 1], pl, cuX, cuY, cLog2, 0, 0, cN, slot0, cN, 0);
                    nj.ctmpl[pl - 1].avail = avail;
                    nj.ctmpl[pl - 1].tu.coeff = coeffDev;
                    nj.ctmpl[pl - 1].tu.resi = coeffDev + (size_t)MAX_JOBS * 1024 * 2; nj.ctmpl[pl - 1].tu.resi_stride = cN;
                    nj.crecon_dst[pl - 1] = reconTileM + (4096 + (size_t)(pl - 1) * 1024) * isz;
                }
                nj.do_chroma = 1;
            }
            if (rp->rdoq_level)
            {
                /* RDOQ: the command makes its bit-estimate tables from nj.ctx (m_rqt[depth].cur: where every candidate and every chroma mode starts) */
                static const bool nxn4Rdoq = !(getenv("X265AMD_NXN4_RDOQ") && atoi(getenv("X265AMD_NXN4_RDOQ")) == 0);
                nj.rdoq_level = (uint8_t)rp->rdoq_level; nj.psy_rdoq_scale = rp->psy_rdoq_scale; nj.rdoq_tu_depth = (uint8_t)initTuDepth;
                nj.rdoq_general = nxn4Rdoq ? 0 : 1;
                x265amd_rdoq_lambda(qpLumaScaled, &nj.rdoq_lambda2[0], &nj.rdoq_lambda[0]);
                x265amd_rdoq_lambda(qpChromaScaled, &nj.rdoq_lambda2[1], &nj.rdoq_lambda[1]);
                nj.rdoq_lambda2[2] = nj.rdoq_lambda2[1]; nj.rdoq_lambda[2] = nj.rdoq_lambda[1];
            }
            if (devLog2 > 3)
            {
                nj.levels_dst = (uint64_t)(uintptr_t)(levelsBuf ? levelsBuf : dDevLevels.p); nj.clevels_dst = (uint64_t)(uintptr_t)(clevelsBuf ? clevelsBuf : dDevCLevels.p);
            }
    }

    /* checkIntraInInter + encodeIntraInInter of this CU as one command (decided on the device) possible? */
    bool devIntraInInter() const
    {
        static const bool on = !(getenv("X265AMD_DEVICE_IININTER") && atoi(getenv("X265AMD_DEVICE_IININTER")) == 0);
        static const bool devRdoq = !(getenv("X265AMD_DEVICE_RDOQ") && atoi(getenv("X265AMD_DEVICE_RDOQ")) == 0);
        return on && (!rp->rdoq_level || devRdoq) && !rp->fast_intra && log2 >= 3 && log2 <= 5 && range[0] == log2 && range[1] >= log2 && !si->tq_bypass_enabled;
    }
    void buildInInterJob(x265amd_intra_nxn_job& nj, uint64_t predTileM, uint64_t reconTileM, uint64_t cand, uint64_t coeffDev, void* levelsBuf, void* clevelsBuf)
    {
        buildDevJob(nj, 0, 0, predTileM, reconTileM, cand, coeffDev, levelsBuf, clevelsBuf);
        nj.pick_sa8d = 1; nj.max_cand = 1;
        nj.no_picture = 1;              /* the picture gets the CU's best mode when the CU is decided (copyToPic); nothing reads this try's samples there before */
    }

    /* Search::estIntraPredQT (search.cpp:1509-1696): per partition the scan, the candidate list, simple RDO of the candidates, then the
     * best mode again with TU splits allowed */
    int estIntraPredQT(int partSize, int rdLevel, sse_t& totalDistortion)
    {
        const int initTuDepth = partSize != 0, numPU = 1 << (2 * initTuDepth), log2TrSize = log2 - initTuDepth, tuSize = 1 << log2TrSize;
        totalDistortion = 0;
        /* An 8x8 CU coded NxN: the four 4x4 units with their decisions are ONE launch (x265amd_intra_nxn) -- a 4x4 unit has no transform split to try and its
         * coefficients are few enough for the device to count their bits, so nothing of the host's enters between the units.  The host repeats the winners'
         * bookkeeping (units, bits, contexts) afterwards.  Not with RDOQ (the quantiser then reads bit estimates of the current contexts per unit). */
        x265amd_intra_nxn_out nxn;
        /* The device routine also takes the same CU coded 2Nx2N -- one 8x8 unit, num_units = 1, chroma decision included: one round trip instead of two and no
         * candidate bits on the host.  (It paid only once the bits of a unit were counted by a wavefront, a lane per context: a single lane needs 45 us for the 64
         * coefficients.)  X265AMD_DEVICE_2Nx2N=0 takes the prediction-unit step with host bits instead. */
        static const bool dev2Nx2N = !(getenv("X265AMD_DEVICE_2Nx2N") && atoi(getenv("X265AMD_DEVICE_2Nx2N")) == 0);
        /* ... and the 16x16 CU coded 2Nx2N with its one 16x16 unit (chroma blocks 8x8): X265AMD_DEVICE_16=0 leaves it to the prediction-unit step */
        static const bool dev16 = !(getenv("X265AMD_DEVICE_16") && atoi(getenv("X265AMD_DEVICE_16")) == 0);
        static const bool dev32 = !(getenv("X265AMD_DEVICE_32") && atoi(getenv("X265AMD_DEVICE_32")) == 0);
        /* RDOQ (round 5): the command makes the bit estimates itself and every chain runs Quant::rdoQuant (x265amd_intra_nxn_job.rdoq_level); X265AMD_DEVICE_RDOQ=0:
         * scan and chains as separate steps with the estimates from the host's contexts, as before */
        static const bool devRdoq = !(getenv("X265AMD_DEVICE_RDOQ") && atoi(getenv("X265AMD_DEVICE_RDOQ")) == 0);
        const bool deviceNxN = (!rp->rdoq_level || devRdoq) && 2 + rdLevel + ((depth + initTuDepth) >> 1) <= MAX_JOBS &&
                               (partSize != 0 ? (log2 == 3 && log2TrSize == 2 && range[0] == 2)
                                              : (dev2Nx2N && (log2 == 3 || (log2 == 4 && dev16) || (log2 == 5 && dev32)) && range[0] == log2 && range[1] >= log2));
        const int devUnits = partSize != 0 ? 4 : 1, devLog2 = partSize != 0 ? 2 : 3, devN = 1 << devLog2;
        if (deviceNxN)
        {
            xa_phase(XA_PH_INTRA_CAND);
            const bool mine = ahead.on && partSize != 0 && ahead.x == cuX && ahead.y == cuY;
            Big& bg = big[log2 == 5 ? 1 : 0];
            const bool mine16 = bg.on && partSize == 0 && (log2 == 4 || log2 == 5) && bg.x == cuX && bg.y == cuY;
            curLevels = (const int16_t*)dDevLevels.p; curCLevels = (const int16_t*)dDevCLevels.p;
            if (mine16)
            {
                /* started before the recursion into the sub-CUs, on the third queue */
                bg.on = false;
                if (xa_stream_sync(bg.q) != hipSuccess) return fail("intra rd: large unit step");
                memcpy(&nxn, bg.out.p, sizeof(nxn));
                curLevels = (const int16_t*)bg.levels.p; curCLevels = (const int16_t*)bg.clevels.p;
                if (xa_stream_fence(st, XA_CMD_ACQUIRE) != hipSuccess) return fail("intra rd: fence");
            }
            else if (mine)
            {
                /* this CU's NxN command has been running on the second queue since the 2Nx2N call: its results, and what it wrote for the first queue to see */
                ahead.on = false;
                if (xa_stream_sync(helper) != hipSuccess) return fail("intra rd: NxN step");
                memcpy(&nxn, dNxnOut2.p, sizeof(nxn));
                if (xa_stream_fence(st, XA_CMD_ACQUIRE) != hipSuccess) return fail("intra rd: fence");
            }
            else
            {
                ahead.on = false;
                x265amd_intra_nxn_job nj;
                /* the NxN command first, on the second queue, when the caller has announced that mode's tiles (an 8x8 CU of an I picture): everything this queue has
                 * written is out (synchronised: a signalling command releases), the other workgroup looks (acquire) */
                const uint32_t loN = (uint32_t)log2 - (uint32_t)(si->tu_max_depth_intra - 1 + 1);           /* the NxN call's transform range, as intra_cu_impl will find it */
                const int rangeN0 = loN < (uint32_t)si->tu_log2_min ? si->tu_log2_min : (loN > (uint32_t)si->tu_log2_max ? si->tu_log2_max : (int)loN);
                const bool goAhead = partSize == 0 && helper && hintPred && hintRecon && rangeN0 == 2 && 2 + rdLevel + ((depth + 1) >> 1) <= MAX_JOBS && dCand2.p;
                if (goAhead)
                {
                    const uint8_t keep = U(cuX, cuY).part_size;
                    U(cuX, cuY).part_size = 3;
                    buildDevJob(nj, 3, rdLevel, hintPred, hintRecon, (uint64_t)(uintptr_t)dCand2.p, (uint64_t)(uintptr_t)dCoeffDev2.p);
                    U(cuX, cuY).part_size = keep;
                    memcpy(dNxnJob2.p, &nj, sizeof(nj));
                    if (xa_follow_or_sync(helper, st) != hipSuccess ||
                        x265amd_intra_nxn(helper, (const x265amd_intra_nxn_job*)dNxnJob2.p, (x265amd_intra_nxn_out*)dNxnOut2.p) != X265AMD_OK)
                        return fail("intra rd: NxN step ahead");
                    ahead.on = true; ahead.x = cuX; ahead.y = cuY;
                }
                hintPred = hintRecon = 0;
                buildDevJob(nj, partSize, rdLevel, predTile, reconTile, (uint64_t)(uintptr_t)dCand.p, (uint64_t)(uintptr_t)dCoeffDev.p);
                nj.no_picture = goAhead ? 1 : 0;
                memcpy(dNxnJob.p, &nj, sizeof(nj));
                if (x265amd_intra_nxn(st, (const x265amd_intra_nxn_job*)dNxnJob.p, (x265amd_intra_nxn_out*)dNxnOut.p) != X265AMD_OK || xa_stream_sync(st) != hipSuccess)
                    return fail("intra rd: NxN step");
                memcpy(&nxn, dNxnOut.p, sizeof(nxn));
            }
            haveNxn = partSize != 0; nxnPsy = nxn.psy_energy; nxnRes = nxn.res_energy;      /* (one 8x8 unit: the unit's own result, haveWhole) */
            haveDevChroma = true; lumaTileDone = true;
            nxnChroma = nxn;
            xa_phase(XA_PH_INTRA_SCAN);
        }
        for (int puIdx = 0; puIdx < numPU; puIdx++)
        {
            const int px = cuX + (initTuDepth ? (puIdx & 1) * tuSize : 0), py = cuY + (initTuDepth ? (puIdx >> 1) * tuSize : 0);
            uint32_t bmode = 0;
            uint64_t candCostList[35]; uint32_t rdModeList[35];
            const int maxCandCount = 2 + rdLevel + ((depth + initTuDepth) >> 1);
            const size_t isz = sizeof(pixel);
            if (deviceNxN)
            {
                /* the unit was decided on the device: its mode, result and levels; its blocks are in place.  codeIntraLumaQT repeats the bookkeeping. */
                const uint32_t bm = nxn.mode[puIdx];
                if (bm > 34) return fail("intra rd: NxN mode");
                for (int yy = 0; yy < tuSize; yy += 4) for (int xx = 0; xx < tuSize; xx += 4) U(px + xx, py + yy).luma_dir = (uint8_t)bm;
                load(cur);
                Cost ic = { 0, 0, 0, 0 };
                pre = Pre{ true, px, py, log2TrSize, nxn.res[puIdx], log2TrSize > 3 ? curLevels : &nxn.levels[0][0] + 16 * puIdx, 0, 0, false };
                const int rq = codeIntraLumaQT(px, py, initTuDepth, true, ic);
                pre.on = false;
                if (rq) return err;
                totalDistortion += ic.distortion;
                continue;
            }
            /* Without RDOQ nothing the host knows enters between the scan and the candidates' transform chains: scan, candidate list and chains are ONE
             * launch (x265amd_intra_pu), the host's share starts with the bits.  (With RDOQ the bit estimates made from the current contexts go to the
             * device first: scan and chains stay two steps.) */
            const bool fused = !rp->rdoq_level && log2TrSize <= range[1] && maxCandCount <= MAX_JOBS;
            int numCand = 0;
            std::vector<x265amd_tu_result> cres;
            std::vector<int16_t> clev;
            uint64_t bcost;
            if (fused)
            {
                xa_phase(XA_PH_INTRA_CAND);
                uint32_t preds[3];
                c->lumaPreds(px, py, preds);
                const uint64_t frac = cur.frac & 32767;
                const uint8_t adi = cur.ctx[C_ADI];
                x265amd_intra_pu_job pj;
                memset(&pj, 0, sizeof(pj));
                const uint64_t slot0 = (uint64_t)(uintptr_t)dCand.p;
                fillJob(pj.tmpl, 0, px, py, log2TrSize, 0, slot0 + 1024 * isz, tuSize, slot0, tuSize, 0);
                pj.tmpl.avail = available(px, py, tuSize);
                pj.lambda = lambda;
                pj.rbits = (uint32_t)((frac + k_bits[adi ^ 0]) >> 15) + 5;           /* as lumaScan below: bitsIntraModeNonMPM / bitsIntraModeMPM */
                pj.mpm_base = (uint32_t)((frac + k_bits[adi ^ 1]) >> 15);
                pj.slot_pixels = 2048; pj.slot_coeffs = 1024;
                for (int i = 0; i < 3; i++) pj.preds[i] = (uint8_t)preds[i];
                pj.max_cand = (uint8_t)maxCandCount;
                memcpy(dPuJob.p, &pj, sizeof(pj));
                if (x265amd_intra_pu(st, (const x265amd_intra_pu_job*)dPuJob.p, (x265amd_intra_pu_out*)dPuOut.p, (x265amd_tu_result*)dRes.p) != X265AMD_OK || xa_stream_sync(st) != hipSuccess)
                    return fail("intra rd: prediction unit step");
                x265amd_intra_pu_out po;
                memcpy(&po, dPuOut.p, sizeof(po));
                xa_phase(XA_PH_INTRA_SCAN);
                numCand = (int)po.num_cand;
                if (numCand < 0 || numCand > maxCandCount) return fail("intra rd: candidate count");
                for (int i = 0; i < numCand; i++) rdModeList[i] = po.modes[i];
                cres.resize((size_t)numCand); clev.resize((size_t)numCand * 1024);
                memcpy(cres.data(), dRes.p, sizeof(x265amd_tu_result) * numCand);
                for (int i = 0; i < numCand; i++) memcpy(clev.data() + (size_t)i * 1024, (const int16_t*)dCoeff.p + 1024 * i, sizeof(int16_t) * tuSize * tuSize);
                mpm0 = preds[0];
            }
            else
            {
            uint64_t modeCosts[35]; uint32_t mb[35], ms[35];
            if (lumaScan(px, py, log2TrSize, modeCosts, mb, ms)) return err;
            bcost = modeCosts[1];
            if (modeCosts[0] < bcost) bcost = modeCosts[0];
            for (int mode = 2; mode < 35; mode++) if (modeCosts[mode] < bcost) bcost = modeCosts[mode];
            for (int i = 0; i < maxCandCount; i++) candCostList[i] = kMaxCost;
            const uint64_t paddedBcost = bcost + (bcost >> 2);
            for (uint32_t mode = 0; mode < 35; mode++)
                if (modeCosts[mode] < paddedBcost || mode == mpm0)
                {
                    /* updateCandList (search.cpp:3953-3972) */
                    uint32_t maxIndex = 0; uint64_t maxValue = 0;
                    for (int i = 0; i < maxCandCount; i++) if (maxValue < candCostList[i]) { maxValue = candCostList[i]; maxIndex = (uint32_t)i; }
                    if (modeCosts[mode] < maxValue) { candCostList[maxIndex] = modeCosts[mode]; rdModeList[maxIndex] = mode; }
                }
            /* the candidates only differ in the mode: their transform chains run as ONE launch, the bits and costs follow on the host in the
             * reference's order; the winner's chain is not run again when it is measured with splits allowed */
            while (numCand < maxCandCount && candCostList[numCand] != kMaxCost) numCand++;
            }
            const bool batch = fused || (log2TrSize <= range[1] && numCand <= MAX_JOBS);
            if (!fused) { cres.resize((size_t)numCand); clev.resize((size_t)numCand * 1024); }
            if (batch && numCand && !fused)
            {
                std::vector<x265amd_intra_tu_job> jobs((size_t)numCand);
                std::vector<int16_t*> lvp((size_t)numCand);
                const uint64_t avail = available(px, py, tuSize);
                for (int i = 0; i < numCand; i++)
                {
                    const uint64_t slot = (uint64_t)(uintptr_t)dCand.p + (size_t)i * 2048 * isz;
                    fillJob(jobs[i], 0, px, py, log2TrSize, (int)rdModeList[i], slot + 1024 * isz, tuSize, slot, tuSize, i);
                    jobs[i].avail = avail;
                    lvp[i] = clev.data() + (size_t)i * 1024;
                }
                if (runJobs(jobs.data(), numCand, cres.data(), lvp.data(), tuSize * tuSize, cur.ctx, initTuDepth)) return err;      /* every candidate starts from m_rqt[depth].cur */
            }
            bcost = kMaxCost;
            int bestIdx = -1;
            for (int i = 0; i < numCand; i++)
            {
                load(cur);
                for (int yy = 0; yy < tuSize; yy += 4) for (int xx = 0; xx < tuSize; xx += 4) U(px + xx, py + yy).luma_dir = (uint8_t)rdModeList[i];
                Cost icosts = { 0, 0, 0, 0 };
                if (batch)
                {
                    const uint64_t slot = (uint64_t)(uintptr_t)dCand.p + (size_t)i * 2048 * isz;
                    pre = Pre{ true, px, py, log2TrSize, cres[i], clev.data() + (size_t)i * 1024, slot, slot + 1024 * isz, false };
                }
                const int rcq = codeIntraLumaQT(px, py, initTuDepth, false, icosts);
                pre.on = false;
                if (rcq) return err;
                if (icosts.rdcost < bcost) { bcost = icosts.rdcost; bmode = rdModeList[i]; bestIdx = i; }
            }
            for (int yy = 0; yy < tuSize; yy += 4) for (int xx = 0; xx < tuSize; xx += 4) U(px + xx, py + yy).luma_dir = (uint8_t)bmode;
            load(cur);
            Cost icosts = { 0, 0, 0, 0 };
            if (batch && bestIdx >= 0)
            {
                const uint64_t slot = (uint64_t)(uintptr_t)dCand.p + (size_t)bestIdx * 2048 * isz;
                pre = Pre{ true, px, py, log2TrSize, cres[bestIdx], clev.data() + (size_t)bestIdx * 1024, slot, slot + 1024 * isz, true };
            }
            const int rcq = codeIntraLumaQT(px, py, initTuDepth, true, icosts);
            pre.on = false;
            if (rcq) return err;
            totalDistortion += icosts.distortion;
        }
        if (numPU > 1)
        {
            uint32_t comb = 0;
            const int half = size >> 1;
            for (int q = 0; q < 4; q++) comb |= cbfBit(cuX + (q & 1) * half, cuY + (q >> 1) * half, 0, 1);
            U(cuX, cuY).cbf[0] |= (uint8_t)comb;
        }
        load(cur);
        return 0;
    }

    /* ---- chroma ---- */
    int codeIntraChromaQt(int x, int y, int tuDepth, Cost& outCost)
    {
        const int log2TrSize = log2 - tuDepth;
        if (tuDepth < U(x, y).tu_depth)
        {
            const int half = 1 << (log2TrSize - 1);
            uint32_t splitCbfU = 0, splitCbfV = 0;
            for (int q = 0; q < 4; q++)
            {
                const int qx = x + (q & 1) * half, qy = y + (q >> 1) * half;
                if (codeIntraChromaQt(qx, qy, tuDepth + 1, outCost)) return err;
                splitCbfU |= cbfBit(qx, qy, 1, tuDepth + 1);
                splitCbfV |= cbfBit(qx, qy, 2, tuDepth + 1);
            }
            U(x, y).cbf[1] |= (uint8_t)(splitCbfU << tuDepth);
            U(x, y).cbf[2] |= (uint8_t)(splitCbfV << tuDepth);
            return 0;
        }
        int log2TrSizeC = log2TrSize - 1;
        if (log2TrSizeC < 2)
        {
            if ((x & 4) || (y & 4)) return 0;
            log2TrSizeC = 2;
        }
        const int areaLuma = 2 << log2TrSizeC, nC = 1 << log2TrSizeC;
        const size_t isz = sizeof(pixel);
        const x265amd_cu_unit& u = U(x, y);
        int mode = u.chroma_dir == 36 ? U(cuX, cuY).luma_dir : u.chroma_dir;
        x265amd_intra_tu_job jobs[2];
        x265amd_tu_result r[2];
        int16_t* lv[2];
        const uint64_t avail = foldChroma(available(x, y, areaLuma), areaLuma >> 2);
        for (int p = 1; p < 3; p++)
        {
            const uint64_t picC = rec[p] + ((uint64_t)(y >> 1) * cstride + (x >> 1)) * isz;
            fillJob(jobs[p - 1], p, x, y, log2TrSizeC, mode, 0, nC, picC, (int)cstride, p - 1);
            jobs[p - 1].avail = avail;
            lv[p - 1] = coeffC[p - 1].data() + (((size_t)zInCu(x, y) << 4) >> 2);
        }
        if (runJobs(jobs, 2, r, lv, nC * nC, c->ctx, U(x, y).tu_depth)) return err;        /* rdoQuant reads cu.m_tuDepth[absPartIdx] for the CBF context */
        for (int p = 1; p < 3; p++)
        {
            setCbf(p, x, y, areaLuma, r[p - 1].num_sig ? 1 << tuDepth : 0);
            outCost.distortion += (sse_t)r[p - 1].nz_dist;          /* scaleChromaDist: weight 256 */
            if (psyRd) outCost.energy += r[p - 1].nz_energy;
        }
        return 0;
    }
    void codeSubdivCbfQTChroma(int x, int y, int tuDepth)
    {
        const bool subdiv = tuDepth < U(x, y).tu_depth;
        const int log2TrSize = log2 - tuDepth;
        if (!(log2TrSize - 1 < 2))
        {
            const int psz = 2 << log2TrSize;
            const int px = cuX + ((x - cuX) & ~(psz - 1)), py = cuY + ((y - cuY) & ~(psz - 1));
            for (int p = 1; p < 3; p++)
                if (!tuDepth || cbfBit(px, py, p, tuDepth - 1))
                {
                    const bool canQuadSplit = log2TrSize - 1 > 2;
                    const int lowest = tuDepth + ((subdiv && !canQuadSplit) ? 1 : 0);
                    c->bin(cbfBit(x, y, p, lowest), C_QT_CBF + tuDepth + 2);
                }
        }
        if (subdiv)
        {
            const int half = 1 << (log2TrSize - 1);
            for (int q = 0; q < 4; q++) codeSubdivCbfQTChroma(x + (q & 1) * half, y + (q >> 1) * half, tuDepth + 1);
        }
    }
    void codeCoeffQTChroma(int x, int y, int tuDepth, int p)
    {
        if (!cbfBit(x, y, p, tuDepth)) return;
        const int log2TrSize = log2 - tuDepth;
        if (tuDepth < U(x, y).tu_depth)
        {
            const int half = 1 << (log2TrSize - 1);
            for (int q = 0; q < 4; q++) codeCoeffQTChroma(x + (q & 1) * half, y + (q >> 1) * half, tuDepth + 1, p);
            return;
        }
        int log2TrSizeC = log2TrSize - 1;
        if (log2TrSizeC < 2)
        {
            if ((x & 4) || (y & 4)) return;
            log2TrSizeC = 2;
        }
        c->coeffNxN(coeffC[p - 1].data() + (((size_t)zInCu(x, y) << 4) >> 2), log2TrSizeC, p, U(x, y));
    }
    int estIntraPredChromaQT(sse_t& totalDistortion)
    {
        const int n4 = size >> 2;
        uint32_t modeList[5] = { 0, 26, 10, 1, 36 };            /* CUData::getAllowedChromaDir (cudata.cpp:889-907) */
        const uint32_t lumaDir = U(cuX, cuY).luma_dir;
        for (int i = 0; i < 4; i++) if (lumaDir == modeList[i]) { modeList[i] = 34; break; }
        uint32_t bestMode = 0; sse_t bestDist = 0; uint64_t bestCost = kMaxCost;
        std::vector<uint8_t> bestCbf(2 * (size_t)n4 * n4, 0);
        const size_t isz = sizeof(pixel);
        if (haveDevChroma)
        {
            /* decided on the device with the luma units (x265amd_intra_nxn, do_chroma): what the loop below leaves behind for the winner -- its blocks are in place */
            const int td1 = U(cuX, cuY).tu_depth;
            const int k = (int)nxnChroma.chroma_best;
            if (k < 0 || k > 4) return fail("intra rd: chroma mode index");
            for (int yy = 0; yy < size; yy += 4) for (int xx = 0; xx < size; xx += 4) U(cuX + xx, cuY + yy).chroma_dir = (uint8_t)modeList[k];
            sse_t dist = 0;
            for (int p = 1; p < 3; p++)
            {
                const x265amd_tu_result& r = nxnChroma.cres[p - 1];
                const int cN = size >> 1;
                const int16_t* lvC = cN > 4 ? curCLevels + (size_t)(p - 1) * cN * cN : nxnChroma.clevels[p - 1];
                memcpy(coeffC[p - 1].data(), lvC, sizeof(int16_t) * cN * cN);
                setCbf(p, cuX, cuY, size, r.num_sig ? 1 << td1 : 0);
                if (td1) U(cuX, cuY).cbf[p] |= (uint8_t)((U(cuX, cuY).cbf[p] >> td1) & 1);
                dist += (sse_t)r.nz_dist;
            }
            coeffCBest[0] = coeffC[0]; coeffCBest[1] = coeffC[1];
            totalDistortion = dist;
            load(cur);
            return 0;
        }
        /* one chroma block per plane for the whole CU (no luma transform split, or an 8x8 CU): the five modes x two planes are independent
         * and run as ONE launch; bits, costs and the choice follow on the host in the reference's order */
        const int td = U(cuX, cuY).tu_depth;
        const bool single = td == 0 || (log2 == 3 && td == 1);
        std::vector<x265amd_tu_result> cres(10);
        std::vector<int16_t> clev((size_t)10 * 1024);
        const int log2C = log2 == 3 ? 2 : log2 - 1 - 0, nC = 1 << (single ? (log2 == 3 ? 2 : log2 - 1) : 2);
        int bestK = -1;
        if (single)
        {
            x265amd_intra_tu_job jobs[10];
            int16_t* lvp[10];
            const uint64_t avail = foldChroma(available(cuX, cuY, size), size >> 2);
            for (int k = 0; k < 5; k++)
                for (int p = 1; p < 3; p++)
                {
                    const int j = k * 2 + p - 1;
                    const int mode = modeList[k] == 36 ? (int)lumaDir : (int)modeList[k];
                    const uint64_t slot = (uint64_t)(uintptr_t)dCand.p + (size_t)j * 2048 * isz;
                    fillJob(jobs[j], p, cuX, cuY, log2 == 3 ? 2 : log2 - 1, mode, 0, nC, slot, nC, j);
                    jobs[j].avail = avail;
                    lvp[j] = clev.data() + (size_t)j * 1024;
                }
            if (runJobs(jobs, 10, cres.data(), lvp, nC * nC, cur.ctx, td)) return err;
            (void)log2C;
        }
        for (int k = 0; k < 5; k++)
        {
            load(cur);
            for (int yy = 0; yy < size; yy += 4) for (int xx = 0; xx < size; xx += 4) U(cuX + xx, cuY + yy).chroma_dir = (uint8_t)modeList[k];
            Cost outCost = { 0, 0, 0, 0 };
            if (single)
            {
                /* codeIntraChromaQt for the one block per plane (and the parent's flag when the block sits one level down, :827-840) */
                for (int p = 1; p < 3; p++)
                {
                    const x265amd_tu_result& r = cres[(size_t)k * 2 + p - 1];
                    memcpy(coeffC[p - 1].data(), clev.data() + (size_t)(k * 2 + p - 1) * 1024, sizeof(int16_t) * nC * nC);
                    setCbf(p, cuX, cuY, size, r.num_sig ? 1 << td : 0);
                    if (td) U(cuX, cuY).cbf[p] |= (uint8_t)((U(cuX, cuY).cbf[p] >> td) & 1);
                    outCost.distortion += (sse_t)r.nz_dist;
                    if (psyRd) outCost.energy += r.nz_energy;
                }
            }
            else if (codeIntraChromaQt(cuX, cuY, 0, outCost)) return err;
            resetBits();
            c->intraDirChroma(U(cuX, cuY));
            codeSubdivCbfQTChroma(cuX, cuY, 0);
            codeCoeffQTChroma(cuX, cuY, 0, 1);
            codeCoeffQTChroma(cuX, cuY, 0, 2);
            const uint32_t b = bits();
            const uint64_t cst = cost(outCost.distortion, b, outCost.energy);
            if (cst < bestCost)
            {
                bestCost = cst; bestDist = outCost.distortion; bestMode = modeList[k];
                /* extractIntraResultChromaQT: levels and reconstruction of this mode */
                coeffCBest[0] = coeffC[0]; coeffCBest[1] = coeffC[1];
                bestK = k;
                if (!single)
                    for (int p = 1; p < 3; p++)
                        copy2D(reconTile + (4096 + (size_t)(p - 1) * 1024) * isz, 32, rec[p] + ((uint64_t)(cuY >> 1) * cstride + (cuX >> 1)) * isz, cstride, size >> 1, size >> 1);
                for (int yy = 0; yy < n4; yy++)
                    for (int xx = 0; xx < n4; xx++)
                    {
                        bestCbf[(size_t)(yy * n4 + xx) * 2] = U(cuX + 4 * xx, cuY + 4 * yy).cbf[1];
                        bestCbf[(size_t)(yy * n4 + xx) * 2 + 1] = U(cuX + 4 * xx, cuY + 4 * yy).cbf[2];
                    }
            }
        }
        for (int yy = 0; yy < n4; yy++)
            for (int xx = 0; xx < n4; xx++)
            {
                x265amd_cu_unit& u = U(cuX + 4 * xx, cuY + 4 * yy);
                u.cbf[1] = bestCbf[(size_t)(yy * n4 + xx) * 2]; u.cbf[2] = bestCbf[(size_t)(yy * n4 + xx) * 2 + 1];
                u.chroma_dir = (uint8_t)bestMode;
            }
        if (single)
        {
            /* the winner's reconstruction is the CU's; the picture keeps the last tried mode's, as after the reference's loop: both planes in one command */
            XaRects r4;
            r4.n = 4;
            for (int p = 1; p < 3; p++)
            {
                const uint64_t bestSlot = (uint64_t)(uintptr_t)dCand.p + (size_t)(bestK * 2 + p - 1) * 2048 * isz;
                const uint64_t lastSlot = (uint64_t)(uintptr_t)dCand.p + (size_t)(4 * 2 + p - 1) * 2048 * isz;
                const int a = 2 * (p - 1);
                r4.dst[a] = reconTile + (4096 + (size_t)(p - 1) * 1024) * isz; r4.src[a] = bestSlot; r4.dst_stride[a] = 32; r4.src_stride[a] = (int16_t)nC;
                r4.dst[a + 1] = rec[p] + ((uint64_t)(cuY >> 1) * cstride + (cuX >> 1)) * isz; r4.src[a + 1] = lastSlot; r4.dst_stride[a + 1] = (int16_t)cstride; r4.src_stride[a + 1] = (int16_t)nC;
                r4.w[a] = r4.w[a + 1] = r4.h[a] = r4.h[a + 1] = (int16_t)nC;
            }
            xa_copy_rects(st, r4);
        }
        totalDistortion = bestDist;
        load(cur);
        return 0;
    }
};

IntraRd::~IntraRd() { chain_block_put(qBlock); }

} // namespace

/* shared body: kind 0 = checkIntraInInter + encodeIntraInInter, kind 1 = checkIntra(part_size) */
/* ws (optional): where the caller keeps this routine's working set between calls -- the CUs of one CTU on one stream / queue use the same buffers
 * one after the other (xa_intra_ws_free releases it) */
static int intra_cu_impl(int kind, int partSize, void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                         const uint64_t* h_src, const uint64_t* h_rec, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu,
                         x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out, int16_t* coeff_out, uint64_t* info, void** ws = nullptr)
{
    if (!si || !rp || !units || !h_src || !h_rec || !cu || (kind < 2 && (!cu_units || !out)) || !d_pred || !d_recon) return xa_fail(X265AMD_EINVAL, "intra rd: null argument");
    if (si->tq_bypass_enabled) return xa_fail(X265AMD_EINVAL, "intra rd: lossless coding is not supported");
    if (partSize != 0 && (partSize != 3 || cu->log2_size != 3 || si->tu_log2_min > 2)) return xa_fail(X265AMD_EINVAL, "intra rd: NxN only for 8x8 CUs with 4x4 transforms");
    xa_phase(XA_PH_ANALYZER);
    IntraRd* ip = ws && *ws ? static_cast<IntraRd*>(*ws) : new IntraRd;
    if (ws) *ws = ip;
    IntraRd& R = *ip;
    R.helper = xa_queue_helper(stream);
    R.big[0].q = R.helper ? xa_queue_helper(R.helper) : nullptr;          /* the third queue rides on the second, the fourth on the third */
    R.big[1].q = R.big[0].q ? xa_queue_helper(R.big[0].q) : nullptr;
    if (R.ahead.on && !(kind == 1 && partSize == 3 && R.ahead.x == cu->x && R.ahead.y == cu->y))
    {
        /* a command started ahead that nobody came for: let it finish before anything else touches what it writes */
        R.ahead.on = false;
        if (xa_stream_sync(R.helper) != hipSuccess || xa_stream_fence(stream, XA_CMD_ACQUIRE) != hipSuccess) return xa_fail(X265AMD_EHIP, "intra rd: second queue");
    }
    if (kind != 1 || partSize != 0 || cu->log2_size != 3) R.hintPred = R.hintRecon = 0;
    R.haveWhole = false; R.haveNxn = false; R.haveDevChroma = false; R.lumaTileDone = false;
    R.st = (hipStream_t)stream; R.si = si; R.rp = rp; R.units = units; R.w4 = si->pic_width >> 2; R.src = h_src; R.rec = h_rec; R.stride = stride; R.cstride = cstride;
    R.cuX = cu->x; R.cuY = cu->y; R.log2 = cu->log2_size; R.size = 1 << R.log2; R.depth = 6 - R.log2; R.qp = cu->qp; R.err = 0;
    R.predTile = d_pred; R.reconTile = d_recon;
    R.pre.on = false;
    int rc = X265AMD_OK;
    if (R.log2 < 3 || R.log2 > 5 || (R.cuX & (R.size - 1)) || (R.cuY & (R.size - 1)) || R.cuX < 0 || R.cuY < 0 || R.cuX + R.size > si->pic_width || R.cuY + R.size > si->pic_height)
        rc = xa_fail(X265AMD_EINVAL, "intra rd: CU outside the picture, misaligned, or not 8..32");
    static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                             32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
    const int bd = 6 * (X265AMD_DEPTH - 8);
    const int qpQuant = R.qp < 0 ? 0 : (R.qp > 51 ? 51 : R.qp);
    int qpC = qpQuant > 57 ? 57 : qpQuant;
    if (qpC >= 30) qpC = chromaScale[qpC];
    R.qpLumaScaled = qpQuant + bd; R.qpChromaScaled = qpC + bd;
    /* CUData::getIntraTUQtDepthRange (cudata.cpp:972-981) */
    {
        /* unsigned in the reference: an 8x8 NxN CU with tu-intra-depth 4 wraps below zero and clips to the MAXIMUM size (no transform splits) */
        const uint32_t lo = (uint32_t)R.log2 - (uint32_t)(si->tu_max_depth_intra - 1 + (partSize != 0));
        R.range[0] = lo < (uint32_t)si->tu_log2_min ? si->tu_log2_min : (lo > (uint32_t)si->tu_log2_max ? si->tu_log2_max : (int)lo);
        R.range[1] = si->tu_log2_max;
    }
    if (rc == X265AMD_OK && !R.dJobs.p &&
        (R.dJobs.alloc(sizeof(x265amd_intra_tu_job) * IntraRd::MAX_JOBS) != hipSuccess || R.dRes.alloc(sizeof(x265amd_tu_result) * IntraRd::MAX_JOBS) != hipSuccess ||
         R.dCoeff.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2) != hipSuccess || R.dResi.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2) != hipSuccess ||
         R.dCand.alloc((size_t)IntraRd::MAX_JOBS * 2048 * sizeof(pixel)) != hipSuccess || R.dLayer.alloc((size_t)4 * 4096 * sizeof(pixel)) != hipSuccess ||
         R.dScan.alloc(35 * 4) != hipSuccess || R.dScanJob.alloc(sizeof(x265amd_intra_job)) != hipSuccess ||
         R.dPuJob.alloc(sizeof(x265amd_intra_pu_job)) != hipSuccess || R.dPuOut.alloc(sizeof(x265amd_intra_pu_out)) != hipSuccess ||
         R.dNxnJob.alloc(sizeof(x265amd_intra_nxn_job)) != hipSuccess || R.dNxnOut.alloc(sizeof(x265amd_intra_nxn_out)) != hipSuccess ||
         R.dDevLevels.alloc(1024 * 2) != hipSuccess || R.dDevCLevels.alloc(2 * 256 * 2) != hipSuccess ||
         R.dCoeffDev.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2 * 2) != hipSuccess ||
         R.mCtx.alloc(X265AMD_CTX_STRIDE) != hipSuccess || R.mEstJob.alloc(sizeof(x265amd_est_job)) != hipSuccess ||
         R.mRdoq.alloc(sizeof(x265amd_tu_rdoq) * IntraRd::MAX_JOBS) != hipSuccess || R.dEst.alloc(sizeof(x265amd_est_bits)) != hipSuccess))
    {
        rc = xa_fail(X265AMD_EHIP, "intra rd: out of device memory");
        R.dJobs.free();         /* a partly made working set is made again next time */
    }
    if (rc == X265AMD_OK && R.helper && !R.dCand2.p &&
        (R.dCand2.alloc((size_t)IntraRd::MAX_JOBS * 2048 * sizeof(pixel)) != hipSuccess || R.dCoeffDev2.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2 * 2) != hipSuccess ||
         R.dNxnJob2.alloc(sizeof(x265amd_intra_nxn_job)) != hipSuccess || R.dNxnOut2.alloc(sizeof(x265amd_intra_nxn_out)) != hipSuccess))
        rc = xa_fail(X265AMD_EHIP, "intra rd: out of device memory");
    for (int b = 0; b < 2 && rc == X265AMD_OK; b++)
    {
        IntraRd::Big& g = R.big[b];
        if (g.q && !g.cand.p &&
            (g.cand.alloc((size_t)IntraRd::MAX_JOBS * 2048 * sizeof(pixel)) != hipSuccess || g.coeffDev.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2 * 2) != hipSuccess ||
             g.job.alloc(sizeof(x265amd_intra_nxn_job)) != hipSuccess || g.out.alloc(sizeof(x265amd_intra_nxn_out)) != hipSuccess ||
             g.levels.alloc(1024 * 2) != hipSuccess || g.clevels.alloc(2 * 256 * 2) != hipSuccess))
            rc = xa_fail(X265AMD_EHIP, "intra rd: out of device memory");
    }
    if (rc == X265AMD_OK && rp->rdoq_level && xa_fill_async(R.st, R.dEst.p, 0, sizeof(x265amd_est_bits)) != hipSuccess)         /* the table is only read by RDOQ */
        rc = xa_fail(X265AMD_EHIP, "intra rd: fill");
    x265amd_cabac* coder = rc == X265AMD_OK ? x265amd_cabac_open(si, units, 1) : nullptr;
    if (coder) coder->ctuInProgress = true;          /* the analysis of a CTU asks (cabac_coder.h: lastQP) */
    if (rc == X265AMD_OK && !coder) rc = xa_fail(X265AMD_EINVAL, "intra rd: slice description");
    if (rc != X265AMD_OK) { if (coder) x265amd_cabac_close(coder); if (!ws) delete ip; return rc; }
    R.c = coder;
    for (int l = 0; l < 4; l++) R.coeffL[l].assign(4096, 0);
    for (int p = 0; p < 2; p++) { R.coeffC[p].assign(1024, 0); R.coeffCBest[p].assign(1024, 0); }
    uint64_t rd[6];
    x265amd_rdcost(cu->reserved[0] ? (int)cu->reserved[0] : R.qp, si->slice_type, rp->psy_rd, 0, 0, 0, rd);          /* (the lambdas' QP: x265amd_rd_cu.reserved[0] above 51) */
    R.lambda2 = rd[0]; R.lambda = rd[1]; R.psyRd = (uint32_t)rd[2];
    memset(&R.cur, 0, sizeof(R.cur));
    memcpy(R.cur.ctx, cu->ctx, X265AMD_CTX_COUNT);
    R.cur.frac = cu->frac_bits;
    const int w4 = R.w4, u4 = R.size >> 2;
    std::vector<x265amd_cu_unit> saved((size_t)u4 * u4);
    for (int yy = 0; yy < u4; yy++) memcpy(&saved[(size_t)yy * u4], &units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2)], sizeof(x265amd_cu_unit) * u4);
    const size_t isz = sizeof(pixel);
    /* the CU as initSubCU + setPartSizeSubParts + setPredModeSubParts leave it */
    for (int yy = 0; yy < u4; yy++)
        for (int xx = 0; xx < u4; xx++)
        {
            x265amd_cu_unit& u = units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2) + xx];
            memset(&u, 0, sizeof(u));
            u.depth = (uint8_t)R.depth; u.pred_mode = X265AMD_MODE_INTRA; u.part_size = (uint8_t)partSize; u.luma_dir = 1; u.chroma_dir = 36; u.qp = (int8_t)R.qp;
            u.ref_idx[0] = u.ref_idx[1] = -1;
        }
    if (kind == 3)
    {
        int started = 0;
        /* on unless switched off (X265AMD_AHEAD_INTER=0).  Round 3 measured it 1-4 % slower (what the try hides, 40 us of device time per CU that gets to it, was eaten by
         * what starting it costs every unskipped CU); since round 4 the CUs that end as skips never get here -- the device's skip chain ends them (inter_chain_dev.h) -- and
         * a CU that does is searched, predicted and coded while its intra try runs beside: 35.8 -> 40.9 frames/s on the bench clip */
        static const bool aheadInter = !(getenv("X265AMD_AHEAD_INTER") && atoi(getenv("X265AMD_AHEAD_INTER")) == 0);
        IntraRd::Big& g = R.inter[R.log2 >= 3 && R.log2 <= 5 ? 5 - R.log2 : 0];
        if (aheadInter && R.log2 >= 3 && R.log2 <= 5 && R.devIntraInInter() && xa_is_queue(R.st))
        {
            /* the queue of this depth: taken when the first CU of the CTU wants it (never waiting for one), given back with the CTU's working set */
            if (!g.q) { g.q = xa_queue_try_acquire(); g.owned = g.q != nullptr; }
            if (g.q && !g.cand.p &&
                (g.cand.alloc((size_t)IntraRd::MAX_JOBS * 2048 * sizeof(pixel)) != hipSuccess || g.coeffDev.alloc((size_t)IntraRd::MAX_JOBS * 1024 * 2 * 2) != hipSuccess ||
                 g.job.alloc(sizeof(x265amd_intra_nxn_job)) != hipSuccess || g.out.alloc(sizeof(x265amd_intra_nxn_out)) != hipSuccess ||
                 g.levels.alloc(1024 * 2) != hipSuccess || g.clevels.alloc(2 * 256 * 2) != hipSuccess))
                rc = xa_fail(X265AMD_EHIP, "intra rd: out of device memory");
        }
        if (rc == X265AMD_OK && aheadInter && R.log2 >= 3 && R.log2 <= 5 && g.q && g.cand.p && R.devIntraInInter())
        {
            if (g.on) { g.on = false; if (xa_stream_sync(g.q) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "intra rd: second queue"); }      /* an earlier try nobody collected */
            if (rc == X265AMD_OK)
            {
                x265amd_intra_nxn_job nj;
                R.buildInInterJob(nj, d_pred, d_recon, (uint64_t)(uintptr_t)g.cand.p, (uint64_t)(uintptr_t)g.coeffDev.p, g.levels.p, g.clevels.p);
                memcpy(g.job.p, &nj, sizeof(nj));
                if (xa_follow_or_sync(g.q, R.st) != hipSuccess ||
                    x265amd_intra_nxn(g.q, (const x265amd_intra_nxn_job*)g.job.p, (x265amd_intra_nxn_out*)g.out.p) != X265AMD_OK)
                    rc = xa_fail(X265AMD_EHIP, "intra rd: intra try ahead");
                else { g.on = true; g.x = R.cuX; g.y = R.cuY; started = 1; }
            }
        }
        for (int yy = 0; yy < u4; yy++) memcpy(&units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2)], &saved[(size_t)yy * u4], sizeof(x265amd_cu_unit) * u4);
        x265amd_cabac_close(coder);
        return rc != X265AMD_OK ? rc : (started ? 1 : 0);
    }
    if (kind == 2)
    {
        /* only start the CU's 2Nx2N command (a 16x16 CU of an I picture, the third queue): the caller recurses into the sub-CUs and comes back with the ordinary call */
        static const bool dev16 = !(getenv("X265AMD_DEVICE_16") && atoi(getenv("X265AMD_DEVICE_16")) == 0);
        static const bool ahead16 = !(getenv("X265AMD_AHEAD_16") && atoi(getenv("X265AMD_AHEAD_16")) == 0);
        static const bool dev32 = !(getenv("X265AMD_DEVICE_32") && atoi(getenv("X265AMD_DEVICE_32")) == 0);
        int started = 0;
        IntraRd::Big& g = R.big[R.log2 == 5 ? 1 : 0];
        static const bool devRdoq = !(getenv("X265AMD_DEVICE_RDOQ") && atoi(getenv("X265AMD_DEVICE_RDOQ")) == 0);
        if (ahead16 && ((R.log2 == 4 && dev16) || (R.log2 == 5 && dev32)) && g.q && g.cand.p && !g.on && (!rp->rdoq_level || devRdoq) && R.range[0] == R.log2 && R.range[1] >= R.log2 &&
            2 + rp->rd_level + (R.depth >> 1) <= IntraRd::MAX_JOBS)
        {
            x265amd_intra_nxn_job nj;
            R.buildDevJob(nj, 0, rp->rd_level, d_pred, d_recon, (uint64_t)(uintptr_t)g.cand.p, (uint64_t)(uintptr_t)g.coeffDev.p, g.levels.p, g.clevels.p);
            nj.no_picture = 1;
            memcpy(g.job.p, &nj, sizeof(nj));
            if (xa_follow_or_sync(g.q, R.st) != hipSuccess ||
                x265amd_intra_nxn(g.q, (const x265amd_intra_nxn_job*)g.job.p, (x265amd_intra_nxn_out*)g.out.p) != X265AMD_OK)
                rc = xa_fail(X265AMD_EHIP, "intra rd: large unit step ahead");
            else { g.on = true; g.x = R.cuX; g.y = R.cuY; started = 1; }
        }
        for (int yy = 0; yy < u4; yy++) memcpy(&units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2)], &saved[(size_t)yy * u4], sizeof(x265amd_cu_unit) * u4);
        x265amd_cabac_close(coder);
        return rc != X265AMD_OK ? rc : (started ? 1 : 0);
    }
    Cost icosts = { 0, 0, 0, 0 };
    sse_t lumaDist = 0, chromaDist = 0;
    xa_phase(XA_PH_INTRA_SETUP);
    if (kind == 0 && R.devIntraInInter())
    {
        /* ---- checkIntraInInter + encodeIntraInInter as one command, the mode picked on the device; started ahead at the CU's entry when a queue was to spare ---- */
        IntraRd::Big& g = R.inter[5 - R.log2];
        x265amd_intra_nxn_out nxn;
        const bool mineAhead = g.on && g.x == R.cuX && g.y == R.cuY;
        R.curLevels = (const int16_t*)R.dDevLevels.p; R.curCLevels = (const int16_t*)R.dDevCLevels.p;
        if (mineAhead)
        {
            g.on = false;
            if (xa_stream_sync(g.q) != hipSuccess || xa_stream_fence(R.st, XA_CMD_ACQUIRE) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "intra rd: intra try");
            memcpy(&nxn, g.out.p, sizeof(nxn));
            R.curLevels = (const int16_t*)g.levels.p; R.curCLevels = (const int16_t*)g.clevels.p;
        }
        else
        {
            x265amd_intra_nxn_job nj;
            R.buildInInterJob(nj, d_pred, d_recon, (uint64_t)(uintptr_t)R.dCand.p, (uint64_t)(uintptr_t)R.dCoeffDev.p, nullptr, nullptr);
            memcpy(R.dNxnJob.p, &nj, sizeof(nj));
            if (x265amd_intra_nxn(R.st, (const x265amd_intra_nxn_job*)R.dNxnJob.p, (x265amd_intra_nxn_out*)R.dNxnOut.p) != X265AMD_OK || xa_stream_sync(R.st) != hipSuccess)
                rc = xa_fail(X265AMD_EHIP, "intra rd: intra try");
            memcpy(&nxn, R.dNxnOut.p, sizeof(nxn));
        }
        if (rc == X265AMD_OK)
        {
            const uint32_t bmode = nxn.mode[0];
            if (bmode > 34) rc = xa_fail(X265AMD_EHIP, "intra rd: intra try mode");
            else
            {
                /* the mode's bits and cost as the scan prices them (loadIntraDirModeLuma + bitsIntraModeMPM / NonMPM) */
                uint32_t preds[3];
                R.c->lumaPreds(R.cuX, R.cuY, preds);
                const uint64_t frac = R.cur.frac & 32767;
                const uint8_t adi = R.cur.ctx[C_ADI];
                uint32_t b = (uint32_t)((frac + k_bits[adi ^ 0]) >> 15) + 5;
                const uint32_t mpmBase = (uint32_t)((frac + k_bits[adi ^ 1]) >> 15);
                for (int i = 0; i < 3; i++) if (preds[i] == bmode) { b = mpmBase + (bmode == preds[0] ? 1u : 2u); break; }
                const uint32_t sad = nxn.chroma_reserved;
                if (info) { info[0] = bmode; info[1] = R.calcRdSADCost(sad, b); info[2] = b; info[3] = sad; }
                for (int yy = 0; yy < R.size; yy += 4) for (int xx = 0; xx < R.size; xx += 4) R.U(R.cuX + xx, R.cuY + yy).luma_dir = (uint8_t)bmode;
                R.load(R.cur);
                R.pre = IntraRd::Pre{ true, R.cuX, R.cuY, R.log2, nxn.res[0], R.log2 > 3 ? R.curLevels : &nxn.levels[0][0], 0, 0, false };
                rc = R.codeIntraLumaQT(R.cuX, R.cuY, 0, false, icosts) ? R.err : X265AMD_OK;
                R.pre.on = false;
                lumaDist = icosts.distortion;
                R.haveDevChroma = true; R.lumaTileDone = true; R.nxnChroma = nxn;
            }
        }
    }
    else if (kind == 0)
    {
        /* ---- checkIntraInInter: DC first, then planar, then the angular modes, strict improvement ---- */
        uint64_t modeCosts[35]; uint32_t mb[35], ms[35];
        rc = R.lumaScan(R.cuX, R.cuY, R.log2, modeCosts, mb, ms);
        if (rc == X265AMD_OK)
        {
            static const uint8_t order[35] = { 1, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34 };
            uint32_t bmode = 1;
            if (rp->fast_intra)
            {
                /* the best angle sampled at distance 5, refined at distance 2 and 1 (search.cpp:1401-1434); DC / planar first as always */
                if (modeCosts[0] < modeCosts[bmode]) bmode = 0;
                uint32_t amode = 5;
                uint64_t acost = kMaxCost;
                for (uint32_t mode = 5; mode < 35; mode += 5) if (modeCosts[mode] < acost) { acost = modeCosts[mode]; amode = mode; }
                for (uint32_t dist = 2; dist >= 1; dist--)
                {
                    const uint32_t lowmode = amode - dist, highmode = amode + dist;
                    if (modeCosts[lowmode] < acost) { acost = modeCosts[lowmode]; amode = lowmode; }
                    if (modeCosts[highmode] < acost) { acost = modeCosts[highmode]; amode = highmode; }
                }
                if (amode == 33 && modeCosts[34] < acost) { acost = modeCosts[34]; amode = 34; }
                if (acost < modeCosts[bmode]) bmode = amode;
            }
            else
                for (int k = 1; k < 35; k++) if (modeCosts[order[k]] < modeCosts[bmode]) bmode = order[k];
            if (info) { info[0] = bmode; info[1] = modeCosts[bmode]; info[2] = mb[bmode]; info[3] = ms[bmode]; }
            for (int yy = 0; yy < R.size; yy += 4) for (int xx = 0; xx < R.size; xx += 4) R.U(R.cuX + xx, R.cuY + yy).luma_dir = (uint8_t)bmode;
            /* ---- encodeIntraInInter ---- */
            R.load(R.cur);
            rc = R.codeIntraLumaQT(R.cuX, R.cuY, 0, false, icosts);
            lumaDist = icosts.distortion;
        }
    }
    else
        rc = R.estIntraPredQT(partSize, rp->rd_level, lumaDist);
    xa_phase(XA_PH_INTRA_CAND);
    std::vector<int16_t> coeffCu(4096 + 2048, 0);
    if (rc == X265AMD_OK)
    {
        R.extractLuma(R.cuX, R.cuY, 0, coeffCu.data());
        if (!R.lumaTileDone) R.copy2D(d_recon, 64, h_rec[0] + ((uint64_t)R.cuY * stride + R.cuX) * isz, stride, R.size, R.size);
        rc = R.estIntraPredChromaQT(chromaDist);
    }
    xa_phase(XA_PH_INTRA_CHROMA);
    if (rc == X265AMD_OK)
    {
        memcpy(coeffCu.data() + 4096, R.coeffCBest[0].data(), sizeof(int16_t) * 1024);
        memcpy(coeffCu.data() + 5120, R.coeffCBest[1].data(), sizeof(int16_t) * 1024);
        x265amd_cu_unit& u0 = units[(R.cuY >> 2) * w4 + (R.cuX >> 2)];
        R.resetBits();
        uint32_t skipFlagBits = 0;
        if (si->slice_type != 2)
        {
            const x265amd_cu_unit* l = coder->at((R.cuX >> 2) - 1, R.cuY >> 2);
            const x265amd_cu_unit* a = coder->at(R.cuX >> 2, (R.cuY >> 2) - 1);
            const int skipCtx = (x265amd_cabac::coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (x265amd_cabac::coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
            coder->bin(0, C_SKIP + skipCtx);
            skipFlagBits = R.bits();
            coder->bin(1, C_PRED_MODE);
        }
        coder->partSize(u0, R.depth, R.size);
        coder->predInfo(R.cuX, R.cuY, R.size, u0);
        const uint32_t mvBits = R.bits() - skipFlagBits;
        bool dqp = si->use_dqp != 0;
        coder->coeffCtu[0] = coeffCu.data(); coder->coeffCtu[1] = coeffCu.data() + 4096; coder->coeffCtu[2] = coeffCu.data() + 5120;
        coder->ctuX0 = R.cuX; coder->ctuY0 = R.cuY;
        coder->transform(R.cuX, R.cuY, R.cuX, R.cuY, 0, R.log2, dqp, R.range);
        memset(out, 0, sizeof(*out));
        out->total_bits = R.bits(); out->mv_bits = mvBits; out->coeff_bits = out->total_bits - mvBits - skipFlagBits;
        out->luma_distortion = (uint32_t)lumaDist; out->chroma_distortion = (uint32_t)chromaDist;
        const sse_t distortion = lumaDist + chromaDist;
        out->distortion = distortion;
        /* psy energy of the reconstruction, residual energy of the prediction (luma) */
        x265amd_rd_cu mc2[2] = { *cu, *cu };
        x265amd_cu_measure m2[2];
        if (R.haveNxn && partSize != 0)
        {
            out->psy_energy = R.psyRd ? R.nxnPsy : 0;
            out->res_energy = R.nxnRes;
        }
        else if (R.haveWhole && partSize == 0)
        {
            /* one transform unit = the CU: psyCost(fenc, recon) and sse(fenc, pred) of the luma block came with the unit's result (rdcost.h:114-117,
             * search.cpp:1279-1283 / :1486-1503 measure exactly these two) */
            out->psy_energy = R.psyRd ? R.wholeRes.nz_energy : 0;
            out->res_energy = (uint32_t)(sse_t)R.wholeRes.zero_dist;
        }
        else
        {
            /* reconstruction and prediction tile in one launch */
            const uint64_t both[2] = { d_recon, d_pred };
            if (x265amd_measure_tile_list(stream, h_src, stride, cstride, mc2, 2, both, m2) != X265AMD_OK) rc = X265AMD_EHIP;
            out->psy_energy = R.psyRd ? m2[0].psy : 0;
            out->res_energy = (uint32_t)(sse_t)m2[1].sse[0];
        }
        out->rd_cost = R.cost(distortion, out->total_bits, out->psy_energy);
        memcpy(out->ctx, coder->ctx, X265AMD_CTX_COUNT);            /* Mode::contexts is stored before checkDQP codes into it */
        out->frac_bits = coder->fracBits;
        /* checkDQP (search.cpp:3974-4003) */
        if (si->use_dqp && R.depth <= si->max_cu_dqp_depth)
        {
            const bool rootCbf = u0.cbf[0] || u0.cbf[1] || u0.cbf[2];
            if (rootCbf)
            {
                if (rp->rd_level >= 3) { R.resetBits(); coder->deltaQP(R.cuX, R.cuY); out->total_bits += R.bits(); memcpy(out->ctx, coder->ctx, X265AMD_CTX_COUNT); out->frac_bits = coder->fracBits; }
                else if (rp->rd_level == 2) out->total_bits++;
                out->rd_cost = R.cost(distortion, out->total_bits, out->psy_energy);
            }
            else
            {
                const int8_t q = (int8_t)coder->refQP(R.cuX, R.cuY);
                for (int yy = 0; yy < u4; yy++) for (int xx = 0; xx < u4; xx++) units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2) + xx].qp = q;
            }
        }
        if (coeff_out) memcpy(coeff_out, coeffCu.data(), sizeof(int16_t) * (4096 + 2048));
    }
    for (int yy = 0; yy < u4; yy++)
    {
        memcpy(&cu_units[(size_t)yy * u4], &units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2)], sizeof(x265amd_cu_unit) * u4);
        memcpy(&units[((R.cuY >> 2) + yy) * w4 + (R.cuX >> 2)], &saved[(size_t)yy * u4], sizeof(x265amd_cu_unit) * u4);
    }
    /* What is still pending are block copies into tiles and the picture; the commands of a stream / queue run in order, so whoever reads them next waits for
     * them anyway.  A caller that keeps the working set (same stream, next CU) goes on at once; the plain entry points return with everything done. */
    if (rc == X265AMD_OK && !ws && xa_stream_sync(R.st) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "intra rd: synchronize");
    x265amd_cabac_close(coder);
    if (!ws) delete ip;
    xa_phase(XA_PH_INTRA_FINAL);
    return rc;
}

/* The four 8x8 CUs of the 16x16 block at (x, y) of an I picture, decided on the device one after the other without the host between them (Analysis::compressIntraCU's
 * loop over the sub-CUs at the last depth, analysis.cpp:596-640, with checkIntra 2Nx2N and NxN per CU, search.cpp:1236-1287).  ctx / frac: the contexts the first CU
 * starts from; split_recon: the parent's reconstruction tile (the winners' samples go there and into the picture); tiles2: prediction / reconstruction tiles for the
 * 2Nx2N evaluations.  results[4] in CU order.  Returns 0, 1 when the chain does not apply here (the caller takes the CUs one by one), or an error code. */
int xa_intra_quad8_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                      intptr_t stride, intptr_t cstride, int x, int y, int qp, const uint8_t* ctx, uint64_t frac, uint64_t split_recon, const uint64_t tilesN[2],
                      const uint64_t tiles2[2], x265amd_intra_cu8_result* results, void** ws, void (*between)(void*), void* between_ctx, int lambda_qp)
{
    static const bool on = !(getenv("X265AMD_INTRA_CHAIN") && atoi(getenv("X265AMD_INTRA_CHAIN")) == 0);
    if (!on || !ws || !*ws || !xa_is_queue(stream)) return 1;
    IntraRd& R = *static_cast<IntraRd*>(*ws);
    void* helper = xa_queue_helper(stream);
    if (!helper || !R.dCand2.p || !R.dLayer.p || !R.dCand.p || R.ahead.on) return 1;
    /* with RDOQ (round 5) the deciding command runs the general form of the evaluation and its own decision (intra_pu_dev.h); X265AMD_CHAIN_RDOQ=0: CU by CU */
    static const bool chainRdoq = !(getenv("X265AMD_DEVICE_RDOQ") && atoi(getenv("X265AMD_DEVICE_RDOQ")) == 0) && !(getenv("X265AMD_CHAIN_RDOQ") && atoi(getenv("X265AMD_CHAIN_RDOQ")) == 0);
    /* delta QP (round 6): the CUs of the chain lie below the quantisation groups' depth, so no QP changes inside it -- but Search::checkIntra counts cu_qp_delta with
     * the coefficients of every CU that has any (codeCoeff with bCodeDQP, search.cpp:1266-1268): the deciding command adds those bins (nxn4_decide), the value is the
     * group's, known here */
    if (si->slice_type != 2 || (si->use_dqp && si->max_cu_dqp_depth > 1) || si->tq_bypass_enabled || (rp->rdoq_level && !chainRdoq) || si->tu_max_depth_intra != 1 || si->tu_log2_min != 2 || si->tu_log2_max < 3 || si->max_cu_depth != 3 ||
        2 + rp->rd_level + 2 > IntraRd::MAX_JOBS || x + 16 > si->pic_width || y + 16 > si->pic_height || (x & 15) || (y & 15))
        return 1;
    if (!R.qJobs.p)
    {
        if (R.qJobs.alloc(8 * sizeof(x265amd_intra_nxn_job)) != hipSuccess || R.qOut.alloc(4 * sizeof(x265amd_intra_cu8_result)) != hipSuccess || !(R.qBlock = chain_block_get()))
            return xa_fail(X265AMD_EHIP, "intra rd: out of device memory");
    }
    XA_HOSTPROF("quad8 (all but the wait)");
    R.st = (hipStream_t)stream; R.si = si; R.rp = rp; R.units = units; R.w4 = si->pic_width >> 2; R.src = h_src; R.rec = h_rec; R.stride = stride; R.cstride = cstride;
    R.log2 = 3; R.size = 8; R.depth = 3; R.qp = qp; R.err = 0; R.helper = helper;
    {
        static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                                 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
        const int bd = 6 * (X265AMD_DEPTH - 8);
        const int qpQuant = qp < 0 ? 0 : (qp > 51 ? 51 : qp);
        int qpC = qpQuant > 57 ? 57 : qpQuant;
        if (qpC >= 30) qpC = chromaScale[qpC];
        R.qpLumaScaled = qpQuant + bd; R.qpChromaScaled = qpC + bd;
    }
    uint64_t rd[6];
    x265amd_rdcost(lambda_qp > 0 ? lambda_qp : qp, si->slice_type, rp->psy_rd, 0, 0, 0, rd);
    R.lambda2 = rd[0]; R.lambda = rd[1]; R.psyRd = (uint32_t)rd[2];
    memset(&R.cur, 0, sizeof(R.cur));
    memcpy(R.cur.ctx, ctx, X265AMD_CTX_COUNT);
    R.cur.frac = frac;
    x265amd_cabac* coder;
    { XA_HOSTPROF("quad8 cabac_open"); coder = x265amd_cabac_open(si, units, 1); }
    if (coder) coder->ctuInProgress = true;          /* the analysis of a CTU asks (cabac_coder.h: lastQP) */
    if (!coder) return xa_fail(X265AMD_EINVAL, "intra rd: slice description");
    R.c = coder;
    x265amd_intra_nxn_job* jobs = static_cast<x265amd_intra_nxn_job*>(R.qJobs.p);
    x265amd_intra_cu8_result* outs = static_cast<x265amd_intra_cu8_result*>(R.qOut.p);
    x265amd_intra_peer* peers = static_cast<x265amd_intra_peer*>(R.qBlock);
    x265amd_intra_chain* chainRec = reinterpret_cast<x265amd_intra_chain*>(peers + 4);
    const size_t isz = sizeof(pixel);
    /* The chain's counts never repeat, in any chain of the process: records come from the pools, and a workgroup that looks at one may still hold (in its L2) what an
     * earlier chain left there -- an old count is smaller than every new one and only makes the reader look again. */
    static std::atomic<uint64_t> tokens{ 1 };
    const uint64_t token0 = tokens.fetch_add(4);
    for (int i = 0; i < 4; i++)
    {
        R.cuX = x + 8 * (i & 1); R.cuY = y + 8 * (i >> 1);
        /* the CU as initSubCU + setPredModeSubParts leave it (the caller writes the decided CU over it afterwards) */
        for (int k = 0; k < 4; k++)
        {
            x265amd_cu_unit& u = R.U(R.cuX + 4 * (k & 1), R.cuY + 4 * (k >> 1));
            memset(&u, 0, sizeof(u));
            u.depth = 3; u.pred_mode = X265AMD_MODE_INTRA; u.luma_dir = 1; u.chroma_dir = 36; u.qp = (int8_t)qp; u.ref_idx[0] = u.ref_idx[1] = -1;
        }
        x265amd_cu_unit& u0 = R.U(R.cuX, R.cuY);
        const uint8_t keepPart = u0.part_size;
        for (int role = 1; role <= 2; role++)
        {
            x265amd_intra_nxn_job nj;
            const int partSize = role == 1 ? 3 : 0;
            R.range[0] = partSize ? 2 : 3; R.range[1] = si->tu_log2_max;
            u0.part_size = (uint8_t)partSize;
            {
            XA_HOSTPROF("quad8 buildDevJob");
            if (role == 1) R.buildDevJob(nj, 3, rp->rd_level, tilesN[0], tilesN[1], (uint64_t)(uintptr_t)R.dCand.p, (uint64_t)(uintptr_t)R.dCoeffDev.p);
            else R.buildDevJob(nj, 0, rp->rd_level, tiles2[0], tiles2[1], (uint64_t)(uintptr_t)R.dCand2.p, (uint64_t)(uintptr_t)R.dCoeffDev2.p);
            }
            nj.no_picture = role == 2;
            nj.chain = (uint64_t)(uintptr_t)chainRec; nj.peer = (uint64_t)(uintptr_t)&peers[i]; nj.cu_out = (uint64_t)(uintptr_t)&outs[i];
            nj.chain_token = token0 + i; nj.chain_role = (uint8_t)role; nj.chain_first = i == 0; nj.chain_index = (uint8_t)i;
            if (si->use_dqp)
            {
                /* Entropy::codeDeltaQP's value for the CUs of this block (entropy.cpp:1737-1756): the group's QP against its prediction, wrapped */
                const int bd = 6 * (X265AMD_DEPTH - 8);
                int dqp = qp - coder->refQP(x, y);
                dqp = (dqp + 78 + bd + (bd / 2)) % (52 + bd) - 26 - (bd / 2);
                nj.reserved[0] = 1; nj.reserved[1] = (uint8_t)(int8_t)dqp;
            }
            /* the neighbour modes inside the block are the chain's: left of units 0 / 2 = units 1 / 3 of the CU to the left, above of units 0 / 1 = units 2 / 3 of the CU above */
            nj.mode_src[0] = (i & 1) ? (uint8_t)(((i - 1) << 2) | 1) : 0xFF; nj.mode_src[1] = (i & 1) ? (uint8_t)(((i - 1) << 2) | 3) : 0xFF;
            nj.mode_src[2] = (i & 2) ? (uint8_t)(((i - 2) << 2) | 2) : 0xFF; nj.mode_src[3] = (i & 2) ? (uint8_t)(((i - 2) << 2) | 3) : 0xFF;
            if (role == 1)
            {
                nj.peer_recon[0] = tiles2[1];
                nj.win_dst[0] = split_recon + ((size_t)(8 * (i >> 1)) * 64 + 8 * (i & 1)) * isz;
                for (int pl = 0; pl < 2; pl++)
                {
                    nj.peer_recon[1 + pl] = tiles2[1] + (4096 + (size_t)pl * 1024) * isz;
                    nj.win_dst[1 + pl] = split_recon + (4096 + (size_t)pl * 1024 + (size_t)(4 * (i >> 1)) * 32 + 4 * (i & 1)) * isz;
                }
            }
            { XA_HOSTPROF("quad8 record push"); memcpy(&jobs[2 * i + (role - 1)], &nj, sizeof(nj)); }
        }
        u0.part_size = keepPart;
        outs[i].status = 0;
    }
    x265amd_cabac_close(coder);
    R.c = nullptr;
    /* what this queue has written is out before the other workgroup looks (a signalling command releases), and that one looks (acquire) */
    int rc = X265AMD_OK;
    if (xa_follow_or_sync(helper, stream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "intra rd: chain fence");
    /* one command per role: the four records of a role run one after the other in the same workgroup */
    if (rc == X265AMD_OK && (x265amd_intra_nxn_list(helper, &jobs[1], 4, 2 * sizeof(x265amd_intra_nxn_job), &peers[0].out) != X265AMD_OK ||
                             x265amd_intra_nxn_list(stream, &jobs[0], 4, 2 * sizeof(x265amd_intra_nxn_job), (x265amd_intra_nxn_out*)R.dNxnOut.p) != X265AMD_OK))
        rc = xa_fail(X265AMD_EHIP, "intra rd: chain commands");
    /* the chain runs by itself for a while: what the caller has to do meanwhile (collecting the enclosing CU's own evaluation: it uses this working set, whose fields for
     * the chain are not needed any more -- the results are read from the mapped record below) */
    if (rc == X265AMD_OK && between) between(between_ctx);
    if (rc == X265AMD_OK && xa_stream_sync(stream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "intra rd: chain");
    if (rc != X265AMD_OK) return rc;
    outs = static_cast<x265amd_intra_cu8_result*>(static_cast<IntraRd*>(*ws)->qOut.p);
    for (int i = 0; i < 4; i++)
    {
        memcpy(&results[i], &outs[i], sizeof(results[i]));
        if (results[i].status != 1)
        {
            /* a command that gave up has left ~0 in a peer's `ready` word: counts only grow, so every later chain on this block would see "the other side gave
             * up" at once.  The block never goes back to the pool (48 KB lost per failed picture; the picture fails anyway) */
            R.qBlock = nullptr;
            return xa_fail(X265AMD_EHIP, "intra rd: the device gave up waiting inside a chain of 8x8 CUs");
        }
    }
    return X265AMD_OK;
}

void xa_intra_ws_free(void* ws)
{
    IntraRd* ip = static_cast<IntraRd*>(ws);
    if (ip && ip->ahead.on && ip->helper) (void)xa_stream_sync(ip->helper);          /* its buffers are about to go back to the pool */
    for (int b = 0; ip && b < 2; b++) if (ip->big[b].on && ip->big[b].q) (void)xa_stream_sync(ip->big[b].q);
    for (int b = 0; ip && b < 3; b++)
    {
        if (ip->inter[b].on && ip->inter[b].q) (void)xa_stream_sync(ip->inter[b].q);
        if (ip->inter[b].owned) { xa_queue_release_helper(ip->inter[b].q); ip->inter[b].q = nullptr; ip->inter[b].owned = false; }
    }
    delete ip;
}
int xa_check_intra_begin_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                            intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, uint64_t d_pred, uint64_t d_recon, void** ws)
{
    if (!ws || !xa_queue_helper(stream) || !xa_queue_helper(xa_queue_helper(stream))) return 0;
    return intra_cu_impl(2, 0, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, nullptr, d_pred, d_recon, nullptr, nullptr, nullptr, ws);
}
int xa_intra_in_inter_begin_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                               intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, uint64_t d_pred, uint64_t d_recon, void** ws)
{
    if (!ws || !xa_is_queue(stream)) return 0;
    return intra_cu_impl(3, 0, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, nullptr, d_pred, d_recon, nullptr, nullptr, nullptr, ws);
}
void xa_intra_ws_hint_nxn(void** ws, uint64_t d_pred_nxn, uint64_t d_recon_nxn)
{
    if (!ws) return;
    if (!*ws) *ws = new IntraRd;
    IntraRd* ip = static_cast<IntraRd*>(*ws);
    ip->hintPred = d_pred_nxn; ip->hintRecon = d_recon_nxn;
}
int xa_check_intra_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                      intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, int part_size, x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon,
                      x265amd_rd_result* out, int16_t* coeff_out, void** ws)
{
    return intra_cu_impl(1, part_size, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, cu_units, d_pred, d_recon, out, coeff_out, nullptr, ws);
}
int xa_intra_in_inter_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                         intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out,
                         int16_t* coeff_out, uint64_t* info, void** ws)
{
    return intra_cu_impl(0, 0, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, cu_units, d_pred, d_recon, out, coeff_out, info, ws);
}

extern "C" int x265amd_intra_in_inter(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                                      const uint64_t* h_src, const uint64_t* h_rec, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu,
                                      x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out, int16_t* coeff_out, uint64_t* info)
{
    return intra_cu_impl(0, 0, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, cu_units, d_pred, d_recon, out, coeff_out, info);
}

extern "C" int x265amd_check_intra(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                                   const uint64_t* h_src, const uint64_t* h_rec, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, int part_size,
                                   x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out, int16_t* coeff_out)
{
    return intra_cu_impl(1, part_size, stream, si, rp, units, h_src, h_rec, stride, cstride, cu, cu_units, d_pred, d_recon, out, coeff_out, nullptr);
}
