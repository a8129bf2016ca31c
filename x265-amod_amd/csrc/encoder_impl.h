/* The encoder object's own types, shared by its translation units (round 6: encoder_api.hip was one file of 2 900 lines):
 *   encoder_api.hip          configuration, open / close / headers, the frame-per-GPU row entries, x265amd_encoder_encode: admission and the order in which pictures start
 *   encoder_lookahead.hip    Lowres, adaptive quantisation, the lookahead's cost estimates, weights, scene cuts, the B-frame trellis, cuTree, slicetypeDecide
 *   encoder_ratecontrol.hip  DPB::prepareEncode with the rate control's QP of a picture and of its quantisation groups
 *   encoder_frame.hip        FrameEncoder::compressFrame: the picture's context, row gates, in-loop filters by rows and by columns, the slice NAL unit
 * Host C++ throughout; compiled by hipcc because every part talks to the HIP runtime. */
#ifndef X265AMD_ENCODER_IMPL_H
#define X265AMD_ENCODER_IMPL_H
#include <hip/hip_runtime.h>
#include "x265amd.h"
#include "x265amd_encoder.h"
#include "x265amd_host.h"
#include "x265amd_ratecontrol.h"
#include <immintrin.h>
#include "xa_fiber.h"
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <future>
#include <memory>
#include <vector>

namespace xaenc {

typedef x265amd_pixel pixel;
enum { TYPE_AUTO = 0, TYPE_IDR = 1, TYPE_I = 2, TYPE_P = 3, TYPE_BREF = 4, TYPE_B = 5 };          /* X265_TYPE_* (x265.h:572-577) */
static inline bool isBType(int t) { return t == TYPE_B || t == TYPE_BREF; }         /* IS_X265_TYPE_B */
enum { RD_TILE_ELEMS = 4096 + 2 * 1024 };

/* The lookahead's device buffers -- motion fields and per-estimate cost arrays, all of one size (a 32-bit word per lowres block) -- come from chunks of 64 of them: a first
 * decision of a few hundred estimates asks for a thousand buffers at once, and a cold general pool answered with a thousand hipMallocs (50 ms at 2160p).  Shared by the encoder
 * and its pictures: a picture keeps the block costs of its estimates on the device (cuTree reads them back when it asks for one) and hands them back when it goes, from
 * whatever thread that happens on. */
struct LaPool
{
    std::mutex mu;
    size_t one = 0;
    std::vector<void*> freeBufs, chunks;
    ~LaPool() { for (void* c : chunks) (void)hipFree(c); }
    void* get()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (freeBufs.empty())
        {
            void* chunk = nullptr;
            if (!one || hipMalloc(&chunk, one * 64) != hipSuccess) return nullptr;
            chunks.push_back(chunk);
            for (int i = 63; i >= 0; i--) freeBufs.push_back((char*)chunk + one * i);
        }
        void* p = freeBufs.back();
        freeBufs.pop_back();
        return p;
    }
    void put(void* p) { if (!p) return; std::lock_guard<std::mutex> lk(mu); freeBufs.push_back(p); }
    /* the motion fields of pictures that are gone (their host vectors' addresses: the keys of x265amd_encoder::laFields): the encoder drops the fields' device copies
     * the next time it looks (a later vector at the same heap address must not find them) */
    std::vector<const void*> deadFields;
    void dead(const void* key) { if (!key) return; std::lock_guard<std::mutex> lk(mu); deadFields.push_back(key); }
};

struct Pic;
typedef std::shared_ptr<Pic> PicP;
struct Pic
{
    int poc = 0, type = 0, sliceQp = 0;
    uint64_t codingOrder = 0;                           /* its place in coding order (the row tasks' priority) */
    bool started = false;
    bool owned = true;                                  /* frame-per-GPU: coded by this object; else its rows are imported (importedRows under `mu`) */
    int importedRows = 0;
    bool hasReferences = false;
    pixel* dSrc = nullptr; pixel* dRec = nullptr;       /* flat Y | U | V padded buffers (pooled device memory) */
    std::vector<x265amd_cu_unit> units;
    std::vector<x265amd_mv_unit> motion;
    int32_t refPoc[2][16];
    /* what DPB::prepareEncode decided for this picture (coding order, main thread) */
    int nalType = 0, lastIDR = 0;
    bool rpsUsed = true;                                /* used_by_curr_pic flags of the picture's RPS: off for an IRAP picture (DPB::computeRPS, dpb.cpp:320) */
    std::vector<PicP> neg, pos, lists[2];
    /* the frame task: result code when the picture is completely coded (reconstruction final, NAL written) */
    std::shared_future<int> done;
    std::vector<uint8_t> nalBytes;
    /* pictures coded in parallel (param.frameNumThreads > 1): the filtered picture is built in dFin (dRec when SAO is off) while the picture is analysed and
     * published to the pictures that reference it as it becomes final (finalX below) -- Frame::m_reconRowFlag (frameencoder.cpp:900-905, framefilter.cpp:654-664),
     * by columns instead of whole rows */
    pixel* dFin = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    int analysedRows = 0;
    /* Publication by columns: finalX[r] luma sample columns of CTU row r are final in the filtered picture (the picture width: the whole row, right margin
     * included); what the pictures referencing this one wait for, CTU by CTU (gateCtuReady / gateRefReady).  analysedCols[r] (under `mu`): CTUs of row r analysed. */
    std::vector<volatile uint64_t*> finalX;     /* counters (xa_fiber.h): the row tasks of other pictures park on them */
    std::vector<int> analysedCols;
    std::atomic<bool> failed{ false };
    const pixel* finalPlanes() const { return dFin ? dFin : dRec; }
    /* Lowres (common/lowres.h) as far as the slice-type decision reads it: the four half-resolution planes, the intra costs per 8x8 block, the frame cost
     * estimates by distance to the reference (costEst[d][0]: P cost against the picture d before; [0][0]: intra), the scene-cut mark */
    pixel* dLowres = nullptr; int32_t* dIntraCost = nullptr;
    int64_t costEst[18]; int intraMbs[18];
    std::vector<int16_t> lowMvs[18];        /* Lowres::lowresMvs[0][d]: the motion field of the estimates against the picture d before (the encoder's searches take a candidate from it) */
    std::vector<int16_t> lowMvs1[18];       /* Lowres::lowresMvs[1][d]: against the picture d behind (B estimates: --b-adapt 2) */
    std::vector<int32_t> lowMvc[18], lowMvc1[18];   /* Lowres::lowresMvCosts: read again when a later estimate uses a field that exists */
    int64_t cost2[18][18];                  /* Lowres::costEst[b - p0][p1 - b] (B estimates scaled as estimateFrameCost does); [d][0] is costEst[d] */
    /* Searched ahead of the trellis, not yet the picture's: fields and estimates the reference makes one at a time when a path asks for them (if it ever does).  They are made
     * side by side in advance and become the picture's -- lowMvs / cost2 / costEst / intraMbs -- at the moment the reference would have made them (x265amd_encoder::frameCostAt),
     * so what exists when a picture is coded, or when the scene-cut check looks for an estimate, is what exists in the reference */
    std::vector<int16_t> specMvs[18], specMvs1[18];
    std::vector<int32_t> specMvc[18], specMvc1[18];
    int64_t specCost2[18][18]; int specIntraMbs[18];
    uint64_t wpSum[3] = { 0, 0, 0 }, wpSsd[3] = { 0, 0, 0 };      /* Lowres::wp_sum / wp_ssd (bEnableWeightedPred) */
    int lumaDenom = 7, chromaDenom = 7;                    /* the slice's pred_weight_table denominators (weightAnalyse) */
    x265amd_weight wp[2][16][3];                           /* slice.m_weightPredTable (weightAnalyse; all zero without weighted prediction) */
    bool weighted = false;                                 /* some reference of this slice carries a weight */
    bool bScenecut = false, bKeyframe = false;
    std::vector<int8_t> tuRecs;         /* --limit-tu 3 / 4: CUData::m_refTuDepth of every CTU (XaTuRecs) */
    /* ---- rate control other than constant QP (round 6): what adaptive quantisation and cuTree keep of a picture's Lowres (common/lowres.h) ---- */
    std::vector<int32_t> intraCostHost;                     /* Lowres::intraCost per lowres block (read back once, in lowresInit) */
    std::vector<double> qpAqOffset, qpCuTreeOffset;         /* Lowres::qpAqOffset / qpCuTreeOffset per 16x16 block (the lowres block grid) */
    std::vector<int32_t> invQscale;                         /* Lowres::invQscaleFactor */
    std::vector<uint16_t> propagateCost;                    /* Lowres::propagateCost */
    /* Lowres::lowresCosts[d0][d1] of the estimates made so far (key d0 * 32 + d1): on the device where the estimate left them (dLc; dSpecLc: of estimates made ahead of their
     * time, see specCost2), on the host once cuTree has asked for them */
    std::map<int, void*> dLc, dSpecLc;
    std::map<int, std::vector<uint16_t> > lcHost;
    std::shared_ptr<LaPool> pool;
    void dropLc(std::map<int, void*>& m, int key) { auto it = m.find(key); if (it != m.end()) { if (pool) pool->put(it->second); m.erase(it); } }
    double avgQpRc = 0;                                     /* FrameData::m_avgQpRc: the rate control's QP before rounding (rateControlStart) */
    bool bLastMiniGopBFrame = false;
    std::vector<int8_t> cuQp;                               /* Analysis::calculateQpforCuSize per quantisation group down to pps.maxCuDQPDepth: per CTU 1 + 4 (+ 16) values in z order */
    const x265amd_mv_unit* regMotion = nullptr;        /* the motion field's mirror in device memory (x265amd_host.h: xa_devmap_*): what the skip chain of this and later pictures reads */
    void registerMotion()
    {
        if (regMotion == motion.data()) return;
        if (regMotion) xa_devmap_unregister(regMotion);
        regMotion = motion.data();
        if (!xa_devmap_register(regMotion, motion.size())) regMotion = nullptr;
    }
    Pic() { memset(refPoc, 0, sizeof(refPoc)); memset(wp, 0, sizeof(wp)); for (int i = 0; i < 18; i++) { costEst[i] = -1; intraMbs[i] = 0; specIntraMbs[i] = 0; for (int j = 0; j < 18; j++) cost2[i][j] = specCost2[i][j] = -1; } }
    ~Pic() { if (pool) { for (auto& e : dLc) pool->put(e.second); for (auto& e : dSpecLc) pool->put(e.second);
                         for (int i = 0; i < 18; i++) for (const std::vector<int16_t>* v : { &lowMvs[i], &lowMvs1[i], &specMvs[i], &specMvs1[i] }) if (!v->empty()) pool->dead(v->data()); } if (regMotion) xa_devmap_unregister(regMotion); xa_scratch_free(dSrc); xa_scratch_free(dRec); xa_scratch_free(dFin); xa_scratch_free(dLowres); xa_scratch_free(dIntraCost); for (volatile uint64_t* c : finalX) xa_counter_free(c); }
    void publish(int row, int x)
    {
        std::atomic_thread_fence(std::memory_order_release); *finalX[row] = (uint64_t)x;
        static const bool pubLog = getenv("X265AMD_PUB_LOG") != nullptr;      /* with the gate's waits (gateCtuWait): who waited for which publication, and when it came */
        if (pubLog) fprintf(stderr, "x265amd pub: poc %d row %d x %d at %.2f\n", poc, row, x, pubClockMs());
    }
    static double pubClockMs() { return fmod(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1e6); }       /* the clock of X265AMD_CTU_LOG */
    int published(int row) const { const int v = (int)*finalX[row]; std::atomic_thread_fence(std::memory_order_acquire); return v; }
    void fail()         /* whoever waits for this picture is released */
    {
        failed.store(true, std::memory_order_release);
        for (volatile uint64_t* c : finalX) *c = 1u << 30;
        { std::lock_guard<std::mutex> lk(mu); }
        cv.notify_all();
    }
};

}
using namespace xaenc;

struct x265amd_encoder
{
    x265amd_param p;
    x265amd_me_ctx* me = nullptr;
    int W = 0, H = 0, w4 = 0, h4 = 0, ctuW = 0, ctuH = 0, nctu = 0;
    int srcW = 0, srcH = 0;             /* x265amd_param's size; W x H is what is coded: srcW x srcH padded to multiples of 8 */
    int marginX = 96, marginY = 80;
    intptr_t stride = 0, cstride = 0;
    size_t org[3] = { 0, 0, 0 }, picElems = 0;
    int qpConstant[3] = { 0, 0, 0 };                    /* indexed by slice type 0 B, 1 P, 2 I */
    int maxDecPicBuffering = 0, numReorderPics = 0;
    int frameCount = 0, lastKeyframe = 0, lastIDR = 0;
    bool haveKeyframe = false, refreshPending = false;  /* open GOPs: a keyframe has been typed (the first one is an IDR picture); DPB::m_bRefreshPending */
    int pocCRA = 0;                                     /* DPB::m_pocCRA */
    bool first = true;
    std::deque<PicP> input;                             /* display order, not yet typed */
    std::deque<PicP> ready;                             /* coding order, typed, not yet prepared */
    std::deque<PicP> inflight;                          /* coding order: prepared pictures, their frame tasks running or (frame-parallel only) still to start */
    uint64_t codingCount = 0;
    int running = 0;                                    /* frame tasks started and not yet collected */
    std::mutex importMu;
    hipStream_t importStream = nullptr;                 /* frame-per-GPU: rows of pictures coded elsewhere are copied in on it */
    std::mutex byCodingMu;
    std::map<uint64_t, PicP> byCoding;                  /* the pictures in flight (and the last few collected) by their place in coding order (row export / import) */
    uint64_t collectedCoding = 0;                       /* pictures collected so far (under byCodingMu) */
    uint64_t statPictures[3] = { 0, 0, 0 }, statReferences = 0;     /* x265amd_encoder_stats: pictures prepared as I / P / B, the sum of their distinct reference pictures */
    uint64_t statEmitted[3] = { 0, 0, 0 }, statBits[3] = { 0, 0, 0 }; double statQpSum[3] = { 0, 0, 0 };       /* ... and of the pictures handed out: count, NAL bits, the sum of their average QPs, by I / P / B */
    std::shared_future<int> lastTask;                   /* the previous picture's task: in-loop filters and SAO run in coding order */
    int frameThreads = 1;
    double uploadMs = 0;        /* X265AMD_TIMING: the callers' time in uploadPicture */
    double firstInMs = -1;      /* X265AMD_HOLD_UNTIL_FLUSH: when the first picture came in (Pic::pubClockMs) */
    std::atomic<uint64_t> cpuPictureNs{ 0 }, cpuFilterNs{ 0 };      /* X265AMD_TIMING: CPU time of the picture threads and the filter threads (CLOCK_THREAD_CPUTIME_ID) */
    bool frameParallel = false;                         /* param.frameNumThreads > 1: the reference's frame-parallel rules (search.cpp:77-92, sao.cpp:264) */
    int refLagRows = 0;                                 /* FrameEncoder::m_refLagRows (frameencoder.cpp:170-175) */
    std::vector<PicP> picList;                          /* front = most recently coded (PicList::pushFront) */
    double depthSaoRate[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    std::vector<uint8_t> headerBytes, outBytes;
    std::vector<x265amd_nal> nals;
    std::vector<pixel> staging;
    pixel* uploadBuf = nullptr;         /* pinned: the padded input picture as uploadPicture puts it together (hipHostFree in the destructor) */
    int32_t* dSaoCount = nullptr; int32_t* dSaoOrg = nullptr; x265amd_sao_ctu* dSaoParams = nullptr; x265amd_deblock_unit* dDbUnits = nullptr;
    pixel* dSaoTmp = nullptr;

    ~x265amd_encoder()
    {
        for (auto& q : inflight) if (q->done.valid()) q->done.wait();
        laFieldsFree();
        if (uploadBuf) (void)hipHostFree(uploadBuf);
        if (getenv("X265AMD_TIMING") && lookahead)
        {
            fprintf(stderr, "x265amd: input: %.1f ms in uploads; cpu of the picture threads %.1f ms, of the filter threads %.1f ms\n", uploadMs, cpuPictureNs.load() / 1e6, cpuFilterNs.load() / 1e6);
            fprintf(stderr, "x265amd: lookahead: %.1f ms in lowres planes + intra costs, %.1f ms in the slice-type decision (%llu estimates, %llu motion searches; %llu batches %.1f ms, %llu single estimates %.1f ms)\n", laInitMs, laDecideMs,
                    (unsigned long long)laJobs, (unsigned long long)laSearches, (unsigned long long)laBatches, laBatchMs, (unsigned long long)laSingles, laSingleMs);
            fprintf(stderr, "x265amd: lookahead estimates by phase: set-up %.1f ms, launch call %.1f, read-back issued %.1f, waited for %.1f, host sums %.1f; %llu weight guesses measured\n", laPhaseMs[0], laPhaseMs[1], laPhaseMs[2], laPhaseMs[3], laPhaseMs[4], (unsigned long long)laWeightJobs);
        }
        if (me) x265amd_me_close(me);
        if (rateCtl) x265amd_rc_close(rateCtl);
        if (dSaoCount) (void)hipFree(dSaoCount);
        if (dSaoOrg) (void)hipFree(dSaoOrg);
        if (dSaoParams) (void)hipFree(dSaoParams);
        if (dDbUnits) (void)hipFree(dDbUnits);
        xa_scratch_free(dSaoTmp);
        if (laStream) (void)hipStreamDestroy(laStream);
        if (importStream) (void)hipStreamDestroy(importStream);
    }
    uint64_t planeAddr(const pixel* base, int k) const { return (uint64_t)(uintptr_t)(base + org[k]); }

    void fillStreamParams(x265amd_stream_params& s) const;
    int uploadPicture(const x265amd_picture* in, Pic& pic, bool onDevice = false);
    void decideMiniGop(bool flush);
    int prepare(const PicP& pic);
    int runFrame(const PicP& pic, std::shared_future<int> prev);
    int runFrameParallel(const PicP& pic);
    /* ---- the lookahead (slicetype.cpp): only when param.scenecutThreshold > 0 ---- */
    bool lookahead = false;
    void* wpEnergy = nullptr; void* wpSums = nullptr; void* wpSumsHost = nullptr; void* wpMvs = nullptr;      /* weighted prediction's device / mapped buffers: the encoder's for good (never back to the pools) */
    int laRowsPerSlice = 0, laNumSlices = 1;            /* Lookahead::m_numRowsPerSlice / m_numCoopSlices (slicetype.cpp:1035-1059) */
    int keyframeMin = 1, lowW = 0, lowH = 0, lowCuW = 0, lowCuH = 0, lowBlocks = 0;
    intptr_t lowStride = 0; size_t lowPlaneElems = 0, lowOrg = 0;
    PicP lastNonB;                                      /* Lookahead::m_lastNonB */
    bool isSceneTransition = false;                     /* Lookahead::m_isSceneTransition */
    hipStream_t laStream = nullptr;
    int lowresInit(Pic& pic);
    void pushMiniGop(int b);
    struct LaWeight { int minscale = 0, mindenom = 0, curScale = 0, curOffset = 0; };
    bool lookaheadWeightGuess(Pic& fenc, Pic& ref, LaWeight& g);
    void lookaheadWeightDecide(const LaWeight& g, const uint32_t costs[2], bool& weighted, int& scale, int& denom, int& offset);
    int sliceWeights(Pic& pic);
    bool keepSources() const { return p.bEnableWeightedPred || p.bEnableWeightedBiPred; }
    int weightRows(struct WPlane& wpl, int r0, int r1);
    int frameCostP(Pic& b, Pic& ref, int dist);         /* CostEstimateGroup::singleCost(p0, p1 = b, b) */
    int frameCostAt(Pic& fenc, Pic& ref0, Pic* ref1, int d0, int d1, int64_t& score);
    struct CostJob { Pic* fenc = nullptr; Pic* ref0 = nullptr; Pic* ref1 = nullptr; int d0 = 0, d1 = 0; bool spec = false; bool whole = false; bool search0 = false, search1 = false; void* dMvs = nullptr; void* dMvc = nullptr; void* dMvs1 = nullptr;
                     void* dMvc1 = nullptr; void* dLc = nullptr; void* dBc = nullptr; void* dW = nullptr; };
    int frameCostMany(std::vector<CostJob>& jobs);
    /* The motion fields' DEVICE copies, by the address of the host copy (Pic::lowMvs and its kin: swapped between vectors, never copied; filled by one place only, the
     * read-back of a search in frameCostMany, which enters the search's own device buffers here -- whatever stood under that address before is replaced): an estimate
     * that reads a field finds it on the device instead of uploading 2 x 130 KB of pageable memory (a 2160p first decision: 51 ms of them).  Bounded: beyond
     * LA_FIELDS_MAX entries the least recently used quarter goes (a field that is gone is uploaded again). */
    struct DevField { void* mv; void* mc; uint64_t used; };
    std::map<const void*, DevField> laFields;
    uint64_t laFieldClock = 0;
    static const size_t LA_FIELDS_MAX = 2048;
    /* the lookahead's device buffers (LaPool above) */
    std::shared_ptr<LaPool> laPool;
    void* laBuf() { return laPool->get(); }
    void laBufPut(void* p) { laPool->put(p); }
    void laFieldPut(const void* key, void* mv, void* mc);
    void laFieldsTrim();
    void laFieldsFree();
    double laPhaseMs[5] = { 0, 0, 0, 0, 0 };        /* frameCostMany: set-up, the launch call, issuing the read-back, waiting for it, the host sums */
    double laInitMs = 0, laDecideMs = 0, laBatchMs = 0, laSingleMs = 0; uint64_t laJobs = 0, laSearches = 0, laBatches = 0, laSingles = 0, laWeightJobs = 0;
    int frameCost(std::vector<Pic*>& frames, int p0, int p1, int b, int64_t& score);       /* CostEstimateGroup::singleCost(p0, p1, b): P (p1 == b) or B estimate */
    int64_t planCost(std::vector<Pic*>& frames, const std::vector<uint8_t>& runs, int64_t limit, int& rc);
    void extendPlans(std::vector<Pic*>& frames, int length, std::vector<std::vector<uint8_t> >& plans, int& rc);
    bool scenecutInternal(std::vector<Pic*>& frames, int p0, int p1, bool real, int& rc);
    bool scenecut(std::vector<Pic*>& frames, int p0, int p1, bool real, int numFrames, int& rc);
    int slicetypeAnalyse(std::vector<Pic*>& frames, bool bKeyframe = false);
    int decideLookahead(bool flush, int maxGops = 1 << 30);
    /* ---- rate control other than constant QP (round 6; include/x265amd_ratecontrol.h) ---- */
    x265amd_rc* rateCtl = nullptr;                         /* RateControl, the constant-rate-factor branch */
    bool useDqp = false; int maxCuDqpDepth = 0;         /* pps.bUseDQP / maxCuDQPDepth (encoder.cpp:3461-3469: with adaptive quantisation) */
    bool aqOn = false;
    void* aqEnergy = nullptr; void* aqEnergyHost = nullptr; void* aqSums = nullptr;      /* x265amd_aq_energy's outputs (the encoder's for good) */
    x265amd_cutree_params treeParams;
    int adaptiveQuant(Pic& pic);
    int runCuTree(std::vector<Pic*>& frames, int numframes, bool bIntra);
    static int cuTreeEstimate(void* ctx, int p0, int p1, int b, const uint16_t** lc, const int16_t** mvs0, const int16_t** mvs1);
    int64_t estimatedPictureCost(Pic& pic);
    void cuQpTable(Pic& pic);
    int filterRows(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags);
    int filterRowsCols(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags);
};

#endif
