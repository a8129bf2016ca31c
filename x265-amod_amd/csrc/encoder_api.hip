/* The encoder object behind include/x265amd_encoder.h: the frame-level host loop of the reference (SURVEY section 8b outer boundary and 8f rank 1),
 * restated for a device-resident encode.  What it follows in the reference:
 *   - mini-GOP formation with bFrameAdaptive 0:   Lookahead::slicetypeDecide        source/encoder/slicetype.cpp:1929-2040
 *   - constant QP per slice type:                  RateControl::init / rateControlStart  source/encoder/ratecontrol.cpp:321-346, :1592-1597
 *   - decoded picture buffer, RPS, NAL type:       DPB::prepareEncode / computeRPS / applyReferencePictureSet / decodingRefreshMarking /
 *                                                  getNalUnitType                  source/encoder/dpb.cpp:134-330, :336-470, :487-510
 *   - reference lists:                             Slice::setRefPicList            source/common/slice.cpp:32-140
 *   - DPB sizes, level:                            Encoder::initVPS/initSPS, determineLevel   source/encoder/encoder.cpp:3340-3470, level.cpp:44-230, :290-300
 *   - per frame: FrameEncoder::compressFrame (analysis rows, deblocking, SAO, border extension, slice NAL)   source/encoder/frameencoder.cpp:470-1130
 * Pictures (source and reconstruction) are padded planes in device memory with the reference's PicYuv margins (maxCUSize + 32 / + 16). */
#include <hip/hip_runtime.h>
#include "x265amd.h"
#include "x265amd_encoder.h"
#include "x265amd_host.h"
#include "x265amd_ratecontrol.h"
#include <immintrin.h>
#include "xa_fiber.h"
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

namespace {

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

struct x265amd_encoder
{
    x265amd_param p;
    x265amd_me_ctx* me = nullptr;
    int W = 0, H = 0, w4 = 0, h4 = 0, ctuW = 0, ctuH = 0, nctu = 0;
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
    int32_t* dSaoCount = nullptr; int32_t* dSaoOrg = nullptr; x265amd_sao_ctu* dSaoParams = nullptr; x265amd_deblock_unit* dDbUnits = nullptr;
    pixel* dSaoTmp = nullptr;

    ~x265amd_encoder()
    {
        for (auto& q : inflight) if (q->done.valid()) q->done.wait();
        laFieldsFree();
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
    int uploadPicture(const x265amd_picture* in, Pic& pic);
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

/* ---- configuration ---- */
extern "C" void x265amd_param_default(x265amd_param* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->fpsNum = 25; p->fpsDenom = 1;
    p->bframes = 0; p->keyframeMax = 250; p->maxNumReferences = 3; p->qp = 30; p->ipFactor = 1.4f; p->pbFactor = 1.3f;      /* (float literals, as common/param.cpp:276-277 has them) */
    p->rateControlMode = X265AMD_RC_CQP; p->rfConstant = 28; p->aqStrength = 1.0; p->qCompress = 0.6; p->aqMode = 0; p->cuTree = 0; p->qgSize = 32;
    p->rdLevel = 3; p->limitReferences = 3; p->bEnableEarlySkip = 1; p->recursionSkipMode = 1; p->bIntraInBFrames = 1; p->psyRd = 2.0;
    p->searchMethod = X265AMD_ME_HEX; p->subpelRefine = 2; p->searchRange = 57; p->maxNumMergeCand = 3;
    p->bEnableSignHiding = 1; p->bEnableStrongIntraSmoothing = 1; p->bEnableTemporalMvp = 1; p->tuQTMaxInterDepth = 1; p->tuQTMaxIntraDepth = 1;
    p->bEnableLoopFilter = 1; p->bEnableSAO = 1; p->bEnableWavefront = 1; p->aspectRatioIdc = 0;
}

void x265amd_encoder::fillStreamParams(x265amd_stream_params& s) const
{
    memset(&s, 0, sizeof(s));
    /* determineLevel (level.cpp:76-230): Main / Main 10, main tier, the lowest level that holds the picture size, rate and DPB */
    static const struct { uint32_t maxLumaSamples, maxLumaSamplesPerSecond; int idc; } levels[] = {
        { 36864, 552960, 30 }, { 122880, 3686400, 60 }, { 245760, 7372800, 63 }, { 552960, 16588800, 90 }, { 983040, 33177600, 93 },
        { 2228224, 66846720, 120 }, { 2228224, 133693440, 123 }, { 8912896, 267386880, 150 }, { 8912896, 534773760, 153 },
        { 8912896, 1069547520, 156 }, { 35651584, 1069547520, 180 }, { 35651584, 2139095040u, 183 }, { 35651584, 4278190080u, 186 } };
    s.profile_idc = X265AMD_DEPTH <= 8 ? 1 : 2;
    s.profile_compatibility_flags = X265AMD_DEPTH <= 8 ? (1u << 1) | (1u << 2) : (1u << 2);
    s.progressive_source = 1; s.frame_only_constraint = 1;
    s.bit_depth_constraint = X265AMD_DEPTH; s.chroma_format_constraint = 1; s.lower_bit_rate_constraint = 1;
    s.intra_constraint = p.keyframeMax <= 1;
    const uint32_t lumaSamples = (uint32_t)(W * H);
    const uint32_t samplesPerSec = (uint32_t)(lumaSamples * ((double)p.fpsNum / p.fpsDenom));
    s.level_idc = 255;
    for (const auto& l : levels)
    {
        if (lumaSamples > l.maxLumaSamples || samplesPerSec > l.maxLumaSamplesPerSecond) continue;
        if (W > sqrt(l.maxLumaSamples * 8.0f) || H > sqrt(l.maxLumaSamples * 8.0f)) continue;
        uint32_t maxDpbSize = 6;
        if (lumaSamples <= (l.maxLumaSamples >> 2)) maxDpbSize = 16;
        else if (lumaSamples <= (l.maxLumaSamples >> 1)) maxDpbSize = 12;
        else if (lumaSamples <= ((3 * l.maxLumaSamples) >> 2)) maxDpbSize = 8;
        if ((uint32_t)maxDecPicBuffering > maxDpbSize) continue;
        s.level_idc = l.idc;
        break;
    }
    s.max_temporal_sub_layers = 1;
    s.max_dec_pic_buffering[0] = maxDecPicBuffering; s.num_reorder_pics[0] = numReorderPics; s.max_latency_increase[0] = p.bframes;
    s.chroma_format_idc = 1; s.pic_width = W; s.pic_height = H; s.bit_depth = X265AMD_DEPTH; s.log2_max_poc_lsb = 8;
    s.log2_min_cu_size = 3; s.log2_diff_max_min_cu_size = 3; s.tu_log2_min = 2; s.tu_log2_max = 5;
    s.tu_max_depth_inter = p.tuQTMaxInterDepth; s.tu_max_depth_intra = p.tuQTMaxIntraDepth;
    s.amp = p.bEnableAMP != 0; s.sao = p.bEnableSAO != 0; s.temporal_mvp = p.bEnableTemporalMvp != 0; s.strong_intra_smoothing = p.bEnableStrongIntraSmoothing != 0;
    s.aspect_ratio_idc = p.aspectRatioIdc;
    s.emit_timing_info = 1; s.num_units_in_tick = p.fpsDenom; s.time_scale = p.fpsNum;
    s.weighted_pred = p.bEnableWeightedPred != 0; s.weighted_bipred = p.bEnableWeightedBiPred != 0;
    s.sign_hide = p.bEnableSignHiding != 0; s.num_ref_idx_default[0] = s.num_ref_idx_default[1] = 1; s.init_qp_minus26 = 0;
    s.use_dqp = useDqp; s.max_cu_dqp_depth = maxCuDqpDepth;           /* Encoder::initPPS (encoder.cpp:3424-3445) */
    s.wpp = p.bEnableWavefront != 0; s.loop_filter_across_slices = 1;
    s.deblocking_filter_control_present = !p.bEnableLoopFilter; s.pic_disable_deblocking = !p.bEnableLoopFilter;
}

extern "C" x265amd_encoder* x265amd_encoder_open(const x265amd_param* p)
{
    if (!p) { xa_fail(X265AMD_EINVAL, "encoder_open: null param"); return nullptr; }
    if (p->sourceWidth < 16 || p->sourceHeight < 16 || (p->sourceWidth & 7) || (p->sourceHeight & 7) || p->sourceWidth > 8192 || p->sourceHeight > 4320)
    { xa_fail(X265AMD_EINVAL, "encoder_open: picture size must be a multiple of 8 (16..8192 x 16..4320)"); return nullptr; }
    {
        /* every field outside the built subset is named (the reference logs "x265 [error]: <what>" per field, encoder/api.cpp:96-239 -> x265_check_params) */
        static thread_local char why[160];
        const char* bad = nullptr;
#define XA_REQUIRE(cond, text) do { if (!bad && !(cond)) bad = text; } while (0)
        XA_REQUIRE(p->fpsNum && p->fpsDenom, "fpsNum / fpsDenom must be non-zero");
        XA_REQUIRE(p->bframes >= 0 && p->bframes <= 16, "bframes outside 0..16");
        XA_REQUIRE(p->keyframeMax >= 1, "keyframeMax below 1");
        XA_REQUIRE(p->maxNumReferences >= 1 && p->maxNumReferences <= 8, "maxNumReferences outside 1..8");
        XA_REQUIRE(p->rateControlMode == 0 || p->rateControlMode == X265AMD_RC_CQP || p->rateControlMode == X265AMD_RC_CRF, "rc.rateControlMode: constant QP (1) and constant rate factor (2) are built, ABR (0 with a bitrate) is not");
        XA_REQUIRE(p->rateControlMode == X265AMD_RC_CRF || (p->qp >= 0 && p->qp <= 51), "qp outside 0..51");
        XA_REQUIRE(p->rateControlMode != X265AMD_RC_CRF || (p->rfConstant >= 0 && p->rfConstant <= 51), "rfConstant outside 0..51");
        XA_REQUIRE(p->aqMode >= 0 && p->aqMode <= 3, "aqMode outside 0..3 (the edge-based modes are not built)");
        XA_REQUIRE(!p->aqMode || p->aqStrength > 0, "aqStrength must be positive with aqMode (0 switches adaptive quantisation off in the reference: say aqMode 0)");
        XA_REQUIRE(!p->aqMode || p->qgSize == 32 || p->qgSize == 64, "qgSize: 64 and 32 are built");
        XA_REQUIRE(!p->cuTree || p->aqMode, "cuTree needs adaptive quantisation (Encoder::configure switches it on with cuTree; say aqMode)");
        XA_REQUIRE(!p->cuTree || p->rateControlMode == X265AMD_RC_CRF, "cuTree needs rate control (the reference switches it off under constant QP, encoder.cpp:3721-3728)");
        XA_REQUIRE(!p->cuTree || p->lookaheadDepth > 0, "cuTree needs the lookahead (lookaheadDepth > 0)");
        XA_REQUIRE(!(p->aqMode && p->rateControlMode != X265AMD_RC_CRF), "adaptive quantisation needs rate control (the reference switches it off under constant QP, encoder.cpp:3721-3728)");
        XA_REQUIRE(p->rdLevel >= 2 && p->rdLevel <= 6, "rdLevel outside 2..6 (rd 0-1 are not built)");
        XA_REQUIRE(p->maxNumMergeCand >= 1 && p->maxNumMergeCand <= 5, "maxNumMergeCand outside 1..5");
        XA_REQUIRE(p->tuQTMaxInterDepth >= 1 && p->tuQTMaxInterDepth <= 4, "tuQTMaxInterDepth outside 1..4");
        XA_REQUIRE(p->tuQTMaxIntraDepth >= 1 && p->tuQTMaxIntraDepth <= 4, "tuQTMaxIntraDepth outside 1..4");
        XA_REQUIRE(p->searchMethod == X265AMD_ME_DIA || p->searchMethod == X265AMD_ME_HEX || p->searchMethod == X265AMD_ME_STAR, "searchMethod: only dia, hex and star are built (no umh / sea / full)");
        XA_REQUIRE(p->subpelRefine >= 0 && p->subpelRefine <= 7, "subpelRefine outside 0..7");
        XA_REQUIRE(p->rdoqLevel >= 0 && p->rdoqLevel <= 2, "rdoqLevel outside 0..2");
        XA_REQUIRE(p->psyRdoqFix8 >= 0, "psyRdoqFix8 negative");
        XA_REQUIRE(p->recursionSkipMode >= 0 && p->recursionSkipMode <= 1, "recursionSkipMode: only 0 and 1 are built (no edge-based rskip)");
        XA_REQUIRE(p->limitReferences >= 0 && p->limitReferences <= 3, "limitReferences outside 0..3");
        XA_REQUIRE(!p->bEnableAMP || p->bEnableRectInter, "bEnableAMP needs bEnableRectInter");
#undef XA_REQUIRE
        if (bad) { snprintf(why, sizeof(why), "encoder_open: %s", bad); xa_fail(X265AMD_EINVAL, why); return nullptr; }
    }
    xa_bind_device();           /* the threads of this library work on the opening thread's GPU */
    std::unique_ptr<x265amd_encoder> e(new x265amd_encoder);
    e->p = *p;
    e->W = p->sourceWidth; e->H = p->sourceHeight; e->w4 = e->W / 4; e->h4 = e->H / 4;
    e->ctuW = (e->W + 63) / 64; e->ctuH = (e->H + 63) / 64; e->nctu = e->ctuW * e->ctuH;
    e->stride = e->W + 2 * e->marginX; e->cstride = e->W / 2 + e->marginX;
    const size_t ysz = (size_t)(e->H + 2 * e->marginY) * e->stride, csz = (size_t)(e->H / 2 + e->marginY) * e->cstride;
    e->org[0] = (size_t)e->marginY * e->stride + e->marginX;
    e->org[1] = ysz + (size_t)(e->marginY / 2) * e->cstride + e->marginX / 2;
    e->org[2] = ysz + csz + (size_t)(e->marginY / 2) * e->cstride + e->marginX / 2;
    e->picElems = ysz + 2 * csz;
    /* RateControl (ratecontrol.cpp:321-346): constant QPs of the three slice types */
    const double ipOffset = 6.0 * log2(p->ipFactor), pbOffset = 6.0 * log2(p->pbFactor);
    auto clipQp = [](int q) { return q < 0 ? 0 : q > 69 ? 69 : q; };
    e->qpConstant[1] = p->qp;
    e->qpConstant[2] = clipQp((int)(p->qp - ipOffset + 0.5));
    e->qpConstant[0] = clipQp((int)(p->qp + pbOffset + 0.5));
    if (e->qpConstant[0] > 51 || e->qpConstant[2] > 51) { xa_fail(X265AMD_EINVAL, "encoder_open: slice QP above 51"); return nullptr; }
    /* level.cpp:290-296 */
    e->numReorderPics = (p->bBPyramid && p->bframes > 1) ? 2 : (p->bframes ? 1 : 0);           /* enforceLevel (level.cpp:295-296) */
    e->maxDecPicBuffering = std::min(16, std::max(e->numReorderPics + 2, p->maxNumReferences) + 1);
    if (p->firstFrame < 0) { xa_fail(X265AMD_EINVAL, "encoder_open: firstFrame"); return nullptr; }
    e->frameCount = p->firstFrame; e->lastKeyframe = p->firstFrame - p->keyframeMax; e->lastIDR = p->firstFrame;
    if (p->scenecutThreshold < 0 || p->scenecutThreshold > 100 || p->lookaheadDepth < 0 || p->lookaheadDepth > 250 || p->keyframeMin < 0 || p->keyframeMin > p->keyframeMax)
    { xa_fail(X265AMD_EINVAL, "encoder_open: scenecutThreshold outside 0..100, lookaheadDepth outside 0..250 or keyframeMin outside 0..keyframeMax"); return nullptr; }
    if (p->shardCount < 0 || p->shardCount > 64 || (p->shardCount > 1 && (p->shardRank < 0 || p->shardRank >= p->shardCount || p->frameNumThreads <= 1)))
    { xa_fail(X265AMD_EINVAL, "encoder_open: shardRank / shardCount (frame-per-GPU needs 0 <= rank < count and frameNumThreads > 1: rows are published by pictures coded in parallel)"); return nullptr; }
    if (p->bFrameAdaptive < 0 || p->bFrameAdaptive > 2) { xa_fail(X265AMD_EINVAL, "encoder_open: bFrameAdaptive: 0 (fixed mini-GOPs), 1 (fast) or 2 (trellis)"); return nullptr; }
    e->lookahead = p->scenecutThreshold > 0 || (p->bFrameAdaptive && p->bframes) || p->cuTree || p->aqMode;
    if ((p->bEnableWeightedPred || p->bEnableWeightedBiPred) && !e->lookahead) { xa_fail(X265AMD_EINVAL, "encoder_open: bEnableWeightedPred needs the lookahead (scenecutThreshold > 0 or bFrameAdaptive 2 with B frames)"); return nullptr; }
    {
        /* Encoder::configure (encoder.cpp:3658-3663) */
        int kmin = p->keyframeMin;
        if (!kmin) { const double fps = (double)p->fpsNum / p->fpsDenom; kmin = std::min((int)fps, p->keyframeMax / 10); }
        e->keyframeMin = std::max(1, kmin);
        /* Lowres::create (lowres.cpp:52-110): half size rounded up to whole 8x8 blocks, the picture's margins, stride a multiple of 32 */
        e->lowCuW = (e->W / 2 + 7) >> 3; e->lowCuH = (e->H / 2 + 7) >> 3;
        if (p->lookaheadSlices > 1 && e->H >= 720)
        {
            e->laRowsPerSlice = std::min(std::max(e->lowCuH / p->lookaheadSlices, 10), e->lowCuH);
            e->laNumSlices = e->lowCuH / e->laRowsPerSlice;
        }
        e->lowW = e->lowCuW * 8; e->lowH = e->lowCuH * 8;
        e->lowBlocks = (e->lowCuW > 2 && e->lowCuH > 2) ? (e->lowCuW - 2) * (e->lowCuH - 2) : e->lowCuW * e->lowCuH;
        e->lowStride = e->W / 2 + 2 * e->marginX;
        e->lowStride += (32 - (e->lowStride & 31)) & 31;
        e->lowPlaneElems = (size_t)(e->lowH + 2 * e->marginY) * e->lowStride;
        e->lowOrg = (size_t)e->marginY * e->lowStride + e->marginX;
        if (e->lookahead)
        {
            /* The lookahead's kernels run for tens of milliseconds (a batch of cost estimates: hundreds of rows chained through progress words), and streams of one priority
             * share hardware queues: a picture's in-loop filter launch -- one workgroup, microseconds -- queued behind such a batch on the same hardware queue waited for it to END
             * (k_deblock_unit: 47 ms at worst in round 4's trace), and every picture that references that row waited with it.  Streams of another priority get hardware queues
             * of their own (device_queue.hip: the resident kernel's is the highest): the lookahead takes the lowest.  X265AMD_LA_PRIORITY=0: the ordinary one, as before. */
            static const bool laLow = !(getenv("X265AMD_LA_PRIORITY") && atoi(getenv("X265AMD_LA_PRIORITY")) == 0);
            int least = 0, greatest = 0;
            if (!laLow || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest || hipStreamCreateWithPriority(&e->laStream, hipStreamNonBlocking, least) != hipSuccess)
                if (hipStreamCreateWithFlags(&e->laStream, hipStreamNonBlocking) != hipSuccess) { xa_fail(X265AMD_EHIP, "encoder_open: stream"); return nullptr; }
        }
    }
    {
        /* pictures whose references are complete are analysed concurrently (B frames of a mini-GOP, the next P): the reference's frame threads, but
         * a picture only starts when its references are final, so the output does not depend on the thread count */
        const char* ft = getenv("X265AMD_FRAME_THREADS");
        e->frameThreads = ft ? atoi(ft) : 3;
        if (e->frameThreads < 1) e->frameThreads = 1;
    }
    if (p->frameNumThreads < 0 || p->frameNumThreads > 16) { xa_fail(X265AMD_EINVAL, "encoder_open: frameNumThreads"); return nullptr; }
    e->frameParallel = p->frameNumThreads > 1;
    const int queues = xa_queues_hint(0);           /* the number of device job queues in force (224 unless X265AMD_QUEUES says otherwise) */
    if (e->frameParallel)
    {
        /* FrameEncoder::init (frameencoder.cpp:170-175): rows of a reference picture that must be final before a row of this picture starts */
        static const int hpelIters[8] = { 1, 1, 1, 2, 3, 1, 2, 3 };         /* MotionEstimate::hpelIterationCount: hpel_iters + qpel_iters / 2 (motion.cpp:48-58, :155) */
        int range = p->searchRange;
        range += p->searchMethod < 2;
        range += 8 / 2;
        range += 2 + (hpelIters[p->subpelRefine] + 1) / 2;
        e->refLagRows = 1 + ((range + 63) / 64);
        if (!getenv("X265AMD_FRAME_THREADS"))
        {
            /* every CTU row in flight holds a device job queue; pictures in flight never wait for one */
            const int rowsInFlight = p->bEnableWavefront ? std::max(1, std::min(e->ctuH, (e->ctuW + 1) / 2)) : 1;
            e->frameThreads = std::max(2, std::min(16, (queues > 0 ? queues : 16) / rowsInFlight));
        }
    }
    e->me = x265amd_me_open();
    if (!e->me) return nullptr;
    e->laPool.reset(new LaPool);
    e->laPool->one = (((size_t)e->lowCuW * e->lowCuH * 4) + 255) & ~(size_t)255;
    /* adaptive quantisation, cuTree, the rate factor (Encoder::configure / initPPS, RateControl::RateControl, Lookahead::Lookahead) */
    e->aqOn = p->aqMode != 0 && p->rateControlMode == X265AMD_RC_CRF;
    e->useDqp = e->aqOn;
    e->maxCuDqpDepth = e->aqOn ? (p->qgSize == 64 ? 0 : 1) : 0;
    memset(&e->treeParams, 0, sizeof(e->treeParams));
    e->treeParams.width8 = e->lowCuW; e->treeParams.height8 = e->lowCuH; e->treeParams.fps_num = p->fpsNum; e->treeParams.fps_denom = p->fpsDenom;
    e->treeParams.b_pyramid = p->bBPyramid != 0; e->treeParams.weighted_bipred = p->bEnableWeightedBiPred != 0; e->treeParams.lookahead_depth = p->lookaheadDepth;
    e->treeParams.strength = 5.0 * (1.0 - p->qCompress);
    if (p->rateControlMode == X265AMD_RC_CRF)
    {
        x265amd_rc_params rp;
        memset(&rp, 0, sizeof(rp));
        rp.width = e->W; rp.height = e->H; rp.fps_num = p->fpsNum; rp.fps_denom = p->fpsDenom; rp.bframes = p->bframes; rp.keyframe_max = p->keyframeMax; rp.cu_tree = p->cuTree != 0;
        rp.qp_min = 0; rp.qp_max = 69; rp.rf_constant = p->rfConstant; rp.q_compress = p->qCompress; rp.ip_factor = p->ipFactor; rp.pb_factor = p->pbFactor;
        e->rateCtl = x265amd_rc_open(&rp);
        if (!e->rateCtl) { xa_fail(X265AMD_EINVAL, "encoder_open: rate control parameters"); return nullptr; }
    }
    const size_t nstat = (size_t)e->nctu * 3 * 5 * 32;
    if (hipMalloc((void**)&e->dSaoCount, nstat * 4) != hipSuccess || hipMalloc((void**)&e->dSaoOrg, nstat * 4) != hipSuccess ||
        hipMalloc((void**)&e->dSaoParams, sizeof(x265amd_sao_ctu) * e->nctu) != hipSuccess ||
        hipMalloc((void**)&e->dDbUnits, sizeof(x265amd_deblock_unit) * e->w4 * e->h4) != hipSuccess ||
        xa_scratch_alloc((void**)&e->dSaoTmp, e->picElems * sizeof(pixel)) != hipSuccess)
    { xa_fail(X265AMD_EHIP, "encoder_open: device allocation"); return nullptr; }
    x265amd_stream_params sp;
    e->fillStreamParams(sp);
    e->headerBytes.resize(512);
    const size_t n = x265amd_write_stream_headers(&sp, e->headerBytes.data(), e->headerBytes.size());
    if (!n || n > e->headerBytes.size()) { xa_fail(X265AMD_EINVAL, "encoder_open: stream headers"); return nullptr; }
    e->headerBytes.resize(n);
    if (p->bEmitInfoSEI)
    {
        /* Encoder::getStreamHeaders' fourth unit (encoder.cpp:3260-3280): who coded this and with what -- the reference's own text names ITS build, this one names this library */
        char text[1024];
        snprintf(text, sizeof(text), "x265amd (x265 build 209 interface) - %s - H.265/HEVC codec on AMD Instinct MI355X - options: %dx%d fps=%u/%u bitdepth=%d %s=%g aq-mode=%d aq-strength=%.2f "
                 "cutree=%d qcomp=%.2f qg-size=%d bframes=%d b-adapt=%d b-pyramid=%d open-gop=%d keyint=%d min-keyint=%d scenecut=%d rc-lookahead=%d lookahead-slices=%d ref=%d limit-refs=%d "
                 "rd=%d rdoq-level=%d psy-rd=%.2f me=%d subme=%d merange=%d max-merge=%d rect=%d amp=%d limit-modes=%d early-skip=%d rskip=%d weightp=%d weightb=%d sao=%d deblock=%d wpp=%d "
                 "tu-intra-depth=%d tu-inter-depth=%d signhide=%d strong-intra-smoothing=%d temporal-mvp=%d b-intra=%d fast-intra=%d",
                 x265amd_version(), p->sourceWidth, p->sourceHeight, p->fpsNum, p->fpsDenom, X265AMD_DEPTH, p->rateControlMode == X265AMD_RC_CRF ? "crf" : "qp",
                 p->rateControlMode == X265AMD_RC_CRF ? p->rfConstant : (double)p->qp, e->aqOn ? p->aqMode : 0, p->aqStrength, p->cuTree != 0, p->qCompress, p->qgSize, p->bframes, p->bFrameAdaptive,
                 p->bBPyramid != 0, p->bOpenGOP != 0, p->keyframeMax, e->keyframeMin, p->scenecutThreshold, p->lookaheadDepth, p->lookaheadSlices, p->maxNumReferences, p->limitReferences,
                 p->rdLevel, p->rdoqLevel, p->psyRd, p->searchMethod, p->subpelRefine, p->searchRange, p->maxNumMergeCand, p->bEnableRectInter != 0, p->bEnableAMP != 0, p->limitModes != 0,
                 p->bEnableEarlySkip != 0, p->recursionSkipMode, p->bEnableWeightedPred != 0, p->bEnableWeightedBiPred != 0, p->bEnableSAO != 0, p->bEnableLoopFilter != 0, p->bEnableWavefront != 0,
                 p->tuQTMaxIntraDepth, p->tuQTMaxInterDepth, p->bEnableSignHiding != 0, p->bEnableStrongIntraSmoothing != 0, p->bEnableTemporalMvp != 0, p->bIntraInBFrames != 0, p->bEnableFastIntra != 0);
        uint8_t sei[1400];
        const size_t m = x265amd_write_info_sei(text, sei, sizeof(sei));
        if (!m) { xa_fail(X265AMD_EINVAL, "encoder_open: info SEI"); return nullptr; }
        e->headerBytes.insert(e->headerBytes.end(), sei, sei + m);
    }
    return e.release();
}

/* ---- frame-per-GPU: rows of a picture between the objects of a set (include/x265amd_encoder.h) ---- */
static PicP picByCoding(x265amd_encoder* e, uint64_t k)
{
    std::lock_guard<std::mutex> lk(e->byCodingMu);
    auto it = e->byCoding.find(k);
    return it == e->byCoding.end() ? PicP() : it->second;
}
static void rowRanges(const x265amd_encoder& e, int row, uint64_t off[3], uint64_t bytes[3])
{
    for (int k = 0; k < 3; k++)
    {
        const int sh = k ? 1 : 0, my = e.marginY >> sh, mx = e.marginX >> sh, h = e.H >> sh, rowH = 64 >> sh;
        const intptr_t st = k ? e.cstride : e.stride;
        const int y0 = row == 0 ? -my : row * rowH, y1 = row == e.ctuH - 1 ? h + my : std::min(h, (row + 1) * rowH);
        const int64_t first = (int64_t)e.org[k] + (int64_t)y0 * st - mx;
        off[k] = (uint64_t)first * sizeof(pixel);
        bytes[k] = (uint64_t)((int64_t)(y1 - y0) * st) * sizeof(pixel);
    }
}
extern "C" int x265amd_encoder_ctu_rows(const x265amd_encoder* e) { return e ? e->ctuH : -1; }
extern "C" int x265amd_encoder_stats(const x265amd_encoder* e, uint64_t* out, int n)
{
    if (!e || !out || n < 4) return xa_fail(X265AMD_EINVAL, "encoder_stats: arguments"), -1;
    out[0] = e->statPictures[0]; out[1] = e->statPictures[1]; out[2] = e->statPictures[2]; out[3] = e->statReferences;
    return 0;
}
extern "C" int x265amd_encoder_row_geometry(const x265amd_encoder* e, int row, x265amd_row_export* out)
{
    if (!e || !out || row < 0 || row >= e->ctuH) return xa_fail(X265AMD_EINVAL, "encoder_row_geometry: bad arguments"), -1;
    memset(out, 0, sizeof(*out));
    out->ctu_row = row;
    rowRanges(*e, row, out->plane_offset, out->plane_bytes);
    const size_t u0 = (size_t)row * 16 * e->w4, u1 = (size_t)std::min(e->h4, (row + 1) * 16) * e->w4;
    out->units_bytes = (u1 - u0) * sizeof(x265amd_cu_unit); out->map_offset_units = u0 * sizeof(x265amd_cu_unit);
    out->motion_bytes = (u1 - u0) * sizeof(x265amd_mv_unit); out->map_offset_motion = u0 * sizeof(x265amd_mv_unit);
    return 0;
}
extern "C" int x265amd_encoder_is_referenced(x265amd_encoder* e, uint64_t k)
{
    if (!e) return xa_fail(X265AMD_EINVAL, "encoder_is_referenced: null"), -1;
    PicP pic = picByCoding(e, k);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (k < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_is_referenced: that picture has been collected and released"), -1;
        return 2;
    }
    return pic->type != TYPE_B;         /* what DPB::prepareEncode fixes with the slice type: a plain B picture is never a reference */
}
extern "C" int x265amd_encoder_owns(const x265amd_encoder* e, uint64_t k) { return e ? (e->p.shardCount <= 1 || (int)(k % (uint64_t)e->p.shardCount) == e->p.shardRank) : 0; }
extern "C" int x265amd_encoder_export_row(x265amd_encoder* e, uint64_t codingIndex, int row, x265amd_row_export* out, int timeoutMs)
{
    if (!e || !out || row < 0 || row >= e->ctuH) return xa_fail(X265AMD_EINVAL, "encoder_export_row: bad arguments"), -1;
    if (!e->frameParallel) return xa_fail(X265AMD_EINVAL, "encoder_export_row: rows are published by objects that code pictures in parallel (frameNumThreads > 1)"), -1;
    PicP pic = picByCoding(e, codingIndex);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (codingIndex < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_export_row: that picture has been collected and released: the pump is more than eight pictures late"), -1;
        return 1;
    }
    if (!pic->owned) return xa_fail(X265AMD_EINVAL, "encoder_export_row: this object does not code that picture"), -1;
    const auto t0 = std::chrono::steady_clock::now();
    while (pic->published(row) < e->W)
    {
        if (pic->failed.load()) return xa_fail(X265AMD_EHIP, "encoder_export_row: the picture failed"), -1;
        if (timeoutMs <= 0) return 2;           /* a look, not a wait: the row is not final yet (the one-thread pump of frame_rows.py asks like this) */
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeoutMs) return xa_fail(X265AMD_EHIP, "encoder_export_row: time-out"), -1;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    /* Pic::fail() sets every counter to the end: a picture that failed before the loop was entered looks published */
    if (pic->failed.load()) return xa_fail(X265AMD_EHIP, "encoder_export_row: the picture failed"), -1;
    memset(out, 0, sizeof(*out));
    out->coding_index = codingIndex; out->ctu_row = row;
    rowRanges(*e, row, out->plane_offset, out->plane_bytes);
    for (int k = 0; k < 3; k++) out->src[k] = (const uint8_t*)pic->finalPlanes() + out->plane_offset[k];
    const size_t u0 = (size_t)row * 16 * e->w4, u1 = (size_t)std::min(e->h4, (row + 1) * 16) * e->w4;
    out->units = pic->units.data() + u0; out->units_bytes = (u1 - u0) * sizeof(x265amd_cu_unit); out->map_offset_units = u0 * sizeof(x265amd_cu_unit);
    out->motion = pic->motion.data() + u0; out->motion_bytes = (u1 - u0) * sizeof(x265amd_mv_unit); out->map_offset_motion = u0 * sizeof(x265amd_mv_unit);
    return 0;
}
extern "C" int x265amd_encoder_import_row(x265amd_encoder* e, const x265amd_row_export* in)
{
    if (!e || !in || in->ctu_row < 0 || in->ctu_row >= e->ctuH || !in->src[0] || !in->src[1] || !in->src[2] || !in->units || !in->motion) return xa_fail(X265AMD_EINVAL, "encoder_import_row: bad arguments"), -1;
    PicP pic = picByCoding(e, in->coding_index);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (in->coding_index < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_import_row: that picture has been collected and released"), -1;
        return 1;
    }
    if (pic->owned) return xa_fail(X265AMD_EINVAL, "encoder_import_row: this object codes that picture itself"), -1;
    uint64_t off[3], bytes[3];
    rowRanges(*e, in->ctu_row, off, bytes);
    for (int k = 0; k < 3; k++)
        if (off[k] != in->plane_offset[k] || bytes[k] != in->plane_bytes[k]) return xa_fail(X265AMD_EINVAL, "encoder_import_row: the row comes from a picture of another geometry"), -1;
    if (in->map_offset_units + in->units_bytes > pic->units.size() * sizeof(x265amd_cu_unit) || in->map_offset_motion + in->motion_bytes > pic->motion.size() * sizeof(x265amd_mv_unit))
        return xa_fail(X265AMD_EINVAL, "encoder_import_row: map range"), -1;
    xa_thread_device();
    uint8_t* dst = (uint8_t*)(pic->dFin ? pic->dFin : pic->dRec);
    {
        /* a stream of the object's own, waited for: a device-to-device hipMemcpy may return before the bytes have landed, and the counter below lets readers in */
        std::lock_guard<std::mutex> lk(e->importMu);
        if (!e->importStream && hipStreamCreateWithFlags(&e->importStream, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_import_row: stream"), -1;
        for (int k = 0; k < 3; k++)
            if (hipMemcpyAsync(dst + off[k], in->src[k], bytes[k], hipMemcpyDefault, e->importStream) != hipSuccess) { pic->fail(); return xa_fail(X265AMD_EHIP, "encoder_import_row: copy"), -1; }
        if (hipStreamSynchronize(e->importStream) != hipSuccess) { pic->fail(); return xa_fail(X265AMD_EHIP, "encoder_import_row: copy"), -1; }
    }
    memcpy((uint8_t*)pic->units.data() + in->map_offset_units, in->units, in->units_bytes);
    memcpy((uint8_t*)pic->motion.data() + in->map_offset_motion, in->motion, in->motion_bytes);
    xa_devmap_push_rows(pic->motion.data(), pic->units.data(), e->w4, in->ctu_row * 16, std::min(e->h4, (in->ctu_row + 1) * 16));      /* the field's mirror in device memory */
    pic->publish(in->ctu_row, e->W);
    { std::lock_guard<std::mutex> lk(pic->mu); pic->importedRows++; }
    pic->cv.notify_all();
    return 0;
}

extern "C" void x265amd_encoder_close(x265amd_encoder* e) { delete e; }

/* splits a byte stream of NAL units behind 4-byte start codes into x265_nal records (payload includes the start code, as the reference's do) */
static void splitNals(std::vector<uint8_t>& bytes, std::vector<x265amd_nal>& nals)
{
    nals.clear();
    size_t start = 0;
    for (size_t i = 4; i + 4 <= bytes.size() + 1; i++)
    {
        const bool sc = i + 4 <= bytes.size() && !bytes[i] && !bytes[i + 1] && !bytes[i + 2] && bytes[i + 3] == 1;
        if (sc || i + 4 > bytes.size())
        {
            const size_t end = sc ? i : bytes.size();
            x265amd_nal n;
            n.type = (bytes[start + 4] >> 1) & 63; n.sizeBytes = (uint32_t)(end - start); n.payload = bytes.data() + start;
            nals.push_back(n);
            start = end;
            if (!sc) break;
        }
    }
}

extern "C" int x265amd_encoder_headers(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal)
{
    if (!e || !ppNal || !piNal) return xa_fail(X265AMD_EINVAL, "encoder_headers: null argument");
    e->outBytes = e->headerBytes;
    splitNals(e->outBytes, e->nals);
    *ppNal = e->nals.data(); *piNal = (uint32_t)e->nals.size();
    return (int)e->outBytes.size();
}

/* ---- pictures ---- */
int x265amd_encoder::uploadPicture(const x265amd_picture* in, Pic& pic)
{
    /* the picture area of each plane with its margins filled by edge replication (PicYuv::copyFromPicture pads, extendPicBorder) */
    staging.assign(picElems, 0);
    for (int k = 0; k < 3; k++)
    {
        const int w = k ? W / 2 : W, h = k ? H / 2 : H, mx = k ? marginX / 2 : marginX, my = k ? marginY / 2 : marginY;
        const intptr_t st = k ? cstride : stride;
        if (!in->planes[k] || in->stride[k] < (int)(w * sizeof(pixel))) return xa_fail(X265AMD_EINVAL, "encoder_encode: input plane");
        pixel* base = staging.data() + org[k];
        for (int y = 0; y < h; y++)
        {
            const pixel* src = (const pixel*)((const uint8_t*)in->planes[k] + (size_t)y * in->stride[k]);
            pixel* row = base + (intptr_t)y * st;
            memcpy(row, src, sizeof(pixel) * w);
            for (int x = 1; x <= mx; x++) { row[-x] = row[0]; row[w - 1 + x] = row[w - 1]; }
        }
        for (int y = 1; y <= my; y++)
        {
            memcpy(base + (intptr_t)(-y) * st - mx, base - mx, sizeof(pixel) * (w + 2 * mx));
            memcpy(base + (intptr_t)(h - 1 + y) * st - mx, base + (intptr_t)(h - 1) * st - mx, sizeof(pixel) * (w + 2 * mx));
        }
    }
    if (xa_scratch_alloc((void**)&pic.dSrc, picElems * sizeof(pixel)) != hipSuccess || xa_scratch_alloc((void**)&pic.dRec, picElems * sizeof(pixel)) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if (hipMemcpy(pic.dSrc, staging.data(), picElems * sizeof(pixel), hipMemcpyHostToDevice) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: upload");
    /* the frame tasks run on their own non-blocking streams: make sure the picture is in place before one can start */
    if (hipMemset(pic.dRec, 0, picElems * sizeof(pixel)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    if (frameParallel && p.bEnableSAO)
    {
        if (xa_scratch_alloc((void**)&pic.dFin, picElems * sizeof(pixel)) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
        if (hipMemset(pic.dFin, 0, picElems * sizeof(pixel)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    }
    return 0;
}

/* ---- the lookahead's slice-type decision with scene-cut detection (param.scenecutThreshold > 0, bFrameAdaptive 0) ----
 * Lowres::init + LookaheadTLD::lowresIntraEstimate for every picture handed in (lowres.cpp:337-403, slicetype.cpp:715-824): x265amd_lowres_init,
 * x265amd_lowres_intra_costs; costEst[0][0] = the intra costs of the blocks that are not on the picture's edge. */
int x265amd_encoder::lowresInit(Pic& pic)
{
    if (xa_scratch_alloc((void**)&pic.dLowres, lowPlaneElems * 4 * sizeof(pixel)) != hipSuccess || xa_scratch_alloc((void**)&pic.dIntraCost, (size_t)lowCuW * lowCuH * 4 + (size_t)lowCuW * lowCuH) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if (hipMemsetAsync(pic.dLowres, 0, lowPlaneElems * 4 * sizeof(pixel), laStream) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    pixel* planes[4];
    for (int k = 0; k < 4; k++) planes[k] = pic.dLowres + (size_t)k * lowPlaneElems + lowOrg;
    int rc = x265amd_lowres_init(laStream, pic.dSrc + org[0], stride, lowW, lowH, planes, lowStride, marginX, marginY);
    if (rc != X265AMD_OK) return rc;
    const int lambda = X265AMD_DEPTH > 8 ? 16 : 1;          /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP], X265_LOOKAHEAD_QP = 12 + 6 * (depth - 8) (common.h:213) */
    uint8_t* dMode = (uint8_t*)(pic.dIntraCost + (size_t)lowCuW * lowCuH);
    rc = x265amd_lowres_intra_costs(laStream, planes[0], lowStride, lowCuW, lowCuH, lambda, pic.dIntraCost, dMode);
    if (rc != X265AMD_OK) return rc;
    /* the picture's sums for the weight analysis are measured in front of the one wait of this function */
    static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");          /* debugging aid: letters s / l / p switch the sums, the lookahead's analysis, the slice's analysis off */
    const bool sums = (p.bEnableWeightedPred || p.bEnableWeightedBiPred) && !(dbgWp && strchr(dbgWp, 's'));
    if (sums && !aqOn)
    {
        /* LookaheadTLD::calcAdaptiveQuantFrame with AQ off (slicetype.cpp:507-513): acEnergyCu over every 16x16 block for Lowres::wp_sum / wp_ssd, then :678-700 */
        const int bw = (W + 15) / 16, bh = (H + 15) / 16;
        if (!wpEnergy && (xa_scratch_alloc(&wpEnergy, (size_t)bw * bh * 4) != hipSuccess || xa_scratch_alloc(&wpSums, 6 * 8) != hipSuccess || xa_mapped_alloc(&wpSumsHost, 6 * 8, true) != hipSuccess ||
                          xa_mapped_alloc(&wpMvs, (size_t)lowCuW * lowCuH * 4, false) != hipSuccess))
            return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
        const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
        rc = x265amd_aq_energy(laStream, srcP, stride, cstride, W, H, 16, (uint32_t*)wpEnergy, (uint64_t*)wpSums);
        if (rc == X265AMD_OK && hipMemcpyAsync(wpSumsHost, wpSums, 6 * 8, hipMemcpyDeviceToHost, laStream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "encoder_encode: picture sums");
        if (rc != X265AMD_OK) return rc;
    }
    if (aqOn && (rc = adaptiveQuant(pic)) != X265AMD_OK) return rc;
    std::vector<int32_t> ic((size_t)lowCuW * lowCuH);
    if (hipMemcpyAsync(ic.data(), pic.dIntraCost, ic.size() * 4, hipMemcpyDeviceToHost, laStream) != hipSuccess || hipStreamSynchronize(laStream) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: lowres intra costs");
    if (aqOn)
    {
        /* the rest of calcAdaptiveQuantFrame (slicetype.cpp:513-640) on the block energies that have arrived with the intra costs */
        const int bw = (W + 15) / 16, bh = (H + 15) / 16, nb = bw * bh;
        pic.qpAqOffset.assign((size_t)nb, 0.0); pic.qpCuTreeOffset.assign((size_t)nb, 0.0); pic.invQscale.assign((size_t)nb, 256);
        rc = x265amd_aq_offsets((const uint32_t*)aqEnergyHost, nb, lowCuW * lowCuH, p.aqMode, p.aqStrength, 1.0, 16, pic.qpAqOffset.data(), pic.qpCuTreeOffset.data(), pic.invQscale.data());
        if (rc != X265AMD_OK) return xa_fail(rc, "encoder_encode: adaptive quantisation");
        if (p.cuTree) { pic.intraCostHost = ic; pic.propagateCost.assign((size_t)lowCuW * lowCuH, 0); }
    }
    int64_t est = 0;
    const bool all = lowCuW <= 2 || lowCuH <= 2;
    for (int y = 0; y < lowCuH; y++)
        for (int x = 0; x < lowCuW; x++)
            if (all || (x > 0 && x < lowCuW - 1 && y > 0 && y < lowCuH - 1)) est += ic[(size_t)y * lowCuW + x];
    pic.costEst[0] = est;
    if (sums)
    {
        uint64_t wp[6];
        memcpy(wp, aqOn ? (const void*)((const char*)aqEnergyHost + (size_t)((W + 15) / 16) * ((H + 15) / 16) * 4) : wpSumsHost, sizeof(wp));
        const int maxCol = ((W + 8) >> 4) << 4, maxRow = ((H + 8) >> 4) << 4;
        const int width[3] = { maxCol, maxCol >> 1, maxCol >> 1 }, height[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
        {
            const uint64_t sum = wp[i], ssd = wp[3 + i];
            pic.wpSum[i] = sum;
            pic.wpSsd[i] = ssd - (sum * sum + (uint64_t)((width[i] * height[i]) / 2)) / (uint64_t)(width[i] * height[i]);
        }
    }
    return X265AMD_OK;
}

/* LookaheadTLD::calcAdaptiveQuantFrame's block loop (slicetype.cpp:560-600): acEnergyCu of every 16x16 block (luma + both chroma blocks) and the picture's sums for the weight
 * analysis, enqueued on the lookahead's stream with their read-back; the caller waits for the stream once (lowresInit) and turns the energies into offsets */
int x265amd_encoder::adaptiveQuant(Pic& pic)
{
    const int bw = (W + 15) / 16, bh = (H + 15) / 16;
    const size_t nb = (size_t)bw * bh;
    if (!aqEnergy && (xa_scratch_alloc(&aqEnergy, nb * 4) != hipSuccess || xa_scratch_alloc(&aqSums, 6 * 8) != hipSuccess || xa_mapped_alloc(&aqEnergyHost, nb * 4 + 6 * 8, true) != hipSuccess))
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if ((p.bEnableWeightedPred || p.bEnableWeightedBiPred) && !wpMvs && xa_mapped_alloc(&wpMvs, (size_t)lowCuW * lowCuH * 4, false) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    int rc = x265amd_aq_energy(laStream, srcP, stride, cstride, W, H, 16, (uint32_t*)aqEnergy, (uint64_t*)aqSums);
    if (rc != X265AMD_OK) return rc;
    if (hipMemcpyAsync(aqEnergyHost, aqEnergy, nb * 4, hipMemcpyDeviceToHost, laStream) != hipSuccess ||
        hipMemcpyAsync((char*)aqEnergyHost + nb * 4, aqSums, 6 * 8, hipMemcpyDeviceToHost, laStream) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: block energies");
    return X265AMD_OK;
}

namespace {
/* bs_size_ue / bs_size_se (common/bitstream.h:94-136) */
inline int bitSizeOf(unsigned v) { int n = 1; while (v > 1) { v >>= 1; n += 2; } return n; }
inline int bsSizeUe(unsigned val) { return bitSizeOf(val + 1); }
inline int bsSizeSe(int val) { int tmp = 1 - val * 2; if (tmp < 0) tmp = val * 2; return tmp < 256 ? bitSizeOf((unsigned)tmp) : bitSizeOf((unsigned)(tmp >> 8)) + 16; }
/* weight_pp_c's arguments for a WeightParam as weightCostLuma / weightCost pass them */
inline x265amd_weight_cand weightCand(int scale, int denom, int offset)
{
    const int correction = 14 - X265AMD_DEPTH;
    x265amd_weight_cand c;
    c.present = 1; c.w0 = scale; c.round = (denom ? 1 << (denom - 1) : 0) << correction; c.shift = denom + correction; c.offset = offset << (X265AMD_DEPTH - 8);
    return c;
}
}

/* LookaheadTLD::weightsAnalyse (slicetype.cpp:879-978) before a list-0 search of `fenc` against `ref`: the early exit when the two do not differ in mean or variance; else the
 * unweighted cost against one candidate (scale from the variances, offset from the means), a smaller denominator if the scale is even, and the 0.998 test.  weighted: the
 * reference's four planes are weighted for the search (scale / 2^denom, offset) */
/* (in two halves, so that the measurements of every search of a batch go out as one launch: the guess -- false: the early exit, no weight --, then the decision from
 * the two costs) */
bool x265amd_encoder::lookaheadWeightGuess(Pic& fenc, Pic& ref, LaWeight& g)
{
    static const float epsilon = 1.f / 128.f;
    float guessScale, fencMean, refMean;
    if (fenc.wpSsd[0] && ref.wpSsd[0]) guessScale = sqrtf((float)fenc.wpSsd[0] / ref.wpSsd[0]);
    else guessScale = 1.0f;
    fencMean = (float)fenc.wpSum[0] / (lowH * lowW) / (1 << (X265AMD_DEPTH - 8));
    refMean = (float)ref.wpSum[0] / (lowH * lowW) / (1 << (X265AMD_DEPTH - 8));
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon) return false;
    {
        /* WeightParam::setFromWeightAndOffset((int)(guessScale * 128 + 0.5f), 0, 7, true) (slice.h:304-316) */
        int w = (int)(guessScale * 128 + 0.5f), d = 7;
        while (d > 0 && w > 127) { d--; w >>= 1; }
        w = std::min(w, 127);
        g.mindenom = d; g.minscale = w;
    }
    g.curScale = g.minscale;
    g.curOffset = (int)(fencMean - refMean * g.curScale / (1 << g.mindenom) + 0.5f);
    if (g.curOffset < -128 || g.curOffset > 127)
    {
        g.curOffset = std::max(-128, std::min(127, g.curOffset));
        g.curScale = (int)((1 << g.mindenom) * (fencMean - g.curOffset) / refMean + 0.5f);
        g.curScale = std::max(0, std::min(127, g.curScale));
    }
    return true;
}
void x265amd_encoder::lookaheadWeightDecide(const LaWeight& g, const uint32_t costs[2], bool& weighted, int& scale, int& denom, int& offset)
{
    weighted = false;
    int minoff = 0, minscale = g.minscale, mindenom = g.mindenom;
    unsigned int minscore = costs[0], origscore = costs[0];
    int found = 0;
    if (!minscore) return;
    const unsigned int sc = costs[1];
    if (sc < minscore) { minscore = sc; minscale = g.curScale; minoff = g.curOffset; found = 1; }
    if (mindenom > 0 && !(minscale & 1))
    {
        const int idx = minscale ? __builtin_ctz((unsigned)minscale) : 32;
        const int shift = std::min(idx, mindenom);
        mindenom -= shift; minscale >>= shift;
    }
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) return;
    weighted = true; scale = minscale; denom = mindenom; offset = minoff;
}
/* weightAnalyse (weightPrediction.cpp:222-540) for a P picture (list 0) or, with weighted bi-prediction, a B picture (both lists): the first reference of each list.  The chroma
 * denominator that fits both chroma scale guesses; per plane: the early exit, else the reference motion compensated with the lookahead's vectors of that distance (mcLuma on the
 * lowres planes, mcChroma on the SOURCE chroma planes) against every candidate scale (+-4 around the guess) and offset (+-2 around the mean's), each with the slice header's cost,
 * a smaller luma denominator if the scale is even, the 0.998 test.  Without a luma weight chroma is not looked at.  Leaves slice.m_weightPredTable in pic.wp and pic.weighted
 * (some reference carries a weight). */
int x265amd_encoder::sliceWeights(Pic& pic)
{
    pic.weighted = false;
    memset(pic.wp, 0, sizeof(pic.wp));
    const int numDirs = isBType(pic.type) ? 2 : 1;
    const float epsilon = 1.f / 128.f;
    const int w16 = ((W + 15) >> 4) << 4, h16 = ((H + 15) >> 4) << 4;
    int numpixels[3];
    numpixels[0] = w16 * h16; numpixels[1] = numpixels[2] = numpixels[0] >> 2;
    auto setW = [](x265amd_weight& w, bool present, int scale, int denom, int off) { w.present = present; w.w = (int16_t)scale; w.denom = (uint8_t)denom; w.o = (int16_t)off; };
    int chromaDenom = 7, lumaDenom = 7;
    const int lambda = X265AMD_DEPTH > 8 ? 16 : 1;          /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP] */
    for (int list = 0; list < numDirs; list++)
    {
        x265amd_weight* weights = pic.wp[list][0];
        Pic& ref = *pic.lists[list][0];
        const int diffPoc = abs(pic.poc - ref.poc);
        float guessScale[3], fencMean[3], refMean[3];
        for (int plane = 0; plane < 3; plane++)
        {
            setW(weights[plane], false, 1, 0, 0);
            const uint64_t fencVar = pic.wpSsd[plane] + !ref.wpSsd[plane], refVar = ref.wpSsd[plane] + !ref.wpSsd[plane];
            guessScale[plane] = sqrt((float)fencVar / refVar);
            fencMean[plane] = (float)pic.wpSum[plane] / (numpixels[plane]) / (1 << (X265AMD_DEPTH - 8));
            refMean[plane] = (float)ref.wpSum[plane] / (numpixels[plane]) / (1 << (X265AMD_DEPTH - 8));
        }
        while (!list && chromaDenom > 0)
        {
            const float thresh = 127.f / (1 << chromaDenom);
            if (guessScale[1] < thresh && guessScale[2] < thresh) break;
            chromaDenom--;
        }
        setW(weights[1], false, 1 << chromaDenom, chromaDenom, 0);
        setW(weights[2], false, 1 << chromaDenom, chromaDenom, 0);
        void* dMvs = nullptr;           /* the field of the luma analysis serves the chroma planes too */
        for (int plane = 0; plane < 3; plane++)
        {
            const int denom = plane ? chromaDenom : lumaDenom;
            if (plane && !weights[0].present) break;
            if (fabsf(refMean[plane] - fencMean[plane]) < 0.5f && fabsf(1.f - guessScale[plane]) < epsilon) { setW(weights[plane], false, 1 << denom, denom, 0); continue; }
            if (plane)
            {
                const int scale = std::max(0, std::min(255, (int)(guessScale[plane] * (1 << denom) + 0.5f)));
                if (scale > 127) continue;
                weights[plane].w = (int16_t)scale;
            }
            else
            {
                /* WeightParam::setFromWeightAndOffset(w, 0, denom, bNormalize = !list) (slice.h:304-316) */
                int w = (int)(guessScale[plane] * (1 << denom) + 0.5f), d = denom;
                while (!list && d > 0 && w > 127) { d--; w >>= 1; }
                w = std::min(w, 127);
                weights[plane].o = 0; weights[plane].denom = (uint8_t)d; weights[plane].w = (int16_t)w;
            }
            int mindenom = weights[plane].denom, minscale = weights[plane].w, minoff = 0;
            if (!plane && diffPoc <= p.bframes + 1)
            {
                const std::vector<int16_t>& f = list ? pic.lowMvs1[diffPoc < 18 ? diffPoc : 0] : pic.lowMvs[diffPoc < 18 ? diffPoc : 0];
                if (diffPoc < 18 && !f.empty())
                {
                    /* (a record the host writes in place: no copy from pageable memory) */
                    if (!wpMvs || f.size() * 2 > (size_t)lowCuW * lowCuH * 4) return xa_fail(X265AMD_EHIP, "encoder_encode: lowres vectors");
                    dMvs = wpMvs;
                    memcpy(dMvs, f.data(), f.size() * 2);
                }
            }
            /* the candidates in the order the reference tries them */
            struct Cand { int scale, off, startOffset, iter; };
            std::vector<Cand> order;
            std::vector<x265amd_weight_cand> cands(1);
            memset(&cands[0], 0, sizeof(cands[0]));
            const int startScale = std::max(0, std::min(127, minscale - 4)), endScale = std::max(0, std::min(127, minscale + 4));
            for (int scale = startScale; scale <= endScale; scale++)
            {
                const int deltaWeight = scale - (1 << mindenom);
                if (deltaWeight > 127 || deltaWeight <= -128) continue;
                int curScale = scale;
                int curOffset = (int)(fencMean[plane] - refMean[plane] * curScale / (1 << mindenom) + 0.5f);
                if (curOffset < -128 || curOffset > 127)
                {
                    curOffset = std::max(-128, std::min(127, curOffset));
                    curScale = (int)((1 << mindenom) * (fencMean[plane] - curOffset) / refMean[plane] + 0.5f);
                    curScale = std::max(0, std::min(127, curScale));
                }
                const int startOffset = std::max(-128, std::min(127, curOffset - 2)), endOffset = std::max(-128, std::min(127, curOffset + 2));
                for (int off = startOffset; off <= endOffset; off++) { order.push_back({ curScale, off, startOffset, scale }); cands.push_back(weightCand(curScale, mindenom, off)); }
            }
            std::vector<uint32_t> costs(cands.size(), 0);
            int rc;
            if (!plane)
            {
                const pixel* refPlanes[4];
                for (int t = 0; t < 4; t++) refPlanes[t] = ref.dLowres + (size_t)t * lowPlaneElems + lowOrg;
                rc = x265amd_lowres_weight_costs(laStream, pic.dLowres + lowOrg, refPlanes, (const int16_t*)dMvs, pic.dIntraCost, lowStride, lowW, lowH, cands.data(), (int)cands.size(), costs.data());
            }
            else
            {
                if (!pic.dSrc || !ref.dSrc) return xa_fail(X265AMD_EHIP, "encoder_encode: weight analysis without the reference's source picture");
                const int cw = ((W >> 4) << 4) >> 1, chh = ((H >> 4) << 4) >> 1;
                rc = x265amd_chroma_weight_costs(laStream, pic.dSrc + org[plane], ref.dSrc + org[plane], (const int16_t*)dMvs, cstride, cw, chh, lowCuW, lowCuH, cands.data(), (int)cands.size(), costs.data());
            }
            if (rc != X265AMD_OK) return rc;
            const uint32_t origscore = costs[0];
            if (!origscore) { setW(weights[plane], false, 1 << denom, denom, 0); continue; }
            uint32_t minscore = origscore;
            bool bFound = false;
            for (size_t k = 0; k < order.size(); k++)
            {
                const Cand& c = order[k];
                /* sliceHeaderCost(&wsp, lambda, !!plane): four times the lambda for chroma (analysed at full resolution), the denominator counted twice for luma */
                const int lam = plane ? lambda * 4 : lambda;
                const int hdr = lam * (10 + bsSizeUe((unsigned)mindenom) * (plane ? 1 : 2) + 2 * (bsSizeSe(c.scale) + bsSizeSe(c.off)));
                const uint32_t sc = costs[k + 1] + (uint32_t)hdr;
                if (sc < minscore) { minscore = sc; minscale = c.scale; minoff = c.off; bFound = true; }
                /* "Don't check any more offsets if the previous one had a lower cost than the current one": the rest of this scale's offsets are skipped */
                if (minoff == c.startOffset && c.off != c.startOffset)
                    while (k + 1 < order.size() && order[k + 1].iter == c.iter) k++;
            }
            if (!(plane || list) && mindenom > 0 && !(minscale & 1))
            {
                const int idx = minscale ? __builtin_ctz((unsigned)minscale) : 32;
                const int shift = std::min(idx, mindenom);
                mindenom -= shift; minscale >>= shift;
            }
            if (!bFound || (minscale == (1 << mindenom) && minoff == 0) || (float)minscore / origscore > 0.998f) setW(weights[plane], false, 1 << denom, denom, 0);
            else setW(weights[plane], true, minscale, mindenom, minoff);
        }
        if (weights[0].present && weights[1].present != weights[2].present)
        {
            /* "make sure both chroma channels match" */
            if (weights[1].present) weights[2] = weights[1]; else weights[1] = weights[2];
        }
        lumaDenom = weights[0].denom; chromaDenom = weights[1].denom;
        for (size_t r = 1; r < pic.lists[list].size(); r++)
        {
            setW(pic.wp[list][r][0], false, 1 << lumaDenom, lumaDenom, 0);
            setW(pic.wp[list][r][1], false, 1 << chromaDenom, chromaDenom, 0);
            setW(pic.wp[list][r][2], false, 1 << chromaDenom, chromaDenom, 0);
        }
        for (int plane = 0; plane < 3; plane++) pic.weighted |= weights[plane].present != 0;
    }
    pic.lumaDenom = pic.wp[0][0][0].denom; pic.chromaDenom = pic.wp[0][0][1].denom;         /* what pred_weight_table() codes once: the first reference's (entropy.cpp:1376-1387) */
    const bool wpLog = getenv("X265AMD_WP_LOG") != nullptr;         /* (read per picture: a test switches it on for one encode) */
    if (wpLog && pic.weighted)
    {
        /* the reference's --log-level full line */
        char buf[512]; int n = snprintf(buf, sizeof(buf), "poc: %d weights:", pic.poc);
        for (int list = 0; list < numDirs; list++)
        {
            const x265amd_weight* w = pic.wp[list][0];
            if (!(w[0].present || w[1].present || w[2].present)) continue;
            n += snprintf(buf + n, sizeof(buf) - n, " [L%d:R0 ", list);
            if (w[0].present) n += snprintf(buf + n, sizeof(buf) - n, "Y{%d/%d%+d}", w[0].w, 1 << w[0].denom, w[0].o);
            if (w[1].present) n += snprintf(buf + n, sizeof(buf) - n, "U{%d/%d%+d}", w[1].w, 1 << w[1].denom, w[1].o);
            if (w[2].present) n += snprintf(buf + n, sizeof(buf) - n, "V{%d/%d%+d}", w[2].w, 1 << w[2].denom, w[2].o);
            n += snprintf(buf + n, sizeof(buf) - n, "]");
        }
        fprintf(stderr, "x265amd: %s\n", buf);
    }
    return X265AMD_OK;
}

/* CostEstimateGroup::singleCost(p0, p1, b = p1) -> estimateFrameCost (slicetype.cpp:3882-4075) for a P candidate `dist` pictures behind its reference: the block
 * loop is x265amd_lowres_frame_cost (motion search of list 0 included: every (picture, distance) pair is estimated once); costEst / intraMbs are the sums over the
 * blocks that are not on the picture's edge (estimateCUCost's tail, :4220-4248) */
int x265amd_encoder::frameCostP(Pic& b, Pic& ref, int dist)
{
    int64_t score;
    return frameCostAt(b, ref, nullptr, dist, 0, score);
}

/* CostEstimateGroup::estimateFrameCost (slicetype.cpp:3975-4075) for candidate `fenc` against `ref0` d0 pictures before it and, for a B estimate, `ref1` d1 pictures behind
 * it: the searches a field still lacks run inside the block loop (bDoSearch), fields that exist are read again; the sum over the blocks that are not on the picture's
 * edge, scaled by 100 / (130 + bFrameBias) for a B estimate; intra blocks are counted for P estimates only */
int x265amd_encoder::frameCostAt(Pic& fenc, Pic& ref0, Pic* ref1, int d0, int d1, int64_t& score)
{
    if (d0 < 1 || d0 > 17 || d1 < 0 || d1 > 17 || (d1 > 0) != (ref1 != nullptr)) return xa_fail(X265AMD_EINVAL, "encoder: lookahead distance");
    if (fenc.cost2[d0][d1] >= 0) { score = fenc.cost2[d0][d1]; return X265AMD_OK; }
    /* the reference makes this estimate now, with the searches its fields still lack: what was searched ahead becomes the picture's */
    if (fenc.lowMvs[d0].empty() && !fenc.specMvs[d0].empty()) { fenc.lowMvs[d0].swap(fenc.specMvs[d0]); fenc.lowMvc[d0].swap(fenc.specMvc[d0]); }
    if (d1 > 0 && fenc.lowMvs1[d1].empty() && !fenc.specMvs1[d1].empty()) { fenc.lowMvs1[d1].swap(fenc.specMvs1[d1]); fenc.lowMvc1[d1].swap(fenc.specMvc1[d1]); }
    if (fenc.specCost2[d0][d1] >= 0 && !fenc.lowMvs[d0].empty() && (d1 == 0 || !fenc.lowMvs1[d1].empty()))
    {
        score = fenc.cost2[d0][d1] = fenc.specCost2[d0][d1];
        {
            /* ... and its block costs (cuTree) */
            const int key = d0 * 32 + d1;
            auto it = fenc.dSpecLc.find(key);
            if (it != fenc.dSpecLc.end()) { fenc.dropLc(fenc.dLc, key); fenc.dLc[key] = it->second; fenc.dSpecLc.erase(it); fenc.lcHost.erase(key); }
        }
        if (d1 == 0) { fenc.costEst[d0] = score; fenc.intraMbs[d0] = fenc.specIntraMbs[d0]; }
        return X265AMD_OK;
    }
    std::vector<CostJob> one(1);
    one[0].fenc = &fenc; one[0].ref0 = &ref0; one[0].ref1 = ref1; one[0].d0 = d0; one[0].d1 = d1;
    const int rc = frameCostMany(one);
    score = fenc.cost2[d0][d1];
    return rc;
}

void x265amd_encoder::laFieldPut(const void* key, void* mv, void* mc)
{
    auto it = laFields.find(key);
    if (it != laFields.end()) { laBufPut(it->second.mv); laBufPut(it->second.mc); it->second = DevField{ mv, mc, ++laFieldClock }; }
    else laFields.emplace(key, DevField{ mv, mc, ++laFieldClock });
}
/* (called when nothing of the lookahead's is in flight: behind frameCostMany's wait) */
void x265amd_encoder::laFieldsTrim()
{
    {
        /* fields of pictures that have gone since the last look */
        std::vector<const void*> dead;
        { std::lock_guard<std::mutex> lk(laPool->mu); dead.swap(laPool->deadFields); }
        for (const void* k : dead)
        {
            auto it = laFields.find(k);
            if (it != laFields.end()) { laBufPut(it->second.mv); laBufPut(it->second.mc); laFields.erase(it); }
        }
    }
    if (laFields.size() <= LA_FIELDS_MAX) return;
    std::vector<uint64_t> ages;
    for (auto& f : laFields) ages.push_back(f.second.used);
    std::nth_element(ages.begin(), ages.begin() + ages.size() / 4, ages.end());
    const uint64_t cut = ages[ages.size() / 4];
    for (auto it = laFields.begin(); it != laFields.end();)
        if (it->second.used < cut) { laBufPut(it->second.mv); laBufPut(it->second.mc); it = laFields.erase(it); } else ++it;
}
void x265amd_encoder::laFieldsFree()
{
    for (auto& f : laFields) { laBufPut(f.second.mv); laBufPut(f.second.mc); }
    laFields.clear();
}

/* Independent estimates side by side: every job on one of a handful of streams (the block loop of an estimate is a few dozen wavefronts chained row to row -- latency,
 * not throughput: a dozen of them overlap on the device), one wait for all, then the host sums.  Jobs of one call must not share a motion field they search or a cost
 * they fill (the callers' batches are by (picture, distance) pairs). */
int x265amd_encoder::frameCostMany(std::vector<CostJob>& jobs)
{
    if (jobs.empty()) return X265AMD_OK;
    laFieldsTrim();             /* (nothing of the lookahead's is in flight here either) */
    const size_t ncu = (size_t)lowCuW * lowCuH;
    int rc = X265AMD_OK;
    const auto tb0 = std::chrono::steady_clock::now();
    struct Tm { x265amd_encoder* e; std::chrono::steady_clock::time_point t0; size_t n; ~Tm() { const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); if (n > 1) { e->laBatchMs += ms; e->laBatches++; } else { e->laSingleMs += ms; e->laSingles++; } } } tm_{ this, tb0, jobs.size() };
    std::vector<x265amd_lowres_cost_job> kj(jobs.size());
    size_t issued = 0;
    /* a field that exists is read where its device copy lies (laFields); one that has none (evicted) is uploaded once and entered */
    auto shared = [&](const std::vector<int16_t>& mv, const std::vector<int32_t>& mc, void*& dMv, void*& dMc) -> bool {
        auto it = laFields.find(mv.data());
        if (it == laFields.end())
        {
            void* a = laBuf(); void* b = laBuf();
            if (!a || !b) { laBufPut(a); laBufPut(b); return false; }
            laFieldPut(mv.data(), a, b);
            it = laFields.find(mv.data());
            if (hipMemcpyAsync(a, mv.data(), ncu * 4, hipMemcpyHostToDevice, laStream) != hipSuccess || hipMemcpyAsync(b, mc.data(), ncu * 4, hipMemcpyHostToDevice, laStream) != hipSuccess)
            { laBufPut(a); laBufPut(b); laFields.erase(mv.data()); return false; }           /* (no entry for a field that did not arrive) */
        }
        it->second.used = ++laFieldClock;
        dMv = it->second.mv; dMc = it->second.mc;
        return true;
    };
    auto tph = std::chrono::steady_clock::now();
    auto phase = [&](int i) { const auto t = std::chrono::steady_clock::now(); laPhaseMs[i] += std::chrono::duration<double, std::milli>(t - tph).count(); tph = t; };
    /* first what every estimate searches, and for the list-0 searches the lookahead's weight guess: their measurements (two candidates each) go out as ONE launch */
    std::vector<LaWeight> lw(jobs.size());
    std::vector<int> wAt(jobs.size(), -1);
    std::vector<x265amd_weight_cost_job> wj;
    for (size_t k = 0; k < jobs.size(); k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        if (!j.spec && (!j.whole || laNumSlices <= 1))
        {
            /* made for good now: a field searched ahead of its time is the one this estimate would search (with cooperative slices only if this estimate is one of those
             * the reference makes in slices too: its batch searches whole pictures) */
            if (fenc.lowMvs[j.d0].empty() && !fenc.specMvs[j.d0].empty()) { fenc.lowMvs[j.d0].swap(fenc.specMvs[j.d0]); fenc.lowMvc[j.d0].swap(fenc.specMvc[j.d0]); }
            if (j.d1 > 0 && fenc.lowMvs1[j.d1].empty() && !fenc.specMvs1[j.d1].empty()) { fenc.lowMvs1[j.d1].swap(fenc.specMvs1[j.d1]); fenc.lowMvc1[j.d1].swap(fenc.specMvc1[j.d1]); }
        }
        /* (an estimate made ahead of its time reads and fills the fields made ahead of their time as well as the picture's own) */
        const bool have0 = !fenc.lowMvs[j.d0].empty() || (j.spec && !fenc.specMvs[j.d0].empty()), have1 = j.d1 > 0 && (!fenc.lowMvs1[j.d1].empty() || (j.spec && !fenc.specMvs1[j.d1].empty()));
        j.search0 = !have0; j.search1 = j.d1 > 0 && !have1;
        static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");
        if (p.bEnableWeightedPred && j.search0 && !(dbgWp && strchr(dbgWp, 'l')) && lookaheadWeightGuess(fenc, *j.ref0, lw[k]))
        {
            x265amd_weight_cost_job w;
            memset(&w, 0, sizeof(w));
            w.d_fenc = fenc.dLowres + lowOrg; w.d_intra_cost = fenc.dIntraCost;
            for (int t = 0; t < 4; t++) w.d_ref[t] = j.ref0->dLowres + (size_t)t * lowPlaneElems + lowOrg;
            w.cands[1] = weightCand(lw[k].curScale, lw[k].mindenom, lw[k].curOffset);
            wAt[k] = (int)wj.size();
            wj.push_back(w);
        }
        laJobs++; laSearches += (j.search0 ? 1 : 0) + (j.search1 ? 1 : 0);
    }
    std::vector<uint32_t> wCosts(2 * wj.size() + 2);
    laWeightJobs += wj.size();
    if (!wj.empty()) rc = x265amd_lowres_weight_costs_many(laStream, wj.data(), (int)wj.size(), lowStride, lowW, lowH, wCosts.data());
    for (size_t k = 0; k < jobs.size() && rc == X265AMD_OK; k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        bool weighted = false; int wScale = 0, wDenom = 0, wOffset = 0;
        if (wAt[k] >= 0) lookaheadWeightDecide(lw[k], &wCosts[2 * wAt[k]], weighted, wScale, wDenom, wOffset);
        /* the estimate's own buffers: its costs, and the fields it searches (the fields it reads are the call's shared copies) */
        void** bufs[6] = { &j.dLc, &j.dBc, &j.dMvs, &j.dMvc, &j.dMvs1, &j.dMvc1 };
        const size_t sizes[6] = { ncu * 2, ncu * 4, ncu * 4, ncu * 4, ncu * 4, ncu * 4 };
        const bool want[6] = { true, true, j.search0, j.search0, j.search1, j.search1 };
        (void)sizes;
        for (int b = 0; b < 6; b++) if (want[b] && !(*bufs[b] = laBuf())) rc = xa_fail(X265AMD_EHIP, "encoder: device allocation");
        issued = k + 1;
        if (rc != X265AMD_OK) break;
        x265amd_lowres_cost_job& q = kj[k];
        memset(&q, 0, sizeof(q));
        q.d_fenc = fenc.dLowres + lowOrg;
        for (int t = 0; t < 4; t++) { q.d_ref0[t] = j.ref0->dLowres + (size_t)t * lowPlaneElems + lowOrg; q.d_ref1[t] = j.ref1 ? j.ref1->dLowres + (size_t)t * lowPlaneElems + lowOrg : nullptr; }
        q.d_intra_cost = fenc.dIntraCost;
        q.d_lowres_costs = (uint16_t*)j.dLc; q.d_bcost = (int32_t*)j.dBc; q.do_search0 = j.search0; q.do_search1 = j.search1;
        if (!j.whole && laNumSlices > 1) { q.rows_per_slice = laRowsPerSlice; q.num_slices = laNumSlices; }
        if (weighted)
        {
            /* the four planes weighted, margins included, for this estimate's list-0 search (slicetype.cpp:962-977) */
            const int correction = 14 - X265AMD_DEPTH;
            if (xa_scratch_alloc(&j.dW, lowPlaneElems * 4 * sizeof(pixel)) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: device allocation"); break; }
            rc = x265amd_weight_buffer(laStream, j.ref0->dLowres, (pixel*)j.dW, lowPlaneElems * 4, wScale, (wDenom ? 1 << (wDenom - 1) : 0) << correction, wDenom + correction, wOffset << (X265AMD_DEPTH - 8));
            if (rc != X265AMD_OK) break;
            for (int t = 0; t < 4; t++) q.d_ref0w[t] = (pixel*)j.dW + (size_t)t * lowPlaneElems + lowOrg;
        }
        bool ok = true;
        const bool own0 = !fenc.lowMvs[j.d0].empty(), own1 = j.d1 > 0 && !fenc.lowMvs1[j.d1].empty();
        void* f0 = j.dMvs; void* c0 = j.dMvc; void* f1 = j.dMvs1; void* c1 = j.dMvc1;
        if (!j.search0) ok = shared((own0 ? fenc.lowMvs : fenc.specMvs)[j.d0], (own0 ? fenc.lowMvc : fenc.specMvc)[j.d0], f0, c0);
        if (ok && j.d1 > 0 && !j.search1) ok = shared((own1 ? fenc.lowMvs1 : fenc.specMvs1)[j.d1], (own1 ? fenc.lowMvc1 : fenc.specMvc1)[j.d1], f1, c1);
        q.d_mvs0 = (int16_t*)f0; q.d_mv_costs0 = (int32_t*)c0; q.d_mvs1 = (int16_t*)f1; q.d_mv_costs1 = (int32_t*)c1;
        if (!ok) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost set-up");
    }
    phase(0);
    /* one launch for all of them (blockIdx.y = the estimate): the device runs as many block rows side by side as it holds */
    if (rc == X265AMD_OK) rc = x265amd_lowres_frame_cost_batch(laStream, me, kj.data(), (int)jobs.size(), lowStride, lowCuW, lowCuH);
    /* the sums over the blocks, on the device too: two numbers per estimate come back instead of its two cost arrays */
    std::vector<int64_t> sums(2 * jobs.size());
    if (rc == X265AMD_OK) rc = x265amd_lowres_cost_sums(laStream, kj.data(), (int)issued, lowCuW, lowCuH, sums.data());
    phase(1);
    for (size_t k = 0; k < issued && rc == X265AMD_OK; k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        bool ok = true;
        if (ok && j.search0)
        {
            std::vector<int16_t>& mv = (j.spec ? fenc.specMvs : fenc.lowMvs)[j.d0]; std::vector<int32_t>& mc = (j.spec ? fenc.specMvc : fenc.lowMvc)[j.d0];
            mv.resize(ncu * 2); mc.resize(ncu);
            ok = hipMemcpyAsync(mv.data(), j.dMvs, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipMemcpyAsync(mc.data(), j.dMvc, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess;
            if (ok) { laFieldPut(mv.data(), j.dMvs, j.dMvc); j.dMvs = j.dMvc = nullptr; }            /* the search's buffers ARE the field's device copy from now on */
        }
        if (ok && j.search1)
        {
            std::vector<int16_t>& mv = (j.spec ? fenc.specMvs1 : fenc.lowMvs1)[j.d1]; std::vector<int32_t>& mc = (j.spec ? fenc.specMvc1 : fenc.lowMvc1)[j.d1];
            mv.resize(ncu * 2); mc.resize(ncu);
            ok = hipMemcpyAsync(mv.data(), j.dMvs1, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipMemcpyAsync(mc.data(), j.dMvc1, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess;
            if (ok) { laFieldPut(mv.data(), j.dMvs1, j.dMvc1); j.dMvs1 = j.dMvc1 = nullptr; }
        }
        if (!ok) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost");
    }
    phase(2);
    if (hipStreamSynchronize(laStream) != hipSuccess && rc == X265AMD_OK) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost");
    phase(3);
    struct Ph { decltype(phase)& f; ~Ph() { f(4); } } ph_{ phase };
    laFieldsTrim();
    for (size_t k = 0; k < issued; k++)
    {
        CostJob& j = jobs[k];
        void* keepLc = (rc == X265AMD_OK && p.cuTree) ? j.dLc : nullptr;            /* Lowres::lowresCosts[d0][d1]: stays with the picture (cuTree reads it) */
        void* bufs[6] = { j.dMvs, j.dMvc, keepLc ? nullptr : j.dLc, j.dBc, j.dMvs1, j.dMvc1 };
        for (void* b : bufs) laBufPut(b);
        xa_scratch_free(j.dW);
        j.dMvs = j.dMvc = j.dLc = j.dBc = j.dMvs1 = j.dMvc1 = j.dW = nullptr;
        if (keepLc)
        {
            const int key = j.d0 * 32 + j.d1;
            std::map<int, void*>& m = j.spec ? j.fenc->dSpecLc : j.fenc->dLc;
            j.fenc->dropLc(m, key);
            m[key] = keepLc;
            if (!j.spec) j.fenc->lcHost.erase(key);
        }
        if (rc != X265AMD_OK)
        {
            if (j.search0) { (j.spec ? j.fenc->specMvs : j.fenc->lowMvs)[j.d0].clear(); (j.spec ? j.fenc->specMvc : j.fenc->lowMvc)[j.d0].clear(); }
            if (j.search1) { (j.spec ? j.fenc->specMvs1 : j.fenc->lowMvs1)[j.d1].clear(); (j.spec ? j.fenc->specMvc1 : j.fenc->lowMvc1)[j.d1].clear(); }
            continue;
        }
        int64_t est = sums[2 * k]; const int imb = (int)sums[2 * k + 1];
        if (j.d1 > 0) est = est * 100 / (130 + 0);          /* param.bFrameBias: the default */
        if (j.spec) { j.fenc->specCost2[j.d0][j.d1] = est; if (j.d1 == 0) j.fenc->specIntraMbs[j.d0] = imb; continue; }
        /* a field made for good replaces whatever was made ahead of its time for the same pair, and the estimates that were built on that */
        if (j.search0) { j.fenc->specMvs[j.d0].clear(); j.fenc->specMvc[j.d0].clear(); for (int t = 0; t < 18; t++) { j.fenc->specCost2[j.d0][t] = -1; j.fenc->dropLc(j.fenc->dSpecLc, j.d0 * 32 + t); } }
        if (j.search1) { j.fenc->specMvs1[j.d1].clear(); j.fenc->specMvc1[j.d1].clear(); for (int t = 0; t < 18; t++) { j.fenc->specCost2[t][j.d1] = -1; j.fenc->dropLc(j.fenc->dSpecLc, t * 32 + j.d1); } }
        j.fenc->cost2[j.d0][j.d1] = est;
        if (j.d1 == 0) { j.fenc->costEst[j.d0] = est; j.fenc->intraMbs[j.d0] = imb; }
    }
    return rc;
}
int x265amd_encoder::frameCost(std::vector<Pic*>& frames, int p0, int p1, int b, int64_t& score)
{
    return frameCostAt(*frames[b], *frames[p0], p1 > b ? frames[p1] : nullptr, b - p0, p1 - b, score);
}

/* The B-frame trellis (X265_B_ADAPT_TRELLIS; what Lookahead::slicetypePath / slicetypePathCost compute, slicetype.cpp:3218-3313).  A plan for the first n pictures of
 * the window is the list of its mini-GOPs' B runs (a run of k: k B pictures, then their P picture); planCost prices one -- per mini-GOP the P picture against the
 * mini-GOP's anchor, then its B pictures (with the pyramid: the middle one between anchor and P picture, the ones in front of it between anchor and middle, the ones behind
 * between middle and P picture) -- and gives up once the sum passes `limit`.  Which estimates are asked for, and in what order, is part of the result (an estimate that
 * is asked for exists afterwards: frameCostAt), so the order of the additions and of the limit checks is the reference's. */
int64_t x265amd_encoder::planCost(std::vector<Pic*>& frames, const std::vector<uint8_t>& runs, int64_t limit, int& rc)
{
    int64_t total = 0;
    int anchor = 0;
    for (size_t g = 0; g < runs.size() && rc == X265AMD_OK; g++)
    {
        const int pPic = anchor + runs[g] + 1;
        int64_t c = 0;
        rc = frameCost(frames, anchor, pPic, pPic, c);
        total += c;
        if (total > limit) break;
        if (p.bBPyramid && runs[g] > 1)
        {
            const int middle = anchor + (pPic - anchor) / 2;
            if (rc == X265AMD_OK) { rc = frameCost(frames, anchor, pPic, middle, c); total += c; }
            for (int b = anchor + 1; b < middle && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, anchor, middle, b, c); total += c; }
            for (int b = middle + 1; b < pPic && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, middle, pPic, b, c); total += c; }
        }
        else
            for (int b = anchor + 1; b < pPic && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, anchor, pPic, b, c); total += c; }
        anchor = pPic;
    }
    return total;
}
/* the cheapest plan for the first `length` pictures: the cheapest plan of a shorter prefix with one more mini-GOP behind it, the last run growing from 0; the cheapest so
 * far is the limit of the next one's pricing; the first of equals stays */
void x265amd_encoder::extendPlans(std::vector<Pic*>& frames, int length, std::vector<std::vector<uint8_t> >& plans, int& rc)
{
    const int longest = std::min(p.bframes, length - 1);
    int64_t cheapest = 1LL << 62;
    std::vector<uint8_t> winner;
    for (int run = 0; run <= longest && rc == X265AMD_OK; run++)
    {
        std::vector<uint8_t> plan = plans[length - (run + 1)];
        plan.push_back((uint8_t)run);
        const int64_t cost = planCost(frames, plan, cheapest, rc);
        if (cost < cheapest) { cheapest = cost; winner.swap(plan); }
    }
    plans[length] = winner;
}

/* Lookahead::scenecutInternal (slicetype.cpp:3016-3047): float / double arithmetic as written there */
bool x265amd_encoder::scenecutInternal(std::vector<Pic*>& frames, int p0, int p1, bool real, int& rc)
{
    Pic* frame = frames[p1];
    if (rc == X265AMD_OK) rc = frameCostP(*frame, *frames[p0], p1 - p0);
    if (rc != X265AMD_OK) return false;
    const int64_t icost = frame->costEst[0], pcost = frame->costEst[p1 - p0];
    const int gopSize = (frame->poc - lastKeyframe) % p.keyframeMax;
    const float threshMax = (float)(p.scenecutThreshold / 100.0);
    float threshMin = (float)(threshMax * 0.25);
    double bias = 5.0 / 100;            /* param.scenecutBias: the default, scaled in Encoder::configure (encoder.cpp:3948) */
    if (real)
    {
        if (keyframeMin == p.keyframeMax) threshMin = threshMax;
        if (gopSize <= keyframeMin / 4) bias = threshMin / 4;
        else if (gopSize <= keyframeMin) bias = threshMin * gopSize / keyframeMin;
        else bias = threshMin + (threshMax - threshMin) * (gopSize - keyframeMin) / (p.keyframeMax - keyframeMin);
    }
    return pcost >= (1.0 - bias) * icost;
}

/* Lookahead::scenecut (slicetype.cpp:2921-3014) */
bool x265amd_encoder::scenecut(std::vector<Pic*>& frames, int p0, int p1, bool real, int numFrames, int& rc)
{
    if (real && p.bframes)
    {
        const int origmaxp1 = p0 + 1 + p.bframes, maxp1 = std::min(origmaxp1, numFrames);
        bool fluctuate = false, noScenecuts = false;
        int64_t avgSatdCost = 0;
        if (frames[p0]->costEst[p1 - p0] > -1) avgSatdCost = frames[p0]->costEst[p1 - p0];
        int cnt = 1;
        for (int cp1 = p1; cp1 <= maxp1; cp1++)
        {
            if (!scenecutInternal(frames, p0, cp1, false, rc))
            {
                for (int i = cp1; i > p0; i--) { frames[i]->bScenecut = false; noScenecuts = false; }
            }
            else if (scenecutInternal(frames, cp1 - 1, cp1, false, rc)) { frames[cp1]->bScenecut = true; noScenecuts = true; }
            if (rc != X265AMD_OK) return false;
            avgSatdCost += frames[cp1]->costEst[cp1 - p0];
            cnt++;
        }
        if (noScenecuts)
        {
            fluctuate = false;
            avgSatdCost /= cnt;
            for (int i = p1; i <= maxp1; i++)
            {
                const int64_t curCost = frames[i]->costEst[i - p0], prevCost = frames[i - 1]->costEst[i - 1 - p0];
                if (fabs((double)(curCost - avgSatdCost)) > 0.1 * avgSatdCost || fabs((double)(curCost - prevCost)) > 0.1 * prevCost)
                {
                    fluctuate = true;
                    if (!isSceneTransition && frames[i]->bScenecut)
                    {
                        isSceneTransition = true;
                        for (int j = i + 1; j <= maxp1; j++) frames[j]->bScenecut = false;
                        break;
                    }
                }
                frames[i]->bScenecut = false;
            }
        }
        if (!fluctuate && !noScenecuts) isSceneTransition = false;
    }
    if (!frames[p1]->bScenecut) return false;
    return scenecutInternal(frames, p0, p1, real, rc);
}

/* Lookahead::slicetypeAnalyse(frames, bKeyframe) (slicetype.cpp:2603-2919) without VBV / zones / gop-lookahead: frames[0] = the last non-B picture, frames[1..] = the
 * undecided pictures of the window.  bKeyframe: the pass behind a keyframe's mini-GOP that cuTree adds (slicetype.cpp:2469-2483): the same analysis with the keyframe as
 * frames[0], cuTree down to the keyframe itself, and every type taken back afterwards */
int x265amd_encoder::slicetypeAnalyse(std::vector<Pic*>& frames, bool bKeyframe)
{
    const int maxSearch = std::min(p.lookaheadDepth, 250);
    int framecnt = 0;
    for (; framecnt < maxSearch; framecnt++)
        if (framecnt + 1 >= (int)frames.size() || frames[framecnt + 1]->type != TYPE_AUTO) break;
    if (!framecnt) return p.cuTree ? runCuTree(frames, 0, bKeyframe) : X265AMD_OK;
    frames.resize((size_t)framecnt + 1);
    const int keyFrameLimit = p.keyframeMax + lastKeyframe - frames[0]->poc - 1, keyintLimit = keyFrameLimit;
    const int origNumFrames = std::min(framecnt, keyintLimit);
    int numFrames = origNumFrames;
    if (p.bOpenGOP && numFrames < framecnt) numFrames++;           /* open GOPs: the window takes in the keyframe (slicetype.cpp:2660-2661) */
    else if (numFrames == 0) { frames[1]->type = TYPE_I; return X265AMD_OK; }
    int rc = X265AMD_OK;
    if (p.bFrameAdaptive == 2 && p.bframes)
    {
        /* m_bBatchMotionSearch (slicetype.cpp:2668-2694; it stays on with a pool of four workers or more): every picture of the window is searched against the pictures
         * 1 .. bframes + 1 before it and, where the window allows, the same distance behind it -- whether or not the trellis below will ask for that pair.  The fields
         * stay with the pictures: the encoder's searches take candidates from them (Search::getLowresMV) */
        std::vector<CostJob> jobs;
        for (int b = 2; b < numFrames; b++)
            for (int i = 1; i <= p.bframes + 1; i++)
            {
                const int p0 = b - i;
                if (p0 < 0 || !frames[b]->lowMvs[i].empty()) continue;
                int p1 = b + i;
                if (p1 >= numFrames || !frames[b]->lowMvs1[i].empty()) p1 = b;
                if (frames[b]->cost2[i][p1 - b] >= 0) continue;
                CostJob j;
                j.fenc = frames[b]; j.ref0 = frames[p0]; j.ref1 = p1 > b ? frames[p1] : nullptr; j.d0 = i; j.d1 = p1 - b; j.whole = true;         /* (batch mode: no cooperative slices, :4004) */
                jobs.push_back(j);
            }
        /* (the first picture of the window is not in the reference's batch: its estimate against the last non-B picture is what the scene-cut check and every path
         * of the trellis start with) */
        if (frames[1]->lowMvs[1].empty()) { CostJob j; j.fenc = frames[1]; j.ref0 = frames[0]; j.d0 = 1; jobs.push_back(j); }
        /* ... nor is the last one, the P picture every path ends with: searched now, side by side with the batch, but AHEAD OF ITS TIME (CostJob::spec) -- the field and the
         * estimate wait in the picture's spec* members until the trellis asks for them, as everything below does */
        auto hasL0 = [](const Pic* f, int d) { return !f->lowMvs[d].empty() || !f->specMvs[d].empty(); };
        auto hasL1 = [](const Pic* f, int d) { return !f->lowMvs1[d].empty() || !f->specMvs1[d].empty(); };
        for (int i = 1; i <= p.bframes + 1 && i <= numFrames && numFrames > 1; i++)
            if (!hasL0(frames[numFrames], i)) { CostJob j; j.fenc = frames[numFrames]; j.ref0 = frames[numFrames - i]; j.d0 = i; j.spec = true; jobs.push_back(j); }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
        /* What the batch leaves to the trellis -- the fields towards pictures behind that it pairs with no distance before (the first picture's; every picture's towards
         * the window's last) -- searched side by side as well instead of one estimate at a time when a path asks.  Whether the reference ever makes one of them depends on
         * the paths it prices and where it gives them up, and with a B pyramid the encoder's pictures reference pictures the trellis did not price them against: so they
         * are made ahead of their time, and only what a path asks for becomes the picture's (frameCostAt). */
        jobs.clear();
        for (int b = 1; b < numFrames; b++)
            for (int jj = 1; jj <= p.bframes; jj++)
            {
                const int p1 = b + jj;
                if (p1 > numFrames) break;
                if (hasL1(frames[b], jj) || !hasL0(frames[b], 1)) continue;
                CostJob j;
                j.fenc = frames[b]; j.ref0 = frames[b - 1]; j.ref1 = frames[p1]; j.d0 = 1; j.d1 = jj; j.spec = true;
                jobs.push_back(j);
            }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
        /* ... and every cost the trellis can ask for of these pictures, side by side.  Those of m_bBatchFrameCosts (:2696-2734: pictures 2 .. numFrames - 1 against fields
         * that exist, the picture behind inside the window; the reference fills them with a pool of more than twelve workers) are the pictures' at once, the rest wait */
        jobs.clear();
        for (int b = 1; b < numFrames; b++)
            for (int i = 1; i <= p.bframes + 1; i++)
            {
                if (b < i || !hasL0(frames[b], i)) continue;
                for (int jj = 0; jj <= p.bframes; jj++)
                {
                    const int p1 = b + jj;
                    if (p1 > numFrames) break;
                    if ((jj && !hasL1(frames[b], jj)) || frames[b]->cost2[i][jj] >= 0 || frames[b]->specCost2[i][jj] >= 0) continue;
                    CostJob j;
                    j.fenc = frames[b]; j.ref0 = frames[b - i]; j.ref1 = jj ? frames[p1] : nullptr; j.d0 = i; j.d1 = jj;
                    j.spec = !(b >= 2 && p1 < numFrames && !frames[b]->lowMvs[i].empty() && (!jj || !frames[b]->lowMvs1[jj].empty()));
                    jobs.push_back(j);
                }
            }
        /* the last picture of the window as a P picture at every distance (the trellis' path ends) */
        for (int i = 1; i <= p.bframes + 1 && i <= numFrames; i++)
            if (hasL0(frames[numFrames], i) && frames[numFrames]->cost2[i][0] < 0 && frames[numFrames]->specCost2[i][0] < 0)
            { CostJob j; j.fenc = frames[numFrames]; j.ref0 = frames[numFrames - i]; j.d0 = i; j.spec = true; jobs.push_back(j); }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
    }
    const bool isScenecut = scenecut(frames, 0, 1, true, origNumFrames, rc);       /* (run whatever the threshold: its estimates and marks stay) */
    if (rc != X265AMD_OK) return rc;
    if (p.scenecutThreshold > 0 && isScenecut) { frames[1]->type = TYPE_I; return X265AMD_OK; }
    int resetStart;
    if (p.bframes)
    {
        int numBFrames = std::min(numFrames - 1, p.bframes);
        if (p.bFrameAdaptive == 2)
        {
            /* X265_B_ADAPT_TRELLIS (slicetype.cpp:2776-2795): the cheapest path of P / B decisions through the window */
            numBFrames = 0;
            if (numFrames > 1)
            {
                std::vector<std::vector<uint8_t> > plans((size_t)numFrames + 1);      /* plans[n]: the cheapest plan for the first n pictures; plans[0] is empty, plans[1] one P picture */
                plans[1].push_back(0);
                for (int j = 2; j <= numFrames && rc == X265AMD_OK; j++) extendPlans(frames, j, plans, rc);
                if (rc != X265AMD_OK) return rc;
                const std::vector<uint8_t>& plan = plans[numFrames];
                numBFrames = plan.empty() ? 0 : plan[0];
                int at = 1;
                for (size_t g = 0; g < plan.size(); g++)
                {
                    for (int k = 0; k < plan[g] && at < numFrames; k++) frames[at++]->type = TYPE_B;
                    if (at < numFrames) frames[at++]->type = TYPE_P;
                }
            }
        }
        else if (p.bFrameAdaptive == 1)
        {
            /* X265_B_ADAPT_FAST (slicetype.cpp:2796-2848): pictures in pairs -- two P pictures when half the second one's blocks are intra, a P picture when P P is cheaper than B P,
             * else B pictures for as long as the P picture behind them stays cheap; every estimate made when it is asked for (no batch: slicetype.cpp:1024) */
            const int cuCount = lowBlocks;
            auto cost = [&](int p0, int p1, int b, bool intraPenalty, int64_t& out) -> int {
                int64_t sc = 0;
                const int r = frameCost(frames, p0, p1, b, sc);
                if (r != X265AMD_OK) return r;
                if (intraPenalty) sc += sc * frames[b]->intraMbs[b - p0] / (cuCount * 8);          /* estimateFrameCost's "arbitrary penalty for I-blocks after B-frames" (:4069-4071) */
                out = sc;
                return X265AMD_OK;
            };
            for (int i = 0; i <= numFrames - 2 && rc == X265AMD_OK; )
            {
                int64_t cost1p0 = 0, cost2p0 = 0, cost1b1 = 0, cost2p1 = 0;
                if ((rc = cost(i + 0, i + 2, i + 2, true, cost2p1)) != X265AMD_OK) break;
                if (frames[i + 2]->intraMbs[2] > cuCount / 2) { frames[i + 1]->type = TYPE_P; frames[i + 2]->type = TYPE_P; i += 2; continue; }
                if ((rc = cost(i + 0, i + 2, i + 1, false, cost1b1)) != X265AMD_OK || (rc = cost(i + 0, i + 1, i + 1, false, cost1p0)) != X265AMD_OK ||
                    (rc = cost(i + 1, i + 2, i + 2, false, cost2p0)) != X265AMD_OK) break;
                if (cost1p0 + cost2p0 < cost1b1 + cost2p1) { frames[i + 1]->type = TYPE_P; i += 1; continue; }
                frames[i + 1]->type = TYPE_B;
                int j;
                for (j = i + 2; j <= std::min(i + p.bframes, numFrames - 1); j++)
                {
                    const int64_t pthresh = std::max(300 - (50 - 0) * (j - i - 1), 300 / 10);          /* INTER_THRESH, P_SENS_BIAS with bFrameBias 0 */
                    int64_t pcost = 0;
                    if ((rc = cost(i + 0, j + 1, j + 1, true, pcost)) != X265AMD_OK) break;
                    if (pcost > pthresh * cuCount || frames[j + 1]->intraMbs[j - i + 1] > cuCount / 3) break;
                    frames[j]->type = TYPE_B;
                }
                if (rc != X265AMD_OK) break;
                frames[j]->type = TYPE_P;
                i = j;
            }
            if (rc != X265AMD_OK) return rc;
            frames[numFrames]->type = TYPE_P;
            numBFrames = 0;
            while (numBFrames < numFrames && frames[numBFrames + 1]->type == TYPE_B) numBFrames++;
        }
        else
            for (int j = 1; j < numFrames; j++) frames[j]->type = (j % (numBFrames + 1)) ? TYPE_B : TYPE_P;
        frames[numFrames]->type = TYPE_P;
        int numAnalyzed = numFrames;
        /* Check scenecut on the first minigop. */
        for (int j = 1; j < numBFrames + 1; j++)
        {
            const bool cut = scenecut(frames, j, j + 1, false, origNumFrames, rc);
            if (rc != X265AMD_OK) return rc;
            if (cut) { frames[j]->type = TYPE_P; numAnalyzed = j; break; }
        }
        resetStart = bKeyframe ? 1 : std::min(numBFrames + 2, numAnalyzed + 1);
    }
    else
    {
        for (int j = 1; j <= numFrames; j++) frames[j]->type = TYPE_P;
        resetStart = bKeyframe ? 1 : 2;
    }
    /* cuTree on the window as it is typed now (slicetype.cpp:2893-2894) */
    if (p.cuTree && (rc = runCuTree(frames, std::min(numFrames, p.keyframeMax), bKeyframe)) != X265AMD_OK) return rc;
    for (int j = keyintLimit + 1; j <= numFrames; j += p.keyframeMax) { frames[j]->type = TYPE_I; resetStart = std::min(resetStart, j + 1); }
    const int maxp1 = std::min(p.bframes + 1, origNumFrames);
    /* Restore frame types for all frames that haven't actually been decided yet. */
    for (int j = resetStart; j <= numFrames; j++)
    {
        frames[j]->type = TYPE_AUTO;
        if (j <= maxp1 && frames[j]->bScenecut && isSceneTransition) isSceneTransition = false;
    }
    return X265AMD_OK;
}

/* CostEstimateGroup::singleCost(p0, p1, b) for Lookahead::cuTree (x265amd_cutree's callback): the estimate is made if it does not exist (with the searches its fields
 * lack), its block costs come back from the device once, the fields are the picture's */
namespace { struct TreeCtx { x265amd_encoder* e; std::vector<Pic*>* frames; }; }
int x265amd_encoder::cuTreeEstimate(void* ctx, int p0, int p1, int b, const uint16_t** lc, const int16_t** mvs0, const int16_t** mvs1)
{
    TreeCtx& t = *(TreeCtx*)ctx;
    x265amd_encoder& e = *t.e;
    std::vector<Pic*>& frames = *t.frames;
    if (p0 < 0 || b < p0 || p1 < b || p1 >= (int)frames.size() || b - p0 > 17 || p1 - b > 17 || b == p0) return xa_fail(X265AMD_EINVAL, "encoder: cuTree estimate");
    int64_t score = 0;
    const int rc = e.frameCost(frames, p0, p1, b, score);
    if (rc != X265AMD_OK) return rc;
    Pic& f = *frames[b];
    const int d0 = b - p0, d1 = p1 - b, key = d0 * 32 + d1;
    auto h = f.lcHost.find(key);
    if (h == f.lcHost.end())
    {
        auto d = f.dLc.find(key);
        if (d == f.dLc.end()) return xa_fail(X265AMD_EHIP, "encoder: cuTree: the block costs of an estimate are gone");
        std::vector<uint16_t> v((size_t)e.lowCuW * e.lowCuH);
        if (hipMemcpyAsync(v.data(), d->second, v.size() * 2, hipMemcpyDeviceToHost, e.laStream) != hipSuccess || hipStreamSynchronize(e.laStream) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: cuTree: block costs");
        h = f.lcHost.emplace(key, std::move(v)).first;
    }
    *lc = h->second.data();
    *mvs0 = f.lowMvs[d0].empty() ? nullptr : f.lowMvs[d0].data();
    *mvs1 = d1 > 0 && !f.lowMvs1[d1].empty() ? f.lowMvs1[d1].data() : nullptr;
    return 0;
}
/* Lookahead::cuTree(frames, numframes, bIntra) (slicetype.cpp:3399-3500): host/fm_ratecontrol.cpp on the pictures' host arrays */
int x265amd_encoder::runCuTree(std::vector<Pic*>& frames, int numframes, bool bIntra)
{
    if (numframes >= (int)frames.size()) numframes = (int)frames.size() - 1;
    std::vector<x265amd_cutree_frame> recs(frames.size());
    std::vector<x265amd_cutree_frame*> ptrs(frames.size());
    for (size_t k = 0; k < frames.size(); k++)
    {
        Pic& f = *frames[k];
        if (f.intraCostHost.empty() || f.qpAqOffset.empty()) return xa_fail(X265AMD_EINVAL, "encoder: cuTree: a picture without block offsets");
        x265amd_cutree_frame& r = recs[k];
        memset(&r, 0, sizeof(r));
        r.slice_type = f.type; r.intra_cost = f.intraCostHost.data(); r.inv_qscale = f.invQscale.data(); r.qp_aq_offset = f.qpAqOffset.data();
        r.qp_cutree_offset = f.qpCuTreeOffset.data(); r.propagate_cost = f.propagateCost.data(); r.weighted_cost_delta = nullptr;       /* (Lowres::weightedCostDelta is an integer quotient below one: always 0, slicetype.cpp:964) */
        ptrs[k] = &r;
    }
    TreeCtx ctx{ this, &frames };
    const int rc = x265amd_cutree(&treeParams, ptrs.data(), numframes, bIntra, cuTreeEstimate, &ctx);
    return rc == X265AMD_OK ? X265AMD_OK : xa_fail(rc, "encoder: cuTree");
}

/* The typed mini-GOP input[0 .. b] goes to `ready` in coding order (slicetype.cpp:2372-2376, :2443-2470): with a B pyramid and two B pictures or more the middle one
 * becomes a reference (Lookahead::placeBref: index (0 + b) / 2); the non-B picture first, then the referenced B picture, then the other B pictures in display order */
void x265amd_encoder::pushMiniGop(int b)
{
    if (p.bBPyramid && b > 1) input[b / 2]->type = TYPE_BREF;
    if (b > 0) input[b - 1]->bLastMiniGopBFrame = true;
    ready.push_back(input[b]);
    for (int i = 0; i < b; i++) if (input[i]->type == TYPE_BREF) ready.push_back(input[i]);
    for (int i = 0; i < b; i++) if (input[i]->type != TYPE_BREF) ready.push_back(input[i]);
    input.erase(input.begin(), input.begin() + b + 1);
    first = false;
}

/* Lookahead::slicetypeDecide (slicetype.cpp:1802-2400) as far as the built subset goes: runs when the input queue holds lookaheadDepth pictures (Lookahead::findJob,
 * m_fullQueueSize; one picture is enough once the caller flushes), types the next mini-GOP and moves it to `ready` in coding order.  Returns 0, or an error code. */
int x265amd_encoder::decideLookahead(bool flush, int maxGops)
{
    const int fullQueue = flush ? 1 : std::max(1, p.lookaheadDepth);
    while ((int)input.size() >= fullQueue && maxGops-- > 0)
    {
        const int maxSearch = std::max(1, std::min(p.lookaheadDepth, 250));
        std::vector<Pic*> frames;
        frames.push_back(lastNonB.get());
        for (int j = 0; j < maxSearch && j < (int)input.size(); j++) frames.push_back(input[j].get());
        const int windowCount = (int)frames.size() - 1;         /* the reference's maxSearch: the pictures this decision looks at */
        if (lastNonB)
        {
            const int rc = slicetypeAnalyse(frames);
            if (rc != X265AMD_OK) return rc;
        }
        const int nlist = std::min((int)input.size(), p.bframes + 2);
        int b = 0;
        for (;; b++)
        {
            Pic& frm = *input[b];
            if (frm.poc - lastKeyframe >= p.keyframeMax && (frm.type == TYPE_AUTO || frm.type == TYPE_I)) frm.type = p.bOpenGOP && haveKeyframe ? TYPE_I : TYPE_IDR;
            if (frm.type == TYPE_I && frm.poc - lastKeyframe >= keyframeMin)
            {
                /* closed GOPs: a keyframe is an IDR picture; open GOPs: it stays an I picture (CRA) and the B pictures in front of it stay (slicetype.cpp:1985-1994) */
                if (p.bOpenGOP) { lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true; }
                else frm.type = TYPE_IDR;
            }
            if (frm.type == TYPE_IDR)
            {
                lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true;
                if (b > 0) { input[b - 1]->type = TYPE_P; b--; }
            }
            Pic& cur = *input[b];          /* (after the step back the tests below see the keyframe in the reference: they do nothing for it; the loop ends at the P picture) */
            if (&cur == &frm)
            {
                if (b == p.bframes || b + 1 >= nlist) { if (frm.type == TYPE_AUTO || frm.type == TYPE_B) frm.type = TYPE_P; }
                if (frm.type == TYPE_AUTO) frm.type = TYPE_B;
                else if (frm.type != TYPE_B) break;
            }
            else break;
        }
        if (p.rateControlMode == X265AMD_RC_CRF)
        {
            /* "calculate the frame costs ahead of time for estimateFrameCost while we still have lowres" (slicetype.cpp:2396-2440): the estimate of every picture of the
             * mini-GOP against the pictures it will reference -- made now if the decision above did not need it, with the searches its fields lack (the encoder's searches
             * take candidates from those fields, and cuTree and the rate control read the estimates) */
            if (p.bBPyramid && b > 1) input[b / 2]->type = TYPE_BREF;          /* placeBref comes first (:2386-2389) */
            std::vector<Pic*> est;
            est.push_back(lastNonB.get());
            for (int i = 0; i <= b; i++) est.push_back(input[i].get());
            int64_t score = 0;
            const int p1n = b + 1;
            const bool isI = est[p1n]->type == TYPE_I || est[p1n]->type == TYPE_IDR;
            if (!isI && est[0])
            {
                const int rc = frameCost(est, 0, p1n, p1n, score);
                if (rc != X265AMD_OK) return rc;
            }
            if (b && est[0])
            {
                int p0 = 0;
                bool isp0available = est[p1n]->type != TYPE_IDR;
                for (int bb = 1; bb <= b; bb++)
                {
                    if (!isp0available) p0 = bb;
                    int p1;
                    if (est[bb]->type == TYPE_B) for (p1 = bb; est[p1]->type == TYPE_B; p1++) { }
                    else p1 = b + 1;
                    if (p0 != bb)
                    {
                        const int rc = frameCost(est, p0, p1, bb, score);
                        if (rc != X265AMD_OK) return rc;
                    }
                    if (est[bb]->type == TYPE_BREF) { p0 = bb; isp0available = true; }
                }
            }
        }
        lastNonB = input[b];
        pushMiniGop(b);
        if (p.cuTree && (lastNonB->type == TYPE_I || lastNonB->type == TYPE_IDR))
        {
            /* the keyframe's own pass (slicetype.cpp:2469-2483): the pictures of the window that are left, behind the keyframe */
            std::vector<Pic*> kf;
            kf.push_back(lastNonB.get());
            const int left = windowCount - (b + 1);
            for (int j = 0; j < left && j < (int)input.size(); j++) kf.push_back(input[j].get());
            const int rc = slicetypeAnalyse(kf, true);
            if (rc != X265AMD_OK) return rc;
        }
    }
    return X265AMD_OK;
}

/* Lookahead::slicetypeDecide with bFrameAdaptive 0 and no scenecut (slicetype.cpp:1929-2040): the next mini-GOP, moved to `ready` in coding order */
void x265amd_encoder::decideMiniGop(bool flush)
{
    while (!input.empty())
    {
        if (!flush && (int)input.size() < p.bframes + 1 && !(first && !input.empty())) return;
        int b = 0;
        for (;; b++)
        {
            Pic& frm = *input[b];
            if (frm.poc - lastKeyframe >= p.keyframeMax) frm.type = p.bOpenGOP && haveKeyframe ? TYPE_I : TYPE_IDR;
            if (frm.type == TYPE_I)
            {
                /* open GOP: the keyframe is an I picture (CRA) that ends the mini-GOP; the B pictures in front of it stay and reference across it */
                lastKeyframe = frm.poc; frm.bKeyframe = true;
                break;
            }
            if (frm.type == TYPE_IDR)
            {
                /* closed GOP: the frame before a keyframe becomes P and ends the mini-GOP; the keyframe opens the next one */
                if (b > 0) { input[b - 1]->type = TYPE_P; b--; break; }
                lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true;
                break;
            }
            if (b == p.bframes || b + 1 >= (int)input.size()) { frm.type = TYPE_P; break; }
            frm.type = TYPE_B;
        }
        pushMiniGop(b);
        if (!flush) return;
    }
}

/* DPB::prepareEncode for the next picture in coding order (main thread): NAL type, reference picture set, reference lists, slice QP */
int x265amd_encoder::prepare(const PicP& picp)
{
    Pic& pic = *picp;
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    /* DPB::getNalUnitType (dpb.cpp:486-506): IDR_N_LP 20; a keyframe of an open GOP CRA 21; pictures in front of the last CRA picture in output order RASL 9 / 8,
     * in front of the last IDR picture RADL 7 / 6; the rest TRAIL 1 / 0 (the second number: B pictures, which nobody references, prepareEncode dpb.cpp:156-172) */
    int nal;
    if (pic.type == TYPE_IDR) nal = 20;
    else if (pic.bKeyframe && p.bOpenGOP) nal = 21;
    else if (pocCRA && pic.poc < pocCRA) nal = 9;
    else if (lastIDR && pic.poc < lastIDR) nal = 7;
    else nal = 1;
    if (pic.type == TYPE_B && nal < 16) nal--;
    pic.nalType = nal;
    pic.rpsUsed = !(nal >= 16 && nal <= 23);
    if (nal == 20) lastIDR = pic.poc;
    pic.lastIDR = lastIDR;
    pic.hasReferences = pic.type != TYPE_B;
    /* recycleUnreferenced: pictures nobody references leave the list */
    picList.erase(std::remove_if(picList.begin(), picList.end(), [](const PicP& q) { return !q->hasReferences; }), picList.end());
    /* decodingRefreshMarking (dpb.cpp:357-399): an IDR picture empties the buffer; after a CRA picture the first picture behind it in output order does, keeping the CRA picture */
    if (nal == 20) { for (auto& q : picList) q->hasReferences = false; }
    else
    {
        if (refreshPending && pic.poc > pocCRA)
        {
            for (auto& q : picList) if (q->poc != pocCRA) q->hasReferences = false;
            refreshPending = false;
        }
        if (nal == 21) { refreshPending = true; pocCRA = pic.poc; }
    }
    std::vector<PicP> rps;                                                                              /* computeRPS */
    for (auto& q : picList)
    {
        if ((int)rps.size() >= maxDecPicBuffering - 1) break;
        if (q->poc != pic.poc && q->hasReferences && (lastIDR >= pic.poc || lastIDR <= q->poc)) rps.push_back(q);
    }
    for (auto& q : picList)                                                                             /* applyReferencePictureSet */
        if (q->hasReferences && std::find(rps.begin(), rps.end(), q) == rps.end()) q->hasReferences = false;
    pic.neg.clear(); pic.pos.clear(); pic.lists[0].clear(); pic.lists[1].clear();
    for (const PicP& q : rps) (q->poc < pic.poc ? pic.neg : pic.pos).push_back(q);
    std::sort(pic.neg.begin(), pic.neg.end(), [](const PicP& a, const PicP& b) { return a->poc > b->poc; });           /* RPS::sortDeltaPOC */
    std::sort(pic.pos.begin(), pic.pos.end(), [](const PicP& a, const PicP& b) { return a->poc < b->poc; });
    if (stype == 2) statPictures[0]++;
    if (stype != 2)
    {
        const int n0 = std::min(std::max(1, (int)pic.neg.size()), p.maxNumReferences), n1 = stype == 0 ? std::min(p.bBPyramid ? 2 : 1, (int)pic.pos.size()) : 0;       /* dpb.cpp:269-273 */
        std::vector<PicP> l0(pic.neg), l1(pic.pos);
        l0.insert(l0.end(), pic.pos.begin(), pic.pos.end()); l1.insert(l1.end(), pic.neg.begin(), pic.neg.end());
        if ((int)l0.size() < n0 || (int)l1.size() < n1 || (stype == 0 && !n1)) return xa_fail(X265AMD_EINVAL, "encoder_encode: reference lists");
        pic.lists[0].assign(l0.begin(), l0.begin() + n0); pic.lists[1].assign(l1.begin(), l1.begin() + n1);
        {
            /* x265amd_encoder_stats: the distinct reference pictures this picture reads (SURVEY section 8d's R) */
            std::vector<const Pic*> seen;
            for (int l = 0; l < 2; l++) for (const PicP& q : pic.lists[l]) if (std::find(seen.begin(), seen.end(), q.get()) == seen.end()) seen.push_back(q.get());
            statPictures[stype == 1 ? 1 : 2]++; statReferences += seen.size();
        }
    }
    static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");
    memset(pic.wp, 0, sizeof(pic.wp)); pic.weighted = false;
    if (((p.bEnableWeightedPred && stype == 1) || (p.bEnableWeightedBiPred && stype == 0)) && !(dbgWp && strchr(dbgWp, 'p')))
    {
        /* FrameEncoder::compressFrame (frameencoder.cpp:553-582): weightAnalyse for P slices with --weightp, for B slices with --weightb */
        const int rcw = sliceWeights(pic);
        if (rcw != X265AMD_OK) return rcw;
    }
    if (rateCtl)
    {
        /* RateControl::rateControlStart, constant rate factor (coding order is the reference's m_startEndOrder; nothing it reads depends on how a picture was coded) */
        x265amd_rc_frame f;
        memset(&f, 0, sizeof(f));
        f.slice_type = stype; f.is_referenced = pic.type != TYPE_B; f.poc = pic.poc; f.scenecut = pic.bScenecut; f.last_minigop_b = pic.bLastMiniGopBFrame;
        if (stype != 2) f.ref0_scenecut = pic.lists[0][0]->bScenecut;
        f.satd_cost = estimatedPictureCost(pic);
        if (stype == 0)
            for (int l = 0; l < 2; l++)
            {
                const Pic& q = *pic.lists[l][0];
                f.ref_slice_type[l] = isBType(q.type) ? 0 : q.type == TYPE_P ? 1 : 2; f.ref_poc[l] = q.poc; f.ref_is_referenced[l] = q.type != TYPE_B; f.ref_avg_qp_rc[l] = q.avgQpRc;
            }
        const int qp = x265amd_rc_start(rateCtl, &f, &pic.avgQpRc);
        if (qp < 0) return xa_fail(X265AMD_EINVAL, "encoder_encode: rate control");
        pic.sliceQp = std::min(qp, 51);             /* FrameEncoder::compressFrame clips the slice QP to the range the syntax carries (frameencoder.cpp:612) */
        if (useDqp) cuQpTable(pic);
        static const bool rcLog = getenv("X265AMD_RC_LOG") != nullptr;
        if (rcLog) fprintf(stderr, "x265amd rc: poc %d type %d qp %d avgQpRc %.9f satd %lld scenecut %d\n", pic.poc, pic.type, pic.sliceQp, pic.avgQpRc, (long long)f.satd_cost, (int)pic.bScenecut);
    }
    else
    pic.sliceQp = pic.type == TYPE_BREF ? (qpConstant[0] + qpConstant[1]) / 2 : qpConstant[stype];                    /* rateControlStart, CQP (ratecontrol.cpp:1594-1597: a referenced B picture lies between B and P) */
    picList.insert(picList.begin(), picp);              /* PicList::pushFront */
    if (frameParallel)
    {
        /* what pictures coded beside this one read of it exists before any task starts: the maps (rows become valid as they are coded) and the POC lists */
        const size_t nUnits = (size_t)w4 * h4;
        pic.units.assign(nUnits, x265amd_cu_unit()); pic.motion.assign(nUnits, x265amd_mv_unit());
        memset(pic.units.data(), 0, sizeof(x265amd_cu_unit) * nUnits); memset(pic.motion.data(), 0, sizeof(x265amd_mv_unit) * nUnits);
        pic.registerMotion();
        memset(pic.refPoc, 0, sizeof(pic.refPoc));
        for (int l = 0; l < 2; l++)
            for (size_t r = 0; r < pic.lists[l].size(); r++) pic.refPoc[l][r] = pic.lists[l][r]->poc;
        pic.finalX.resize(ctuH);
        for (int r = 0; r < ctuH; r++) pic.finalX[r] = xa_counter_alloc();
        pic.analysedCols.assign(ctuH, 0);
    }
    return 0;
}

/* Lookahead::getEstimatedPictureCost (slicetype.cpp:1327-1439) as far as the constant rate factor reads it (Lowres::satdCost: RateControl only asks whether it is zero unless
 * cuTree is off): I and P pictures -- with cuTree the estimate's block costs rescaled by the cuTree offsets (frameCostRecalculate), without it the plain estimate (the
 * reference's costEstAq differs from it by the AQ weights; B pictures, whose QP does not read it, get the plain estimate too) */
int64_t x265amd_encoder::estimatedPictureCost(Pic& pic)
{
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    if (stype == 2)
    {
        if (!p.cuTree || pic.intraCostHost.empty()) return std::max<int64_t>(pic.costEst[0], 1);
        std::vector<uint16_t> lc(pic.intraCostHost.size());
        for (size_t i = 0; i < lc.size(); i++) lc[i] = (uint16_t)std::min(pic.intraCostHost[i], (1 << 14) - 1);        /* lowresIntraEstimate: lowresCosts[0][0] (slicetype.cpp:806) */
        return x265amd_frame_cost_recalculate(&treeParams, lc.data(), pic.qpCuTreeOffset.data());
    }
    const int d0 = pic.poc - pic.lists[0][0]->poc;
    if (stype == 1 && p.cuTree && d0 > 0 && d0 < 18)
    {
        const int key = d0 * 32;
        auto h = pic.lcHost.find(key);
        if (h == pic.lcHost.end())
        {
            auto d = pic.dLc.find(key);
            if (d != pic.dLc.end())
            {
                std::vector<uint16_t> v((size_t)lowCuW * lowCuH);
                if (hipMemcpyAsync(v.data(), d->second, v.size() * 2, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipStreamSynchronize(laStream) == hipSuccess)
                    h = pic.lcHost.emplace(key, std::move(v)).first;
            }
        }
        if (h != pic.lcHost.end()) return x265amd_frame_cost_recalculate(&treeParams, h->second.data(), pic.qpCuTreeOffset.data());
    }
    if (d0 > 0 && d0 < 18 && pic.costEst[d0] > 0) return pic.costEst[d0];
    return 1;
}

/* Analysis::calculateQpforCuSize (analysis.cpp:3634-3714) for every quantisation group of the picture, ahead of the analysis: per CTU the QP of the 64x64 CU, then (qgSize 32)
 * of its four 32x32 CUs in z order -- values up to 69 (what setLambdaFromQP is given; it clips the QP that is coded to 51).  A referenced picture with cuTree takes the cuTree
 * offsets, every other picture the adaptive quantisation's. */
void x265amd_encoder::cuQpTable(Pic& pic)
{
    const int per = maxCuDqpDepth >= 1 ? 5 : 1;
    pic.cuQp.assign((size_t)nctu * per, (int8_t)pic.sliceQp);
    const double* offs = (p.cuTree && pic.type != TYPE_B) ? pic.qpCuTreeOffset.data() : pic.qpAqOffset.data();
    for (int a = 0; a < nctu; a++)
    {
        const int x = (a % ctuW) * 64, y = (a / ctuW) * 64;
        pic.cuQp[(size_t)a * per] = (int8_t)x265amd_cu_qp(pic.avgQpRc, offs, W, H, x, y, 64, 0, 69);
        for (int q = 0; q < 4 && per == 5; q++)
        {
            const int cx = x + (q & 1) * 32, cy = y + (q >> 1) * 32;
            if (cx < W && cy < H) pic.cuQp[(size_t)a * per + 1 + q] = (int8_t)x265amd_cu_qp(pic.avgQpRc, offs, W, H, cx, cy, 32, 0, 69);
        }
    }
}

/* what the analysis and the slice header of one picture need, derived from the picture's lists (DPB::prepareEncode has run) */
/* MotionReference with weights (reference.cpp:51-185): the weighted copy of a reference picture that the motion searches of ONE slice read (luma; chroma too when the
 * sub-sample refinement measures chroma, subme > 2), made CTU row by CTU row as the reference picture's rows become final (applyWeight).  The copy is the pointwise
 * weight_pp_c of the padded plane: the margins of the reconstruction repeat its edge samples, so weighting them is what extending the weighted rows gives. */
struct WPlane
{
    Pic* src = nullptr; x265amd_weight w[3]; pixel* buf = nullptr; bool chroma[3] = { false, false, false };
    std::mutex mu; std::vector<uint8_t> rowDone; hipStream_t st = nullptr;
    ~WPlane() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } xa_scratch_free(buf); }
};
struct FrameCtx
{
    int stype = 0;
    std::vector<uint64_t> planes;           /* reference pictures (distinct), then the weighted copies of this slice (wplanes), then the reconstruction, then the source: 3 addresses each */
    int numRefs = 0;
    std::vector<std::unique_ptr<WPlane>> wplanes;
    int32_t mePic[2][16];
    x265amd_mvpred_info info;
    x265amd_inter_search_params sp;
    x265amd_slice_info si;
    x265amd_analysis_params ap;
    const Pic* colPic = nullptr;
    bool failed = false;
};

static void frameContext(const x265amd_encoder& e, Pic& pic, FrameCtx& c)
{
    const x265amd_param& p = e.p;
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    c.stype = stype;
    const std::vector<PicP>* lists = pic.lists;
    std::vector<Pic*> index;
    int32_t refPic[2][16];
    memset(refPic, 0, sizeof(refPic));
    if (!e.frameParallel) memset(pic.refPoc, 0, sizeof(pic.refPoc));        /* coded in parallel: set by prepare(), other pictures' tasks may be reading it */
    for (int l = 0; l < 2; l++)
        for (size_t r = 0; r < lists[l].size(); r++)
        {
            Pic* q = lists[l][r].get();
            size_t k = std::find(index.begin(), index.end(), q) - index.begin();
            if (k == index.size()) { index.push_back(q); for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(q->finalPlanes(), cc)); }
            refPic[l][r] = (int32_t)k;
            if (!e.frameParallel) pic.refPoc[l][r] = q->poc;
        }
    c.numRefs = (int)index.size();
    memcpy(c.mePic, refPic, sizeof(c.mePic));
    if (pic.weighted)
        for (int l = 0; l < 2; l++)
            for (size_t r = 0; r < lists[l].size(); r++)
            {
                if (!pic.wp[l][r][0].present) continue;         /* MotionReference::init is given weights only when the luma weight is there (frameencoder.cpp:573-577) */
                std::unique_ptr<WPlane> wpl(new WPlane);
                wpl->src = lists[l][r].get();
                memcpy(wpl->w, pic.wp[l][r], sizeof(wpl->w));
                wpl->rowDone.assign(e.ctuH, 0);
                if (xa_scratch_alloc((void**)&wpl->buf, e.picElems * sizeof(pixel)) != hipSuccess || hipStreamCreateWithFlags(&wpl->st, hipStreamNonBlocking) != hipSuccess) { c.failed = true; return; }
                wpl->chroma[0] = true;
                for (int cc = 1; cc < 3; cc++) wpl->chroma[cc] = p.subpelRefine > 2 && pic.wp[l][r][cc].present;     /* numInterpPlanes (reference.cpp:56) */
                c.mePic[l][r] = (int32_t)(c.planes.size() / 3);
                for (int cc = 0; cc < 3; cc++) c.planes.push_back(wpl->chroma[cc] ? e.planeAddr(wpl->buf, cc) : e.planeAddr(wpl->src->finalPlanes(), cc));
                c.wplanes.push_back(std::move(wpl));
            }
    for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(pic.dRec, cc));
    for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(pic.dSrc, cc));

    x265amd_mvpred_info& info = c.info;
    memset(&info, 0, sizeof(info));
    info.pic_width = e.W; info.pic_height = e.H; info.is_inter_b = stype == 0; info.max_num_merge_cand = p.maxNumMergeCand;
    info.num_ref_idx[0] = (int32_t)lists[0].size(); info.num_ref_idx[1] = (int32_t)lists[1].size();
    info.temporal_mvp = p.bEnableTemporalMvp != 0; info.col_from_l0 = stype != 0; info.check_ldc = stype != 0; info.poc = pic.poc;
    memcpy(info.ref_poc, pic.refPoc, sizeof(info.ref_poc));
    c.colPic = stype == 2 ? nullptr : (stype == 1 ? lists[0][0].get() : lists[1][0].get());
    if (c.colPic) { info.col_poc = c.colPic->poc; memcpy(info.col_ref_poc, c.colPic->refPoc, sizeof(info.col_ref_poc)); }

    x265amd_inter_search_params& sp = c.sp;
    memset(&sp, 0, sizeof(sp));
    sp.search_method = p.searchMethod; sp.subpel_refine = p.subpelRefine; sp.search_range = p.searchRange; sp.qp = pic.sliceQp; sp.chroma_mc = 1;
    sp.frame_parallel = e.frameParallel;
    memcpy(sp.ref_pic, refPic, sizeof(sp.ref_pic));
    memcpy(sp.me_pic, c.mePic, sizeof(sp.me_pic));
    if (pic.weighted) { sp.weighted = stype == 1 ? 1 : 2; memcpy(sp.wp, pic.wp, sizeof(sp.wp)); }
    sp.lowres_blocks_in_row = e.lowCuW;
    for (int l = 0; l < 2; l++)
        for (size_t r = 0; r < lists[l].size(); r++)
        {
            const int diffPoc = abs(pic.poc - lists[l][r]->poc);
            const std::vector<int16_t>& f = l ? pic.lowMvs1[diffPoc < 18 ? diffPoc : 0] : pic.lowMvs[diffPoc < 18 ? diffPoc : 0];
            if (diffPoc <= p.bframes + 1 && diffPoc < 18 && !f.empty()) sp.lowres_mvs[l][r] = (uint64_t)(uintptr_t)f.data();
        }

    x265amd_slice_info& si = c.si;
    memset(&si, 0, sizeof(si));
    si.pic_width = e.W; si.pic_height = e.H; si.slice_type = stype; si.slice_qp = pic.sliceQp;
    si.num_ref_idx[0] = info.num_ref_idx[0]; si.num_ref_idx[1] = info.num_ref_idx[1];
    si.max_num_merge_cand = p.maxNumMergeCand; si.sign_hide = p.bEnableSignHiding != 0; si.wpp = p.bEnableWavefront != 0;
    si.use_dqp = e.useDqp; si.max_cu_dqp_depth = e.maxCuDqpDepth;
    si.max_cu_depth = 3; si.max_amp_depth = p.bEnableAMP ? 3 : 0; si.tu_log2_min = 2; si.tu_log2_max = 5;
    si.tu_max_depth_inter = p.tuQTMaxInterDepth; si.tu_max_depth_intra = p.tuQTMaxIntraDepth;

    x265amd_analysis_params& ap = c.ap;
    memset(&ap, 0, sizeof(ap));
    ap.psy_rd = p.psyRd; ap.rd_level = p.rdLevel; ap.early_skip = p.bEnableEarlySkip != 0; ap.rskip = p.recursionSkipMode; ap.limit_refs = p.limitReferences;
    ap.b_intra = p.bIntraInBFrames != 0; ap.rect = p.bEnableRectInter != 0; ap.amp = p.bEnableAMP != 0; ap.limit_modes = p.limitModes != 0;
    ap.strong_intra_smoothing = p.bEnableStrongIntraSmoothing != 0; ap.use_sao = p.bEnableSAO != 0;
    ap.fast_intra = p.bEnableFastIntra != 0;
    ap.rdoq_level = p.rdoqLevel; ap.psy_rdoq_scale = p.rdoqLevel ? p.psyRdoqFix8 : 0;      /* encoder.cpp:3667: no psy-rdoq without RDOQ */
}

/* slice header (Entropy::codeSliceHeader inputs as DPB / Encoder set them) + the sub-streams -> the picture's NAL unit */
static int sliceNal(const x265amd_encoder& e, Pic& pic, const FrameCtx& c, const int32_t* saoFlags, const std::vector<uint8_t>& data, const std::vector<uint32_t>& sizes, int nsub)
{
    const x265amd_param& p = e.p;
    x265amd_slice_header h;
    memset(&h, 0, sizeof(h));
    h.nal_unit_type = pic.nalType; h.temporal_id_plus1 = 1; h.first_in_access_unit = 1;
    h.slice_type = c.stype; h.poc = pic.poc; h.last_idr_poc = pic.lastIDR; h.log2_max_poc_lsb = 8; h.rps_idx = -1; h.num_rps_in_sps = 0;
    h.num_negative = (int32_t)pic.neg.size(); h.num_positive = (int32_t)pic.pos.size();
    {
        int j = 0;
        for (const PicP& q : pic.neg) { h.delta_poc[j] = q->poc - pic.poc; h.used[j++] = pic.rpsUsed; }
        for (const PicP& q : pic.pos) { h.delta_poc[j] = q->poc - pic.poc; h.used[j++] = pic.rpsUsed; }
    }
    h.temporal_mvp_enabled = p.bEnableTemporalMvp != 0;
    h.use_sao = p.bEnableSAO != 0; h.sao_luma = saoFlags[0]; h.sao_chroma = saoFlags[1];
    h.num_ref_idx[0] = c.info.num_ref_idx[0]; h.num_ref_idx[1] = c.info.num_ref_idx[1]; h.num_ref_idx_default[0] = h.num_ref_idx_default[1] = 1;
    h.col_from_l0 = c.stype != 0; h.col_ref_idx = 0; h.max_num_merge_cand = p.maxNumMergeCand;
    h.slice_qp = pic.sliceQp; h.pps_init_qp = 26; h.deblocking_disabled = !p.bEnableLoopFilter;
    h.slfase_flag = (0x5f4e4a53u >> (pic.poc % 31)) & 1;                                              /* SLFASE_CONSTANT (dpb.cpp:294) */
    h.wpp = p.bEnableWavefront != 0;
    h.weighted_pred = p.bEnableWeightedPred != 0; h.weighted_bipred = p.bEnableWeightedBiPred != 0; h.luma_log2_weight_denom = pic.lumaDenom; h.chroma_log2_weight_denom = pic.chromaDenom;
    memcpy(h.wp, pic.wp, sizeof(h.wp));
    size_t dataBytes = 0;
    for (int s = 0; s < nsub; s++) dataBytes += sizes[s];
    pic.nalBytes.assign(dataBytes * 3 / 2 + 4096, 0);
    const size_t n = x265amd_write_slice_nal(&h, data.data(), sizes.data(), nsub, pic.nalBytes.data(), pic.nalBytes.size());
    if (!n || n > pic.nalBytes.size()) return xa_fail(X265AMD_EINVAL, "encoder: slice NAL");
    pic.nalBytes.resize(n);
    return 0;
}

/* FrameEncoder::compressFrame for one picture (its own thread and HIP stream).  The analysis starts when every reference picture is final; the
 * in-loop filters, SAO (its decision carries state from picture to picture, SAO::m_depthSaoRate) and the shared filter scratch run in coding
 * order, i.e. after the previous picture's task. */
int x265amd_encoder::runFrame(const PicP& picp, std::shared_future<int> prev)
{
    Pic& pic = *picp;
    for (int l = 0; l < 2; l++)
        for (const PicP& q : pic.lists[l]) if (q->done.valid() && q->done.get() != 0) return X265AMD_EHIP;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const std::vector<PicP>* lists = pic.lists;
    FrameCtx fc;
    frameContext(*this, pic, fc);
    if (fc.failed) return xa_fail(X265AMD_EHIP, "encoder: weighted reference planes");
    for (auto& wpl : fc.wplanes)            /* one picture at a time: the reference pictures are complete, so are their weighted copies */
        if (weightRows(*wpl, 0, ctuH - 1) != X265AMD_OK) return X265AMD_EHIP;
    const int stype = fc.stype;
    std::vector<uint64_t>& planes = fc.planes;
    x265amd_mvpred_info& info = fc.info;
    x265amd_inter_search_params& sp = fc.sp;
    x265amd_slice_info& si = fc.si;
    x265amd_analysis_params& ap = fc.ap;
    const Pic* colPic = fc.colPic;
    (void)stype;

    const size_t nUnits = (size_t)w4 * h4;
    pic.units.assign(nUnits, x265amd_cu_unit()); pic.motion.assign(nUnits, x265amd_mv_unit());
    memset(pic.units.data(), 0, sizeof(x265amd_cu_unit) * nUnits); memset(pic.motion.data(), 0, sizeof(x265amd_mv_unit) * nUnits);
    std::vector<x265amd_mv_unit> noCol;
    if (!colPic) { noCol.resize(nUnits); memset(noCol.data(), 0, sizeof(x265amd_mv_unit) * nUnits); }
    std::vector<uint8_t> refDepth(2 * nUnits, 0);
    std::vector<int8_t> refQp0(2 * (size_t)nctu, 0);
    for (int l = 0; l < 2; l++)
        if (!lists[l].empty())
        {
            const Pic* q = lists[l][0].get();
            for (size_t i = 0; i < nUnits; i++) refDepth[l * nUnits + i] = q->units[i].depth;
            for (int i = 0; i < nctu; i++) refQp0[(size_t)l * nctu + i] = useDqp ? q->units[(size_t)(i / ctuW) * 16 * w4 + (size_t)(i % ctuW) * 16].qp : (int8_t)q->sliceQp;      /* CUData::m_qp[0] of the co-located CTU */
        }
    std::vector<x265amd_cu_stat> stat((size_t)nctu + 1);
    memset(stat.data(), 0, sizeof(x265amd_cu_stat) * stat.size());
    std::vector<int16_t> coeff((size_t)nctu * RD_TILE_ELEMS, 0);
    std::vector<uint8_t> data((size_t)W * H * 3 + (1u << 16));
    std::vector<uint32_t> sizes((size_t)ctuH + 1, 0);
    int nsub = 0;
    const bool sao = p.bEnableSAO != 0;
    if (useDqp && pic.cuQp.empty()) return xa_fail(X265AMD_EINVAL, "encoder: the picture has no CU QPs");
    int rc = xa_analyse_frame(me, st, &info, &sp, &si, &ap, pic.units.data(), pic.motion.data(), colPic ? colPic->motion.data() : noCol.data(),
                              refDepth.data(), refQp0.data(), planes.data(), (int)(planes.size() / 3), stride, cstride, stat.data(), coeff.data(), nullptr,
                              sao ? nullptr : data.data(), data.size(), sizes.data(), &nsub, nullptr, useDqp ? pic.cuQp.data() : nullptr);
    if (rc != X265AMD_OK) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: analysis");

    /* ---- from here on in coding order ---- */
    if (prev.valid() && prev.get() != 0) return X265AMD_EHIP;
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    if (p.bEnableLoopFilter)
    {
        std::vector<x265amd_deblock_unit> dbu(nUnits);
        rc = x265amd_deblock_units(&si, &info, pic.units.data(), pic.motion.data(), dbu.data());
        if (rc != X265AMD_OK) return rc;
        if (hipMemcpyAsync(dDbUnits, dbu.data(), sizeof(x265amd_deblock_unit) * nUnits, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: deblock upload");
        rc = x265amd_deblock_picture(st, recY, recU, recV, stride, cstride, W, H, dDbUnits, 0, 0, 0, 0, 0, 3);
        if (rc != X265AMD_OK) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: deblock");
    }
    int32_t saoFlags[2] = { 0, 0 };
    if (sao)
    {
        const size_t nstat = (size_t)nctu * 3 * 5 * 32;
        const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
        const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
        if (hipMemsetAsync(dSaoCount, 0, nstat * 4, st) != hipSuccess || hipMemsetAsync(dSaoOrg, 0, nstat * 4, st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: sao memset");
        rc = x265amd_sao_stats(st, recP, srcP, stride, cstride, W, H, dSaoCount, dSaoOrg);
        if (rc != X265AMD_OK) return rc;
        std::vector<int32_t> cnt(nstat), orgs(nstat);
        if (hipMemcpyAsync(cnt.data(), dSaoCount, nstat * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(orgs.data(), dSaoOrg, nstat * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: sao download");
        std::vector<x265amd_sao_ctu> sparams((size_t)nctu);
        memset(sparams.data(), 0, sizeof(x265amd_sao_ctu) * nctu);
        rc = x265amd_sao_rdo(&si, pic.type != TYPE_B ? 1 : 0, 1, 0, 69,       /* IS_REFERENCED: fixed by the type (hasReferences changes as later pictures are prepared) */
                             pic.units.data(), cnt.data(), orgs.data(), depthSaoRate, sparams.data(), saoFlags);
        if (rc != X265AMD_OK) return rc;
        if (hipMemcpyAsync(dSaoParams, sparams.data(), sizeof(x265amd_sao_ctu) * nctu, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dSaoTmp, pic.dRec, picElems * sizeof(pixel), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: sao upload");
        const uint64_t dstP[3] = { planeAddr(dSaoTmp, 0), planeAddr(dSaoTmp, 1), planeAddr(dSaoTmp, 2) };
        rc = x265amd_sao_apply(st, recP, dstP, stride, cstride, W, H, dSaoParams);
        if (rc != X265AMD_OK) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: sao");
        std::swap(pic.dRec, dSaoTmp);
        recY = pic.dRec + org[0]; recU = pic.dRec + org[1]; recV = pic.dRec + org[2];
        rc = x265amd_encode_slice_data(&si, pic.units.data(), coeff.data(), sparams.data(), saoFlags, data.data(), data.size(), sizes.data(), &nsub);
        if (rc != X265AMD_OK) return rc;
    }
    /* the reconstruction becomes a reference: extend its borders */
    rc = x265amd_extend_pic_border(st, recY, stride, W, H, marginX, marginY);
    if (rc == X265AMD_OK) rc = x265amd_extend_pic_border(st, recU, cstride, W / 2, H / 2, marginX / 2, marginY / 2);
    if (rc == X265AMD_OK) rc = x265amd_extend_pic_border(st, recV, cstride, W / 2, H / 2, marginX / 2, marginY / 2);
    if (rc != X265AMD_OK) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: border extension");

    rc = sliceNal(*this, pic, fc, saoFlags, data, sizes, nsub);
    if (rc) return rc;
    /* the source is no longer needed (unless the weight analysis of later pictures reads its chroma planes: weightAnalyse works on source pictures); the reconstruction
     * stays while the picture is referenced.  The reference lists are only needed by pictures that are still to come through their own lists */
    if (!keepSources()) { xa_scratch_free(pic.dSrc); pic.dSrc = nullptr; }
    return 0;
}

/* ---- pictures coded in parallel (param.frameNumThreads > 1) ----
 * FrameEncoder::compressFrame as the reference runs it with several frame encoders (frameencoder.cpp:880-960, :1930-1960; framefilter.cpp:559-664): a CTU row
 * of this picture starts when every reference picture has finished the rows down to refLagRows below it, and the in-loop filters follow the analysis row by
 * row so that the rows of this picture become available to the pictures that reference it while its lower rows are still being analysed. */
struct RowGate { x265amd_encoder* e; Pic* pic; std::vector<Pic*> refs; std::vector<uint8_t>* refDepth; size_t nUnits; FrameCtx* fc = nullptr; std::vector<int8_t>* refQp0 = nullptr; };

/* What a CTU may read of a reference picture, and when.  The reference waits for whole rows: row + refLagRows rows of every reference picture before a row
 * starts (frameencoder.cpp:893-908).  What the row's commands can actually read is less -- vectors end searchRange samples below the block (search.cpp:92,
 * :2763; merge / AMVP candidates beyond are left out), plus sub-sample steps and interpolation taps: the rows row - 2 .. row + 2 at most -- and pictures here
 * are published by COLUMNS (Pic::finalX, filterRowsCols): a CTU starts when those rows of every reference picture are final two CTUs to its right, which covers
 * ordinary vectors, the co-located CTUs' motion and the co-located depths; every command that reads reference samples first asks gateRefReady with its exact
 * reach (xa_ref_guard_*), so a long vector waits for exactly what it needs.  A picture therefore follows its reference pictures a few CTUs behind instead of
 * rows behind, and the slow last CTU row of a picture (cut CTUs when the height is no multiple of 64) no longer holds up every picture behind it.  Results do
 * not depend on any of this: a sample is only ever read when it is final.  (Deadlock freedom with few device queues: a row takes its queue when it starts, and
 * it starts only when finalX of the rows it will follow is above zero, i.e. when those rows hold their queues and run.) */
/* the per-CTU gate: the rows row - 2 .. row + 1 of every reference picture final up to 56 samples beyond the CTU to the right (vectors up to that length, the
 * co-located CTUs' motion and depths), row + 2 begun (a vector reaching into its first lines is rare: gateRefWait then waits for it, and the row holds its queue) */
static inline int gateNeed(const x265amd_encoder& e, int col) { return std::min(e.W, 64 * col + 120); }
static int gateCtuReady(void* ctx, int row, int col)        /* polled (the start condition of a row task): 1 yes, 0 not yet, -1 a reference picture failed */
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    const int need = gateNeed(e, col), r0 = std::max(0, row - 2), r1 = std::min(e.ctuH - 1, row + 1);
    for (Pic* q : g.refs)
    {
        if (q->failed.load(std::memory_order_acquire)) return -1;
        if (row + 2 < e.ctuH && q->published(row + 2) < 1) return 0;
        for (int r = r1; r >= r0; r--) if (q->published(r) < need) return 0;
    }
    return 1;
}
static int gateRowReady(void* ctx, int row) { return gateCtuReady(ctx, row, 0); }
static int gateCtuWait(void* ctx, int row, int col)         /* blocking: the task parks on the counters */
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    const int need = gateNeed(e, col), r0 = std::max(0, row - 2), r1 = std::min(e.ctuH - 1, row + 1);
    static const bool pubLog = getenv("X265AMD_PUB_LOG") != nullptr;
    for (Pic* q : g.refs)
        for (int r = r1; r >= r0; r--)
        {
            if (q->published(r) < need)
            {
                const double t0 = pubLog ? Pic::pubClockMs() : 0;
                xa_wait_counter(q->finalX[r], (uint64_t)need);
                if (pubLog) fprintf(stderr, "x265amd gate: poc %d row %d col %d waited from %.2f to %.2f for poc %d row %d x %d\n", g.pic->poc, row, col, t0, Pic::pubClockMs(), q->poc, r, need);
            }
            if (q->failed.load(std::memory_order_acquire)) return -1;
        }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
static int gateRefWait(void* ctx, int picIdx, int yMin, int yMax, int xMax)
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    if (picIdx >= (int)g.refs.size() && g.fc && picIdx - (int)g.refs.size() < (int)g.fc->wplanes.size())
    {
        /* a weighted copy (a motion search of a slice with weights): whole CTU rows of the reference picture, then the copy's rows (MotionReference::applyWeight at the
         * row's start, frameencoder.cpp:900-908) */
        WPlane& wpl = *g.fc->wplanes[picIdx - (int)g.refs.size()];
        Pic* q = wpl.src;
        const int r0 = std::min(std::max(yMin, 0), e.H - 1) >> 6, r1 = std::min(std::max(yMax, 0), e.H - 1) >> 6;
        const int wc = xa_task_wait_class(3);
        for (int r = r1; r >= r0; r--)
        {
            if (q->published(r) < e.W) xa_wait_counter(q->finalX[r], (uint64_t)e.W);
            if (q->failed.load(std::memory_order_acquire)) { xa_task_wait_class(wc); return -1; }
        }
        xa_task_wait_class(wc);
        std::atomic_thread_fence(std::memory_order_acquire);
        return g.e->weightRows(wpl, r0, r1) == X265AMD_OK ? 0 : -1;
    }
    if (picIdx < 0 || picIdx >= (int)g.refs.size()) return 0;          /* the picture itself / the source: not a reference */
    Pic* q = g.refs[picIdx];
    const int need = xMax >= e.W - 1 ? e.W : std::max(0, xMax + 1);
    const int r0 = std::min(std::max(yMin, 0), e.H - 1) >> 6, r1 = std::min(std::max(yMax, 0), e.H - 1) >> 6;
    const int wc = xa_task_wait_class(3);
    for (int r = r1; r >= r0; r--)
    {
        if (q->published(r) < need) xa_wait_counter(q->finalX[r], (uint64_t)need);
        if (q->failed.load(std::memory_order_acquire)) { xa_task_wait_class(wc); return -1; }
    }
    xa_task_wait_class(wc);
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
/* what gateCtuWait(row, col) has waited for (XaRowHooks::ctu_reach) */
static void gateCtuReach(void* ctx, int row, int col, int* r0, int* r1, int* need)
{
    const x265amd_encoder& e = *((RowGate*)ctx)->e;
    *need = gateNeed(e, col); *r0 = std::max(0, row - 2); *r1 = std::min(e.ctuH - 1, row + 1);
}
static void gateBeforeRow(void*, int) {}
static void gateBeforeCtu(void* ctx, int row, int col)
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    /* the co-located CTU's depths (topSkipMinDepth reads refFrameList[l][0] at this CTU's address): that CTU of the reference picture is coded now */
    for (int l = 0; l < 2; l++)
        if (!g.pic->lists[l].empty())
        {
            const Pic* q = g.pic->lists[l][0].get();
            const int y0 = row * 16, y1 = std::min(e.h4, y0 + 16), x0 = col * 16, x1 = std::min(e.w4, x0 + 16);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) (*g.refDepth)[l * g.nUnits + (size_t)y * e.w4 + x] = q->units[(size_t)y * e.w4 + x].depth;
            /* ... and its first unit's QP (topSkipMinDepth's previousQP: CUData::m_qp[0] of that CTU as it was coded) */
            if (e.useDqp && g.refQp0) (*g.refQp0)[(size_t)l * e.nctu + (size_t)row * e.ctuW + col] = q->units[(size_t)y0 * e.w4 + x0].qp;
        }
}
static void gateAfterCtu(void* ctx, int row, int col)
{
    RowGate& g = *(RowGate*)ctx;
    { std::lock_guard<std::mutex> lk(g.pic->mu); g.pic->analysedCols[row] = col + 1; }
    g.pic->cv.notify_all();
}
static void gateAfterRow(void* ctx, int row)
{
    RowGate& g = *(RowGate*)ctx;
    { std::lock_guard<std::mutex> lk(g.pic->mu); g.pic->analysedRows = row + 1; }
    g.pic->cv.notify_all();
}

/* CTU rows r0 .. r1 of a weighted copy (MotionReference::applyWeight, reference.cpp:118-185): weight_pp_c over the rows' padded lines -- with the first row the top margin, with
 * the last the bottom margin -- of the planes that carry a weight.  The rows of the reference picture are final (the caller has waited for them). */
int x265amd_encoder::weightRows(WPlane& wpl, int r0, int r1)
{
    std::lock_guard<std::mutex> lk(wpl.mu);
    bool any = false;
    xa_thread_device();
    for (int r = r0; r <= r1; r++)
    {
        if (r < 0 || r >= ctuH || wpl.rowDone[r]) continue;
        for (int cc = 0; cc < 3; cc++)
        {
            if (!wpl.chroma[cc]) continue;
            const int sh = cc ? 1 : 0, h = H >> sh, my = marginY >> sh, mx = marginX >> sh, rows = 64 >> sh;
            const intptr_t st = cc ? cstride : stride;
            const int y0 = r == 0 ? -my : rows * r, y1 = r == ctuH - 1 ? h + my : rows * (r + 1);
            const intptr_t at = (intptr_t)org[cc] + (intptr_t)y0 * st - mx;
            const x265amd_weight& w = wpl.w[cc];
            const int correction = 14 - X265AMD_DEPTH;
            const int rc = x265amd_weight_buffer(wpl.st, wpl.src->finalPlanes() + at, wpl.buf + at, (size_t)(y1 - y0) * st, w.w, (w.denom ? 1 << (w.denom - 1) : 0) << correction,
                                                 w.denom + correction, w.o * (1 << (X265AMD_DEPTH - 8)));
            if (rc != X265AMD_OK) return rc;
        }
        any = true;
    }
    if (any && hipStreamSynchronize(wpl.st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: weighted reference rows");
    for (int r = std::max(r0, 0); r <= r1 && r < ctuH; r++) wpl.rowDone[r] = 1;
    return X265AMD_OK;
}

/* The filter thread of a picture: FrameFilter::processRow / processPostRow for each CTU row as the analysis delivers it.  Row r is deblocked when row r + 1
 * is analysed (its vertical edges, then its horizontal edges, which reach three samples up into row r - 1); its SAO statistics follow (they leave out the samples
 * the rows below still change) and its parameters are decided; row r - 1 can then be offset (its last lines and the line below them are final), its borders
 * extended and the row published.  The last row publishes itself. */
int x265amd_encoder::filterRows(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags)
{
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const bool sao = p.bEnableSAO != 0, dbl = p.bEnableLoopFilter != 0;
    const size_t nUnits = (size_t)w4 * h4, nstat = (size_t)nctu * 3 * 5 * 32, rowStat = (size_t)ctuW * 3 * 5 * 32;
    struct Scratch { void* p = nullptr; ~Scratch() { xa_scratch_free(p); } } dDb, dCnt, dOrg, dPar;
    std::vector<x265amd_deblock_unit> dbu;
    std::vector<int32_t> cnt, orgs;
    if (dbl) { if (xa_scratch_alloc(&dDb.p, sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: device allocation"); dbu.resize(nUnits); }
    if (sao)
    {
        if (xa_scratch_alloc(&dCnt.p, nstat * 4) != hipSuccess || xa_scratch_alloc(&dOrg.p, nstat * 4) != hipSuccess || xa_scratch_alloc(&dPar.p, sizeof(x265amd_sao_ctu) * nctu) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: device allocation");
        cnt.resize(nstat); orgs.resize(nstat);
        saoFlags[0] = saoFlags[1] = 1;          /* SAO::startSlice: never switched off when pictures are coded in parallel (sao.cpp:264) */
    }
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    pixel* fin = sao ? pic.dFin : pic.dRec;
    const uint64_t finP[3] = { planeAddr(fin, 0), planeAddr(fin, 1), planeAddr(fin, 2) };
    double unusedRate[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int rc = X265AMD_OK;
    /* offsets, borders, publication of CTU row k */
    auto finish = [&](int k) -> int {
        const int y0 = k * 64, y1 = std::min(H, y0 + 64);
        if (sao)
        {
            int r = x265amd_sao_apply_rows(st, recP, finP, stride, cstride, W, H, (const x265amd_sao_ctu*)dPar.p, k, k + 1);
            if (r != X265AMD_OK) return r;
        }
        int r = x265amd_extend_border_rows(st, fin + org[0], stride, W, H, marginX, marginY, y0, y1);
        if (r == X265AMD_OK) r = x265amd_extend_border_rows(st, fin + org[1], cstride, W / 2, H / 2, marginX / 2, marginY / 2, y0 / 2, y1 / 2);
        if (r == X265AMD_OK) r = x265amd_extend_border_rows(st, fin + org[2], cstride, W / 2, H / 2, marginX / 2, marginY / 2, y0 / 2, y1 / 2);
        if (r != X265AMD_OK) return r;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: row filters");
        pic.publish(k, W);
        return X265AMD_OK;
    };
    static const bool timing = getenv("X265AMD_TIMING") != nullptr;
    double tPh[6] = { 0, 0, 0, 0, 0, 0 };
    auto tLast = std::chrono::steady_clock::now();
    auto stamp = [&](int k) { if (!timing) return; const auto n = std::chrono::steady_clock::now(); tPh[k] += std::chrono::duration<double, std::milli>(n - tLast).count(); tLast = n; };
    struct Report { const bool& on; double* t; int poc; ~Report() { if (on) fprintf(stderr, "x265amd: filter rows of poc %d (ms): waiting %.1f, deblock units + upload %.1f, deblock %.1f, sao statistics %.1f, sao decision + upload %.1f, offsets + borders %.1f\n", poc, t[0], t[1], t[2], t[3], t[4], t[5]); } } report{ timing, tPh, pic.poc };
    for (int r = 0; r < ctuH && rc == X265AMD_OK; r++)
    {
        stamp(5);
        {
            std::unique_lock<std::mutex> lk(pic.mu);
            /* intra prediction of row r + 1 reads the unfiltered last line of row r: FrameEncoder::m_filterRowDelay (frameencoder.cpp:124-126, :1936-1950) */
            const int needRows = (dbl || sao) ? std::min(ctuH, r + 2) : r + 1;
            pic.cv.wait(lk, [&] { return pic.analysedRows >= needRows || pic.failed; });
            if (pic.failed) return X265AMD_EHIP;
        }
        stamp(0);
        const int y4b = r * 16, y4e = std::min(h4, y4b + 16);
        if (dbl)
        {
            rc = x265amd_deblock_units_rows(&si, &info, pic.units.data(), pic.motion.data(), dbu.data(), y4b, y4e);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync((x265amd_deblock_unit*)dDb.p + (size_t)y4b * w4, dbu.data() + (size_t)y4b * w4, sizeof(x265amd_deblock_unit) * (size_t)(y4e - y4b) * w4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: deblock upload"); break; }
            stamp(1);
            rc = x265amd_deblock_rows(st, recY, recU, recV, stride, cstride, W, H, (const x265amd_deblock_unit*)dDb.p, 0, 0, 0, 0, 0, 3, y4b, y4e);
            if (rc != X265AMD_OK) break;
        }
        if (sao)
        {
            if (hipMemsetAsync((int32_t*)dCnt.p + r * rowStat, 0, rowStat * 4, st) != hipSuccess || hipMemsetAsync((int32_t*)dOrg.p + r * rowStat, 0, rowStat * 4, st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao memset"); break; }
            rc = x265amd_sao_stats_rows(st, recP, srcP, stride, cstride, W, H, (int32_t*)dCnt.p, (int32_t*)dOrg.p, r, r + 1);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync(cnt.data() + r * rowStat, (int32_t*)dCnt.p + r * rowStat, rowStat * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(orgs.data() + r * rowStat, (int32_t*)dOrg.p + r * rowStat, rowStat * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao download"); break; }
            stamp(3);
            int32_t flags[2] = { 1, 1 };
            rc = x265amd_sao_rdo_rows(&si, pic.type != TYPE_B ? 1 : 0, 2, 0, 69, pic.units.data(), cnt.data(), orgs.data(), unusedRate, sparams.data(), flags, r, r + 1);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync((x265amd_sao_ctu*)dPar.p + (size_t)r * ctuW, sparams.data() + (size_t)r * ctuW, sizeof(x265amd_sao_ctu) * ctuW, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao upload"); break; }
            stamp(4);
        }
        if (!dbl && !sao) { rc = finish(r); continue; }         /* nothing below changes this row */
        if (r > 0) rc = finish(r - 1);
        if (rc == X265AMD_OK && r == ctuH - 1) rc = finish(r);
    }
    return rc;
}

/* Waiting for a stream without burning a core: hipStreamSynchronize polls flat out, and the filter threads of twenty pictures in flight did that beside the worker
 * threads -- past the CPU quota of the box (16 cores), where the kernel then freezes every thread of the process for the rest of its 100 ms period (cgroup cpu.stat:
 * nr_throttled; a dozen milliseconds each time, in the middle of the encode).  An event, a short poll for the common case (the work is a few kernels), then naps. */
static hipError_t streamWaitPolite(hipStream_t st, hipEvent_t ev)
{
    static const bool off = getenv("X265AMD_FILTER_SPIN") && atoi(getenv("X265AMD_FILTER_SPIN")) != 0;
    static const int spinUs = getenv("X265AMD_FILTER_SPIN_US") ? atoi(getenv("X265AMD_FILTER_SPIN_US")) : 30;
    if (off || !ev) return hipStreamSynchronize(st);
    hipError_t e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;)
    {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spinUs)) { for (int k = 0; k < 16; k++) __builtin_ia32_pause(); continue; }
        struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr);
        /* a filter stream that stands for seconds: say so once (the kernels behind the event are a few microseconds each) */
        static std::atomic<int> said{ 0 };
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3) && said.fetch_add(1) < 4)
            fprintf(stderr, "x265amd: a filter stream has not reached its event for 3 s (stream %p, hipStreamQuery says %s)\n", (void*)st, hipGetErrorName(hipStreamQuery(st)));
    }
}

/* The filter thread of a picture, by columns.  A UNIT is a CTU row r and a range of its CTU columns [c0, c1): the deblocking of the unit's edges (vertical edges
 * right of c0's left boundary up to and including c1's left boundary, then the horizontal edges of the columns, the top one reaching three samples into row
 * r - 1), the SAO statistics and decisions of its CTUs; behind it row r - 1 (and the last row itself) is offset, its borders extended and its columns published
 * up to eight samples short of the unit's right end (the offsets of those need the next unit's horizontal edges).  A unit is ready when
 *   - row r is analysed through CTU c1 (the vertical edge at its right boundary reads both sides' coding data),
 *   - row r + 1 is analysed through CTU c1 (its intra prediction has then read everything it needs of row r's last line UNFILTERED:
 *     FrameEncoder::m_filterRowDelay, frameencoder.cpp:124-126, :1936-1950),
 *   - the units of row r - 1 cover the columns (their vertical edges precede this unit's top horizontal edge, their decisions are the merge-up candidates).
 * The same samples as FrameFilter's row order produce (framefilter.cpp:559-664): vertical edges lie eight samples apart and touch three on either side, so
 * their order is free; a horizontal edge reads its own columns behind the vertical edges on both sides; a CTU's statistics leave out what the CTUs to its right
 * and below still change (sao.cpp:760-776); an offset sample is written when it and its neighbours are final.  Units are taken as large as the analysis
 * allows (a row the filter falls behind on is caught up in one unit), at least `minChunk` CTUs. */
int x265amd_encoder::filterRowsCols(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags)
{
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    hipEvent_t ev = nullptr;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    struct EventGuard { hipEvent_t e; ~EventGuard() { if (e) (void)hipEventDestroy(e); } } evGuard{ ev };
    const bool sao = p.bEnableSAO != 0, dbl = p.bEnableLoopFilter != 0;
    const size_t nUnits = (size_t)w4 * h4, ctuStat = (size_t)3 * 5 * 32, nstat = (size_t)nctu * ctuStat;
    /* device: deblocking records, SAO parameters.  Pinned host memory the device reads / writes in place (no staging copies, no synchronisation to free a
     * staging buffer): the records as the host derives them (copied to the device in stream order), the statistics as the kernel stores them, the parameters as
     * decided (copied in stream order). */
    struct Scratch { void* p = nullptr; ~Scratch() { xa_scratch_free(p); } } dDb, dPar;
    XaMapped hDb, hPar; XaMappedOut hCnt, hOrg;
    if (dbl && (xa_scratch_alloc(&dDb.p, sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess || hDb.alloc(sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess))
        return xa_fail(X265AMD_EHIP, "encoder: device allocation");
    if (sao)
    {
        if (xa_scratch_alloc(&dPar.p, sizeof(x265amd_sao_ctu) * nctu) != hipSuccess || hPar.alloc(sizeof(x265amd_sao_ctu) * nctu) != hipSuccess ||
            hCnt.alloc(nstat * 4) != hipSuccess || hOrg.alloc(nstat * 4) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: device allocation");
        saoFlags[0] = saoFlags[1] = 1;          /* SAO::startSlice: never switched off when pictures are coded in parallel (sao.cpp:264) */
    }
    x265amd_deblock_unit* dbu = (x265amd_deblock_unit*)hDb.p;
    int32_t* cnt = (int32_t*)hCnt.p; int32_t* orgs = (int32_t*)hOrg.p;
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    pixel* fin = sao ? pic.dFin : pic.dRec;
    const uint64_t finP[3] = { planeAddr(fin, 0), planeAddr(fin, 1), planeAddr(fin, 2) };
    static const bool timing = getenv("X265AMD_TIMING") != nullptr;
    double tWait = 0, tWork = 0; int numUnits = 0, numSweeps = 0;
    auto tLast = std::chrono::steady_clock::now();
    auto lap = [&](double& acc) { if (!timing) return; const auto n = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(n - tLast).count(); tLast = n; };
    struct Report { const bool& on; double& w; double& k; int& n; int& sw; int poc; ~Report() { if (on) fprintf(stderr, "x265amd: filter units of poc %d: %d units in %d sweeps, %.1f ms waiting for the analysis, %.1f ms filtering\n", poc, n, sw, w, k); } } report{ timing, tWait, tWork, numUnits, numSweeps, pic.poc };
    /* a picture nobody references is waited for by nobody: whole rows */
    /* the offsets' parameters are read by the kernel where the host wrote them (mapped memory: device memory behind the BAR unless X265AMD_PUSH_RECORDS=0 put the pools
     * into host memory, where a read per sample would cross PCIe: then they are copied as before) */
    static const bool parCopy = (getenv("X265AMD_SAO_PARAMS_COPY") && atoi(getenv("X265AMD_SAO_PARAMS_COPY")) != 0) || (getenv("X265AMD_PUSH_RECORDS") && atoi(getenv("X265AMD_PUSH_RECORDS")) == 0);
    static const int minChunkEnv = getenv("X265AMD_FILTER_CHUNK") ? atoi(getenv("X265AMD_FILTER_CHUNK")) : 0;
    const int minChunk = pic.type == TYPE_B ? ctuW : (minChunkEnv > 0 ? minChunkEnv : 2);
    /* the last rows are where a chain of pictures waits for each other (they finish last, and cut CTUs make the last row the slowest): every CTU of them at once */
    auto minChunkOf = [&](int r) { return (pic.type != TYPE_B && r >= ctuH - 3) ? 1 : minChunk; };
    std::vector<int> doneTop((size_t)ctuH, 0), doneFull((size_t)ctuH, 0), pubX((size_t)ctuH, 0), a((size_t)ctuH, 0);
    std::vector<uint8_t> carry((size_t)ctuH * (X265AMD_CTX_STRIDE + 8), 0);
    struct Unit { int r, c0, c1; };
    std::vector<Unit> todoTop, todoFull;
    int rc = X265AMD_OK;
    /* offsets and borders of the sample columns [pubX[k], newX) of CTU row k (enqueued; published behind the sweep's synchronisation) */
    auto finishCols = [&](int k, int newX) -> int {
        const int x0 = pubX[k];
        if (newX <= x0) return X265AMD_OK;
        const int y0 = k * 64, y1 = std::min(H, y0 + 64);
        if (sao)
        {
            int r = x265amd_sao_apply_rows_cols(st, recP, finP, stride, cstride, W, H, parCopy ? (const x265amd_sao_ctu*)dPar.p : (const x265amd_sao_ctu*)hPar.p, k, k + 1, x0, newX);
            if (r != X265AMD_OK) return r;
        }
        return xa_extend_border_band_420(st, fin + org[0], fin + org[1], fin + org[2], stride, cstride, W, H, marginX, marginY, y0, y1, x0, newX, x0 == 0, newX == W);
    };
    /* A CTU row's unit in two steps (round 4).  TOP: the vertical edges of the row's first eight lines and its top horizontal edge -- which completes the deblocking of
     * the row ABOVE -- as soon as the row itself is analysed (nothing of this touches the row's last line, which the row below still reads unfiltered); the row above can
     * then be offset, extended and published: one CTU row earlier than when everything waited for the row below.  FULL: the other vertical edges, the inner horizontal
     * edges, the statistics and the decisions, when the row below is analysed (FrameEncoder::m_filterRowDelay).  Vertical edges are decided per four lines and touch only
     * their own lines, the top horizontal edge touches lines 0-2: the samples are those of the reference's order (X265AMD_FILTER_EARLY_TOP=0: both steps together). */
    static const bool earlyTop = !(getenv("X265AMD_FILTER_EARLY_TOP") && atoi(getenv("X265AMD_FILTER_EARLY_TOP")) == 0);
    auto colsOf = [&](const std::vector<int>& an, int r) -> int { return an[r] == ctuW ? ctuW : an[r] - 1; };
    auto limTop = [&](const std::vector<int>& an, int r) -> int {
        int lim = colsOf(an, r);
        if (r > 0) lim = std::min(lim, doneFull[r - 1]);
        if (!earlyTop && r + 1 < ctuH) lim = std::min(lim, colsOf(an, r + 1));
        return lim;
    };
    auto limFull = [&](const std::vector<int>& an, int r) -> int {
        int lim = doneTop[r];
        if (r + 1 < ctuH) lim = std::min(lim, colsOf(an, r + 1));
        return lim;
    };
    auto chunkOk = [&](int r, int c0, int c1) { return c1 > c0 && (c1 == ctuW || c1 - c0 >= minChunkOf(r)); };
    for (;;)
    {
        {
            std::unique_lock<std::mutex> lk(pic.mu);
            /* something to do? (a snapshot of the analysis: the rows only advance) */
            auto ready = [&]() -> bool {
                if (pic.failed) return true;
                bool allDone = true;
                for (int r = 0; r < ctuH; r++)
                {
                    if (doneFull[r] == ctuW) continue;
                    allDone = false;
                    if (chunkOk(r, doneTop[r], limTop(pic.analysedCols, r))) return true;
                    /* (a FULL step may become possible through the TOP step of the same sweep: the TOP test above covers that case) */
                    if (chunkOk(r, doneFull[r], limFull(pic.analysedCols, r))) return true;
                }
                return allDone;
            };
            pic.cv.wait(lk, ready);
            if (pic.failed) return X265AMD_EHIP;
            a = pic.analysedCols;
        }
        lap(tWait);
        /* ---- one sweep: every step that is ready, top row first (the stream orders them: FULL of row r - 1, TOP of row r, FULL of row r).  First the edges and the
         * statistics of all of them, one synchronisation, then the decisions on the host, then offsets + borders, a second synchronisation, then the publication:
         * two waits per sweep however many rows are in flight. ---- */
        todoTop.clear(); todoFull.clear();
        bool all = true;
        static const bool dbCopy = getenv("X265AMD_DEBLOCK_UNITS_COPY") && atoi(getenv("X265AMD_DEBLOCK_UNITS_COPY")) != 0;
        for (int r = 0; r < ctuH && rc == X265AMD_OK; r++)
        {
            if (doneFull[r] == ctuW) continue;
            all = false;
            const int y4b = r * 16, y4e = std::min(h4, y4b + 16), y4t = std::min(y4e, y4b + 2);
            {
                const int c0 = doneTop[r], c1 = limTop(a, r);
                if (chunkOk(r, c0, c1))
                {
                    const int x4b = c0 * 16, x4e = std::min(w4, c1 * 16 + 1);       /* + the unit column right of the boundary edge */
                    if (dbl)
                    {
                        /* the edge records of the whole row height (both steps read them) where the kernels read them: mapped memory, no copy */
                        rc = x265amd_deblock_units_rect(&si, &info, pic.units.data(), pic.motion.data(), dbu, y4b, y4e, x4b, x4e);
                        if (rc != X265AMD_OK) break;
                        _mm_sfence();       /* the records went through the write-combining BAR mapping: out of this core's buffers before the launch that reads them */
                        if (dbCopy && hipMemcpy2DAsync((x265amd_deblock_unit*)dDb.p + (size_t)y4b * w4 + x4b, sizeof(x265amd_deblock_unit) * w4, dbu + (size_t)y4b * w4 + x4b, sizeof(x265amd_deblock_unit) * w4,
                                             sizeof(x265amd_deblock_unit) * (size_t)(x4e - x4b), (size_t)(y4e - y4b), hipMemcpyHostToDevice, st) != hipSuccess)
                        { rc = xa_fail(X265AMD_EHIP, "encoder: deblock upload"); break; }
                        rc = x265amd_deblock_rows_cols(st, recY, recU, recV, stride, cstride, W, H, dbCopy ? (const x265amd_deblock_unit*)dDb.p : dbu, 0, 0, 0, 0, 0, 3, y4b, y4t, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    todoTop.push_back(Unit{ r, c0, c1 });
                    doneTop[r] = c1;
                }
            }
            {
                const int c0 = doneFull[r], c1 = limFull(a, r);
                if (chunkOk(r, c0, c1))
                {
                    if (dbl && y4e > y4t)
                    {
                        rc = x265amd_deblock_rows_cols(st, recY, recU, recV, stride, cstride, W, H, dbCopy ? (const x265amd_deblock_unit*)dDb.p : dbu, 0, 0, 0, 0, 0, 3, y4t, y4e, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    if (sao)
                    {
                        /* every workgroup stores all 160 sums and counts of its (CTU, plane): nothing to clear; the host reads them where the kernel leaves them */
                        rc = x265amd_sao_stats_rows_cols(st, recP, srcP, stride, cstride, W, H, cnt, orgs, r, r + 1, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    todoFull.push_back(Unit{ r, c0, c1 });
                    doneFull[r] = c1;
                }
            }
        }
        if (rc != X265AMD_OK) break;
        if (todoTop.empty() && todoFull.empty()) { if (all) break; continue; }
        numUnits += (int)(todoTop.size() + todoFull.size()); numSweeps++;
        if (sao && !todoFull.empty())
        {
            if (streamWaitPolite(st, ev) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: sao statistics"); break; }
            for (const Unit& u : todoFull)
            {
                int32_t flags[2] = { 1, 1 };
                rc = x265amd_sao_rdo_cols(&si, pic.type != TYPE_B ? 1 : 0, 2, 0, 69, pic.units.data(), cnt, orgs, sparams.data(), flags, u.r, u.c0, u.c1,
                                          carry.data() + (size_t)u.r * (X265AMD_CTX_STRIDE + 8));
                if (rc != X265AMD_OK) break;
                const size_t off = (size_t)u.r * ctuW + u.c0, n = (size_t)(u.c1 - u.c0);
                memcpy((x265amd_sao_ctu*)hPar.p + off, sparams.data() + off, sizeof(x265amd_sao_ctu) * n);
                _mm_sfence();               /* as for the deblocking records above */
                if (parCopy && hipMemcpyAsync((x265amd_sao_ctu*)dPar.p + off, (const x265amd_sao_ctu*)hPar.p + off, sizeof(x265amd_sao_ctu) * n, hipMemcpyHostToDevice, st) != hipSuccess)
                { rc = xa_fail(X265AMD_EHIP, "encoder: sao upload"); break; }
            }
            if (rc != X265AMD_OK) break;
        }
        /* final now: the row above a TOP step (its parameters were decided by its own FULL step, in this sweep at the latest), and the last row behind its FULL step --
         * up to eight samples short of the step's right end */
        for (const Unit& u : todoTop)
        {
            if (u.r == 0) continue;
            rc = finishCols(u.r - 1, u.c1 == ctuW ? W : 64 * u.c1 - 8);
            if (rc != X265AMD_OK) break;
        }
        if (rc != X265AMD_OK) break;
        for (const Unit& u : todoFull)
        {
            if (u.r != ctuH - 1) continue;
            rc = finishCols(u.r, u.c1 == ctuW ? W : 64 * u.c1 - 8);
            if (rc != X265AMD_OK) break;
        }
        if (rc != X265AMD_OK) break;
        if (streamWaitPolite(st, ev) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: row filters"); break; }
        for (const Unit& u : todoTop)
        {
            const int newX = u.c1 == ctuW ? W : 64 * u.c1 - 8;
            if (u.r > 0 && newX > pubX[u.r - 1]) { pubX[u.r - 1] = newX; pic.publish(u.r - 1, newX); }
        }
        for (const Unit& u : todoFull)
        {
            const int newX = u.c1 == ctuW ? W : 64 * u.c1 - 8;
            if (u.r == ctuH - 1 && newX > pubX[u.r]) { pubX[u.r] = newX; pic.publish(u.r, newX); }
        }
        lap(tWork);
    }
    return rc;
}

static uint64_t thread_cpu_ns()
{
    struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
int x265amd_encoder::runFrameParallel(const PicP& picp)
{
    Pic& pic = *picp;
    /* whatever happens, the pictures waiting for rows of this one are released */
    struct Release { Pic& pic; int* rc; ~Release() { if (*rc) pic.fail(); } };
    int rc = X265AMD_EHIP;
    Release release{ pic, &rc };
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const std::vector<PicP>* lists = pic.lists;
    FrameCtx fc;
    frameContext(*this, pic, fc);
    const Pic* colPic = fc.colPic;
    const size_t nUnits = (size_t)w4 * h4;
    std::vector<x265amd_mv_unit> noCol;
    if (!colPic) { noCol.resize(nUnits); memset(noCol.data(), 0, sizeof(x265amd_mv_unit) * nUnits); }
    std::vector<uint8_t> refDepth(2 * nUnits, 0);
    std::vector<int8_t> refQp0(2 * (size_t)nctu, 0);
    if (fc.failed) return xa_fail(X265AMD_EHIP, "encoder: weighted reference planes");
    RowGate gate{ this, &pic, {}, &refDepth, nUnits, &fc, &refQp0 };
    if (useDqp && pic.cuQp.empty()) return xa_fail(X265AMD_EINVAL, "encoder: the picture has no CU QPs");
    for (int l = 0; l < 2; l++)
    {
        for (const PicP& q : lists[l]) if (std::find(gate.refs.begin(), gate.refs.end(), q.get()) == gate.refs.end()) gate.refs.push_back(q.get());
        if (!lists[l].empty()) for (int i = 0; i < nctu; i++) refQp0[(size_t)l * nctu + i] = (int8_t)lists[l][0]->sliceQp;
    }
    std::vector<x265amd_cu_stat> stat((size_t)nctu + 1);
    memset(stat.data(), 0, sizeof(x265amd_cu_stat) * stat.size());
    std::vector<int16_t> coeff((size_t)nctu * RD_TILE_ELEMS, 0);
    std::vector<uint8_t> data((size_t)W * H * 3 + (1u << 16));
    std::vector<uint32_t> sizes((size_t)ctuH + 1, 0);
    int nsub = 0;
    const bool sao = p.bEnableSAO != 0;
    std::vector<x265amd_sao_ctu> sparams((size_t)nctu);
    memset(sparams.data(), 0, sizeof(x265amd_sao_ctu) * nctu);
    int32_t saoFlags[2] = { 0, 0 };
    int filterRc = X265AMD_OK;
    /* by columns when there is something to filter and the rows run as a wavefront; the row-by-row form otherwise */
    static const bool colsOff = getenv("X265AMD_FILTER_COLS") && atoi(getenv("X265AMD_FILTER_COLS")) == 0;
    const bool byCols = !colsOff && p.bEnableWavefront && (p.bEnableLoopFilter || p.bEnableSAO) && ctuH > 1 && ctuW > 1;
    std::thread filters([&, byCols] { xa_thread_device(); filterRc = byCols ? filterRowsCols(pic, fc.si, fc.info, sparams, saoFlags) : filterRows(pic, fc.si, fc.info, sparams, saoFlags); if (filterRc) pic.fail();
                                      cpuFilterNs += thread_cpu_ns(); });
    /* the rows' priority among the row tasks of all pictures in flight: the picture's place in coding order -- an I picture some places earlier (X265AMD_I_BOOST): its
     * chain of 8x8 CUs is the longest thing in flight, nothing it needs comes from another picture, and the pictures behind the scene cut wait for it */
    static const uint64_t iBoost = getenv("X265AMD_I_BOOST") ? (uint64_t)atoi(getenv("X265AMD_I_BOOST")) : 0;
    const bool isI = pic.type == TYPE_IDR || pic.type == TYPE_I;
    const uint64_t rowOrder = isI ? (pic.codingOrder + 1 > iBoost ? pic.codingOrder + 1 - iBoost : 1) : pic.codingOrder + 1;
    const XaRowHooks hooks{ &gate, gateRowReady, gateBeforeRow, gateAfterRow, gateCtuWait, gateBeforeCtu, gateAfterCtu, gateRefWait, rowOrder, gateCtuReach };
    int arc = xa_analyse_frame(me, st, &fc.info, &fc.sp, &fc.si, &fc.ap, pic.units.data(), pic.motion.data(), colPic ? colPic->motion.data() : noCol.data(),
                               refDepth.data(), refQp0.data(), fc.planes.data(), (int)(fc.planes.size() / 3), stride, cstride, stat.data(), coeff.data(), nullptr,
                               sao ? nullptr : data.data(), data.size(), sizes.data(), &nsub, &hooks, useDqp ? pic.cuQp.data() : nullptr);
    if (arc != X265AMD_OK) pic.fail();
    filters.join();
    if (arc != X265AMD_OK) return rc = arc;
    if (filterRc != X265AMD_OK) return rc = filterRc;
    if (sao)
    {
        arc = x265amd_encode_slice_data(&fc.si, pic.units.data(), coeff.data(), sparams.data(), saoFlags, data.data(), data.size(), sizes.data(), &nsub);
        if (arc != X265AMD_OK) return rc = arc;
    }
    arc = sliceNal(*this, pic, fc, saoFlags, data, sizes, nsub);
    if (arc) return rc = arc;
    if (!keepSources()) { xa_scratch_free(pic.dSrc); pic.dSrc = nullptr; }
    return rc = 0;
}

extern "C" int x265amd_encoder_encode(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal, const x265amd_picture* picIn, x265amd_picture* picOut)
{
    if (!e) return xa_fail(X265AMD_EINVAL, "encoder_encode: null encoder");
    if (ppNal) *ppNal = nullptr;
    if (piNal) *piNal = 0;
    if (picIn)
    {
        PicP pic(new Pic);
        pic->pool = e->laPool;
        if (e->firstInMs < 0) e->firstInMs = Pic::pubClockMs();
        pic->poc = e->frameCount++;
        const auto tu0 = std::chrono::steady_clock::now();
        int rc = e->uploadPicture(picIn, *pic);
        if (rc) return -1;
        const auto tl0 = std::chrono::steady_clock::now();
        e->uploadMs += std::chrono::duration<double, std::milli>(tl0 - tu0).count();
        if (e->lookahead && (rc = e->lowresInit(*pic)) != X265AMD_OK) return -1;
        e->laInitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count();
        e->input.push_back(pic);
    }
    const bool flushing = picIn == nullptr;
    /* The slice-type decisions.  With pictures coming in, whatever a full queue allows.  When the caller flushes, ONE mini-GOP at a time (below): each decision of
     * the lookahead takes as long as a P picture, and the pictures of the first mini-GOP have no reason to wait for the decisions about the last -- a clip shorter than
     * the lookahead is decided entirely while it is flushed, and its first P picture used to start when the last decision was made (X265AMD_FLUSH_DECIDE_ALL=1: that
     * form).  The decisions themselves do not depend on when they are made. */
    static const bool decideAll = getenv("X265AMD_FLUSH_DECIDE_ALL") && atoi(getenv("X265AMD_FLUSH_DECIDE_ALL")) != 0;
    auto decide = [e, flushing]() -> int {
        const auto tl1 = std::chrono::steady_clock::now();
        int rc = X265AMD_OK;
        if (e->lookahead) rc = e->decideLookahead(flushing, flushing && !decideAll ? 1 : 1 << 30);
        else e->decideMiniGop(flushing);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl1).count();
        e->laDecideMs += ms;
        static const bool timingD = getenv("X265AMD_TIMING") != nullptr;
        if (timingD && ms > 0.5) fprintf(stderr, "x265amd: decision: %.1f ms, %d pictures typed, %d still in the lookahead%s\n", ms, (int)e->ready.size(), (int)e->input.size(), flushing ? " (flushing)" : "");
        return rc;
    };
    /* every typed picture is prepared in coding order here (DPB::prepareEncode is bookkeeping: it does not wait for any picture to be coded); the frame itself is a task */
    auto admit = [e]() -> int {
        while (!e->ready.empty())
        {
            PicP pic = e->ready.front();
            e->ready.pop_front();
            if (e->prepare(pic)) return -1;
            pic->codingOrder = e->codingCount++;
            pic->owned = e->p.shardCount <= 1 || (int)(pic->codingOrder % (uint64_t)e->p.shardCount) == e->p.shardRank;
            e->inflight.push_back(pic);
            { std::lock_guard<std::mutex> lk(e->byCodingMu); e->byCoding[pic->codingOrder] = pic; }      /* stays until the picture has been collected (below): however many pictures are in flight */
        }
        return 0;
    };
    if (decide() != X265AMD_OK || admit()) return -1;
    const bool timing = getenv("X265AMD_TIMING") != nullptr;
    auto start = [e, timing](const PicP& pic) {
        std::shared_future<int> prev = e->lastTask;
        pic->started = true;
        if (timing) fprintf(stderr, "x265amd: poc %d handed to a frame task at %.1f ms (%d running)\n", pic->poc, Pic::pubClockMs(), e->running);
        pic->done = std::async(std::launch::async, [e, pic, prev, timing]() {
            xa_thread_device();
            const auto t0 = std::chrono::steady_clock::now();
            if (!pic->owned)
            {
                /* another object codes this picture: its rows arrive through x265amd_encoder_import_row -- unless nobody will ever read them (a plain B picture is
                 * no reference: the row pump does not send it) */
                if (pic->type == TYPE_B) { if (!e->keepSources()) { xa_scratch_free(pic->dSrc); pic->dSrc = nullptr; } return (int)X265AMD_OK; }
                std::unique_lock<std::mutex> lk(pic->mu);
                static const int importWaitS = getenv("X265AMD_IMPORT_WAIT_S") ? atoi(getenv("X265AMD_IMPORT_WAIT_S")) : 300;       /* (debugging a stalled pump: a short wait shows where it stands) */
                const bool ok = pic->cv.wait_for(lk, std::chrono::seconds(importWaitS), [&] { return pic->importedRows >= e->ctuH || pic->failed.load(); });
                if (!ok || pic->failed.load()) { lk.unlock(); pic->fail(); return xa_fail(X265AMD_EHIP, "encoder: a picture coded elsewhere did not arrive"); }
                lk.unlock();
                /* (weightAnalyse of later pictures reads the chroma planes of its references' SOURCE pictures on every object of the set: sliceWeights) */
                if (!e->keepSources()) { xa_scratch_free(pic->dSrc); pic->dSrc = nullptr; }
                return (int)X265AMD_OK;
            }
            const int rc = e->frameParallel ? e->runFrameParallel(pic) : e->runFrame(pic, prev);
            { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); e->cpuPictureNs += (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
            if (timing)
                fprintf(stderr, "x265amd: poc %d type %d qp %d: %.2f ms\n", pic->poc, pic->type, pic->sliceQp,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            return rc;
        }).share();
        e->lastTask = pic->done;
        e->running++;
    };
    /* Tasks start in coding order while fewer than frameThreads + 1 run.  Coded in parallel, a picture without references does not wait for its turn: nothing it needs
     * comes from another picture, and an I picture takes as long as a dozen of the others -- started when the lookahead hands it over, it is coded beside the pictures
     * in front of it instead of holding up the ones behind it (X265AMD_EARLY_I=0: in turn).  Output stays in coding order. */
    static const bool earlyI = !(getenv("X265AMD_EARLY_I") && atoi(getenv("X265AMD_EARLY_I")) == 0);
    static const int earlyIMax = getenv("X265AMD_EARLY_I_MAX") ? atoi(getenv("X265AMD_EARLY_I_MAX")) : 1 << 20;
    static const bool earlyP = !(getenv("X265AMD_EARLY_P") && atoi(getenv("X265AMD_EARLY_P")) == 0);
    static const bool earlyPAlways = getenv("X265AMD_EARLY_P") && atoi(getenv("X265AMD_EARLY_P")) == 2;
    static const bool earlyBref = getenv("X265AMD_EARLY_BREF") && atoi(getenv("X265AMD_EARLY_BREF")) != 0;      /* a referenced B picture is a link of the same chain */
    static const int earlyPMax = getenv("X265AMD_EARLY_P_MAX") ? atoi(getenv("X265AMD_EARLY_P_MAX")) : 6;
    /* (measured, profiles/r05_early_b_sweep.txt: 2160p clips 15-22 % shorter with 12, 8-bit, Main 10 and --preset slow alike; the 1080p clips unchanged or -- the
     * sixty-frame clip with its two scene cuts -- 15 % longer: there an I picture behind a scene cut shares the device with a dozen pictures more.  So: by size) */
    const int earlyBMax = getenv("X265AMD_EARLY_B_MAX") ? atoi(getenv("X265AMD_EARLY_B_MAX")) : (e->ctuH > 24 ? 12 : 0);
    auto launch = [&]() {
    const bool headLong = !e->inflight.empty() && (e->inflight.front()->type == TYPE_IDR || e->inflight.front()->type == TYPE_I) && e->inflight.front()->started &&
                          e->inflight.front()->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready;
    for (auto& q : e->inflight)
    {
        if (q->started) continue;
        /* the first picture in coding order always runs: it is the one collected next, whatever started ahead of its turn */
        if (e->running <= e->frameThreads || q == e->inflight.front()) { start(q); if (e->frameThreads <= 1) q->done.wait(); continue; }
        if (!(e->frameParallel && earlyI)) break;
        if (q->type == TYPE_IDR || q->type == TYPE_I)
        {
            /* (X265AMD_EARLY_I_MAX: no more of them at once than this -- an experiment of round 5's end: while P pictures started ahead of their turn all the time, a limit
             * of one helped a long 2160p clip; with the P pictures held to their turn outside an I picture's time it does not, and there is none.  profiles/r05_sched_sweep.txt) */
            int runningI = 0;
            for (auto& o : e->inflight)
                if (o->started && (o->type == TYPE_IDR || o->type == TYPE_I) && o->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready) runningI++;
            if (runningI < earlyIMax) start(q);
        }
        /* The P pictures are the chain every other picture hangs on (each follows its reference by a few CTU rows, the B pictures between two of them follow both): a P
         * picture held back until the B pictures in front of it have been collected starts with nothing to trail and takes its full latency, so it starts when the
         * lookahead hands it over, too (every picture it references is in front of it in coding order and therefore started; X265AMD_EARLY_P=0: in turn). */
        /* (round 5's end: like the B pictures below, only while a running I picture holds the head of the coding order -- X265AMD_EARLY_P=2: always, as round 4 had it.
         * With P pictures ahead of their turn ALL the time a long clip stood at 52 frames/s where it reaches 100 without: the pictures far ahead held the places
         * and the queues that the pictures collected next were waiting for; profiles/r05_sched_sweep.txt) */
        else if (earlyP && (headLong || earlyPAlways) && (q->type == TYPE_P || (earlyBref && q->type == TYPE_BREF)) && e->running <= e->frameThreads + earlyPMax) start(q);
        /* The B pictures, too (round 5), while an I picture that still runs holds the head of the coding order: `running` counts every picture that trails it and is not
         * collected yet (collection is in coding order), and the B pictures of the mini-GOPs whose P pictures ran already waited for the I picture's END although their
         * references were rows ahead of them -- at 2160p a third of a twenty-frame clip, at --preset slow more.  How many pictures run side by side changes nothing in
         * what they code (the vertical reach of the vectors follows from the parameter frameNumThreads, not from this count): up to X265AMD_EARLY_B_MAX more than the
         * parameter (0: in turn).  Only then, and only for large pictures (see earlyBMax above). */
        else if (earlyBMax > 0 && headLong && e->running <= e->frameThreads + earlyBMax) start(q);
    }
    };
    static const bool holdUntilFlush = getenv("X265AMD_HOLD_UNTIL_FLUSH") != nullptr;      /* an experiment: no picture starts before the caller flushes (what the clip costs when every decision is made beforehand) */
    if (holdUntilFlush)
    {
        if (!flushing) return 0;
        while (!e->input.empty()) { const size_t before = e->input.size(); if (decide() != X265AMD_OK || admit()) return -1; if (e->input.size() >= before) break; }
        static bool said = false;
        if (!said) { said = true; fprintf(stderr, "x265amd: every decision made %.1f ms after the encoder's first picture came in; the pictures start now\n", Pic::pubClockMs() - e->firstInMs); }
    }
    launch();
    /* flushing: the next mini-GOP is decided while the picture the caller will get next is still being coded */
    while (flushing && !e->input.empty())
    {
        if (!e->inflight.empty() && e->inflight.front()->started && e->inflight.front()->done.wait_for(std::chrono::seconds(0)) == std::future_status::ready) break;
        const size_t before = e->input.size();
        if (decide() != X265AMD_OK || admit()) return -1;
        launch();
        if (e->input.size() >= before) break;
    }
    if (e->inflight.empty()) return 0;
    PicP front = e->inflight.front();
    if (!front->started) { xa_fail(X265AMD_EINVAL, "encoder_encode: the first picture in coding order has no task"); return -1; }
    int waiting = 0;
    for (auto& q : e->inflight) waiting += !q->started;
    /* the caller is held when enough pictures run and enough wait behind them (the lookahead may run ahead of the frame tasks by a window of its own) */
    const bool mustWait = !picIn || (e->running > e->frameThreads && waiting > 2 * e->p.lookaheadDepth + 8);
    if (!mustWait && front->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return 0;
    const int rc = front->done.get();
    e->inflight.pop_front();
    e->running--;
    {
        /* a collected picture is complete -- every row exported or imported -- so the row pump has no more business with it; a few stay for a pump that asks late */
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        e->collectedCoding = front->codingOrder + 1;
        while (!e->byCoding.empty() && e->byCoding.begin()->first + 8 < e->collectedCoding) e->byCoding.erase(e->byCoding.begin());
    }
    if (rc) { xa_fail(rc, "encoder_encode: a frame task failed"); return -1; }
    if (const char* dumpPath = getenv("X265AMD_RC_DUMP"))
    {
        /* debugging aid: the picture's record in the layout of oracle/ref_rc_dump.cpp (the reference's decisions for the same picture), appended to the file named */
        if (FILE* f = fopen(dumpPath, "ab"))
        {
            const Pic& q = *front;
            const int blocks16 = ((e->W + 15) / 16) * ((e->H + 15) / 16), lowresBlocks = e->lowCuW * e->lowCuH;
            int32_t hdr[44];
            memset(hdr, 0, sizeof(hdr));
            hdr[0] = 0x52434450; hdr[1] = q.poc; hdr[2] = q.type; hdr[3] = q.type != TYPE_B; hdr[4] = q.sliceQp; hdr[5] = q.bScenecut;
            for (int l = 0; l < 2; l++) { hdr[6 + l] = (int32_t)q.lists[l].size(); for (size_t r = 0; r < q.lists[l].size() && r < 16; r++) hdr[8 + 16 * l + r] = q.lists[l][r]->poc; }
            hdr[40] = e->w4; hdr[41] = e->h4; hdr[42] = blocks16; hdr[43] = lowresBlocks;
            fwrite(hdr, sizeof(hdr), 1, f);
            const int64_t satd = 0; fwrite(&satd, 8, 1, f);
            const double qq[2] = { q.avgQpRc, 0 }; fwrite(qq, 8, 2, f);
            std::vector<double> zd((size_t)blocks16, 0.0); std::vector<int32_t> zi((size_t)std::max(blocks16, lowresBlocks), 0); std::vector<uint16_t> zs((size_t)lowresBlocks, 0);
            fwrite(q.qpAqOffset.size() == (size_t)blocks16 ? q.qpAqOffset.data() : zd.data(), 8, blocks16, f);
            fwrite(q.qpCuTreeOffset.size() == (size_t)blocks16 ? q.qpCuTreeOffset.data() : zd.data(), 8, blocks16, f);
            fwrite(q.invQscale.size() == (size_t)blocks16 ? q.invQscale.data() : zi.data(), 4, blocks16, f);
            fwrite(q.intraCostHost.size() == (size_t)lowresBlocks ? q.intraCostHost.data() : zi.data(), 4, lowresBlocks, f);
            fwrite(q.propagateCost.size() == (size_t)lowresBlocks ? q.propagateCost.data() : zs.data(), 2, lowresBlocks, f);
            const size_t n = (size_t)e->w4 * e->h4;
            std::vector<uint8_t> b(n);
            for (size_t i = 0; i < n; i++) b[i] = (uint8_t)q.units[i].qp; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].depth; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].pred_mode; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].cbf[0]; fwrite(b.data(), 1, n, f);
            fclose(f);
        }
    }
    e->outBytes.swap(front->nalBytes);
    splitNals(e->outBytes, e->nals);
    if (picOut)
    {
        e->staging.resize(e->picElems);
        if (hipMemcpy(e->staging.data(), front->finalPlanes(), e->picElems * sizeof(pixel), hipMemcpyDeviceToHost) != hipSuccess) { xa_fail(X265AMD_EHIP, "encoder: recon download"); return -1; }
        for (int k = 0; k < 3; k++)
        {
            if (!picOut->planes[k]) continue;
            const int w = k ? e->W / 2 : e->W, hh = k ? e->H / 2 : e->H;
            const intptr_t st = k ? e->cstride : e->stride;
            for (int y = 0; y < hh; y++)
                memcpy((uint8_t*)picOut->planes[k] + (size_t)y * picOut->stride[k], e->staging.data() + e->org[k] + (intptr_t)y * st, sizeof(pixel) * w);
        }
        picOut->poc = front->poc; picOut->sliceType = front->type; picOut->qp = front->sliceQp;
    }
    /* a finished picture that nobody references any more releases its lists (and with them the pictures only it kept alive) */
    front->lists[0].clear(); front->lists[1].clear(); front->neg.clear(); front->pos.clear();
    if (ppNal) *ppNal = e->nals.data();
    if (piNal) *piNal = (uint32_t)e->nals.size();
    return 1;
}
