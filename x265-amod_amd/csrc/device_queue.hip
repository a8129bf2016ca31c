/* Device job queues (x265amd_host.h: xa_q_*, xa_stream_*): the resident server kernel and its host side.  See xa_queue.h for the why and the
 * protocol.  The host loop being replaced is the reference's worker thread calling primitives one block at a time
 * (reference: source/encoder/analysis.cpp:1146-1848 -> source/common/primitives.h:239-433); the queue keeps that call order per CTU row and removes
 * the launch + synchronise pair from every call.
 */
#define XA_SERVER_BYTES
#include "x265amd_dev.h"
#include "x265amd_host.h"
#include "xa_queue.h"
#include "xa_fiber.h"
#ifndef XA_PREFETCH_SLOT
#define XA_PREFETCH_SLOT 0          /* fetching the next slot while a command runs: measured, no gain (the load competes with the command's own first loads) */
#endif
/* stage stamps (X265AMD_QUEUE_PROF): thread 0 of the workgroup adds the time since its previous stamp to the stage's total; [15] is "outside".  A stamp is a clock read
 * and an LDS update by the wavefront everybody else waits for at the next barrier -- forty of them in an 8x8 CU of an I picture -- so they are compiled in only when the
 * library is built for that report (X265AMD_EXTRA_FLAGS=-DXA_PROFILE_STAGES bash build.sh); otherwise the report's stage lines read zero. */
__shared__ unsigned long long xa_stage_acc[22];      /* [0..15] transform chains and the prediction-unit step, [16..21] the NxN step */
__shared__ long long xa_stage_prev;
/* the fused intra command's stages on a clock of its own (the chains' stamps above do not disturb it): [kind][stage], thread 0 */
__shared__ unsigned long long xa_nxn_acc[4][10];
__shared__ unsigned long long xa_chain_acc[8];
__shared__ long long xa_chain_prev;
__shared__ long long xa_nxn_prev;
__shared__ int xa_nxn_kind;
#ifdef XA_PROFILE_STAGES
#define XA_STAGE(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); xa_stage_acc[k] += (unsigned long long)(t_ - xa_stage_prev); xa_stage_prev = t_; } } while (0)
#define XA_CHAIN_START() do { if (threadIdx.x == 0) xa_chain_prev = wall_clock64(); } while (0)
#define XA_CHAIN(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); xa_chain_acc[k] += (unsigned long long)(t_ - xa_chain_prev); xa_chain_prev = t_; } } while (0)
#define XA_NXN_START(kind) do { if (threadIdx.x == 0) { xa_nxn_kind = (kind); xa_nxn_prev = wall_clock64(); } } while (0)
#define XA_NXN(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); xa_nxn_acc[xa_nxn_kind][k] += (unsigned long long)(t_ - xa_nxn_prev); xa_nxn_prev = t_; } } while (0)
#define XA_LINK_START()
#define XA_LINK(k)
#define XA_LINK_T(k)
#else
#define XA_STAGE(k)
#define XA_CHAIN_START()
#define XA_CHAIN(k)
#define XA_NXN_START(kind)
#define XA_NXN(k)
/* what stays in every build: the two waits of a chained 8x8 CU that say which of its two evaluations the chain waits for ([6] the deciding command for the other
 * evaluation, [7] the other command for the chain): two clock reads per CU by a lane that is waiting anyway */
#define XA_LINK_START() do { if (threadIdx.x == 0) xa_chain_prev = wall_clock64(); } while (0)
#define XA_LINK(k) do { if (threadIdx.x == 0) xa_chain_acc[k] += (unsigned long long)(wall_clock64() - xa_chain_prev); } while (0)
#define XA_LINK_T(k) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); xa_chain_acc[k] += (unsigned long long)(t_ - xa_chain_prev); xa_chain_prev = t_; } } while (0)
#endif
#include "tu_dev.h"
#include "intra_dev.h"
#include "mc_dev.h"
#include "me_dev.h"
#include "measure_dev.h"
#include "entropy_dev.h"
#include "intra_pu_dev.h"
#define XA_CHAIN_DEVICE
#include "inter_chain_dev.h"
#define XA_SEARCH_DEVICE
#include "inter_search_dev.h"
#include <immintrin.h>
#include <signal.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string.h>
#include <thread>
#include <vector>

/* =========================================================================================================
 * device side
 * ======================================================================================================= */
extern __shared__ __attribute__((aligned(16))) char xa_smem[];
__shared__ int xa_q_index;                                            /* the queue this workgroup serves (k_job_server: base + blockIdx.x) */
__device__ uint64_t* xa_dbg_area[256];                                /* per workgroup: XaRingHost::dbg (debugging aid) */
#define XA_DBG(c, slot, v) do { if (((c).reserved & 2) && (threadIdx.x & 63) == 0) __hip_atomic_store(&xa_dbg_area[xa_q_index][((threadIdx.x >> 6) * 8 + (slot)) & 63], (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)      /* aliases tu_smem / me_smem: one dynamic LDS block, laid out per command */

XA_DEV uint64_t xa_sys_load(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
XA_DEV void xa_sys_store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

/* byte copies by the whole workgroup: 16 bytes per lane where both sides allow it.  HOSTDST: the destination is pinned host memory (results the host
 * reads as soon as the command has signalled): system-scope stores, 8 bytes per lane */
template<bool HOSTDST> XA_DEV void block_copy(char* dst, const char* src, size_t bytes, int tid, int nthr)
{
    if (HOSTDST)
    {
        if ((((uintptr_t)dst | (uintptr_t)src) & 7) == 0)
        {
            const size_t n8 = bytes >> 3;
            for (size_t i = tid; i < n8; i += nthr) xa_sys_store(reinterpret_cast<uint64_t*>(dst) + i, reinterpret_cast<const uint64_t*>(src)[i]);
            for (size_t i = (n8 << 3) + tid; i < bytes; i += nthr) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        else
            for (size_t i = tid; i < bytes; i += nthr) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0)
    {
        const size_t n16 = bytes >> 4;
        for (size_t i = tid; i < n16; i += nthr) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (size_t i = (n16 << 4) + tid; i < bytes; i += nthr) dst[i] = src[i];
    }
    else
        for (size_t i = tid; i < bytes; i += nthr) dst[i] = src[i];
}

XA_DEV void block_copy_rects(const XaArgsRects& r, int tid, int nthr)
{
    for (int k = 0; k < r.n; k++)
    {
        const pixel* src = reinterpret_cast<const pixel*>(r.src[k]);
        pixel* dst = reinterpret_cast<pixel*>(r.dst[k]);
        const int w = r.w[k], total = w * r.h[k];
        for (int i = tid; i < total; i += nthr)
        {
            const int y = i / w, x = i - y * w;
            dst[(size_t)y * r.dst_stride[k] + x] = src[(size_t)y * r.src_stride[k] + x];
        }
    }
}

/* every command body is a function of its own: the register allocation of one does not weigh on the others */
XA_DEV void xa_op_copy(const XaCmd& c, int tid)        /* small: inlined into the server loop (no call frame) */
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsCopy& a = *reinterpret_cast<const XaArgsCopy*>(c.args);
        block_copy<false>(reinterpret_cast<char*>(a.dst), reinterpret_cast<const char*>(a.src), a.bytes, tid, NT);
        if (tid == 0) XA_BYTES(2 * a.bytes);
    }
}

XA_DEV void xa_op_copy2d(const XaCmd& c, int tid)        /* small: inlined into the server loop (no call frame) */
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsCopy2D& a = *reinterpret_cast<const XaArgsCopy2D*>(c.args);
        for (uint64_t y = wv; y < a.height; y += XA_SERVER_WAVES)
            block_copy<false>(reinterpret_cast<char*>(a.dst + y * a.dpitch), reinterpret_cast<const char*>(a.src + y * a.spitch), a.width, lane, 64);
        if (tid == 0) XA_BYTES(2 * a.width * a.height);
    }
}

XA_DEV void xa_op_fill(const XaCmd& c, int tid)        /* small: inlined into the server loop (no call frame) */
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsFill& a = *reinterpret_cast<const XaArgsFill*>(c.args);
        char* d = reinterpret_cast<char*>(a.dst);
        for (size_t i = tid; i < a.bytes; i += NT) d[i] = (char)a.value;
        if (tid == 0) XA_BYTES(a.bytes);
    }
}

XA_DEV void xa_op_copy_rects(const XaCmd& c, int tid)        /* small: inlined into the server loop (no call frame) */
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
        block_copy_rects(*reinterpret_cast<const XaArgsRects*>(c.args), tid, NT);
        if (tid == 0)
        {
            const XaArgsRects& r = *reinterpret_cast<const XaArgsRects*>(c.args);
            unsigned long long b = 0;
            for (int k = 0; k < r.n; k++) b += 2ull * (unsigned)r.w[k] * (unsigned)r.h[k] * sizeof(pixel);
            XA_BYTES(b);
        }
}

__device__ __noinline__ void xa_op_mc(const XaCmd& c, int tid)
{
    const int lane = tid & 63, wv = tid >> 6;
    const XaArgsMc a = *reinterpret_cast<const XaArgsMc*>(c.args);
    if (c.op == XA_OP_MC_COST)
    {
        for (int ji = wv; ji < a.n; ji += XA_SERVER_WAVES) wave_mc_job<true>(a, ji, lane);
        return;
    }
    /* prediction only: every sample stands alone, so the wavefronts are dealt out over the jobs (one job: all eight on it) */
    const int wpj = a.n >= XA_SERVER_WAVES ? 1 : XA_SERVER_WAVES / a.n, perRound = XA_SERVER_WAVES / wpj;
    for (int base = 0; base < a.n; base += perRound)
    {
        const int ji = base + wv / wpj, sub = wv % wpj;
        if (wv < perRound * wpj && ji < a.n) wave_mc_job<false>(a, ji, sub * 64 + lane, wpj * 64);
    }
}

__device__ __noinline__ void xa_op_cu_measure(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    const CuMeasureJob* jobs = reinterpret_cast<const CuMeasureJob*>(a.a);
    CuMeasure* out = reinterpret_cast<CuMeasure*>(a.b);
    static_assert(sizeof(CuMeasureLds) <= XA_SERVER_LDS, "LDS budget");
    if (a.n <= 2 * XA_SERVER_WAVES)
    {
        /* a few candidates of one CU: the whole workgroup on each in turn */
        CuMeasureLds& s = *reinterpret_cast<CuMeasureLds*>(xa_smem);
        for (int ji = 0; ji < a.n; ji++)
        {
            const CuMeasureJob j = xa_ld_record(jobs + ji);
            block_cu_measure_job(j, out + ji, s, tid, NT);
        }
        return;
    }
    constexpr int W = (XA_SERVER_LDS / (64 * 64 * (int)sizeof(pixel))) < XA_SERVER_WAVES ? (XA_SERVER_LDS / (64 * 64 * (int)sizeof(pixel))) : XA_SERVER_WAVES;
    if (wv < W)
        for (int ji = wv; ji < a.n; ji += W) wave_cu_measure_job(jobs, ji, out, reinterpret_cast<pixel*>(xa_smem) + wv * 64 * 64, lane);
}

__device__ __noinline__ void xa_op_tu_chain(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    static_assert((XA_SERVER_WAVES + 1) * sizeof(TuLds) + 64 <= XA_SERVER_LDS, "LDS budget");
    const x265amd_tu_job* jobs = reinterpret_cast<const x265amd_tu_job*>(a.a);
    x265amd_tu_result* out = reinterpret_cast<x265amd_tu_result*>(a.c);
    /* The units of an inter CU: a few large ones (16x16, 32x32) and some small ones.  A large unit on one wavefront is tens of microseconds the row
     * waits for; the whole workgroup takes those one after the other, then the small ones go one per wavefront. */
    TuLds& big = reinterpret_cast<TuLds*>(xa_smem)[XA_SERVER_WAVES];
    unsigned long long* red = reinterpret_cast<unsigned long long*>(xa_smem + (XA_SERVER_WAVES + 1) * sizeof(TuLds));
    const bool few = a.n <= 4 * XA_SERVER_WAVES;
    if (few)
        for (int ji = 0; ji < a.n; ji++)
            if (__hip_atomic_load(&jobs[ji].log2_tr_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= 4) block_tu_chain_job(jobs, ji, out, big, red, tid, NT);
    TuLds& s = reinterpret_cast<TuLds*>(xa_smem)[wv];
    int k = 0;
    for (int ji = 0; ji < a.n; ji++)
    {
        if (few && __hip_atomic_load(&jobs[ji].log2_tr_size, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= 4) continue;
        if ((k++ % XA_SERVER_WAVES) == wv) wave_tu_chain_job<false>(jobs, nullptr, ji, out, s, nullptr, lane);
    }
}

__device__ __noinline__ void xa_op_tu_chain_rdoq(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsJobs4& a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
        constexpr int W = XA_SERVER_LDS / (int)(sizeof(TuLds) + sizeof(RdoqLds));
        static_assert(W >= 1, "LDS budget");
        if (wv < W)
        {
            TuLds& s = reinterpret_cast<TuLds*>(xa_smem)[wv];
            char* rdoqLds = xa_smem + W * sizeof(TuLds) + wv * sizeof(RdoqLds);
            for (int ji = serial ? (wv ? a.n : 0) : wv; ji < a.n; ji += serial ? 1 : W)
                wave_tu_chain_job<true>(reinterpret_cast<const x265amd_tu_job*>(a.a), reinterpret_cast<const x265amd_tu_rdoq*>(a.b), ji, reinterpret_cast<x265amd_tu_result*>(a.c), s, rdoqLds, lane);
        }
    }
}

__device__ __noinline__ void xa_op_intra_tu_chain(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsJobs4& a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
        static_assert(XA_SERVER_WAVES * (sizeof(TuLds) + sizeof(IntraTuLds)) <= XA_SERVER_LDS, "LDS budget");
        TuLds& s = reinterpret_cast<TuLds*>(xa_smem)[wv];
        IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(xa_smem + XA_SERVER_WAVES * sizeof(TuLds))[wv];
        for (int ji = serial ? (wv ? a.n : 0) : wv; ji < a.n; ji += serial ? 1 : XA_SERVER_WAVES)
            wave_intra_tu_chain_job<false>(reinterpret_cast<const x265amd_intra_tu_job*>(a.a), nullptr, ji, reinterpret_cast<x265amd_tu_result*>(a.c), s, ip, nullptr, lane);
    }
}

__device__ __noinline__ void xa_op_intra_tu_chain_rdoq(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsJobs4& a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
        constexpr int W = XA_SERVER_LDS / (int)(sizeof(TuLds) + sizeof(IntraTuLds) + sizeof(RdoqLds));
        static_assert(W >= 1, "LDS budget");
        if (wv < W)
        {
            TuLds& s = reinterpret_cast<TuLds*>(xa_smem)[wv];
            IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(xa_smem + W * sizeof(TuLds))[wv];
            char* rdoqLds = xa_smem + W * (sizeof(TuLds) + sizeof(IntraTuLds)) + wv * sizeof(RdoqLds);
            for (int ji = serial ? (wv ? a.n : 0) : wv; ji < a.n; ji += serial ? 1 : W)
                wave_intra_tu_chain_job<true>(reinterpret_cast<const x265amd_intra_tu_job*>(a.a), reinterpret_cast<const x265amd_tu_rdoq*>(a.b), ji,
                                              reinterpret_cast<x265amd_tu_result*>(a.c), s, ip, rdoqLds, lane);
        }
    }
}

__device__ __noinline__ void xa_op_intra_scan(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    static_assert(XA_SERVER_WAVES * sizeof(IntraLds) <= XA_SERVER_LDS && sizeof(IntraScanLds) <= XA_SERVER_LDS, "LDS budget");
    const x265amd_intra_job* jobs = reinterpret_cast<const x265amd_intra_job*>(a.a);
    int32_t* out = reinterpret_cast<int32_t*>(a.b);
    pixel* nbOut = reinterpret_cast<pixel*>(a.c);
    if (a.n <= 2 * XA_SERVER_WAVES)
    {
        /* the usual case is ONE block (the next CU of the row): the whole workgroup on it */
        IntraScanLds& s = *reinterpret_cast<IntraScanLds*>(xa_smem);
        for (int ji = 0; ji < a.n; ji++)
        {
            const x265amd_intra_job j = xa_ld_record(jobs + ji);
            block_intra_scan_job(j, out + (size_t)ji * 35, nbOut ? nbOut + (size_t)ji * 2 * 129 : nullptr, s, tid, NT);
        }
        return;
    }
    IntraLds& s = reinterpret_cast<IntraLds*>(xa_smem)[wv];
    for (int ji = wv; ji < a.n; ji += XA_SERVER_WAVES) wave_intra_scan_job(jobs, ji, out, nbOut, s, lane);
}

__device__ __noinline__ void xa_op_est_bit(const XaCmd& c, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    (void)NT; (void)lane; (void)wv;
    const bool serial = (c.reserved & 1) != 0;      /* debugging: one wavefront runs every job */
    (void)serial;
    {
        const XaArgsJobs4& a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
        for (int ji = serial ? (wv ? a.n : 0) : wv; ji < a.n; ji += serial ? 1 : XA_SERVER_WAVES) wave_est_bit_job(reinterpret_cast<const x265amd_est_job*>(a.a), ji, lane);
    }
}

__device__ __noinline__ void xa_op_intra_pu(const XaCmd& c, int tid)
{
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    static_assert(sizeof(IntraScanLds) <= XA_SERVER_LDS && XA_SERVER_WAVES * (sizeof(TuLds) + sizeof(IntraTuLds)) <= XA_SERVER_LDS, "LDS budget");
    block_intra_pu(reinterpret_cast<const x265amd_intra_pu_job*>(a.a), reinterpret_cast<x265amd_intra_pu_out*>(a.b), reinterpret_cast<x265amd_tu_result*>(a.c), xa_smem, tid,
                   64 * XA_SERVER_WAVES);
}

__device__ __noinline__ void xa_op_intra_nxn(const XaCmd& c, int tid)
{
    static_assert(XA_SERVER_WAVES * (sizeof(TuLds) + sizeof(IntraTuLds)) + sizeof(Nxn4Lds) <= XA_SERVER_LDS, "LDS budget (block_intra_nxn: chromaAhead)");
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    /* a.n records a.c bytes apart, one after the other (the chained CUs of a block: include/x265amd.h, x265amd_intra_nxn_list) */
    for (int i = 0; i < (a.n > 0 ? a.n : 1); i++)
        block_intra_nxn(reinterpret_cast<const x265amd_intra_nxn_job*>(a.a + (uint64_t)i * a.c), reinterpret_cast<x265amd_intra_nxn_out*>(a.b), xa_smem, tid, 64 * XA_SERVER_WAVES, XA_SERVER_LDS);
}

__device__ __noinline__ void xa_op_inter_chain(const XaCmd& c, int tid)
{
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    block_inter_chain(reinterpret_cast<const XaChainJob*>(a.a), xa_smem, tid);
}

/* the groups of a launch one after the other: each stages its window, its jobs go to the wavefronts */
/* ldsBytes: what of the workgroup's LDS the search may use (the fused search command keeps its own record at the end: inter_search_dev.h) */
template<int WHICH> __device__ __noinline__ void xa_op_me(const XaCmd& c, int tid, size_t ldsBytes = XA_SERVER_LDS)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const MeParams p = *reinterpret_cast<const MeParams*>(c.args);
    const int groups = (int)c.count;
    if (WHICH != 2 && me_multi_fits(p, groups, XA_SERVER_WAVES, ldsBytes))
    {
        /* one job per group (the searches of one prediction unit, a reference picture each): side by side, a wavefront each */
        bool single = true;
        for (int vb = 0; vb < groups; vb++) single &= p.groups[vb].num_jobs == 1;
        if (single)
        {
            if (WHICH == 0) block_me_search_multi<false>(p, groups, tid, NT); else block_me_search_multi<true>(p, groups, tid, NT);
            __syncthreads();
            return;
        }
    }
    for (int vb = 0; vb < groups; vb++)
    {
        if (WHICH == 0) block_me_search<false>(p, vb, tid, NT);
        else if (WHICH == 1) block_me_search<true>(p, vb, tid, NT);
        else block_me_deferred(p, vb, tid, NT);
        __syncthreads();
    }
}

/* XA_OP_WAIT: this queue goes on when another queue has finished its command number `target` (a release: xa_queue_follow), and looks at memory afresh */
__shared__ int xa_wait_failed;          /* set by xa_op_wait: the leader stood; the server loop poisons the queue (XaRingHost::fault) */
__device__ __noinline__ void xa_op_wait(const XaCmd& c, int tid)
{
    const XaArgsWait a = *reinterpret_cast<const XaArgsWait*>(c.args);
    if (tid == 0)
    {
        const long long t0 = wall_clock64();
        while (xa_sys_load(reinterpret_cast<const uint64_t*>(a.word)) < a.target)
        {
            __builtin_amdgcn_s_sleep(1);
            /* two seconds: the other queue stands.  Going on would run this queue's commands against whatever the leader has written so far, so the queue
             * is poisoned instead: nothing behind this command runs, and the host's waits on it fail (q_wait looks at XaRingHost::fault) */
            if (wall_clock64() - t0 > 200000000ll) { xa_wait_failed = 1; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

/* the search bodies for the fused search command (inter_search_dev.h) */
__device__ __noinline__ void xa_op_me_call(int which, const XaCmd& c, int tid)
{
    /* the records the caller has just written are read with scalar loads too: that cache is coherent with nothing */
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (which == 0) xa_op_me<0>(c, tid, XA_SEARCH_LDS_BELOW);
    else if (which == 1) xa_op_me<1>(c, tid, XA_SEARCH_LDS_BELOW);
    else xa_op_me<2>(c, tid, XA_SEARCH_LDS_BELOW);
}
__device__ __noinline__ void xa_op_inter_search(const XaCmd& c, int tid)
{
    const XaArgsJobs4 a = *reinterpret_cast<const XaArgsJobs4*>(c.args);
    block_inter_search(reinterpret_cast<const XaSearchJob*>(a.a), xa_smem, tid);
}

XA_DEV void xa_dispatch(const XaCmd& c, int tid)
{
    switch (c.op)
    {
    case XA_OP_COPY:
        xa_op_copy(c, tid);
        break;
    case XA_OP_COPY2D:
        xa_op_copy2d(c, tid);
        break;
    case XA_OP_FILL:
        xa_op_fill(c, tid);
        break;
    case XA_OP_COPY_RECTS:
        xa_op_copy_rects(c, tid);
        break;
    case XA_OP_MC:
    case XA_OP_MC_COST:
        xa_op_mc(c, tid);
        break;
    case XA_OP_CU_MEASURE:
        xa_op_cu_measure(c, tid);
        break;
    case XA_OP_TU_CHAIN:
        xa_op_tu_chain(c, tid);
        break;
    case XA_OP_TU_CHAIN_RDOQ:
        xa_op_tu_chain_rdoq(c, tid);
        break;
    case XA_OP_INTRA_TU_CHAIN:
        xa_op_intra_tu_chain(c, tid);
        break;
    case XA_OP_INTRA_TU_CHAIN_RDOQ:
        xa_op_intra_tu_chain_rdoq(c, tid);
        break;
    case XA_OP_INTRA_SCAN:
        xa_op_intra_scan(c, tid);
        break;
    case XA_OP_EST_BIT:
        xa_op_est_bit(c, tid);
        break;
    case XA_OP_INTRA_PU:
        xa_op_intra_pu(c, tid);
        break;
    case XA_OP_INTRA_NXN:
        xa_op_intra_nxn(c, tid);
        break;
    case XA_OP_INTER_CHAIN:
        xa_op_inter_chain(c, tid);
        break;
    case XA_OP_INTER_SEARCH:
        xa_op_inter_search(c, tid);
        break;
    case XA_OP_WAIT:
        xa_op_wait(c, tid);
        break;
    case XA_OP_ME_SEARCH: xa_op_me<0>(c, tid); break;
    case XA_OP_ME_SEARCH_STAR: xa_op_me<1>(c, tid); break;
    case XA_OP_ME_DEFERRED: xa_op_me<2>(c, tid); break;
    default:
        break;
    }
}

/* One workgroup per queue.  Wavefront 0 polls the slot of the next command (relaxed system-scope loads of its own device memory), copies it into LDS
 * and the workgroup runs it.  A workgroup leaves when its queue's quit word is set, or when it has seen no command for `idleTicks` of the 100 MHz
 * wall clock AND the host's heartbeat (hosts[0].alive: bumped with every queue taken and every 256th command of any queue) has stood still for that
 * long -- nothing resident outlives a host that went away, and no queue of a server in use loses its workgroup because its own row was idle (a queue
 * handed out later would have been written to with nobody reading). */
__global__ __launch_bounds__(64 * XA_SERVER_WAVES) void k_job_server(XaRingDev* rings, XaRingHost* hosts, long long idleTicks, uint64_t generation, int base)
{
    __shared__ XaCmd s_cmd;
    __shared__ int s_go;
    __shared__ unsigned long long s_prof[64];
    __shared__ unsigned long long s_sized[24];
    /* base: the first queue of this launch (the server's queues are served by two launches: the second starts when the first's queues are all taken -- Server::startSecond) */
    const int qIdx = base + (int)blockIdx.x;
    XaRingDev* rd = rings + qIdx;
    XaRingHost* rh = hosts + qIdx;
    const int tid = threadIdx.x;
    uint64_t seen = 0, signalled = 0, pre = 0, lastAlive = 0;
    bool havePre = false;
    __shared__ unsigned long long s_bytes[32];
    const long long tResident0 = wall_clock64();
    const long long cResident0 = clock64();
    if (tid < 64) s_prof[tid] = 0;
    if (tid < 24) s_sized[tid] = 0;
    if (tid < 32) s_bytes[tid] = 0;
    if (tid == 0) xa_bytes_acc = 0;
    if (tid < 22) xa_stage_acc[tid] = 0;
    if (tid < 40) (&xa_nxn_acc[0][0])[tid] = 0;
    if (tid < 8) xa_chain_acc[tid] = 0;
    if (tid == 0) xa_stage_prev = wall_clock64();
    if (tid == 0) { xa_sys_store(&rh->state, 1); xa_dbg_area[qIdx] = rh->dbg; xa_wait_failed = 0; xa_q_index = qIdx; }
    __syncthreads();
    for (;;)
    {
        if (tid < 64)
        {
            /* Wavefront 0 polls the next command's SLOT (relaxed system-scope loads of the workgroup's own uncached memory): the slot's check word closes
             * over its other fifteen words, this command's number and the server generation, so a slot that passes is the command, complete -- no
             * separate head word to read first (one memory latency less per command), and a slot caught half written just fails and is read again. */
            int go = 0;
            const uint64_t* slot = reinterpret_cast<const uint64_t*>(&rd->cmd[seen % XA_RING]);
            const long long t0 = wall_clock64();
            long long tIdle = t0;
            for (unsigned spins = 1;; spins++)
            {
                /* lanes 0..15: the slot; lane 16: the doorbell (the command number the host is waiting for).  The first look uses what was fetched while the
                 * previous command ran: a command queued behind another costs no memory latency of its own. */
                uint64_t w;
                if (spins == 1 && havePre) w = pre;
                else w = tid < 16 ? xa_sys_load(slot + tid) : (tid == 16 ? xa_sys_load(&rd->head) : 0);
                havePre = false;
                /* position-weighted sum of the fifteen words (each times its own odd constant): a slot caught between two commands -- some words of the
                 * old one, some of the new -- does not pass.  A plain xor did: tile-to-tile copies change `dst` and `src` by the same bits, the two
                 * changes cancelled, and a half-arrived slot ran with the old addresses (one corrupted stream in five at 832x480). */
                uint64_t x = tid < 15 ? w * (XA_CHECK_MUL * (uint64_t)(2 * tid + 1)) : 0;
                x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
                const uint64_t want = __shfl(x, 0, 64) ^ (XA_CHECK_MUL * (seen + 1)) ^ generation;
                if (__shfl(w, 15, 64) == want)
                {
                    if (tid < 16) reinterpret_cast<uint64_t*>(&s_cmd)[tid] = w;
                    xa_wave_sync();
                    go = 1;
                    break;
                }
                /* nothing to run: if the host waits for what has been run (doorbell) and has not been told yet, tell it.  (A doorbell seen ahead of a
                 * command still on its way only makes this report early; the next one follows when the queue is empty again.) */
                if (__shfl(w, 16, 64) >= seen && signalled < seen) { go = 2; break; }
                if ((spins & 63) == 0)
                {
                    int stop = 0;
                    if (tid == 0)
                    {
                        stop = xa_sys_load(&rd->quit) != 0;
                        if (!stop && wall_clock64() - tIdle > idleTicks)
                        {
                            const uint64_t a = xa_sys_load(&hosts[0].alive);
                            if (a == lastAlive) stop = 1; else { lastAlive = a; tIdle = wall_clock64(); }
                        }
                    }
                    if (__shfl(stop, 0, 64)) break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (tid == 0) s_prof[62] += (unsigned long long)(wall_clock64() - t0);
            if (tid == 0)
            {
                s_go = go;
                /* Caches and the resident workgroup.  Job records are read with system-scope loads (xa_ld_record) and what a command produces is
                 * read back by this same CU, so most commands need no cache maintenance.  The exceptions: a command that asks for it (XA_CMD_ACQUIRE:
                 * other rows', pictures' and copy engines' writes), copies (their source may be the pinned staging area the host refills in place) and
                 * the RDOQ / estBit commands, which read host-written tables with plain loads.  Without the invalidation the second use of such a
                 * buffer reads the first use's bytes (measured).  The scalar data cache is never touched by a fence: see below. */
                const uint32_t op = go == 1 ? reinterpret_cast<const uint32_t*>(&s_cmd)[0] : 0, fl = go == 1 ? reinterpret_cast<const uint32_t*>(&s_cmd)[1] : 0;
                if (go == 1 && ((fl & XA_CMD_ACQUIRE) || (reinterpret_cast<const uint32_t*>(&s_cmd)[3] & 32) || op == XA_OP_COPY || op == XA_OP_EST_BIT || op == XA_OP_TU_CHAIN_RDOQ || op == XA_OP_INTRA_TU_CHAIN_RDOQ))
                {
                    const long long tf = wall_clock64();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                    __builtin_amdgcn_s_dcache_inv();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    s_prof[63] += (unsigned long long)(wall_clock64() - tf);
                }
            }
        }
        __syncthreads();
        if (!s_go) break;
        if (s_go == 2)
        {
            /* the queue is empty and the host waits: every wavefront's stores leave, then the count goes out (as after a signalling command, below) */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
            {
                const long long te = wall_clock64();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                xa_sys_store(&rh->tail, seen);
                s_prof[63] += (unsigned long long)(wall_clock64() - te);
            }
            signalled = seen;
            continue;
        }
        const uint32_t flags = s_cmd.flags;
        if (s_cmd.op == XA_OP_EXIT) break;
        const long long td = wall_clock64();
        /* the slot behind this command and the doorbell, fetched while the command runs */
        if (XA_PREFETCH_SLOT && tid < 64)
        {
            const uint64_t* nextSlot = reinterpret_cast<const uint64_t*>(&rd->cmd[(seen + 1) % XA_RING]);
            pre = tid < 16 ? xa_sys_load(nextSlot + tid) : (tid == 16 ? xa_sys_load(&rd->head) : 0);
            havePre = true;
        }
        /* a queue whose XA_OP_WAIT gave up runs nothing more: its commands still count as finished (the host's ring bookkeeping goes on), their results
         * stay unwritten and XaRingHost::fault tells every wait on this queue that they are */
        if (flags & XA_CMD_RESET)
        {
            /* the poison of a wait that gave up ends with the owner it happened to (xa_queue_acquire sends this in front of a new owner's first command) */
            __syncthreads();
            if (tid == 0) { xa_wait_failed = 0; xa_sys_store(&rh->fault, 0ull); }
            __syncthreads();
        }
        if (!xa_wait_failed) xa_dispatch(s_cmd, tid);
        if (s_cmd.op == XA_OP_WAIT && tid == 0 && xa_wait_failed) xa_sys_store(&rh->fault, seen + 1);
        /* before the workgroup reports or publishes, every wavefront's stores have left (results live in host memory, read as soon as the count
         * moves).  Between two commands of the queue the barrier is enough: the CU's vector memory path keeps the order of one workgroup's accesses,
         * which is all the compiler itself relies on for a workgroup-scope release in this execution mode. */
        if (flags & (XA_CMD_RELEASE | XA_CMD_SIGNAL)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        /* what a copy or fill has just written may be read with SCALAR loads by the commands behind it (plane tables, group lists: wave-uniform
         * addresses); the scalar data cache is coherent with nothing, so it goes now */
        if (tid == 0 && (s_cmd.op == XA_OP_COPY || s_cmd.op == XA_OP_COPY2D || s_cmd.op == XA_OP_FILL || s_cmd.op == XA_OP_COPY_RECTS))
        {
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        seen++;
        if (tid == 0)
        {
            const long long te = wall_clock64();
            const int pslot = (int)(s_cmd.op & 31) < 31 ? (int)(s_cmd.op & 31) : 30;
            s_prof[2 * pslot] += 1; s_prof[2 * pslot + 1] += (unsigned long long)(te - td);
            s_bytes[s_cmd.op & 31] += xa_bytes_acc; xa_bytes_acc = 0;          /* behind the barrier: every lane's contribution is in */
            if (s_cmd.reserved & 16)        /* X265AMD_QUEUE_DEBUG & 16: single-job commands of the three hot kinds by block size */
            {
                const XaArgsJobs4& a = *reinterpret_cast<const XaArgsJobs4*>(s_cmd.args);
                int b = -1;
                if (s_cmd.op == XA_OP_INTRA_SCAN && a.n == 1) b = 20 + (reinterpret_cast<const x265amd_intra_job*>(a.a)->log2_tr_size - 2);
                else if (s_cmd.op == XA_OP_INTRA_PU) b = 20 + (reinterpret_cast<const x265amd_intra_pu_job*>(a.a)->tmpl.tu.log2_tr_size - 2);
                else if (s_cmd.op == XA_OP_INTRA_TU_CHAIN) b = 24 + (reinterpret_cast<const x265amd_intra_tu_job*>(a.a)->tu.log2_tr_size - 2);
                else if (s_cmd.op == XA_OP_CU_MEASURE && a.n == 1) b = 28 + (reinterpret_cast<const CuMeasureJob*>(a.a)->log2_size - 3);
                if (b >= 20 && b < 31) { s_sized[2 * (b - 20)] += 1; s_sized[2 * (b - 20) + 1] += (unsigned long long)(te - td); }
            }
            if (flags & (XA_CMD_RELEASE | XA_CMD_SIGNAL))
            {
                /* what the host (results in pinned memory) and other workgroups (pictures) will read leaves this XCD's L2 now: the L2 keeps the lines a
                 * workgroup has stored to host memory, and nothing but a release writes them back while the kernel is resident.  (Storing the
                 * results with system scope instead was measured: slower, every store then waits for the host's acknowledgement.) */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                s_prof[63] += (unsigned long long)(wall_clock64() - te);
            }
            if (flags & XA_CMD_SIGNAL) xa_sys_store(&rh->tail, seen);
            xa_sys_store(&rd->done, seen);          /* for a queue that follows this one (XA_OP_WAIT): behind the release when the command carried one */
        }
        if (flags & XA_CMD_SIGNAL) signalled = seen;
    }
    __syncthreads();
    __syncthreads();
    if (tid >= 128 && tid < 150) xa_sys_store(&rh->stage[tid - 128], rh->stage[tid - 128] + xa_stage_acc[tid - 128]);
    if (tid >= 160 && tid < 184) xa_sys_store(&rh->sized[tid - 160], rh->sized[tid - 160] + s_sized[tid - 160]);
    if (tid < 64) xa_sys_store(&rh->prof[tid], rh->prof[tid] + s_prof[tid]);       /* totals over the server generations (the host clears them) */
    if (tid < 32) xa_sys_store(&rh->bytes[tid], rh->bytes[tid] + s_bytes[tid]);
    if (tid >= 64 && tid < 104) xa_sys_store(&rh->nxn[tid - 64], rh->nxn[tid - 64] + (&xa_nxn_acc[0][0])[tid - 64]);
    if (tid >= 104 && tid < 112) xa_sys_store(&rh->chain[tid - 104], rh->chain[tid - 104] + xa_chain_acc[tid - 104]);
    if (tid == 32) { xa_sys_store(&rh->resident, rh->resident + (unsigned long long)(wall_clock64() - tResident0)); xa_sys_store(&rh->cycles, rh->cycles + (unsigned long long)(clock64() - cResident0)); }
    if (tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); xa_sys_store(&rh->state, 0); }
}

/* =========================================================================================================
 * host side
 * ======================================================================================================= */
void xa_bind_device();
void xa_thread_device();
namespace {

const size_t kStagingBytes = 1 << 20;

struct Deferred { void* dst; const void* src; size_t bytes; };      /* staging -> pageable destination once the queue has drained */

}

struct XaQueue
{
    int idx = 0;
    XaRingDev* rd = nullptr;            /* device memory through the BAR */
    XaRingHost* rh = nullptr;           /* pinned host memory */
    uint64_t submitted = 0;             /* commands written so far */
    uint64_t lastSignal = 0;            /* `submitted` after the last command that carried XA_CMD_SIGNAL */
    uint64_t generation = 0;            /* of the server kernel this queue talks to: part of every command's check word, so that a slot left over from an
                                           earlier kernel (the rings are not cleared, and command numbers restart) never passes for a new command */
    char* staging = nullptr; size_t stagingUsed = 0, stagingUsedOut = 0;
    std::vector<Deferred> deferred;
    bool busy = false;
    std::chrono::steady_clock::time_point acquired;
    /* X265AMD_QUEUE_LOG: the events of one row (xa_queue_log): wall time, the task's running time, kind ('E' command written, 'W' wait begins, 'R' wait over), op */
    struct Ev { uint64_t wallNs, runNs; char kind; int op; };
    std::vector<Ev> log; bool logging = false; int logPoc = 0, logRow = 0;
    void* helper = nullptr;             /* a second queue the holder of this one may use beside it (xa_queue_set_helper) */
    void* aux = nullptr;                /* another one, for the searches a row of a P picture starts ahead (xa_queue_set_aux) */
    uint32_t nextFlags = 0;             /* flags the next command gets on top of its own (xa_q_next_flags) */
    std::vector<void*> laterMapped;     /* pushed-record blocks that commands still in the queue read: back to the pool at the next synchronisation (xa_q_free_mapped_later) */
    void ev(char kind, int op)
    {
        if (!logging) return;
        static const auto t00 = std::chrono::steady_clock::now();
        log.push_back(Ev{ (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t00).count(), xa_task_run_ns(), kind, op });
    }
};

namespace {

void dump_debug_areas(int);
std::atomic<uint64_t> g_waitNs(0), g_heldNs(0), g_waits(0), g_depNs(0);
const bool g_prof = getenv("X265AMD_QUEUE_PROF") != nullptr;

/* the number of queues when X265AMD_QUEUES does not say: a resident workgroup on 224 of the 256 CUs.  Measured: 1080p encodes the same with 128 and with 224
 * (26.1 / 26.2 frames/s over three alternating runs each); 2160p, whose I pictures want up to four queues for each of 34 CTU rows, 9.9 against 12.8 */
std::atomic<int> g_queuesHint{ 224 };

struct Server
{
    std::mutex m;
    std::condition_variable freed;
    int numQueues = 0;
    XaRingDev* rings = nullptr;
    XaRingHost* hosts = nullptr;
    char* staging = nullptr;
    std::vector<XaQueue> q;
    hipStream_t stream = nullptr;
    /* The queues are served by TWO launches of the resident kernel (round 5): the first `firstCount` queues from the start, the rest from the moment a queue beyond them is
     * handed out (queues are taken lowest index first).  A resident workgroup holds a whole compute unit (144 KB of LDS, 256 vector registers per lane): with all 224 resident
     * from the first picture on, everything else -- the lookahead's cost estimates above all, whose first decision is 384 estimates -- ran on the 32 units left (527 ms at
     * 2160p against 298 on an idle device, longer than the I picture beside it).  While only the I picture runs, its rows need a fraction of the queues.  The second launch
     * has a stream of the OTHER extreme priority: like the first it must not share a hardware queue with anything (see init). */
    hipStream_t stream2 = nullptr;
    int firstCount = 0;
    bool running2 = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;        /* around every launch of the resident kernel, on its stream: its duration by HIP events (bench.py's roofline) */
    double kernelMs = 0.0; uint64_t launches = 0;
    bool running = false;
    int refs = 0;
    uint64_t generation = 0;
    bool disabled = false, ringsInHost = false;
    volatile uint64_t freeCount = 0;            /* queues not taken (what parked row tasks watch; changed under the lock) */

    int init()
    {
        if (numQueues) return 0;
        xa_bind_device();
        xa_thread_device();
        const char* e = getenv("X265AMD_QUEUES");
        int n = e ? atoi(e) : g_queuesHint.load();
        if (n <= 0) { disabled = true; return -1; }
        if (n > 224) n = 224;
        ringsInHost = getenv("X265AMD_RING_HOST") != nullptr;
        if (ringsInHost ? hipHostMalloc((void**)&rings, sizeof(XaRingDev) * n, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess
                        : hipExtMallocWithFlags((void**)&rings, sizeof(XaRingDev) * n, hipDeviceMallocUncached) != hipSuccess)
            return -1;
        if (!ringsInHost && (hipMemset(rings, 0, sizeof(XaRingDev) * n) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) return -1;
        if (ringsInHost) memset((void*)rings, 0, sizeof(XaRingDev) * n);
        if (hipHostMalloc((void**)&hosts, sizeof(XaRingHost) * n, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
        if (hipHostMalloc((void**)&staging, kStagingBytes * n, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
        /* The resident kernel must not share a hardware queue with anything: a kernel launched behind it on the same hardware queue waits until it
         * leaves (the in-loop filters of a finished picture would wait for every other picture's analysis).  Streams of another priority are given
         * hardware queues of their own, and nothing else in this library asks for a priority. */
        {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest) != hipSuccess)
                if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return -1;
        }
        {
            const char* f = getenv("X265AMD_QUEUES_FIRST");
            /* Measured (profiles/r05_queues_first_sweep.txt): with 96 or 128 first the lookahead's first decision falls from 135 to 89 ms at 1080p (518 -> 499 at 2160p, whose
             * I picture wants more than that many queues within 170 ms), but the 20-frame encodes are not faster (0.38 -> 0.39 s at 1080p: a queue handed out whose workgroup has
             * still to find a free compute unit among the lookahead's long-running rows stalls its row) -- so the default stays ONE launch of everything; X265AMD_QUEUES_FIRST=n
             * is the experiment */
            firstCount = f ? atoi(f) : n;
            if (firstCount <= 0 || firstCount > n) firstCount = n;
            int least = 0, greatest = 0;
            if (firstCount < n && (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest || hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, least) != hipSuccess))
            { stream2 = nullptr; firstCount = n; }         /* no second priority to be had: one launch, as before */
        }
        if (hipFuncSetAttribute((const void*)k_job_server, hipFuncAttributeMaxDynamicSharedMemorySize, XA_SERVER_LDS) != hipSuccess) return -1;
        if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) return -1;
        memset((void*)hosts, 0, sizeof(XaRingHost) * n);
        q.resize(n);
        for (int i = 0; i < n; i++) { q[i].idx = i; q[i].rd = rings + i; q[i].rh = hosts + i; q[i].staging = staging + kStagingBytes * i; }
        numQueues = n;
        freeCount = n;
        if (getenv("X265AMD_QUEUE_DEBUG") && (atoi(getenv("X265AMD_QUEUE_DEBUG")) & 2)) signal(SIGABRT, dump_debug_areas);
        return 0;
    }
    /* called with the lock held */
    int start()
    {
        if (running) return 0;
        xa_thread_device();
        /* every workgroup starts counting at 0; the head and quit words are reset through the BAR (posted writes, ordered before the launch's doorbell) */
        generation += 0x0123456789ABCDEFull;
        for (int i = 0; i < numQueues; i++)
        {
            XaQueue& x = q[i];
            x.submitted = 0; x.lastSignal = 0; x.generation = generation;
            x.rh->tail = 0; x.rh->state = 0; x.rh->fault = 0; x.rh->dbg[63] = 0;
            *reinterpret_cast<volatile uint64_t*>(&rings[i].quit) = 0;
            *reinterpret_cast<volatile uint64_t*>(&rings[i].head) = 0;
            *reinterpret_cast<volatile uint64_t*>(&rings[i].done) = 0;
        }
        _mm_sfence();
        (void)hipEventRecord(ev0, stream);
        hipLaunchKernelGGL(k_job_server, dim3(firstCount), dim3(64 * XA_SERVER_WAVES), XA_SERVER_LDS, stream, rings, hosts, 100000000LL * 60, generation, 0);
        if (hipGetLastError() != hipSuccess) return -1;
        (void)hipEventRecord(ev1, stream);
        running = true; running2 = false;
        return 0;
    }
    /* called with the lock held, the first launch running: the queues from firstCount on get their workgroups */
    int startSecond()
    {
        if (running2 || firstCount >= numQueues) return 0;
        xa_thread_device();
        hipLaunchKernelGGL(k_job_server, dim3(numQueues - firstCount), dim3(64 * XA_SERVER_WAVES), XA_SERVER_LDS, stream2, rings, hosts, 100000000LL * 60, generation, firstCount);
        if (hipGetLastError() != hipSuccess) return -1;
        running2 = true;
        return 0;
    }
    void stop()
    {
        if (!running) return;
        for (int i = 0; i < numQueues; i++) rings[i].quit = 1;
        _mm_sfence();
        (void)hipStreamSynchronize(stream);
        if (running2) (void)hipStreamSynchronize(stream2);
        running = false; running2 = false;
        { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) { kernelMs += ms; launches++; } }
        static const bool prof = getenv("X265AMD_QUEUE_PROF") != nullptr;
        if (prof) profile_report(false);
    }
    void profile_report(bool final)
    {
        static const char* const names[XA_OP_COUNT] = { "nop/fence", "exit", "copy", "copy2d", "fill", "copy_rects", "mc", "mc_cost", "cu_measure", "tu_chain", "tu_chain_rdoq", "intra_tu_chain",
                                                         "intra_tu_chain_rdoq", "intra_scan", "me_search", "me_search_star", "me_deferred", "est_bit", "intra_pu", "intra_nxn", "inter_chain", "inter_search", "wait" };
        uint64_t tot[64] = { 0 }, stage[24] = { 0 };
        for (int i = 0; i < numQueues; i++) for (int k = 0; k < 64; k++) tot[k] += hosts[i].prof[k];
        for (int i = 0; i < numQueues; i++) for (int k = 0; k < 22; k++) stage[k] += hosts[i].stage[k];
        uint64_t cmds = 0, ticks = 0;
        for (int op = 0; op < XA_OP_COUNT; op++) { cmds += tot[2 * op]; ticks += tot[2 * op + 1]; }
        if (!final && cmds < lastReported + 2000000) return;
        lastReported = cmds;
        fprintf(stderr, "x265amd host threads: queues held %.1f ms in all, of which %.1f ms waiting for the row above, %.1f ms waiting for the device in %llu waits (%.2f us each)\n",
                g_heldNs.load() / 1e6, g_depNs.load() / 1e6, g_waitNs.load() / 1e6, (unsigned long long)g_waits.load(), g_waits.load() ? g_waitNs.load() / 1e3 / g_waits.load() : 0.0);
        fprintf(stderr, "x265amd job server: %llu commands, %.1f ms in command bodies, %.1f ms in fences, %.1f ms polling (all queues; 100 MHz clock)\n", (unsigned long long)cmds,
                ticks / 1e5, tot[63] / 1e5, tot[62] / 1e5);
        {
            uint64_t res = 0, cyc = 0;
            for (int i = 0; i < numQueues; i++) { res += hosts[i].resident; cyc += hosts[i].cycles; }
            if (res) fprintf(stderr, "  shader clock while resident: %.0f MHz (s_memtime against the 100 MHz clock, all workgroups)\n", 100.0 * cyc / res);
        }
        for (int op = 0; op < XA_OP_COUNT; op++)
            if (tot[2 * op]) fprintf(stderr, "  %-20s %9llu x %7.2f us = %8.1f ms\n", names[op], (unsigned long long)tot[2 * op], tot[2 * op + 1] / 100.0 / tot[2 * op], tot[2 * op + 1] / 1e5);
        fprintf(stderr, "  stages of the transform chains as wavefront 0 saw them (ms): record %.1f, neighbours %.1f, prediction %.1f, residual %.1f, transforms %.1f, quantisation %.1f, sign hiding %.1f, "
                "levels out + sse %.1f, psy %.1f, inverse %.1f, reconstruction %.1f, sse + psy %.1f, result %.1f, pu record / select %.1f, pu scan %.1f, elsewhere %.1f\n", stage[0] / 1e5, stage[1] / 1e5, stage[2] / 1e5, stage[3] / 1e5, stage[4] / 1e5,
                stage[5] / 1e5, stage[6] / 1e5, stage[7] / 1e5, stage[8] / 1e5, stage[9] / 1e5, stage[10] / 1e5, stage[11] / 1e5, stage[12] / 1e5, stage[13] / 1e5, stage[14] / 1e5, stage[15] / 1e5);
        fprintf(stderr, "  stages of the NxN step (ms): record + predictors %.1f, scan %.1f, candidate list %.1f, chains %.1f, bits %.1f, choice + blocks + measurements + chroma %.1f\n",
                stage[16] / 1e5, stage[17] / 1e5, stage[18] / 1e5, stage[19] / 1e5, stage[20] / 1e5, stage[21] / 1e5);
        {
            uint64_t nx[40] = { 0 };
            for (int i = 0; i < numQueues; i++) for (int k = 0; k < 40; k++) nx[k] += hosts[i].nxn[k];
            static const char* const kinds[4] = { "4 x 4x4", "1 x 8x8", "1 x 16x16", "1 x 32x32" };
            for (int kd = 0; kd < 4; kd++)
            {
                uint64_t t = 0; for (int k = 0; k < 10; k++) t += nx[kd * 10 + k];
                if (!t) continue;
                fprintf(stderr, "  intra_nxn %-10s (ms): record %.1f, predictors %.1f, scan %.1f, neighbours -> lds + candidate list %.1f, chains + bits (wave 0) %.1f, waiting for the other waves %.1f, "
                        "choice %.1f, winner's blocks %.1f, luma measurements %.1f, chroma %.1f\n", kinds[kd], nx[kd * 10] / 1e5, nx[kd * 10 + 1] / 1e5, nx[kd * 10 + 2] / 1e5, nx[kd * 10 + 3] / 1e5,
                        nx[kd * 10 + 4] / 1e5, nx[kd * 10 + 5] / 1e5, nx[kd * 10 + 6] / 1e5, nx[kd * 10 + 7] / 1e5, nx[kd * 10 + 8] / 1e5, nx[kd * 10 + 9] / 1e5);
            }
        }
        {
            uint64_t cx[8] = { 0 };
            for (int i = 0; i < numQueues; i++) for (int k = 0; k < 8; k++) cx[k] += hosts[i].chain[k];
            if (cx[6] || cx[7])
                fprintf(stderr, "  chained 8x8 CUs (ms, all workgroups): the deciding command: its evaluation %.1f, waiting for the other evaluation %.1f, the decision up to the chain's word %.1f; "
                        "the other command: waiting for the chain %.1f, its evaluation up to its word %.1f\n", cx[5] / 1e5, cx[6] / 1e5, cx[4] / 1e5, cx[7] / 1e5, cx[3] / 1e5);
            if (cx[3])
                fprintf(stderr, "  chained 8x8 CUs (ms): the deciding command waiting for the chain %.1f, the other command waiting for the chain %.1f, waiting for the other evaluation %.1f, "
                        "its record + both CUs' bits %.1f, costs + the winner's samples + the result %.1f, publishing %.1f\n", cx[0] / 1e5, cx[1] / 1e5, cx[2] / 1e5, cx[3] / 1e5, cx[4] / 1e5, cx[5] / 1e5);
        }
        static const char* const sized[11] = { "scan / pu 4", "scan / pu 8", "scan / pu 16", "scan / pu 32", "intra_tu* 4", "intra_tu* 8", "intra_tu* 16", "intra_tu* 32",
                                               "cu_measure 8", "cu_measure 16", "cu_measure 32" };
        {
            uint64_t sz[24] = { 0 };
            for (int i = 0; i < numQueues; i++) for (int k = 0; k < 24; k++) sz[k] += hosts[i].sized[k];
            for (int b = 0; b < 11; b++)
                if (sz[2 * b]) fprintf(stderr, "  (n = 1) %-12s %9llu x %7.2f us = %8.1f ms\n", sized[b], (unsigned long long)sz[2 * b], sz[2 * b + 1] / 100.0 / sz[2 * b], sz[2 * b + 1] / 1e5);
        }
    }
    uint64_t lastReported = 0;
};

Server& server() { static Server* s = new Server; return *s; }

/* X265AMD_QUEUE_DEBUG & 2: the runtime aborts the process on a GPU memory fault; say what the wavefronts had announced */
void dump_debug_areas(int)
{
    Server& S = server();
    for (int i = 0; i < S.numQueues; i++)
    {
        if (!S.q[i].busy) continue;
        fprintf(stderr, "x265amd queue %d debug area (wave x slot):\n", i);
        for (int w = 0; w < 8; w++)
        {
            fprintf(stderr, "  wave %d:", w);
            for (int k = 0; k < 8; k++) fprintf(stderr, " %llx", (unsigned long long)S.hosts[i].dbg[w * 8 + k]);
            fprintf(stderr, "\n");
        }
    }
    fflush(stderr);
    signal(SIGABRT, SIG_DFL);
    abort();
}

inline XaQueue* as_queue(void* st) { return reinterpret_cast<XaQueue*>((uintptr_t)st & ~(uintptr_t)1); }

/* a queue whose XA_OP_WAIT timed out (k_job_server: XaRingHost::fault) has skipped every command since: whatever the caller waits for was not produced */
static int q_faulted(XaQueue* q)
{
    const uint64_t f = *reinterpret_cast<const volatile uint64_t*>(&q->rh->fault);
    if (!f) return 0;
    fprintf(stderr, "x265amd queue %d: command %llu (XA_OP_WAIT) gave up waiting for the queue it follows; the commands behind it were not run\n", q->idx, (unsigned long long)f);
    return -1;
}

int q_wait(XaQueue* q, uint64_t target)
{
    const volatile uint64_t* tail = &q->rh->tail;
    if (*tail >= target) return q_faulted(q);
    q->ev('W', 0);
    struct EvEnd { XaQueue* q; ~EvEnd() { q->ev('R', 0); } } evEnd{ q };
    const auto t0 = std::chrono::steady_clock::now();
    struct Acc { std::chrono::steady_clock::time_point t0; ~Acc() { if (g_prof) { g_waitNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_waits++; } } } acc{ t0 };
    if (xa_in_task())
    {
        /* a row task: park; the worker thread runs another row meanwhile (xa_fiber.h).  A server that went away shows as a wait without end: the time limit
         * (longer than any command: 120 s) turns it into an error, as the 30 s of the ordinary path below does for direct callers */
        if (xa_wait_counter_deadline(tail, target, 120ull * 1000000000ull))
        {
            fprintf(stderr, "x265amd queue %d: no answer: submitted %llu, waiting for %llu, finished %llu, resident %llu\n", q->idx,
                    (unsigned long long)q->submitted, (unsigned long long)target, (unsigned long long)*tail, (unsigned long long)q->rh->state);
            return -1;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return q_faulted(q);
    }
    for (unsigned spins = 0;; spins++)
    {
        if (*tail >= target) break;
        _mm_pause();
        if ((spins & 0xFFFFF) == 0xFFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
        {
            fprintf(stderr, "x265amd queue %d: no answer: submitted %llu, waiting for %llu, finished %llu, resident %llu, slot re-reads %llu\n", q->idx,
                    (unsigned long long)q->submitted, (unsigned long long)target, (unsigned long long)*tail, (unsigned long long)q->rh->state, (unsigned long long)q->rh->dbg[63]);
            return -1;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return q_faulted(q);
}

void xa_server_alive();
std::atomic<uint64_t> g_pushNs{ 0 }, g_pushN{ 0 };
int q_push(XaQueue* q, uint32_t op, uint32_t flags, uint32_t count, const void* args, size_t argBytes)
{
    static const bool timing = getenv("X265AMD_TIMING") != nullptr;
    struct PushTimer { bool on; std::chrono::steady_clock::time_point t0; ~PushTimer() { if (on) { g_pushNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_pushN++; } } }
        pt{ timing, timing ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point() };
    if (argBytes > sizeof(uint64_t) * XA_CMD_ARG_WORDS) return -1;
    if (q->submitted - q->lastSignal >= XA_RING - 2) flags |= XA_CMD_SIGNAL;
    if (q->submitted >= XA_RING && q_wait(q, q->submitted - XA_RING + 1)) return -1;        /* the slot must have been consumed */
    XaCmd c;
    memset(&c, 0, sizeof(c));
    static const uint32_t debugBits = getenv("X265AMD_QUEUE_DEBUG") ? (uint32_t)atoi(getenv("X265AMD_QUEUE_DEBUG")) : 0;
    c.op = op; c.flags = flags; c.count = count; c.reserved = debugBits;
    if (argBytes) memcpy(c.args, args, argBytes);
    {
        const uint64_t* w = reinterpret_cast<const uint64_t*>(&c);
        uint64_t x = 0;
        for (int i = 0; i < 15; i++) x += w[i] * (XA_CHECK_MUL * (uint64_t)(2 * i + 1));
        c.check = x ^ (XA_CHECK_MUL * (q->submitted + 1)) ^ q->generation;
    }
    /* everything the command refers to (job records pushed through the BAR, staging and other host memory) is globally visible before the slot:
     * the slot is what the workgroup polls, and write-combining buffers drain in no particular order */
    _mm_sfence();
    XaCmd* slot = &q->rd->cmd[q->submitted % XA_RING];
    /* 128 bytes through the write-combining BAR mapping, then the doorbell behind a store fence */
    const __m128i* s = reinterpret_cast<const __m128i*>(&c);
    __m128i* d = reinterpret_cast<__m128i*>(slot);
    for (int i = 0; i < 8; i++) _mm_store_si128(d + i, _mm_load_si128(s + i));
    _mm_sfence();                       /* the slot is the doorbell: out of the write-combining buffer now */
    q->submitted++;
    if ((q->submitted & 255) == 0) xa_server_alive();
    if (flags & XA_CMD_SIGNAL) q->lastSignal = q->submitted;
    q->ev('E', (int)op);
    return 0;
}

} // namespace

namespace { void xa_server_alive() { Server& S = server(); if (S.hosts) __atomic_fetch_add(&S.hosts[0].alive, 1, __ATOMIC_RELAXED); } }
void xa_prof_dependency_wait(uint64_t ns) { if (g_prof) g_depNs += ns; }

/* ---- the process's device.  HIP's current device is per thread and defaults to 0; a process of a multi-GPU job selects its GPU on its main thread only
 * (torch.cuda.set_device(LOCAL_RANK)).  The first encoder object / job server records that thread's device, and every thread this library creates
 * (row-task workers, picture tasks, filter threads) selects it before its first HIP call. ---- */
namespace { std::atomic<int> g_device{ -1 }; }
void xa_bind_device()
{
    int expect = -1, dev = 0;
    if (g_device.load() >= 0 || hipGetDevice(&dev) != hipSuccess) return;
    g_device.compare_exchange_strong(expect, dev);
    xa_fiber_set_thread_init(xa_thread_device);
}
void xa_thread_device() { const int d = g_device.load(); if (d >= 0) (void)hipSetDevice(d); }

/* ---- X265AMD_TIMING: host phases of the row tasks ---- */
namespace {
std::atomic<uint64_t> g_phaseNs[XA_PH_COUNT], g_phaseN[XA_PH_COUNT];
const bool g_phases = getenv("X265AMD_TIMING") != nullptr;
}
void xa_phase(int k)
{
    if (!g_phases) return;
    uint64_t* mark = xa_task_mark();
    if (!mark) return;
    const uint64_t now = xa_task_run_ns();
    if (*mark && now >= *mark) { g_phaseNs[k] += now - *mark; g_phaseN[k]++; }
    *mark = now;
}
void xa_phase_report(void)
{
    if (!g_phases) return;
    static const char* const names[XA_PH_COUNT] = { "other", "intra setup", "intra scan", "intra candidates", "intra bits", "intra chroma", "intra final", "push", "row coder", "analyzer",
                                                    "inter search", "inter rd", "merge predict + measure", "merge candidates", "merge rd", "rd plan + launch", "rd skip host", "rd walk" };
    fprintf(stderr, "x265amd: command pushes: %llu, %.1f ms in all (%.2f us each)\n", (unsigned long long)g_pushN.load(), g_pushNs.load() / 1e6, g_pushN.load() ? g_pushNs.load() / 1e3 / g_pushN.load() : 0.0);
    fprintf(stderr, "x265amd: host phases of the row tasks (ms, stamps):");
    for (int k = 0; k < XA_PH_COUNT; k++) fprintf(stderr, " %s %.1f (%llu)", names[k], g_phaseNs[k].load() / 1e6, (unsigned long long)g_phaseN[k].load());
    fprintf(stderr, "\n");
}

/* A caller that knows it will want more queues than the default (pictures above 1080 lines: more CTU rows in flight, and the rows of I pictures take up to
 * four) says so before the job server starts; afterwards the call changes nothing.  Returns the number a server started now would have. */
int xa_queues_hint(int n)
{
    if (n > 224) n = 224;
    int cur = g_queuesHint.load();
    while (n > cur && !g_queuesHint.compare_exchange_weak(cur, n)) {}
    const char* e = getenv("X265AMD_QUEUES");
    return e ? atoi(e) : g_queuesHint.load();
}

bool xa_queues_enabled()
{
    Server& S = server();
    std::lock_guard<std::mutex> g(S.m);
    return !S.disabled && S.init() == 0;
}

/* A queue whose XA_OP_WAIT gave up skips every command behind it and says so to every wait (XaRingHost::fault).  That ends with the owner it happened to: the next one
 * starts with a command that clears the workgroup's flag and the fault word, and waits for it (rare: the queue stood for two seconds under its last owner). */
static void xa_queue_clear_fault(void* st)
{
    XaQueue* q = as_queue(st);
    if (!*reinterpret_cast<const volatile uint64_t*>(&q->rh->fault)) return;
    fprintf(stderr, "x265amd queue %d: handed out again after a wait that gave up; the fault is cleared for its new owner\n", q->idx);
    if (xa_q_enqueue(st, XA_OP_NOP, nullptr, 0, 1, XA_CMD_RESET | XA_CMD_SIGNAL) != hipSuccess) return;
    const volatile uint64_t* fw = &q->rh->fault;
    for (int i = 0; i < 20000 && *fw; i++) std::this_thread::sleep_for(std::chrono::microseconds(100));
}

void* xa_queue_acquire()
{
    Server& S = server();
    std::unique_lock<std::mutex> g(S.m);
    if (S.disabled || S.init() != 0) return nullptr;
    /* All queues taken: wait for one.  Falling back to a HIP stream here would put ordinary kernels beside the resident one, and a row that runs on
     * launches while the rows around it run on queues can starve behind the resident kernel.  The rows of a picture take their queues in row order,
     * so the row everybody else waits for always holds one: the wait ends. */
    XaQueue* f = nullptr;
    for (;;)
    {
        for (XaQueue& x : S.q) if (!x.busy) { f = &x; break; }
        if (f) break;
        if (xa_in_task())
        {
            /* a row task never sleeps on a condition variable: it parks until a queue is given back */
            g.unlock();
            xa_wait_counter(&S.freeCount, 1);
            g.lock();
            continue;
        }
        if (S.freed.wait_for(g, std::chrono::seconds(120)) == std::cv_status::timeout) return nullptr;
    }
    if (S.start() != 0 || (f->idx >= S.firstCount && S.startSecond() != 0)) return nullptr;
    __atomic_fetch_add(&S.hosts[0].alive, 1, __ATOMIC_RELAXED);
    f->busy = true; f->stagingUsed = 0; f->stagingUsedOut = 0; f->deferred.clear(); f->helper = nullptr; f->aux = nullptr;
    S.freeCount = S.freeCount - 1;
    S.refs++;
    xa_scratch_local_begin();           /* the calling thread is the one that uses the queue */
    f->acquired = std::chrono::steady_clock::now();
    void* st = reinterpret_cast<void*>((uintptr_t)f | 1);
    g.unlock();
    xa_queue_clear_fault(st);
    return st;
}

/* a second queue for the holder of a first one, if one is free right now: never waits (the rows of a picture take their FIRST queues in row order so that
 * waiting for one always ends; a second queue is a bonus) */
void* xa_queue_try_acquire() { return xa_queue_try_acquire_spare(-1); }
/* ... with the number of queues that must stay free given by the caller (-1: X265AMD_HELPER_SPARE, default 24): the extra queues of a P picture's rows are a
 * convenience and leave half of the queues alone, the extra queues of an I picture's rows halve its time and take what there is */
void* xa_queue_try_acquire_spare(int spareWanted)
{
    Server& S = server();
    std::unique_lock<std::mutex> g(S.m);
    if (S.disabled || S.init() != 0) return nullptr;
    static const int spareDefault = getenv("X265AMD_HELPER_SPARE") ? atoi(getenv("X265AMD_HELPER_SPARE")) : 24;       /* queues left to the rows that need a first one */
    const int spare = spareWanted >= 0 ? spareWanted : spareDefault;
    int freeN = 0;
    XaQueue* f = nullptr;
    for (XaQueue& x : S.q) if (!x.busy) { freeN++; if (!f) f = &x; }
    if (!f || freeN <= spare) return nullptr;
    if (S.start() != 0 || (f->idx >= S.firstCount && S.startSecond() != 0)) return nullptr;
    f->busy = true; f->stagingUsed = 0; f->stagingUsedOut = 0; f->deferred.clear(); f->helper = nullptr; f->aux = nullptr;
    S.freeCount = S.freeCount - 1;
    S.refs++;
    f->acquired = std::chrono::steady_clock::now();
    void* st = reinterpret_cast<void*>((uintptr_t)f | 1);
    g.unlock();
    xa_queue_clear_fault(st);
    return st;
}
void xa_queue_release_helper(void* st)
{
    if (!xa_is_queue(st)) return;
    XaQueue* q = as_queue(st);
    (void)xa_stream_fence(st, XA_CMD_RELEASE);
    (void)xa_stream_sync(st);
    Server& S = server();
    std::lock_guard<std::mutex> g(S.m);
    q->busy = false;
    S.freeCount = S.freeCount + 1;
    S.freed.notify_one();
    if (--S.refs == 0) S.stop();
}
void xa_queue_set_helper(void* st, void* helper) { if (xa_is_queue(st)) as_queue(st)->helper = helper; }
void* xa_queue_helper(void* st) { return xa_is_queue(st) ? as_queue(st)->helper : nullptr; }
void xa_queue_set_aux(void* st, void* aux) { if (xa_is_queue(st)) as_queue(st)->aux = aux; }
void* xa_queue_aux(void* st) { return xa_is_queue(st) ? as_queue(st)->aux : nullptr; }

void xa_queue_log(void* st, int poc, int row)
{
    if (!xa_is_queue(st)) return;
    XaQueue* q = as_queue(st);
    q->logging = true; q->logPoc = poc; q->logRow = row; q->log.clear(); q->log.reserve(1 << 16);
}

void xa_queue_release(void* st)
{
    if (!xa_is_queue(st)) return;
    XaQueue* q = as_queue(st);
    (void)xa_stream_fence(st, XA_CMD_RELEASE);
    (void)xa_stream_sync(st);
    if (q->logging)
    {
        q->logging = false;
        for (const XaQueue::Ev& e : q->log) fprintf(stderr, "x265amd qlog poc %d row %d: %.2f %.2f %c %d\n", q->logPoc, q->logRow, e.wallNs / 1e3, e.runNs / 1e3, e.kind, e.op);
        q->log.clear();
    }
    xa_scratch_local_end();
    Server& S = server();
    std::lock_guard<std::mutex> g(S.m);
    q->busy = false;
    S.freeCount = S.freeCount + 1;
    S.freed.notify_one();
    if (g_prof) g_heldNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - q->acquired).count();
    if (getenv("X265AMD_QUEUE_DEBUG") && q->rh->dbg[63]) fprintf(stderr, "x265amd queue %d: %llu command slot re-reads so far\n", q->idx, (unsigned long long)q->rh->dbg[63]);
    if (--S.refs == 0) S.stop();
}

/* the next command of the queue also carries `flags` (XA_CMD_ACQUIRE before a command that reads tables the host has just pushed into reused memory: the
 * scalar data cache may still hold the block's previous contents) -- cheaper than a command of its own; streams: nothing to do */
/* a block of pushed records (xa_mapped_alloc) that a command not yet waited for reads: it goes back to the pool when the queue is next synchronised.  Streams:
 * returns false, the caller synchronises and frees as usual. */
bool xa_q_free_mapped_later(void* st, void* p)
{
    if (!xa_is_queue(st) || !p) return false;
    as_queue(st)->laterMapped.push_back(p);
    return true;
}
void xa_q_next_flags(void* st, int flags) { if (xa_is_queue(st)) as_queue(st)->nextFlags |= (uint32_t)flags; }

hipError_t xa_q_enqueue(void* st, int op, const void* args, size_t argBytes, int count, int flags)
{
    static const bool trace = getenv("X265AMD_QUEUE_TRACE") != nullptr;     /* debugging: name every command and wait for it */
    XaQueue* q = as_queue(st);
    flags |= (int)q->nextFlags; q->nextFlags = 0;
    if (trace)
    {
        static const auto t00 = std::chrono::steady_clock::now();
        const double tEnq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t00).count();
        fprintf(stderr, "x265amd queue %d: t %.1f op %d count %d flags %d args", q->idx, tEnq, op, count, flags);
        for (size_t i = 0; i < argBytes / 8; i++) fprintf(stderr, " %llx", (unsigned long long)reinterpret_cast<const uint64_t*>(args)[i]);
        fprintf(stderr, "\n");
        fflush(stderr);
        if (q_push(q, (uint32_t)op, (uint32_t)flags | XA_CMD_SIGNAL, (uint32_t)count, args, argBytes) || q_wait(q, q->submitted)) return hipErrorUnknown;
        fprintf(stderr, "  took %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t00).count() - tEnq);
        static const bool verbose = getenv("X265AMD_QUEUE_TRACE")[0] == '2';
        if (!verbose) return hipSuccess;
        if (op == XA_OP_CU_MEASURE || op == XA_OP_MC)
            for (int w = 0; w < 8; w++)
            {
                fprintf(stderr, "  wave %d:", w);
                for (int k = 0; k < 8; k++) fprintf(stderr, " %llx", (unsigned long long)q->rh->dbg[w * 8 + k]);
                fprintf(stderr, "\n");
            }
        if (op == XA_OP_CU_MEASURE)
            for (int k = 0; k < count; k++)
            {
                const uint64_t* r = reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint64_t*>(args)[0]) + 9 * k;
                fprintf(stderr, "  host job %d:", k);
                for (int w = 0; w < 9; w++) fprintf(stderr, " %llx", (unsigned long long)r[w]);
                fprintf(stderr, "\n");
            }
        return hipSuccess;
    }
    return q_push(q, (uint32_t)op, (uint32_t)flags, (uint32_t)count, args, argBytes) == 0 ? hipSuccess : hipErrorUnknown;
}

hipError_t xa_stream_sync(void* st)
{
    if (!xa_is_queue(st)) return hipStreamSynchronize((hipStream_t)st);
    XaQueue* q = as_queue(st);
    if (q->lastSignal != q->submitted)
    {
        /* the doorbell: the number of the command the host waits for, stored into the queue's device memory (one posted 8-byte write).  The workgroup reads it
         * together with the next slot and reports as soon as it has run that far and finds nothing to run -- no empty signalling command to fetch (1.45 us). */
        *reinterpret_cast<volatile uint64_t*>(&q->rd->head) = q->submitted;
        _mm_sfence();
        q->lastSignal = q->submitted;
    }
    if (q_wait(q, q->submitted))
    {
        /* the server is gone: nothing reads the pushed records any more */
        for (void* p : q->laterMapped) xa_mapped_free(p);
        q->laterMapped.clear(); q->deferred.clear();
        xa_fail(X265AMD_EHIP, "device queue: no answer from the job server");
        return hipErrorUnknown;
    }
    for (const Deferred& d : q->deferred) memcpy(d.dst, d.src, d.bytes);
    q->deferred.clear();
    for (void* p : q->laterMapped) xa_mapped_free(p);
    q->laterMapped.clear();
    q->stagingUsed = 0; q->stagingUsedOut = 0;
    return hipSuccess;
}

/* everything enqueued on `leader` so far happens before what is enqueued on `follower` from now on -- on the device, neither queue's host waits: the leader gets a
 * releasing command, the follower a command that polls the leader's count of finished commands and then acquires */
hipError_t xa_queue_follow(void* follower, void* leader)
{
    if (!xa_is_queue(follower) || !xa_is_queue(leader)) return hipErrorInvalidValue;
    XaQueue* L = as_queue(leader);
    if (q_push(L, XA_OP_NOP, XA_CMD_RELEASE, 0, nullptr, 0)) return hipErrorUnknown;
    const XaArgsWait a = { (uint64_t)(uintptr_t)&L->rd->done, L->submitted };
    return q_push(as_queue(follower), XA_OP_WAIT, 0, 0, &a, sizeof(a)) ? hipErrorUnknown : hipSuccess;
}

hipError_t xa_stream_fence(void* st, int flags)
{
    if (!xa_is_queue(st)) return hipSuccess;        /* kernel boundaries of a stream are release / acquire points already */
    return xa_q_enqueue(st, XA_OP_NOP, nullptr, 0, 0, flags);
}

/* the staging area in halves by direction (see XaMapped in x265amd_host.h: lines the workgroup has stored to are not to be read by it later) */
static char* q_stage(XaQueue* q, size_t bytes, bool deviceWrites)
{
    const size_t need = (bytes + 127) & ~(size_t)127;
    size_t& used = deviceWrites ? q->stagingUsedOut : q->stagingUsed;
    if (used + need > kStagingBytes / 2) return nullptr;
    char* p = q->staging + (deviceWrites ? kStagingBytes / 2 : 0) + used;
    used += need;
    return p;
}

hipError_t xa_copy_async(void* st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    if (!xa_is_queue(st)) return hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)st);
    if (!bytes) return hipSuccess;
    XaQueue* q = as_queue(st);
    XaArgsCopy a = { (uint64_t)(uintptr_t)dst, (uint64_t)(uintptr_t)src, bytes, kind == hipMemcpyDeviceToHost ? 1u : 0u };
    if (kind == hipMemcpyHostToDevice)
    {
        /* pageable source: through this queue's pinned staging area (the call returns with the source free to change, as hipMemcpyAsync does) */
        char* s = q_stage(q, bytes, false);
        if (!s)
        {
            if (xa_stream_sync(st) != hipSuccess) return hipErrorUnknown;
            s = q_stage(q, bytes, false);
            if (!s) return hipMemcpy(dst, src, bytes, kind);
        }
        memcpy(s, src, bytes);
        a.src = (uint64_t)(uintptr_t)s;
    }
    else if (kind == hipMemcpyDeviceToHost)
    {
        char* s = q_stage(q, bytes, true);
        if (!s)
        {
            if (xa_stream_sync(st) != hipSuccess) return hipErrorUnknown;
            s = q_stage(q, bytes, true);
            if (!s) return hipMemcpy(dst, src, bytes, kind);
        }
        a.dst = (uint64_t)(uintptr_t)s;
        q->deferred.push_back(Deferred{ dst, s, bytes });
    }
    return xa_q_enqueue(st, XA_OP_COPY, &a, sizeof(a), 1, 0);
}

hipError_t xa_copy2d_to_mapped_async(void* st, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
    if (!xa_is_queue(st)) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)st);
    XaArgsCopy2D a = { (uint64_t)(uintptr_t)dst, (uint64_t)(uintptr_t)src, dpitch, spitch, width, height };
    return xa_q_enqueue(st, XA_OP_COPY2D, &a, sizeof(a), 1, 0);
}

hipError_t xa_fill_async(void* st, void* dst, int value, size_t bytes)
{
    if (!xa_is_queue(st)) return hipMemsetAsync(dst, value, bytes, (hipStream_t)st);
    XaArgsFill a = { (uint64_t)(uintptr_t)dst, bytes, (uint32_t)value };
    return xa_q_enqueue(st, XA_OP_FILL, &a, sizeof(a), 1, 0);
}

extern "C" void* x265amd_queue_acquire(void) { return xa_queue_acquire(); }
extern "C" void x265amd_queue_release(void* queue) { xa_queue_release(queue); }
/* Counters of the resident kernel since the last reset (valid while no queue is held: the workgroups write them when they leave).  out[0] commands,
 * [1] ticks (100 MHz) in command bodies, [2] in fences, [3] polling, [4] algorithmic bytes of all commands, [5] resident ticks summed over the workgroups,
 * [6] launches of k_job_server, [7] their duration by HIP events in microseconds (summed), [8] workgroups per launch, [9] reserved;
 * then per command kind k < 32: [10 + 3 k] count, [11 + 3 k] body ticks, [12 + 3 k] algorithmic bytes.  n: words available in out (>= 106 for everything). */
extern "C" int x265amd_queue_stats(uint64_t* out, int n, int reset)
{
    Server& S = server();
    std::lock_guard<std::mutex> g(S.m);
    if (!out || n < 10) return xa_fail(X265AMD_EINVAL, "queue_stats: arguments");
    for (int i = 0; i < n; i++) out[i] = 0;
    if (!S.numQueues) return X265AMD_OK;
    if (S.running && S.refs == 0) S.stop();
    for (int i = 0; i < S.numQueues; i++)
    {
        const XaRingHost& h = S.hosts[i];
        for (int k = 0; k < 32; k++)
        {
            const uint64_t cnt = k < 31 ? h.prof[2 * k] : 0, tk = k < 31 ? h.prof[2 * k + 1] : 0;
            if (k < XA_OP_COUNT) { out[0] += cnt; out[1] += tk; }
            out[4] += h.bytes[k];
            if (10 + 3 * k + 2 < n && k < XA_OP_COUNT) { out[10 + 3 * k] += cnt; out[11 + 3 * k] += tk; out[12 + 3 * k] += h.bytes[k]; }
        }
        out[2] += h.prof[63]; out[3] += h.prof[62]; out[5] += h.resident;
    }
    out[6] = S.launches; out[7] = (uint64_t)(S.kernelMs * 1000.0); out[8] = (uint64_t)S.numQueues;
    if (reset)
    {
        for (int i = 0; i < S.numQueues; i++) { memset((void*)S.hosts[i].prof, 0, sizeof(S.hosts[i].prof)); memset((void*)S.hosts[i].bytes, 0, sizeof(S.hosts[i].bytes)); S.hosts[i].resident = 0; S.hosts[i].cycles = 0;
                                                   memset((void*)S.hosts[i].stage, 0, sizeof(S.hosts[i].stage)); memset((void*)S.hosts[i].sized, 0, sizeof(S.hosts[i].sized));
                                                   memset((void*)S.hosts[i].nxn, 0, sizeof(S.hosts[i].nxn)); memset((void*)S.hosts[i].chain, 0, sizeof(S.hosts[i].chain)); }
        S.kernelMs = 0.0; S.launches = 0;
    }
    return X265AMD_OK;
}

extern "C" void x265amd_queue_profile_report(void) { Server& S = server(); std::lock_guard<std::mutex> g(S.m); if (S.numQueues) S.profile_report(true); }


/* ---- coherence probe (dbg): does a resident workgroup see what a copy engine / another kernel wrote into a buffer it has read before?
 * mode 0: hipMemcpy host -> device; mode 1: a kernel on another stream.  Returns the number of stale rounds out of `rounds`. ---- */
__global__ void k_probe_fill(uint32_t* p, int n, uint32_t v) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v + (uint32_t)i; }
extern "C" int x265amd_queue_coherence_probe(int rounds, int mode, int bytes)
{
    void* st = xa_queue_acquire();
    if (!st) return -1;
    void* dX = nullptr; void* mapped = nullptr;
    const int n = bytes / 4;
    if (hipMalloc(&dX, bytes) != hipSuccess || xa_mapped_alloc(&mapped, bytes, true) != hipSuccess) return -2;
    hipStream_t other; (void)hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
    std::vector<uint32_t> h(n);
    int stale = 0;
    for (int r = 0; r < rounds; r++)
    {
        const uint32_t v = 0x1000000u * (uint32_t)(r + 1);
        if (mode == 0) { for (int i = 0; i < n; i++) h[i] = v + (uint32_t)i; (void)hipMemcpy(dX, h.data(), bytes, hipMemcpyHostToDevice); }
        else { hipLaunchKernelGGL(k_probe_fill, dim3(64), dim3(256), 0, other, (uint32_t*)dX, n, v); (void)hipStreamSynchronize(other); }
        (void)xa_stream_fence(st, XA_CMD_ACQUIRE);
        XaArgsCopy a = { (uint64_t)(uintptr_t)mapped, (uint64_t)(uintptr_t)dX, (uint64_t)bytes, 1 };
        (void)xa_q_enqueue(st, XA_OP_COPY, &a, sizeof(a), 1, 0);
        (void)xa_stream_sync(st);
        const uint32_t* got = (const uint32_t*)mapped;
        bool bad = false;
        for (int i = 0; i < n; i++) if (got[i] != v + (uint32_t)i) { bad = true; break; }
        stale += bad;
    }
    (void)hipStreamDestroy(other);
    xa_queue_release(st);
    xa_mapped_free(mapped); (void)hipFree(dX);
    return stale;
}

/* ---- round-trip probe (dbg): nanoseconds per host -> queue -> host round trip.  mode 0: one empty signalling command; mode 1: a 64-byte fill, then the wait
 * (xa_stream_sync: what every block operation of the analysis pays); mode 2: three fills, then the wait ---- */
extern "C" double x265amd_queue_rtt_ns(int iters, int mode)
{
    void* st = xa_queue_acquire();
    if (!st || iters <= 0) return -1.0;
    void* d = nullptr;
    if (xa_scratch_alloc(&d, 4096) != hipSuccess) return -1.0;
    for (int warm = 0; warm < 2; warm++)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++)
        {
            if (mode == 0) { (void)xa_stream_fence(st, XA_CMD_SIGNAL); (void)xa_stream_sync(st); }
            else
            {
                for (int k = 0; k < (mode == 1 ? 1 : 3); k++) (void)xa_fill_async(st, (char*)d + 64 * k, i & 255, 64);
                (void)xa_stream_sync(st);
            }
        }
        const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / iters;
        if (warm) { xa_scratch_free(d); xa_queue_release(st); return ns; }
    }
    return -1.0;
}

/* ---- self test (tests/test_device_queue.py): copies, fills and rectangle copies through a queue against the same through a stream ---- */
extern "C" int x265amd_queue_selftest_wait_fault(void)
{
    void* keep = xa_queue_acquire();            /* held throughout: the resident kernel stays (a restart would clear every fault by itself) */
    void* a = keep ? xa_queue_acquire() : nullptr;
    void* b = a ? xa_queue_acquire() : nullptr;
    if (!a || !b) { if (a) xa_queue_release(a); if (keep) xa_queue_release(keep); return xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: no queue"); }
    void* d = nullptr;
    if (xa_scratch_alloc(&d, 4096) != hipSuccess) { xa_queue_release(a); xa_queue_release(b); xa_queue_release(keep); return xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: memory"); }
    const int idxA = as_queue(a)->idx;
    /* `a` waits for a count `b` never reaches */
    const XaArgsWait w = { (uint64_t)(uintptr_t)&as_queue(b)->rd->done, as_queue(b)->submitted + 1000 };
    int rc = X265AMD_OK;
    uint8_t back[64];
    if (q_push(as_queue(a), XA_OP_WAIT, 0, 0, &w, sizeof(w))) rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: push");
    if (rc == X265AMD_OK)
    {
        (void)xa_fill_async(a, d, 0x5a, 64);
        if (xa_stream_sync(a) == hipSuccess) rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: the wait behind a stalled queue did not fail");
    }
    xa_queue_release(a); xa_queue_release(b);
    if (rc == X265AMD_OK)
    {
        /* the same queue again (the first free one): a new owner, no fault */
        a = xa_queue_acquire();
        if (!a) rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: no queue the second time");
        else
        {
            if (as_queue(a)->idx != idxA) rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: another queue was handed out");
            else if (xa_fill_async(a, d, 0xa5, 64) != hipSuccess || xa_copy_async(a, back, d, 64, hipMemcpyDeviceToHost) != hipSuccess || xa_stream_fence(a, XA_CMD_RELEASE) != hipSuccess ||
                     xa_stream_sync(a) != hipSuccess)
                rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: the queue's next owner inherited the fault");
            else
                for (int i = 0; i < 64; i++) if (back[i] != 0xa5) rc = xa_fail(X265AMD_EHIP, "queue_selftest_wait_fault: the next owner's commands did not run");
            xa_queue_release(a);
        }
    }
    xa_scratch_free(d);
    xa_queue_release(keep);
    return rc;
}

extern "C" int x265amd_queue_selftest(int rounds, int numQueues)
{
    if (rounds <= 0 || numQueues <= 0) return xa_fail(X265AMD_EINVAL, "queue_selftest: arguments");
    std::vector<void*> qs;
    for (int i = 0; i < numQueues; i++)
    {
        void* q = xa_queue_acquire();
        if (!q) { for (void* p : qs) xa_queue_release(p); return xa_fail(X265AMD_EHIP, "queue_selftest: no queue"); }
        qs.push_back(q);
    }
    std::atomic<int> bad(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < numQueues; t++)
        pool.emplace_back([&, t] {
            void* st = qs[t];
            const size_t n = 4096 + 64 * t;
            void *dA = nullptr, *dB = nullptr, *mapped = nullptr;
            if (xa_scratch_alloc(&dA, n) != hipSuccess || xa_scratch_alloc(&dB, n) != hipSuccess || xa_mapped_alloc(&mapped, n, true) != hipSuccess) { bad++; return; }
            std::vector<uint8_t> src(n), back(n);
            for (int r = 0; r < rounds; r++)
            {
                for (size_t i = 0; i < n; i++) src[i] = (uint8_t)(i * 7 + r * 13 + t);
                /* the first command of a round sees whatever other agents wrote; the last one publishes */
                if (xa_stream_fence(st, XA_CMD_ACQUIRE) != hipSuccess) { bad++; break; }
                if (xa_copy_async(st, dA, src.data(), n, hipMemcpyHostToDevice) != hipSuccess) { bad++; break; }
                XaRects rc;
                memset(&rc, 0, sizeof(rc));
                rc.n = 1; rc.dst[0] = (uint64_t)(uintptr_t)dB; rc.src[0] = (uint64_t)(uintptr_t)dA; rc.dst_stride[0] = rc.src_stride[0] = 64; rc.w[0] = 64;
                rc.h[0] = (int16_t)(n / 64 / sizeof(x265amd_pixel));
                xa_copy_rects(st, rc);
                if (xa_fill_async(st, dA, r & 255, 128) != hipSuccess) { bad++; break; }
                /* same data three ways back: pageable (deferred), mapped in place */
                if (xa_copy_async(st, back.data(), dB, n, hipMemcpyDeviceToHost) != hipSuccess) { bad++; break; }
                if (xa_copy2d_to_mapped_async(st, mapped, 64, dB, 64, 64, n / 64) != hipSuccess) { bad++; break; }
                if (xa_stream_fence(st, XA_CMD_RELEASE) != hipSuccess || xa_stream_sync(st) != hipSuccess) { bad++; break; }
                const size_t copied = (size_t)rc.h[0] * 64 * sizeof(x265amd_pixel);
                if (memcmp(back.data(), src.data(), copied) != 0 || memcmp(mapped, src.data(), copied) != 0) { bad++; break; }
            }
            xa_scratch_free(dA); xa_scratch_free(dB); xa_mapped_free(mapped);
        });
    for (auto& th : pool) th.join();
    for (void* p : qs) xa_queue_release(p);
    return bad.load() ? xa_fail(X265AMD_EHIP, "queue_selftest: mismatch") : X265AMD_OK;
}
