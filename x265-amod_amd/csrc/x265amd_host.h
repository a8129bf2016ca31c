/* Host-side plumbing shared by the C-ABI translation units of libx265amd. */
#ifndef X265AMD_HOST_H
#define X265AMD_HOST_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/x265amd.h"

/* records the message for x265amd_last_error() and returns `code` */
int xa_fail(int code, const char* msg);

/* Device scratch for the host orchestrators: blocks are kept in size classes and handed out again instead of going back to hipFree
 * (hipMalloc / hipFree cost more than the kernels of a CU-sized step).  A block may be reused as soon as it is released: every user
 * enqueues its work on one stream, so reuse is stream-ordered.  x265amd_release_scratch() returns everything to the runtime. */
hipError_t xa_scratch_alloc(void** p, size_t bytes);
void xa_scratch_free(void* p);
/* While a thread holds a device job queue its freed blocks wait in a list of its own and serve its own next allocations: the queue's workgroup may
 * still own dirty L2 lines of a block (a resident workgroup writes back when told to, XA_CMD_RELEASE, not at a kernel boundary), and a block handed to
 * another queue on another XCD in that state is overwritten by the late write-back.  xa_scratch_local_end() returns the list to the shared pool; the
 * queue has been released (fenced + drained) by then. */
void xa_scratch_local_begin();
void xa_scratch_local_end();

/* Host-visible staging for the orchestrators' small job and result records: pinned host memory that the kernels read and write in place
 * (hipHostMalloc, mapped + coherent), pooled like the device scratch.  It replaces a hipMemcpyAsync per record array: the host fills the
 * jobs, launches, synchronises the stream and reads the results where the kernel left them.  Only for data a kernel touches once
 * (fine-grained host memory is not cached on the device).
 * Two pools, by direction: a block is either written by the host and read by the device (XaMapped: job records) or written by the device and
 * read by the host (XaMappedOut: results, levels).  A resident workgroup (device job queue) keeps the lines it has STORED to host memory in its L2,
 * and no acquire drops them: a block that served as a result array and came back from the pool as a job array was read as the old results
 * (measured, dbg/README.md).  Kernel boundaries hid that; keeping the directions apart removes it. */
hipError_t xa_mapped_alloc(void** p, size_t bytes, bool deviceWrites = false);
void xa_mapped_free(void* p);
#ifdef __cplusplus
struct XaMapped
{
    void* p = nullptr;
    bool deviceWrites;
    explicit XaMapped(bool deviceWrites_ = false) : deviceWrites(deviceWrites_) {}
    ~XaMapped() { xa_mapped_free(p); }
    hipError_t alloc(size_t bytes) { return xa_mapped_alloc(&p, bytes ? bytes : 16, deviceWrites); }
};
struct XaMappedOut : XaMapped { XaMappedOut() : XaMapped(true) {} };
#endif

/* ---- streams and device job queues ----
 * The `stream` argument of the batch entry points is either a hipStream_t or a device job queue (xa_queue.h): a queue handle has its low bit set.
 * A queue is a resident workgroup bound to the calling host thread; enqueueing a command is a store into device memory, waiting for it is a spin on
 * pinned host memory -- no runtime call either way.  The helpers below take either kind, so the orchestrators are written once:
 *   xa_stream_sync          hipStreamSynchronize / wait until the queue has run everything enqueued (then finish the deferred host copies)
 *   xa_copy_async           hipMemcpyAsync / a copy command; pageable host memory goes through the queue's pinned staging area
 *   xa_copy2d_to_mapped_async  device rows -> pinned host memory (hipMemcpy2DAsync / a copy command)
 *   xa_fill_async           hipMemsetAsync / a fill command
 *   xa_stream_fence         queue only: XA_CMD_ACQUIRE before reading what other workgroups, kernels or copies wrote, XA_CMD_RELEASE after writing what
 *                           they will read (the boundaries of a kernel do both; a resident workgroup has to be told)
 *   XA_LAUNCH               a kernel launch on a stream, or the same body as a command on a queue */
inline bool xa_is_queue(const void* st) { return ((uintptr_t)st & 1) != 0; }
bool xa_queues_enabled();
int xa_queues_hint(int n);              /* before the job server's first use: at least n queues (unless X265AMD_QUEUES says otherwise); returns the number in force */
void* xa_queue_acquire();               /* NULL when queues are off (X265AMD_QUEUES=0) or all are taken: use a stream then */
void xa_queue_release(void* st);
void* xa_queue_try_acquire();           /* a second queue for the holder of a first one, or NULL at once: never waits */
void xa_queue_release_helper(void* st);
void xa_queue_set_helper(void* st, void* helper);       /* the second queue rides on the first: whoever gets `st` finds it with xa_queue_helper */
void* xa_queue_helper(void* st);
int xa_extend_border_band_420(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride, int width, int height, int marginX, int marginY,
                              int y_begin, int y_end, int x_begin, int x_end, int left, int right);       /* csrc/plane_kernels.hip: the three planes' margins in one launch */
void* xa_queue_try_acquire_spare(int spare);          /* xa_queue_try_acquire that leaves at least `spare` queues free (-1: the default) */
void xa_queue_set_aux(void* st, void* aux);             /* a further queue riding on `st`, for its holder's own use (ctu_analysis.hip: searches started ahead) */
void* xa_queue_aux(void* st);
void xa_queue_log(void* st, int poc, int row);      /* X265AMD_QUEUE_LOG=poc,row: the command / wait timeline of that row goes to stderr when the queue is given back */
hipError_t xa_stream_sync(void* st);
hipError_t xa_stream_fence(void* st, int flags);
hipError_t xa_queue_follow(void* follower, void* leader);     /* two device job queues: what `leader` holds so far happens before what `follower` gets from now on; no host wait */
hipError_t xa_copy_async(void* st, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t xa_copy2d_to_mapped_async(void* st, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);
hipError_t xa_fill_async(void* st, void* dst, int value, size_t bytes);
hipError_t xa_q_enqueue(void* st, int op, const void* args, size_t argBytes, int count, int flags);
void xa_q_next_flags(void* st, int flags);
bool xa_q_free_mapped_later(void* st, void* p);   /* queue only (false otherwise): the pushed-record block goes back to the pool at the queue's next synchronisation */         /* queue only: the next command also carries these flags */
void xa_prof_dependency_wait(uint64_t ns);        /* X265AMD_QUEUE_PROF: time a row spent waiting for the row above (queue held, nobody working) */
#define XA_LAUNCH(ERR, st, OP, COUNT, ARGS, KERNEL, GRID, BLOCK, LDS, ...)                                                   \
    do {                                                                                                                     \
        if (xa_is_queue(st)) (ERR) = xa_q_enqueue((st), (OP), &(ARGS), sizeof(ARGS), (COUNT), 0);                           \
        else { hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, (hipStream_t)(st), __VA_ARGS__); (ERR) = hipGetLastError(); }   \
    } while (0)

/* up to four 2-D sample copies (device to device) as ONE launch: the three planes of a tile, a prediction / reconstruction pair, a winner's blocks.
 * Strides and sizes in samples. */
struct XaRects { uint64_t dst[4], src[4]; int16_t dst_stride[4], src_stride[4], w[4], h[4]; int32_t n; };         /* up to four; strides < 32768 samples */
void xa_copy_rects(void* st, const XaRects& r);

/* x265amd_analyse_frame with row hooks for pictures coded in parallel (csrc/ctu_analysis.hip; used by the encoder object): row_ready(ctx, row) says whether
 * the reference pictures have finished the rows CTU row `row` may read (1 yes, 0 not yet, -1 a reference picture failed; polled, no blocking, no side
 * effects), before_row(ctx, row) runs once before the row's first CTU, after_row(ctx, row) once the row is analysed and its reconstruction is in device
 * memory.  Rows finish in order. */
struct XaRowHooks
{
    void* ctx; int (*row_ready)(void* ctx, int row); void (*before_row)(void* ctx, int row); void (*after_row)(void* ctx, int row);
    /* optional (NULL: not used), per CTU: ctu_wait(ctx, row, col) BLOCKS until the reference pictures let CTU (row, col) start (they are published column by
     * column) and returns 0, or -1 when one of them failed -- it waits on counters (xa_wait_counter: a row task parks, cheaply, however many rows wait);
     * before_ctu / after_ctu around the CTU's analysis (after_ctu: its reconstruction is in device memory, its units and motion are in the maps);
     * ref_wait(ctx, pic, y_min, y_max, x_max): blocks until the samples of reference picture `pic` (index into the plane table) of the lines y_min .. y_max, right
     * to column x_max (picture coordinates; beyond the picture edge = its margin) may be read; 0, or -1 when the picture failed.  Called before every command
     * that reads reference samples (xa_ref_guard_*): the per-CTU gate covers ordinary vectors, this covers the long ones. */
    int (*ctu_wait)(void* ctx, int row, int col); void (*before_ctu)(void* ctx, int row, int col); void (*after_ctu)(void* ctx, int row, int col);
    int (*ref_wait)(void* ctx, int pic, int y_min, int y_max, int x_max);
    /* the picture's place in coding order + 1 (0: take the order of the calls): its rows' priority among the row tasks of all pictures in flight -- a picture
     * started ahead of its turn (an I picture: nothing to wait for) must not push aside the rows of the pictures everybody else waits for */
    uint64_t order;
    /* optional: what ctu_wait(ctx, row, col) has waited for -- the CTU rows *r0 .. *r1 of every reference picture final up to luma column *need (the picture
     * width: to the end, margin included).  The skip chain checks its vectors against it on the device (inter_chain_dev.h); NULL: nothing is known, every
     * vector goes through ref_wait on the host */
    void (*ctu_reach)(void* ctx, int row, int col, int* r0, int* r1, int* need);
};
/* device-resident mirrors of motion fields (csrc/ctu_analysis.hip; inter_chain_dev.h): one per host field, written by the host through the BAR as CUs are decided
 * and by the device's skip chain; NULL when device job queues are off */
void* xa_devmap_register(const x265amd_mv_unit* host, size_t units);
void xa_devmap_unregister(const x265amd_mv_unit* host);
void* xa_devmap_find(const x265amd_mv_unit* host);
void xa_devmap_push_rows(const x265amd_mv_unit* host, const x265amd_cu_unit* units, int w4, int y4a, int y4b);
/* the guards (csrc/ctu_analysis.hip): wait until the row task's hooks (if it has any) let the jobs' reference samples be read; 0, or -1 when a picture failed */
int xa_ref_guard_mc(const x265amd_mc_job* jobs, int n);
int xa_ref_guard_me(const x265amd_me_job* jobs, const int* pics, int n);
/* --limit-tu 3 / 4: CUData::m_refTuDepth of every CTU (21 values each: the CUs of depth 0, 1 and 2 in raster order inside a depth; -1: nothing decided there) -- this picture's,
 * written as CUs are decided and read from the CTUs to the left and above, and those of the first reference picture of each list (read at the co-located CTU, which is coded
 * by then: the CTU gate) */
struct XaTuRecs { int8_t* cur; const int8_t* ref[2]; };
int xa_analyse_frame(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                     const x265amd_slice_info* si, const x265amd_analysis_params* A, x265amd_cu_unit* units, x265amd_mv_unit* cur,
                     const x265amd_mv_unit* col, const uint8_t* ref_depth, const int8_t* ref_qp0, const uint64_t* h_planes, int num_pics,
                     intptr_t stride, intptr_t cstride, x265amd_cu_stat* cu_stat, int16_t* coeff_out, x265amd_ctu_result* results,
                     uint8_t* slice_data, size_t cap, uint32_t* substream_sizes, int* num_substreams, const XaRowHooks* hooks, const int8_t* cu_qp = nullptr,
                     const XaTuRecs* tu_recs = nullptr);
/* cu_qp (with si->use_dqp): Analysis::calculateQpforCuSize for every quantisation group, per CTU in z order -- 1 value (max_cu_dqp_depth 0) or 1 + 4 (depth 1): the QP of the
 * 64x64 CU, then of its four 32x32 CUs; the encoder object fills it from the rate control's QP and the adaptive quantisation / cuTree offsets of the picture */

/* x265amd_check_intra / x265amd_intra_in_inter with a working set the caller keeps between the CUs of one CTU (csrc/intra_rd.hip): *ws starts as NULL */
int xa_check_intra_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                      intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, int part_size, x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon,
                      x265amd_rd_result* out, int16_t* coeff_out, void** ws);
int xa_intra_in_inter_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                         intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, x265amd_rd_result* out,
                         int16_t* coeff_out, uint64_t* info, void** ws);
void xa_intra_ws_free(void* ws);
/* P / B pictures: starts the intra try of a CU (what xa_intra_in_inter_ws will compute, with the same contexts in `cu` and the same tiles) on the queue of the CU's
 * depth and returns 1, or 0 when that is not possible; the later xa_intra_in_inter_ws call for the same CU collects it, a try nobody collects is dropped. */
int xa_intra_in_inter_begin_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                               intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, uint64_t d_pred, uint64_t d_recon, void** ws);
/* before xa_check_intra_ws(.., part_size 0, ..) of an 8x8 CU that will be tried as NxN next: the NxN mode's tiles.  With a second queue on the stream
 * (xa_queue_helper) the NxN evaluation then starts beside the 2Nx2N one; the NxN call that follows collects it. */
void xa_intra_ws_hint_nxn(void** ws, uint64_t d_pred_nxn, uint64_t d_recon_nxn);
int xa_intra_quad8_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                      intptr_t stride, intptr_t cstride, int x, int y, int qp, const uint8_t* ctx, uint64_t frac, uint64_t split_recon, const uint64_t tilesN[2],
                      const uint64_t tiles2[2], x265amd_intra_cu8_result* results, void** ws, void (*between)(void*) = nullptr, void* between_ctx = nullptr, int lambda_qp = 0);
/* (lambda_qp: the QP of the lambdas when it is not `qp` -- QPs above 51, x265amd_rd_cu.reserved[0]; 0: it is) */
/* starts the 2Nx2N evaluation of a 16x16 CU on the stream's third queue (the helper's helper) and returns 1, or returns 0 when that is not possible (then nothing
 * has happened); < 0: error.  The caller evaluates the CU's sub-CUs, then calls xa_check_intra_ws(.., part_size 0, ..) for the same CU with the same tiles, which
 * collects the result.  The contexts in `cu` are those the later call passes. */
int xa_check_intra_begin_ws(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, const uint64_t* h_rec,
                            intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu, uint64_t d_pred, uint64_t d_recon, void** ws);

/* the skip and the residual measurement of a merge candidate together (csrc/inter_rd.hip) */
int xa_merge_rd(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, intptr_t stride, intptr_t cstride,
                const x265amd_rd_cu* cu, x265amd_cu_unit* skip_units, x265amd_cu_unit* merge_units, uint64_t d_pred, uint64_t d_recon_skip, uint64_t d_recon_merge,
                x265amd_rd_result* out_skip, x265amd_rd_result* out_merge, int16_t* coeff_out, int* merge_is_skip);

int xa_inter_residual_rd_lazy(void* stream, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, intptr_t stride,
                              intptr_t cstride, const x265amd_rd_cu* cu, x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes,
                              x265amd_rd_result* out, int16_t* coeff_out);

/* The process's GPU (csrc/device_queue.hip): xa_bind_device records the calling thread's current device (first call wins); xa_thread_device selects it on
 * the calling thread -- every thread the library creates calls it before its first HIP call (HIP's current device is per thread, default 0). */
void xa_bind_device();
void xa_thread_device();

/* X265AMD_TIMING: host time of a row task by phase (running time only: the clock stops while the task is parked).  XA_PHASE(k) charges the time since the
 * previous stamp of this task to phase k; the totals are printed per frame. */
enum { XA_PH_OTHER = 0, XA_PH_INTRA_SETUP, XA_PH_INTRA_SCAN, XA_PH_INTRA_CAND, XA_PH_INTRA_BITS, XA_PH_INTRA_CHROMA, XA_PH_INTRA_FINAL, XA_PH_PUSH, XA_PH_CABAC_CTU, XA_PH_ANALYZER,
       XA_PH_INTER_SEARCH, XA_PH_INTER_RD, XA_PH_MERGE, XA_PH_MERGE_CAND, XA_PH_MERGE_RD, XA_PH_RD_PLAN, XA_PH_RD_SKIPHOST, XA_PH_RD_WALK, XA_PH_COUNT };
void xa_phase(int k);
void xa_phase_report(void);

/* X265AMD_HOSTPROF=1: host CPU time by named scope (csrc/table_setup.hip: a table of (name, calls, nanoseconds of task running time), printed by
 * xa_hostprof_report).  XA_HOSTPROF("name") at the top of a scope; the clock is the row task's running time, so parked time does not count. */
struct XaHostProfScope { int id; uint64_t t0; explicit XaHostProfScope(int id_); ~XaHostProfScope(); };
int xa_hostprof_id(const char* name);
void xa_hostprof_report(void);
extern bool g_xaHostProf;
#define XA_HOSTPROF_CAT2(a, b) a##b
#define XA_HOSTPROF_CAT(a, b) XA_HOSTPROF_CAT2(a, b)
#define XA_HOSTPROF(name) static const int XA_HOSTPROF_CAT(xa_hp_id_, __LINE__) = xa_hostprof_id(name); XaHostProfScope XA_HOSTPROF_CAT(xa_hp_, __LINE__)(XA_HOSTPROF_CAT(xa_hp_id_, __LINE__))

/* device address of the centre (MVD 0) of x265amd_me_ctx's MV cost table for `qp` (BitCost::s_costs[qp]) */
const uint16_t* xa_me_device_mvcost(x265amd_me_ctx* ctx, int qp);
const float* xa_me_device_bitsize(x265amd_me_ctx* ctx);      /* BitCost::s_bitsizes on the device, index |d| */
const uint16_t* xa_me_device_tables(x265amd_me_ctx* ctx);    /* all MV cost tables (MeParams::tables) */

/* The reference's primitive slots cannot report failure (primitives.h:133-236), so a HIP error inside a
 * per-slot entry point is fatal: there is deliberately no CPU fallback. */
#define XA_HIP_FATAL(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            fprintf(stderr, "x265amd: fatal: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            abort();                                                                                         \
        }                                                                                                    \
    } while (0)

#define XA_HIP_CHECK(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e_));                           \
    } while (0)

#endif
