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

/* Host-visible staging for the orchestrators' small job and result records: pinned host memory that the kernels read and write in place
 * (hipHostMalloc, mapped + coherent), pooled like the device scratch.  It replaces a hipMemcpyAsync per record array: the host fills the
 * jobs, launches, synchronises the stream and reads the results where the kernel left them.  Only for data a kernel touches once
 * (fine-grained host memory is not cached on the device). */
hipError_t xa_mapped_alloc(void** p, size_t bytes);
void xa_mapped_free(void* p);
#ifdef __cplusplus
struct XaMapped
{
    void* p = nullptr;
    ~XaMapped() { xa_mapped_free(p); }
    hipError_t alloc(size_t bytes) { return xa_mapped_alloc(&p, bytes ? bytes : 16); }
};
#endif

/* up to three 2-D sample copies (device to device) as ONE launch: the three planes of a tile, or a prediction / reconstruction pair.
 * Strides and sizes in samples. */
struct XaRects { uint64_t dst[3], src[3]; int32_t dst_stride[3], src_stride[3], w[3], h[3]; int32_t n; };
void xa_copy_rects(hipStream_t st, const XaRects& r);

/* device address of the centre (MVD 0) of x265amd_me_ctx's MV cost table for `qp` (BitCost::s_costs[qp]) */
const uint16_t* xa_me_device_mvcost(x265amd_me_ctx* ctx, int qp);

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
