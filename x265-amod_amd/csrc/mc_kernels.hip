/* Layer 3: inter prediction (include/x265amd.h, `x265amd_motion_compensation`).
 *
 * Device restatement of Predict::motionCompensation (reference: source/common/predict.cpp:77-243) and its helpers
 * predInterLuma/Chroma Pixel/Short (:245-408), addWeightBi/Uni (:411-577), Yuv::addAvg (yuv.cpp:189-211), and
 * CUData::clipMv (cudata.cpp:1915-1928), 4:2:0.  One 64-lane wavefront per PU, lanes strided over the output samples;
 * every sample recomputes its (separable) interpolation with the reference's intermediate roundings (the int16
 * row-extended horizontal pass of the hv cases included), so results are bit-exact.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

#include "mc_dev.h"
#include "xa_queue.h"

template<bool COST>
__global__ __launch_bounds__(64 * MC_WAVES) void k_motion_compensation(XaArgsMc a)
{
    const int ji = blockIdx.x * MC_WAVES + (threadIdx.x >> 6);
    if (ji >= a.n) return;
    wave_mc_job<COST>(a, ji, xa_lane());
}

extern "C" int x265amd_motion_compensation(void* stream, const uint64_t* d_planes, intptr_t stride, intptr_t cstride, int pic_w, int pic_h,
                                           const x265amd_mc_job* d_jobs, int n)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_planes || !d_jobs) return xa_fail(X265AMD_EINVAL, "x265amd_motion_compensation: bad arguments");
    const XaArgsMc qa = { d_planes, (long)stride, (long)cstride, pic_w, pic_h, d_jobs, n, nullptr, 0L, 0L, nullptr };
    hipError_t e;
    XA_LAUNCH(e, stream, XA_OP_MC, n, qa, k_motion_compensation<false>, dim3((n + MC_WAVES - 1) / MC_WAVES), dim3(64 * MC_WAVES), 0, qa);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_inter_cost(void* stream, const uint64_t* d_planes, intptr_t stride, intptr_t cstride, int pic_w, int pic_h,
                                  const x265amd_mc_job* d_jobs, int n, const uint64_t* d_fenc_planes, intptr_t fenc_stride, intptr_t fenc_cstride,
                                  uint32_t* d_cost)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_planes || !d_jobs || !d_fenc_planes || !d_cost) return xa_fail(X265AMD_EINVAL, "x265amd_inter_cost: bad arguments");
    const XaArgsMc qa = { d_planes, (long)stride, (long)cstride, pic_w, pic_h, d_jobs, n, d_fenc_planes, (long)fenc_stride, (long)fenc_cstride, d_cost };
    hipError_t e;
    XA_LAUNCH(e, stream, XA_OP_MC_COST, n, qa, k_motion_compensation<true>, dim3((n + MC_WAVES - 1) / MC_WAVES), dim3(64 * MC_WAVES), 0, qa);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
