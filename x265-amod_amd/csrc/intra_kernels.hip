/* Layer 3: intra neighbour set + 35-mode luma scan (include/x265amd.h, `x265amd_intra_scan`).
 *
 * Device restatement of Predict::fillReferenceSamples / initAdiPattern (reference: source/common/predict.cpp:600-649,
 * :736-877) and of the mode scan of Search::estIntraPredQT (source/encoder/search.cpp:1566-1613).  One 64-lane wavefront
 * per block: the 4N+1 neighbours are gathered and substituted in parallel (one lane per sample, the source of an
 * unavailable sample found with bit scans over the availability mask), smoothed, and the 35 x (N/8)^2 (mode, 8x8 tile)
 * pairs are spread over the lanes; each lane predicts its tile sample by sample and takes the 8x8 Hadamard in registers.
 * 16x16 groups are rounded once, as sa8d_16x16 does (pixel.cpp:347-357).  Integer arithmetic, bit-exact.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

#include "intra_dev.h"
#include "xa_queue.h"

__global__ __launch_bounds__(64 * IN_WAVES) void k_intra_scan(const x265amd_intra_job* jobs, int n, int32_t* out, pixel* nbOut)
{
    __shared__ IntraLds lds[IN_WAVES];
    const int lane = xa_lane(), wv = threadIdx.x >> 6;
    const int ji = blockIdx.x * IN_WAVES + wv;
    if (ji >= n) return;
    wave_intra_scan_job(jobs, ji, out, nbOut, lds[wv], lane);
}

/* one block per workgroup (block_intra_scan_job): small batches, where the time of a block matters more than the blocks per second */
__global__ __launch_bounds__(64 * IN_WG_WAVES) void k_intra_scan_wg(const x265amd_intra_job* jobs, int n, int32_t* out, pixel* nbOut)
{
    __shared__ IntraScanLds lds;
    const int ji = blockIdx.x;
    if (ji >= n) return;
    const x265amd_intra_job j = xa_ld_record(jobs + ji);
    block_intra_scan_job(j, out + (size_t)ji * 35, nbOut ? nbOut + (size_t)ji * 2 * 129 : nullptr, lds, threadIdx.x, 64 * IN_WG_WAVES);
}

extern "C" int x265amd_intra_scan(void* stream, const x265amd_intra_job* d_jobs, int n, int32_t* d_sa8d, x265amd_pixel* d_neighbours)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs || !d_sa8d) return xa_fail(X265AMD_EINVAL, "x265amd_intra_scan: bad arguments");
    const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)d_jobs, (uint64_t)(uintptr_t)d_sa8d, (uint64_t)(uintptr_t)d_neighbours, 0, n };
    hipError_t e;
    /* two forms of the same scan: a workgroup per block below X265AMD_SCAN_WG_MAX blocks (default 256), a wavefront per block above */
    static const int wgMax = getenv("X265AMD_SCAN_WG_MAX") ? atoi(getenv("X265AMD_SCAN_WG_MAX")) : 256;
    if (n <= wgMax)
        XA_LAUNCH(e, stream, XA_OP_INTRA_SCAN, n, qa, k_intra_scan_wg, dim3(n), dim3(64 * IN_WG_WAVES), 0, d_jobs, n, d_sa8d, d_neighbours);
    else
        XA_LAUNCH(e, stream, XA_OP_INTRA_SCAN, n, qa, k_intra_scan, dim3((n + IN_WAVES - 1) / IN_WAVES), dim3(64 * IN_WAVES), 0, d_jobs, n, d_sa8d, d_neighbours);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
