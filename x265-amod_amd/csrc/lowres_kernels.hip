/* Lookahead lowres pipeline, first stage (include/x265amd.h: x265amd_lowres_init, x265amd_lowres_intra_costs).
 *
 *  - Lowres::init (reference: source/common/lowres.cpp:337-403): the half-resolution picture and its three half-pel companions
 *    (frame_init_lowres_core, source/common/pixel.cpp:605-628), each with extended borders;
 *  - LookaheadTLD::lowresIntraEstimate (source/encoder/slicetype.cpp:715-824): for every 8x8 block of the lowres picture the cheapest intra
 *    prediction by SATD -- DC, planar, every fifth angular mode, then +-2 and +-1 around the best angle -- plus the fixed penalties.
 *
 * k_lowres_init is HBM-bound streaming: a thread produces one sample of each of the four planes from a 3x3 neighbourhood of source samples
 * (9 reads, L2-served overlaps, 4 writes); algorithmic bytes per lowres sample = 4 source + 4 written = 8 x sizeof(pixel).
 * k_lowres_intra: one 64-lane wavefront per block (one lane per sample), neighbours and predictions in LDS, 12 predictions + SATDs in the
 * reference's order (the refinement depends on the coarse scan's winner).  Blocks are independent: one launch covers the picture. */
#include "x265amd_dev.h"
#include "x265amd_host.h"

__global__ __launch_bounds__(256) void k_lowres_init(const pixel* src, long srcStride, pixel* d0, pixel* dh, pixel* dv, pixel* dc, long dstStride, int width, int height)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= width || y >= height) return;
    const pixel* s0 = src + (long)(2 * y) * srcStride + 2 * x;
    const pixel* s1 = s0 + srcStride;
    const pixel* s2 = s1 + srcStride;
    const int a0 = s0[0], a1 = s0[1], a2 = s0[2], b0 = s1[0], b1 = s1[1], b2 = s1[2], c0 = s2[0], c1 = s2[1], c2 = s2[2];
    /* slower than a plain bilinear filter, but it is what the reference computes: pixel.cpp:615 */
    auto filt = [](int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; };
    const long o = (long)y * dstStride + x;
    d0[o] = (pixel)filt(a0, b0, a1, b1);
    dh[o] = (pixel)filt(a1, b1, a2, b2);
    dv[o] = (pixel)filt(b0, c0, b1, c1);
    dc[o] = (pixel)filt(b1, c1, b2, c2);
}

extern "C" int x265amd_lowres_init(void* stream, const x265amd_pixel* d_src, intptr_t src_stride, int width, int height, x265amd_pixel* const d_planes[4],
                                   intptr_t stride, int marginX, int marginY)
{
    if (!d_src || !d_planes || !d_planes[0] || !d_planes[1] || !d_planes[2] || !d_planes[3] || width <= 0 || height <= 0 || marginX < 0 || marginY < 0)
        return xa_fail(X265AMD_EINVAL, "x265amd_lowres_init: bad arguments");
    hipLaunchKernelGGL(k_lowres_init, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream, (const pixel*)d_src, (long)src_stride,
                       (pixel*)d_planes[0], (pixel*)d_planes[1], (pixel*)d_planes[2], (pixel*)d_planes[3], (long)stride, width, height);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    for (int k = 0; k < 4; k++)
    {
        const int rc = x265amd_extend_pic_border(stream, d_planes[k], stride, width, height, marginX, marginY);
        if (rc != X265AMD_OK) return rc;
    }
    return X265AMD_OK;
}

/* one wavefront per 8x8 block */
__global__ __launch_bounds__(64) void k_lowres_intra(const pixel* plane, long stride, int widthInCU, int heightInCU, int penalty, int32_t* costOut, uint8_t* modeOut)
{
    __shared__ pixel nb[2][33];         /* [0] samples, [1] filtered: above-left, 16 above, 16 left */
    __shared__ pixel swapped[33];
    __shared__ pixel pred[64];
    __shared__ pixel fenc[64];
    const int lane = xa_lane(), cu = blockIdx.x;
    if (cu >= widthInCU * heightInCU) return;
    const int cuX = cu % widthInCU, cuY = cu / widthInCU;
    const pixel* pix = plane + (long)(8 * cuY) * stride + 8 * cuX;
    fenc[lane] = pix[(long)(lane >> 3) * stride + (lane & 7)];
    const pixel* corner = pix - stride - 1;
    if (lane < 17) nb[0][lane] = corner[lane];                              /* above-left + top (2N) */
    if (lane >= 17 && lane < 33) nb[0][lane] = corner[(long)(lane - 16) * stride];     /* left (2N): rows 1..16 below the corner */
    xa_wave_sync();
    wave_intra_filter(nb[0], nb[1], 8, lane);
    xa_wave_sync();
    auto tryMode = [&](int mode, const pixel* s, int bFilter) {
        wave_intra_pred(s, swapped, 1, mode, bFilter, pred, 8, false, lane);
        xa_wave_sync();
        const int c = xa_wave_satd(fenc, 8, pred, 8, 8, 8, lane);
        xa_wave_sync();
        return c;
    };
    int icost = 0x7FFFFFFF, ilow = 0;           /* me.COST_MAX is not reachable by an 8x8 SATD */
    int c = tryMode(1, nb[0], 1);               /* DC_IDX, filtered edges (cuSize <= 16) */
    if (c < icost) { icost = c; ilow = 1; }
    c = tryMode(0, nb[1], 0);                   /* PLANAR_IDX on the filtered neighbours */
    if (c < icost) { icost = c; ilow = 0; }
    int acost = 0x7FFFFFFF, alow = 4;
    for (int mode = 5; mode < 35; mode += 5)
    {
        c = tryMode(mode, nb[(xa_intra_filter_flags(mode) & 8) != 0], 1);
        if (c < acost) { acost = c; alow = mode; }
    }
    for (int dist = 2; dist >= 1; dist--)
    {
        const int minusmode = alow - dist, plusmode = alow + dist;
        c = tryMode(minusmode, nb[(xa_intra_filter_flags(minusmode) & 8) != 0], 1);
        if (c < acost) { acost = c; alow = minusmode; }
        c = tryMode(plusmode, nb[(xa_intra_filter_flags(plusmode) & 8) != 0], 1);
        if (c < acost) { acost = c; alow = plusmode; }
    }
    if (acost < icost) { icost = acost; ilow = alow; }
    if (lane == 0) { costOut[cu] = icost + penalty; modeOut[cu] = (uint8_t)ilow; }
}

extern "C" int x265amd_lowres_intra_costs(void* stream, const x265amd_pixel* d_plane, intptr_t stride, int width_in_cu, int height_in_cu, int lambda,
                                          int32_t* d_cost, uint8_t* d_mode)
{
    if (!d_plane || !d_cost || !d_mode || width_in_cu <= 0 || height_in_cu <= 0) return xa_fail(X265AMD_EINVAL, "x265amd_lowres_intra_costs: bad arguments");
    const int penalty = 5 * lambda + 4;         /* intraPenalty + lowresPenalty (slicetype.cpp:722-724) */
    hipLaunchKernelGGL(k_lowres_intra, dim3(width_in_cu * height_in_cu), dim3(64), 0, (hipStream_t)stream, (const pixel*)d_plane, (long)stride, width_in_cu, height_in_cu,
                       penalty, d_cost, d_mode);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
