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
#include <string.h>
#include <vector>

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


/* ---------------- lookahead frame cost: CostEstimateGroup::estimateCUCost for every 8x8 block (slicetype.cpp:4077-4249) ----------------
 * with MotionEstimate::motionEstimate in its lowres form (motion.cpp:764-860 start, :879-987 hexagon, :1469-1525 sub-pel on the four half-pel
 * planes) and ReferencePlanes::lowresMC / lowresQPelCost (lowres.h:71-124).
 *
 * The blocks of a frame are chained: a block's MV predictor candidates are the MVs found for its right, lower, lower-left and lower-right
 * neighbours (reverse raster order).  One wavefront owns one block ROW and walks it right to left; it starts block x when the row below has
 * finished block max(x - 1, 0) (progress counters in global memory, release / acquire).  Rows are launched bottom row first, so the wavefronts a
 * row waits for are always dispatched before it.  Lane = sample of the 8x8 block; SAD / SATD through LDS. */
struct LowresCostParams
{
    const pixel* fenc; const pixel* ref[2][4];     /* sample (0,0) of the planes: fenc fpel; per list fpel, H, V, C */
    const pixel* refW[4];                          /* weighted copies of list 0's planes for its motion search (estimateCUCost's wfref0), or NULL */
    long stride;
    int widthInCU, heightInCU, bidir, doSearch[2], merange;
    const int32_t* intraCost;
    int16_t* mvs[2]; int32_t* mvCosts[2];       /* Lowres::lowresMvs / lowresMvCosts of (list, distance): read when !doSearch, written otherwise */
    uint16_t* lowresCosts; int32_t* bcost;
    const uint16_t* cost;                       /* MV cost table centre */
    int* progress;                              /* per row: the leftmost finished block (widthInCU = none yet) */
    int rowsPerSlice, numSlices;                /* cooperative slices (CostEstimateGroup::processTasks, slicetype.cpp:3957-3970): numSlices > 1 = block rows [k rowsPerSlice, (k + 1) rowsPerSlice)
                                                 * (the last slice to the bottom) are chains of their own: a slice's bottom row takes no predictors from the row below */
};

/* Round 5: a candidate block never touches LDS.  The wavefront's lanes are the 64 samples of the 8x8 block in TILE order (lanes 16 t .. 16 t + 15 = the 4x4 tile t,
 * raster inside it), the source sample sits in a register; a candidate is one (quarter-sample positions: two) global load per lane, its SAD a DPP sum, its SATD the
 * cross-lane 4x4 Hadamard of each 16-lane row (pixel.cpp:210-297: satd_8x4 twice = the four tiles' sums, each halved -- a tile's sum of absolute Hadamard coefficients is
 * even).  And the candidates of one step of the search are LOADED TOGETHER: the hexagon's six points, the three of an iteration, the eight of the square, the four of a
 * sub-sample step cost one memory latency per step instead of one per point (the kernel is latency bound: a wavefront per block row, a block a chain of some thirty
 * dependent evaluations).  Same values, same comparisons in the same order as before (tests/test_lowres.py). */
struct LrBlock
{
    const LowresCostParams* p;
    int list, lane, lx, ly;                     /* lx, ly: this lane's sample inside the block */
    int f;                                      /* the source sample */
    const pixel* const* planes;                 /* the planes lr_ld reads: the list's, or list 0's weighted copies while it is searched */
    long off;                                   /* blockOffset */
    int mvpx, mvpy;
};

XA_DEV int lr_mvcost(const LrBlock& b, int qx, int qy) { return (uint16_t)(b.p->cost[qx - b.mvpx] + b.p->cost[qy - b.mvpy]); }

/* lowresMC: this lane's sample of the candidate block of quarter-pel MV (qx, qy) */
XA_DEV int lr_ld(const LrBlock& b, int qx, int qy)
{
    const pixel* const* plane = b.planes;
    const long so = b.off + (long)b.ly * b.p->stride + b.lx;
    const int hpelA = (qy & 2) | ((qx & 2) >> 1);
    int v = plane[hpelA][so + (qx >> 2) + (long)(qy >> 2) * b.p->stride];
    if ((qx | qy) & 1)
    {
        const int qx2 = qx + (qx & 1), qy2 = qy + (qy & 1);
        const int hpelB = (qy2 & 2) | ((qx2 & 2) >> 1);
        v = (v + plane[hpelB][so + (qx2 >> 2) + (long)(qy2 >> 2) * b.p->stride] + 1) >> 1;          /* pixelavg_pp */
    }
    return v;
}
XA_DEV int lr_sad_v(const LrBlock& b, int v) { return xa_wave_sum(abs(b.f - v)); }
XA_DEV int lr_satd_v(const LrBlock& b, int v)
{
    const int h = xa_lane_had4x4(b.f - v, b.lane);
    const int s = xa_row16_sum(abs(h)) >> 1;
    return __builtin_amdgcn_readlane(s, 0) + __builtin_amdgcn_readlane(s, 16) + __builtin_amdgcn_readlane(s, 32) + __builtin_amdgcn_readlane(s, 48);
}
XA_DEV int lr_sad_q(const LrBlock& b, int qx, int qy) { return lr_sad_v(b, lr_ld(b, qx, qy)); }
XA_DEV int lr_satd_q(const LrBlock& b, int qx, int qy) { return lr_satd_v(b, lr_ld(b, qx, qy)); }
/* N full-pel candidates at once: COST_MV of each (SAD on the fpel plane + MV cost) */
template<int N> XA_DEV void lr_cost_f_n(const LrBlock& b, const int (&mx)[N], const int (&my)[N], int (&c)[N])
{
    int v[N];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = lr_ld(b, mx[k] * 4, my[k] * 4);
#pragma unroll
    for (int k = 0; k < N; k++) c[k] = lr_sad_v(b, v[k]) + lr_mvcost(b, mx[k] * 4, my[k] * 4);
}

__device__ const int8_t lr_hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
__device__ const uint8_t lr_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
__device__ const int8_t lr_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };

/* MotionEstimate::motionEstimate(ref, mvmin, mvmax, qmvp, 0, NULL, merange, outQMv) for a lowres reference, HEX search, subpelRefine 1 */
XA_DEV int lr_motion_estimate(LrBlock& b, int mnx, int mny, int mxx, int mxy, int qmvpx, int qmvpy, int merange, int& outx, int& outy)
{
    b.mvpx = qmvpx; b.mvpy = qmvpy;                                     /* setMVP */
    const int qmnx = mnx * 4, qmny = mny * 4, qmxx = mxx * 4, qmxy = mxy * 4;
    int pmx = min(max(qmvpx, qmnx), qmxx), pmy = min(max(qmvpy, qmny), qmxy);       /* clipped */
    const int bestprex = pmx, bestprey = pmy;
    int bx = (pmx + 2) >> 2, by = (pmy + 2) >> 2;                       /* roundToFPel */
    /* the predictor, its full-sample neighbour and the zero vector: loaded together (the latter two are only priced when the reference prices them) */
    const int vPre = lr_ld(b, pmx, pmy), vFpel = lr_ld(b, bx * 4, by * 4), vZero = lr_ld(b, 0, 0);
    const int bprecost = lr_sad_v(b, vPre);
    int bcost = bprecost;
    if ((pmx | pmy) & 3) bcost = lr_sad_v(b, vFpel) + lr_mvcost(b, bx * 4, by * 4);
    if (pmx | pmy)
    {
        const int c = lr_sad_v(b, vZero) + lr_mvcost(b, 0, 0);
        if (c < bcost) { bcost = c; bx = 0; by = max(min(0, mxy), mny); }
    }
    auto inRange = [&](int x, int y) { return x >= mnx && x <= mxx && y >= mny && y <= mxy; };
    /* hexagon search (motion.cpp:879-987) */
    {
        int c6[6];
        {
            const int hx[6] = { bx - 2, bx - 1, bx + 1, bx + 2, bx + 1, bx - 1 }, hy[6] = { by, by + 2, by + 2, by, by - 2, by - 2 };
            lr_cost_f_n<6>(b, hx, hy, c6);
        }
        int c0 = c6[0], c1 = c6[1], c2 = c6[2];
        int packed = bcost << 3;
        if (by >= mny && by <= mxy && (c0 << 3) + 2 < packed) packed = (c0 << 3) + 2;
        if (by + 2 >= mny && by + 2 <= mxy)
        {
            if ((c1 << 3) + 3 < packed) packed = (c1 << 3) + 3;
            if ((c2 << 3) + 4 < packed) packed = (c2 << 3) + 4;
        }
        c0 = c6[3]; c1 = c6[4]; c2 = c6[5];
        if (by >= mny && by <= mxy && (c0 << 3) + 5 < packed) packed = (c0 << 3) + 5;
        if (by - 2 >= mny && by - 2 <= mxy)
        {
            if ((c1 << 3) + 6 < packed) packed = (c1 << 3) + 6;
            if ((c2 << 3) + 7 < packed) packed = (c2 << 3) + 7;
        }
        if (packed & 7)
        {
            int dir = (packed & 7) - 2;
            if (by + lr_hex2[dir + 1][1] >= mny && by + lr_hex2[dir + 1][1] <= mxy)
            {
                bx += lr_hex2[dir + 1][0]; by += lr_hex2[dir + 1][1];
                for (int i = (merange >> 1) - 1; i > 0 && inRange(bx, by); i--)
                {
                    int cc[3];
                    {
                        const int hx[3] = { bx + lr_hex2[dir][0], bx + lr_hex2[dir + 1][0], bx + lr_hex2[dir + 2][0] }, hy[3] = { by + lr_hex2[dir][1], by + lr_hex2[dir + 1][1], by + lr_hex2[dir + 2][1] };
                        lr_cost_f_n<3>(b, hx, hy, cc);
                    }
                    packed &= ~7;
                    for (int k = 0; k < 3; k++)
                        if (by + lr_hex2[dir + k][1] >= mny && by + lr_hex2[dir + k][1] <= mxy && (cc[k] << 3) + k + 1 < packed) packed = (cc[k] << 3) + k + 1;
                    if (!(packed & 7)) break;
                    dir += (packed & 7) - 2;
                    dir = lr_mod6m1[dir + 1];
                    bx += lr_hex2[dir + 1][0]; by += lr_hex2[dir + 1][1];
                }
            }
        }
        bcost = packed >> 3;
        /* square refine */
        int dir = 0;
        const bool upOk = by - 1 >= mny && by - 1 <= mxy, dnOk = by + 1 >= mny && by + 1 <= mxy;
        int c8[8];
        {
            const int sx[8] = { bx, bx, bx - 1, bx + 1, bx - 1, bx - 1, bx + 1, bx + 1 }, sy[8] = { by - 1, by + 1, by, by, by - 1, by + 1, by - 1, by + 1 };
            lr_cost_f_n<8>(b, sx, sy, c8);
        }
        if (upOk && c8[0] < bcost) { bcost = c8[0]; dir = 1; }
        if (dnOk && c8[1] < bcost) { bcost = c8[1]; dir = 2; }
        if (c8[2] < bcost) { bcost = c8[2]; dir = 3; }
        if (c8[3] < bcost) { bcost = c8[3]; dir = 4; }
        if (upOk && c8[4] < bcost) { bcost = c8[4]; dir = 5; }
        if (dnOk && c8[5] < bcost) { bcost = c8[5]; dir = 6; }
        if (upOk && c8[6] < bcost) { bcost = c8[6]; dir = 7; }
        if (dnOk && c8[7] < bcost) { bcost = c8[7]; dir = 8; }
        bx += lr_square1[dir][0]; by += lr_square1[dir][1];
    }
    int qx, qy;
    if (bprecost < bcost) { qx = bestprex; qy = bestprey; bcost = bprecost; }
    else { qx = bx * 4; qy = by * 4; }
    if (!bcost) bcost = lr_mvcost(b, qx, qy);
    else
    {
        /* lowres sub-pel refinement (motion.cpp:1496-1525), workload[1]: 4 half-pel SADs, 4 quarter-pel SATDs -- each group of four loaded together */
        int bdir = 0;
        {
            int v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = lr_ld(b, qx + lr_square1[i + 1][0] * 2, qy + lr_square1[i + 1][1] * 2);
#pragma unroll
            for (int i = 1; i <= 4; i++)
            {
                const int tx = qx + lr_square1[i][0] * 2, ty = qy + lr_square1[i][1] * 2;
                if (ty < qmny || ty > qmxy) continue;
                const int c = lr_sad_v(b, v[i - 1]) + lr_mvcost(b, tx, ty);
                if (c < bcost) { bcost = c; bdir = i; }
            }
        }
        qx += lr_square1[bdir][0] * 2; qy += lr_square1[bdir][1] * 2;
        {
            const int v0 = lr_ld(b, qx, qy);
            int v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = lr_ld(b, qx + lr_square1[i + 1][0], qy + lr_square1[i + 1][1]);
            bcost = lr_satd_v(b, v0) + lr_mvcost(b, qx, qy);
            bdir = 0;
#pragma unroll
            for (int i = 1; i <= 4; i++)
            {
                const int tx = qx + lr_square1[i][0], ty = qy + lr_square1[i][1];
                if (ty < qmny || ty > qmxy) continue;
                const int c = lr_satd_v(b, v[i - 1]) + lr_mvcost(b, tx, ty);
                if (c < bcost) { bcost = c; bdir = i; }
            }
        }
        qx += lr_square1[bdir][0]; qy += lr_square1[bdir][1];
    }
    outx = qx; outy = qy;
    return bcost;
}

XA_DEV void lowres_cost_row(const LowresCostParams& p, int row, pixel* fencT, pixel* buf, pixel* buf2);
__global__ __launch_bounds__(64) void k_lowres_cost(LowresCostParams p)
{
    __shared__ pixel fencT[64];
    __shared__ pixel buf[64];
    __shared__ pixel buf2[64];
    lowres_cost_row(p, (int)blockIdx.x, fencT, buf, buf2);
}
/* many estimates as one launch: blockIdx.y = the estimate, blockIdx.x = its block row (bottom row first).  A row waits only for the row dispatched before it, so the
 * rows of an estimate need not all be resident: thousands of rows (hundreds of estimates) go into one grid and the device runs as many side by side as it holds */
__global__ __launch_bounds__(64) void k_lowres_cost_batch(const LowresCostParams* jobs)
{
    __shared__ pixel fencT[64];
    __shared__ pixel buf[64];
    __shared__ pixel buf2[64];
    __shared__ LowresCostParams sp;
    static_assert(sizeof(LowresCostParams) % 8 == 0, "copied in 64-bit words");
    for (int i = threadIdx.x; i < (int)(sizeof(LowresCostParams) / 8); i += 64) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(jobs + blockIdx.y)[i];
    __syncthreads();
    if ((int)blockIdx.x >= sp.heightInCU) return;
    lowres_cost_row(sp, (int)blockIdx.x, fencT, buf, buf2);
}
XA_DEV void lowres_cost_row(const LowresCostParams& p, int row, pixel* fencT, pixel* buf, pixel* buf2)
{
    const int lane = xa_lane();
    const int cuY = p.heightInCU - 1 - row;            /* bottom row first */
    const bool lastRow = cuY == p.heightInCU - 1 || (p.numSlices > 1 && (cuY + 1) % p.rowsPerSlice == 0 && (cuY + 1) / p.rowsPerSlice < p.numSlices);
    const int W = p.widthInCU;
    LrBlock b;
    b.p = &p; b.lane = lane;
    b.ly = ((lane >> 5) << 2) + ((lane >> 2) & 3); b.lx = (((lane >> 4) & 1) << 2) + (lane & 3);        /* tile order: lanes 16 t .. 16 t + 15 are the 4x4 tile t */
    (void)fencT; (void)buf; (void)buf2;
    for (int cuX = W - 1; cuX >= 0; cuX--)
    {
        /* (an estimate that only measures -- both motion fields exist -- has no order between its blocks: the neighbours' vectors are read, not made) */
        if (!lastRow && (p.doSearch[0] || p.doSearch[1]))
        {
            const int need = cuX > 0 ? cuX - 1 : 0;
            /* What crosses from one wavefront to another inside the launch is a handful of words per block -- the progress word and the neighbours' vectors --, and
             * every one of them is stored and loaded by relaxed atomics at agent scope (write-through / past the caches).  So NO cache maintenance is needed: the
             * acquire fence that stood here (invalidating the L2's lines of the planes for every block) and the release fence at the end of a block (writing the L2
             * back) were what made a batch of many estimates ten times slower per block than one estimate alone (round 5: section 4.25 of DESIGN.md). */
            if (lane == 0) while (__hip_atomic_load(&p.progress[cuY + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > need) __builtin_amdgcn_s_sleep(8);
            xa_wave_sync();
        }
        const int cuXY = cuX + cuY * W;
        b.off = 8L * cuX + 8L * cuY * p.stride;
        b.f = p.fenc[b.off + (long)b.ly * p.stride + b.lx];
        const int mnx = -cuX * 8 - 8, mny = -cuY * 8 - 8, mxx = (W - cuX - 1) * 8 + 8, mxy = (p.heightInCU - cuY - 1) * 8 + 8;
        int bcost = 1 << 28, listused = 0;          /* MotionEstimate::COST_MAX */
        int mvx[2] = { 0, 0 }, mvy[2] = { 0, 0 };
        for (int i = 0; i < 1 + p.bidir; i++)
        {
            b.list = i; b.planes = (i == 0 && p.refW[0]) ? p.refW : p.ref[i];
            int fencCost;
            if (!p.doSearch[i])
            {
                fencCost = p.mvCosts[i][cuXY];
                mvx[i] = p.mvs[i][2 * cuXY]; mvy[i] = p.mvs[i][2 * cuXY + 1];
                if (fencCost < bcost) { bcost = fencCost; listused = i + 1; }
                continue;
            }
            /* reverse-order MV prediction: the cheapest neighbour MV by SATD is the predictor */
            int numc = 0, mcx[4], mcy[4];
            const int16_t* mv = p.mvs[i];
            /* written by other wavefronts during this launch: read past the caches */
            auto add = [&](int idx) {
                const int v = __hip_atomic_load(reinterpret_cast<const int*>(mv) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mcx[numc] = (int16_t)(v & 0xFFFF); mcy[numc] = (int16_t)(v >> 16); numc++;
            };
            if (cuX < W - 1) add(cuXY + 1);
            if (!lastRow)
            {
                add(cuXY + W);
                if (cuX > 0) add(cuXY + W - 1);
                if (cuX < W - 1) add(cuXY + W + 1);
            }
            int mvpx = 0, mvpy = 0, skipCost = 0x7FFFFFFF;
            if (numc)
            {
                int mvpcost = 1 << 28;
                int vc[4];
                for (int k = 0; k < numc; k++) vc[k] = lr_ld(b, mcx[k], mcy[k]);         /* the (at most four) neighbour vectors' blocks loaded together */
                for (int k = 0; k < numc; k++)
                {
                    const int c = lr_satd_v(b, vc[k]);
                    if (c < mvpcost) { mvpcost = c; mvpx = mcx[k]; mvpy = mcy[k]; }
                    if (!(mvpx | mvpy) && p.bidir) skipCost = c;
                }
            }
            int ox, oy;
            fencCost = lr_motion_estimate(b, mnx, mny, mxx, mxy, mvpx, mvpy, p.merange, ox, oy);
            if (skipCost < 64 && skipCost < fencCost && p.bidir) { fencCost = skipCost; ox = 0; oy = 0; }
            mvx[i] = ox; mvy[i] = oy;
            if (lane == 0)
            {
                /* (the vector as ONE word, written through: the rows above and the next block of this row read it during the launch) */
                __hip_atomic_store(reinterpret_cast<int*>(p.mvs[i]) + cuXY, (int)(((uint32_t)ox & 0xFFFFu) | ((uint32_t)oy << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p.mvCosts[i][cuXY] = fencCost;
            }
            if (fencCost < bcost) { bcost = fencCost; listused = i + 1; }
        }
        if (p.bidir)
        {
            /* avg(l0-mv, l1-mv), then the co-located average */
            b.list = 0; b.planes = p.ref[0];
            const int a0 = lr_ld(b, mvx[0], mvy[0]);
            b.list = 1; b.planes = p.ref[1];
            const int a1 = lr_ld(b, mvx[1], mvy[1]);
            const long o = b.off + (long)b.ly * p.stride + b.lx;
            const int z0 = p.ref[0][0][o], z1 = p.ref[1][0][o];
            int bicost = lr_satd_v(b, (a0 + a1 + 1) >> 1);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bicost = lr_satd_v(b, (z0 + z1 + 1) >> 1);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bcost += 4;
        }
        else
        {
            bcost += 4;
            if (p.intraCost[cuXY] < bcost) { bcost = p.intraCost[cuXY]; listused = 0; }
        }
        if (lane == 0)
        {
            p.bcost[cuXY] = bcost;
            p.lowresCosts[cuXY] = (uint16_t)(min(bcost, 0x3FFF) | (listused << 14));
            /* INVARIANT of this hand-over (round 5, section 4.25 of DESIGN.md): every word another wavefront of the launch reads -- the row's vectors (mvs) and the progress
             * word -- is written and read with AGENT-SCOPE ATOMICS, which go to the L2 past the per-CU caches; bcost / lowresCosts / mvCosts above are plain stores because
             * nobody inside the launch reads them.  A new cross-row read MUST be such an atomic too: there is no release / acquire fence here to cover a plain one.
             * The wait below orders the vectors' stores in front of the progress word only because stores count in vmcnt on this ISA (gfx9 family; gfx10+ has vscnt). */
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "lowres_cost_row's hand-over relies on gfx950's vmcnt covering stores"
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            /* the vectors' stores have arrived before the progress word leaves: a wait, no cache write-back */
            __hip_atomic_store(&p.progress[cuY], cuX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int x265amd_lowres_frame_cost_batch(void* stream, x265amd_me_ctx* me, const x265amd_lowres_cost_job* jobs, int n, intptr_t stride, int width_in_cu, int height_in_cu)
{
    if (n <= 0) return X265AMD_OK;
    if (!me || !jobs || width_in_cu <= 0 || height_in_cu <= 0 || n > 65535) return xa_fail(X265AMD_EINVAL, "x265amd_lowres_frame_cost_batch: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    std::vector<LowresCostParams> ps((size_t)n);
    void* dParams = nullptr; void* dProgress = nullptr;
    if (xa_scratch_alloc(&dParams, sizeof(LowresCostParams) * n) != hipSuccess || xa_scratch_alloc(&dProgress, sizeof(int32_t) * (size_t)n * height_in_cu) != hipSuccess)
    { xa_scratch_free(dParams); xa_scratch_free(dProgress); return xa_fail(X265AMD_EHIP, "x265amd_lowres_frame_cost_batch: device allocation"); }
    for (int i = 0; i < n; i++)
    {
        const x265amd_lowres_cost_job& j = jobs[i];
        if (!j.d_fenc || !j.d_ref0[0] || !j.d_intra_cost || !j.d_mvs0 || !j.d_mv_costs0 || !j.d_lowres_costs || !j.d_bcost || (j.d_ref1[0] && (!j.d_mvs1 || !j.d_mv_costs1)))
        { xa_scratch_free(dParams); xa_scratch_free(dProgress); return xa_fail(X265AMD_EINVAL, "x265amd_lowres_frame_cost_batch: bad job"); }
        LowresCostParams& p = ps[(size_t)i];
        memset(&p, 0, sizeof(p));
        p.fenc = (const pixel*)j.d_fenc;
        for (int k = 0; k < 4; k++) { p.ref[0][k] = (const pixel*)j.d_ref0[k]; p.ref[1][k] = (const pixel*)j.d_ref1[k]; }
        p.stride = (long)stride; p.widthInCU = width_in_cu; p.heightInCU = height_in_cu; p.bidir = j.d_ref1[0] != nullptr;
        p.doSearch[0] = j.do_search0 != 0; p.doSearch[1] = p.bidir && j.do_search1;
        p.merange = 16;
        p.intraCost = j.d_intra_cost;
        p.mvs[0] = j.d_mvs0; p.mvCosts[0] = j.d_mv_costs0; p.mvs[1] = j.d_mvs1; p.mvCosts[1] = j.d_mv_costs1;
        p.lowresCosts = j.d_lowres_costs; p.bcost = j.d_bcost;
        p.cost = xa_me_device_mvcost(me, 12 + 6 * (XA_DEPTH - 8));
        p.progress = (int*)dProgress + (size_t)i * height_in_cu;
        p.rowsPerSlice = j.rows_per_slice; p.numSlices = j.rows_per_slice > 0 ? j.num_slices : 0;
        for (int k = 0; k < 4; k++) p.refW[k] = j.d_ref0w[0] ? (const pixel*)j.d_ref0w[k] : nullptr;
    }
    std::vector<int32_t> init((size_t)n * height_in_cu, width_in_cu);
    hipError_t e = hipMemcpyAsync(dProgress, init.data(), sizeof(int32_t) * init.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dParams, ps.data(), sizeof(LowresCostParams) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(k_lowres_cost_batch, dim3(height_in_cu, n), dim3(64), 0, st, (const LowresCostParams*)dParams);
        e = hipGetLastError();
    }
    /* the records and the rows' progress words are read until the kernel ends: the caller's wait for the results is ours too */
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    xa_scratch_free(dParams); xa_scratch_free(dProgress);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

/* x265amd_lowres_cost_sums: a workgroup per estimate */
struct LowresSumJob { const int32_t* bcost; const uint16_t* lc; };
__global__ __launch_bounds__(256) void k_lowres_cost_sums(const LowresSumJob* jobs, int W, int H, long long* out)
{
    __shared__ long long s_est[4];
    __shared__ int s_imb[4];
    const LowresSumJob j = jobs[blockIdx.x];
    const bool all = W <= 2 || H <= 2;
    long long est = 0; int imb = 0;
    for (int i = threadIdx.x; i < W * H; i += 256)
    {
        const int y = i / W, x = i - y * W;
        if (all || (x > 0 && x < W - 1 && y > 0 && y < H - 1)) { est += j.bcost[i]; imb += (j.lc[i] >> 14) == 0; }
    }
    est = xa_wave_sum(est); imb = xa_wave_sum(imb);
    if ((threadIdx.x & 63) == 0) { s_est[threadIdx.x >> 6] = est; s_imb[threadIdx.x >> 6] = imb; }
    __syncthreads();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = s_est[0] + s_est[1] + s_est[2] + s_est[3]; out[2 * blockIdx.x + 1] = s_imb[0] + s_imb[1] + s_imb[2] + s_imb[3]; }
}
extern "C" int x265amd_lowres_cost_sums(void* stream, const x265amd_lowres_cost_job* jobs, int n, int width_in_cu, int height_in_cu, int64_t* sums)
{
    if (n <= 0) return X265AMD_OK;
    if (!jobs || !sums || width_in_cu <= 0 || height_in_cu <= 0) return xa_fail(X265AMD_EINVAL, "x265amd_lowres_cost_sums: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    std::vector<LowresSumJob> js((size_t)n);
    for (int i = 0; i < n; i++)
    {
        if (!jobs[i].d_bcost || !jobs[i].d_lowres_costs) return xa_fail(X265AMD_EINVAL, "x265amd_lowres_cost_sums: bad job");
        js[(size_t)i].bcost = jobs[i].d_bcost; js[(size_t)i].lc = jobs[i].d_lowres_costs;
    }
    void* dJobs = nullptr; void* dOut = nullptr;
    if (xa_scratch_alloc(&dJobs, sizeof(LowresSumJob) * n) != hipSuccess || xa_scratch_alloc(&dOut, sizeof(long long) * 2 * n) != hipSuccess)
    { xa_scratch_free(dJobs); xa_scratch_free(dOut); return xa_fail(X265AMD_EHIP, "x265amd_lowres_cost_sums: device allocation"); }
    hipError_t e = hipMemcpyAsync(dJobs, js.data(), sizeof(LowresSumJob) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);          /* (js is pageable and goes out of scope) */
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(k_lowres_cost_sums, dim3(n), dim3(256), 0, st, (const LowresSumJob*)dJobs, width_in_cu, height_in_cu, (long long*)dOut);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(sums, dOut, sizeof(long long) * 2 * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    xa_scratch_free(dJobs); xa_scratch_free(dOut);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_lowres_frame_cost(void* stream, x265amd_me_ctx* me, const x265amd_pixel* d_fenc, const x265amd_pixel* const d_ref0[4],
                                         const x265amd_pixel* const d_ref1[4], intptr_t stride, int width_in_cu, int height_in_cu, int do_search0, int do_search1,
                                         const int32_t* d_intra_cost, int16_t* d_mvs0, int32_t* d_mv_costs0, int16_t* d_mvs1, int32_t* d_mv_costs1,
                                         uint16_t* d_lowres_costs, int32_t* d_bcost, int32_t* d_progress)
{
    if (!me || !d_fenc || !d_ref0 || !d_intra_cost || !d_mvs0 || !d_mv_costs0 || !d_lowres_costs || !d_bcost || !d_progress || width_in_cu <= 0 || height_in_cu <= 0 ||
        (d_ref1 && (!d_mvs1 || !d_mv_costs1)))
        return xa_fail(X265AMD_EINVAL, "x265amd_lowres_frame_cost: bad arguments");
    LowresCostParams p;
    memset(&p, 0, sizeof(p));
    p.fenc = (const pixel*)d_fenc;
    for (int k = 0; k < 4; k++) { p.ref[0][k] = (const pixel*)d_ref0[k]; p.ref[1][k] = d_ref1 ? (const pixel*)d_ref1[k] : nullptr; }
    p.stride = (long)stride; p.widthInCU = width_in_cu; p.heightInCU = height_in_cu; p.bidir = d_ref1 != nullptr;
    p.doSearch[0] = do_search0 != 0; p.doSearch[1] = d_ref1 && do_search1;
    p.merange = 16;                                     /* CostEstimateGroup::s_merange */
    p.intraCost = d_intra_cost;
    p.mvs[0] = d_mvs0; p.mvCosts[0] = d_mv_costs0; p.mvs[1] = d_mvs1; p.mvCosts[1] = d_mv_costs1;
    p.lowresCosts = d_lowres_costs; p.bcost = d_bcost;
    p.cost = xa_me_device_mvcost(me, 12 + 6 * (XA_DEPTH - 8));          /* X265_LOOKAHEAD_QP */
    p.progress = d_progress;
    hipStream_t st = (hipStream_t)stream;
    /* progress[row] = width_in_cu: nothing finished yet (0x01010101-style memset cannot express it: a tiny fill) */
    std::vector<int32_t> init((size_t)height_in_cu, width_in_cu);
    XA_HIP_CHECK(hipMemcpyAsync(d_progress, init.data(), sizeof(int32_t) * height_in_cu, hipMemcpyHostToDevice, st));
    XA_HIP_CHECK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_lowres_cost, dim3(height_in_cu), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}


/* ---------------- weighted prediction, the analysis' measurements ----------------
 * weightCostLuma (slicetype.cpp:826-858) and weightCost's luma branch (weightPrediction.cpp:172-218): the sum over the picture's 8x8 lowres blocks of
 * min(SATD(source block, reference block), intra cost of the block), the reference block optionally motion compensated with the lookahead's vectors (mcLuma,
 * weightPrediction.cpp:58-90: the vector clipped to the picture + 8 samples, Lowres::lowresMC) and optionally weighted (weight_pp_c, pixel.cpp:519-538).
 * blockIdx.x = the block, blockIdx.y = the candidate weight: every candidate of a decision in one launch (the reference tries them one after the other; a
 * candidate's sum does not depend on the others).  The sums are uint32 and wrap like the reference's. */
enum { kWeightCostParts = 48 };
struct WeightCostParams
{
    const pixel* fenc; const pixel* ref[4]; long stride; int width, height, blocksX;
    const int16_t* mvs; const int32_t* intraCost; const x265amd_weight_cand* cands; uint32_t* costs;
    int chroma, lowCuW, lowCuH;     /* chroma != 0: weightCost's 4:2:0 chroma branch on full-resolution chroma planes with mcChroma (weightPrediction.cpp:93-159, :205-208) */
};
XA_DEV void lowres_weight_cost_body(const WeightCostParams& p, pixel* fencT, pixel* refT)
{
    /* a few dozen wavefronts per candidate, each walking its share of the blocks and leaving one partial sum: no flood of one-block workgroups, no atomics */
    const int lane = threadIdx.x, lx = lane & 7, ly = lane >> 3;
    const x265amd_weight_cand w = p.cands[blockIdx.y];
    const int numBlocks = p.blocksX * ((p.height + 7) >> 3);
    uint32_t sum = 0;
    for (int cu = blockIdx.x; cu < numBlocks; cu += gridDim.x)
    {
        const int bx = cu % p.blocksX, by = cu / p.blocksX;
        const int x = bx * 8, y = by * 8;
        const long off = (long)y * p.stride + x;
        xa_wave_sync();
        fencT[lane] = p.fenc[off + (long)ly * p.stride + lx];
        int v;
        if (p.chroma)
        {
            /* mcChroma as the reference has it: the vector of lowres block (row = the block's first SAMPLE row, column = the block's index) for the blocks whose
             * sample position lies inside the lowres block grid, taken at a quarter for the position and an eighth for the fraction; a plain copy elsewhere */
            const pixel* a = p.ref[0] + off;
            if (p.mvs && x < p.lowCuW && y < p.lowCuH)
            {
                const int v32 = reinterpret_cast<const int*>(p.mvs)[y * p.lowCuW + bx];
                int qx = (int16_t)(v32 & 0xFFFF), qy = (int16_t)(v32 >> 16);
                qx = min(max(qx, (-x - 8) * 4), (p.width - x - 1 + 8) * 4);
                qy = min(max(qy, (-y - 8) * 4), (p.height - y - 1 + 8) * 4);
                a += (qx >> 2) + (long)(qy >> 2) * p.stride;
                v = mc_sample<4, false>(a + (long)ly * p.stride + lx, p.stride, qx & 7, qy & 7);
            }
            else v = a[(long)ly * p.stride + lx];
        }
        else if (p.mvs)
        {
            const int v32 = reinterpret_cast<const int*>(p.mvs)[cu];
            int qx = (int16_t)(v32 & 0xFFFF), qy = (int16_t)(v32 >> 16);
            qx = min(max(qx, (-x - 8) * 4), (p.width - x - 1 + 8) * 4);
            qy = min(max(qy, (-y - 8) * 4), (p.height - y - 1 + 8) * 4);
            const int hpelA = (qy & 2) | ((qx & 2) >> 1);
            const pixel* a = p.ref[hpelA] + off + (qx >> 2) + (long)(qy >> 2) * p.stride;
            v = a[(long)ly * p.stride + lx];
            if ((qx | qy) & 1)
            {
                const int qx2 = qx + (qx & 1), qy2 = qy + (qy & 1);
                const int hpelB = (qy2 & 2) | ((qx2 & 2) >> 1);
                const pixel* c = p.ref[hpelB] + off + (qx2 >> 2) + (long)(qy2 >> 2) * p.stride;
                v = (v + c[(long)ly * p.stride + lx] + 1) >> 1;
            }
        }
        else
            v = p.ref[0][off + (long)ly * p.stride + lx];
        if (w.present)
        {
            const int val = (int16_t)(v << (XA_IF_INTERNAL_PREC - XA_DEPTH));
            v = xa_clip3(0, XA_PIXEL_MAX, ((w.w0 * val + w.round) >> w.shift) + w.offset);
        }
        refT[lane] = (pixel)v;
        xa_wave_sync();
        int c = xa_wave_satd(refT, 8, fencT, 8, 8, 8, lane);
        if (p.intraCost) c = min(c, p.intraCost[cu]);
        sum += (uint32_t)c;
    }
    if (lane == 0) p.costs[blockIdx.y * gridDim.x + blockIdx.x] = sum;
}
__global__ __launch_bounds__(64) void k_lowres_weight_cost(WeightCostParams p)
{
    __shared__ pixel fencT[64];
    __shared__ pixel refT[64];
    lowres_weight_cost_body(p, fencT, refT);
}
/* many decisions' candidates as one launch: blockIdx.z = the decision (its own pictures, candidates and sums) */
__global__ __launch_bounds__(64) void k_lowres_weight_cost_many(const WeightCostParams* jobs)
{
    __shared__ pixel fencT[64];
    __shared__ pixel refT[64];
    __shared__ WeightCostParams sp;
    static_assert(sizeof(WeightCostParams) % 8 == 0, "copied in 64-bit words");
    for (int i = threadIdx.x; i < (int)(sizeof(WeightCostParams) / 8); i += 64) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(jobs + blockIdx.z)[i];
    __syncthreads();
    lowres_weight_cost_body(sp, fencT, refT);
}
/* debugging aid (X265AMD_WP_FLOOD=<mode>,<workgroups>): the dispatch pattern of this kernel's first form -- thousands of one-wave workgroups, mode 1: each ending in an atomicAdd on
 * one word, mode 2: each reading a little memory, mode 0: nothing at all -- beside whatever else runs; DESIGN.md section 8 (the intra chain's open sensitivity) */
__global__ __launch_bounds__(64) void k_flood(uint32_t* word, int mode, const uint32_t* src)
{
    __shared__ uint32_t t[64];
    t[threadIdx.x] = mode == 2 ? src[(blockIdx.x * 64 + threadIdx.x) & 1023] : threadIdx.x;
    __syncthreads();
    uint32_t v = t[63 - threadIdx.x];
    v = xa_wave_sum(v);
    if (mode == 1 && threadIdx.x == 0) atomicAdd(word + (blockIdx.y & 63), v);
}
extern "C" int x265amd_lowres_weight_costs(void* stream, const x265amd_pixel* d_fenc, const x265amd_pixel* const d_ref[4], const int16_t* d_mvs, const int32_t* d_intra_cost,
                                           intptr_t stride, int width, int height, const x265amd_weight_cand* cands, int n, uint32_t* costs)
{
    if (!d_fenc || !d_ref || !d_ref[0] || (d_mvs && (!d_ref[1] || !d_ref[2] || !d_ref[3])) || !cands || !costs || n <= 0 || n > 256 || width <= 0 || height <= 0 || (width & 7))
        return xa_fail(X265AMD_EINVAL, "x265amd_lowres_weight_costs: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    /* the candidates go to the device through a record the host writes in place, the partial sums come back in pinned memory the kernel writes: no copies.  The two blocks are
     * the calling thread's for good */
    static thread_local void* mC = nullptr; static thread_local void* mOut = nullptr;
    if (!mC && (xa_mapped_alloc(&mC, sizeof(x265amd_weight_cand) * 256, false) != hipSuccess || xa_mapped_alloc(&mOut, 4 * 256 * kWeightCostParts, true) != hipSuccess))
    { xa_mapped_free(mC); xa_mapped_free(mOut); mC = mOut = nullptr; return xa_fail(X265AMD_EHIP, "x265amd_lowres_weight_costs: allocation"); }
    memcpy(mC, cands, sizeof(x265amd_weight_cand) * n);
    WeightCostParams p;
    memset(&p, 0, sizeof(p));
    p.fenc = (const pixel*)d_fenc;
    for (int k = 0; k < 4; k++) p.ref[k] = (const pixel*)d_ref[k];
    p.stride = (long)stride; p.width = width; p.height = height; p.blocksX = width >> 3;
    p.mvs = d_mvs; p.intraCost = d_intra_cost; p.cands = (const x265amd_weight_cand*)mC; p.costs = (uint32_t*)mOut;
    const char* const flood = getenv("X265AMD_WP_FLOOD");          /* (read at every call: tests/test_encoder_api.py switches it on for one encode) */
    if (flood)
    {
        static thread_local void* dF = nullptr;
        int mode = 1, wgs = 11040;
        sscanf(flood, "%d,%d", &mode, &wgs);
        if (!dF && xa_scratch_alloc(&dF, 8192) != hipSuccess) dF = nullptr;
        if (dF) hipLaunchKernelGGL(k_flood, dim3(wgs / 46 > 0 ? wgs / 46 : 1, 46), dim3(64), 0, st, (uint32_t*)dF, mode, (const uint32_t*)dF + 1024);
    }
    hipLaunchKernelGGL(k_lowres_weight_cost, dim3(kWeightCostParts, n), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    const uint32_t* part = (const uint32_t*)mOut;
    for (int i = 0; i < n; i++) { uint32_t t = 0; for (int k = 0; k < kWeightCostParts; k++) t += part[i * kWeightCostParts + k]; costs[i] = t; }
    return X265AMD_OK;
}
extern "C" int x265amd_lowres_weight_costs_many(void* stream, const x265amd_weight_cost_job* jobs, int n, intptr_t stride, int width, int height, uint32_t* costs)
{
    if (n <= 0) return X265AMD_OK;
    if (!jobs || !costs || n > 65535 || width <= 0 || height <= 0 || (width & 7)) return xa_fail(X265AMD_EINVAL, "x265amd_lowres_weight_costs_many: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t bytesP = sizeof(WeightCostParams) * n, bytesC = sizeof(x265amd_weight_cand) * 2 * n, bytesO = sizeof(uint32_t) * 2 * kWeightCostParts * n;
    void* dev = nullptr;
    if (xa_scratch_alloc(&dev, bytesP + bytesC + bytesO) != hipSuccess) return xa_fail(X265AMD_EHIP, "x265amd_lowres_weight_costs_many: device allocation");
    std::vector<char> host(bytesP + bytesC);
    WeightCostParams* ps = reinterpret_cast<WeightCostParams*>(host.data());
    x265amd_weight_cand* cs = reinterpret_cast<x265amd_weight_cand*>(host.data() + bytesP);
    for (int i = 0; i < n; i++)
    {
        const x265amd_weight_cost_job& j = jobs[i];
        if (!j.d_fenc || !j.d_ref[0] || (j.d_mvs && (!j.d_ref[1] || !j.d_ref[2] || !j.d_ref[3]))) { xa_scratch_free(dev); return xa_fail(X265AMD_EINVAL, "x265amd_lowres_weight_costs_many: bad job"); }
        WeightCostParams& p = ps[i];
        memset(&p, 0, sizeof(p));
        p.fenc = (const pixel*)j.d_fenc;
        for (int k = 0; k < 4; k++) p.ref[k] = (const pixel*)j.d_ref[k];
        p.stride = (long)stride; p.width = width; p.height = height; p.blocksX = width >> 3;
        p.mvs = j.d_mvs; p.intraCost = j.d_intra_cost;
        p.cands = reinterpret_cast<const x265amd_weight_cand*>((char*)dev + bytesP) + 2 * i;
        p.costs = reinterpret_cast<uint32_t*>((char*)dev + bytesP + bytesC) + 2 * kWeightCostParts * i;
        cs[2 * i] = j.cands[0]; cs[2 * i + 1] = j.cands[1];
    }
    std::vector<uint32_t> part((size_t)2 * kWeightCostParts * n);
    hipError_t e = hipMemcpyAsync(dev, host.data(), host.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(k_lowres_weight_cost_many, dim3(kWeightCostParts, 2, n), dim3(64), 0, st, (const WeightCostParams*)dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), (char*)dev + bytesP + bytesC, bytesO, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    xa_scratch_free(dev);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    for (int i = 0; i < 2 * n; i++) { uint32_t t = 0; for (int k = 0; k < kWeightCostParts; k++) t += part[(size_t)i * kWeightCostParts + k]; costs[i] = t; }
    return X265AMD_OK;
}
/* the chroma planes' form (weightAnalyse's planes 1 and 2, weightPrediction.cpp:348-375): d_fenc / d_ref = sample (0, 0) of the current and the reference picture's SOURCE chroma
 * plane (margins extended), width x height = the plane clamped to whole 16x16 luma blocks, d_mvs = the lookahead's field of the luma analysis (or NULL), low_cu_w / low_cu_h
 * = the lowres picture in 8x8 blocks */
extern "C" int x265amd_chroma_weight_costs(void* stream, const x265amd_pixel* d_fenc, const x265amd_pixel* d_ref, const int16_t* d_mvs, intptr_t stride, int width, int height,
                                           int low_cu_w, int low_cu_h, const x265amd_weight_cand* cands, int n, uint32_t* costs)
{
    if (!d_fenc || !d_ref || !cands || !costs || n <= 0 || n > 256 || width <= 0 || height <= 0 || (width & 7) || (height & 7))
        return xa_fail(X265AMD_EINVAL, "x265amd_chroma_weight_costs: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    static thread_local void* mC = nullptr; static thread_local void* mOut = nullptr;
    if (!mC && (xa_mapped_alloc(&mC, sizeof(x265amd_weight_cand) * 256, false) != hipSuccess || xa_mapped_alloc(&mOut, 4 * 256 * kWeightCostParts, true) != hipSuccess))
    { xa_mapped_free(mC); xa_mapped_free(mOut); mC = mOut = nullptr; return xa_fail(X265AMD_EHIP, "x265amd_chroma_weight_costs: allocation"); }
    memcpy(mC, cands, sizeof(x265amd_weight_cand) * n);
    WeightCostParams p;
    memset(&p, 0, sizeof(p));
    p.fenc = (const pixel*)d_fenc; p.ref[0] = (const pixel*)d_ref;
    p.stride = (long)stride; p.width = width; p.height = height; p.blocksX = width >> 3;
    p.mvs = d_mvs; p.cands = (const x265amd_weight_cand*)mC; p.costs = (uint32_t*)mOut;
    p.chroma = 1; p.lowCuW = low_cu_w; p.lowCuH = low_cu_h;
    hipLaunchKernelGGL(k_lowres_weight_cost, dim3(kWeightCostParts, n), dim3(64), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    const uint32_t* part = (const uint32_t*)mOut;
    for (int i = 0; i < n; i++) { uint32_t t = 0; for (int k = 0; k < kWeightCostParts; k++) t += part[i * kWeightCostParts + k]; costs[i] = t; }
    return X265AMD_OK;
}
/* weight_pp_c over a whole padded buffer (LookaheadTLD::weightsAnalyse's weighted copies of the four lowres planes, slicetype.cpp:971-975) */
__global__ void k_weight_buffer(const pixel* src, pixel* dst, size_t n, int w0, int round, int shift, int offset)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int val = (int16_t)((int)src[i] << (XA_IF_INTERNAL_PREC - XA_DEPTH));
    dst[i] = (pixel)xa_clip3(0, XA_PIXEL_MAX, ((w0 * val + round) >> shift) + offset);
}
extern "C" int x265amd_weight_buffer(void* stream, const x265amd_pixel* d_src, x265amd_pixel* d_dst, size_t count, int w0, int round, int shift, int offset)
{
    if (!d_src || !d_dst || !count) return xa_fail(X265AMD_EINVAL, "x265amd_weight_buffer: bad arguments");
    hipLaunchKernelGGL(k_weight_buffer, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const pixel*)d_src, (pixel*)d_dst, count, w0, round, shift, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}


/* ---------------- adaptive quantisation, the data-parallel half: LookaheadTLD::acEnergyCu for every quantisation group ----------------
 * (slicetype.cpp:48-92, :264-283; cu[].var = pixel_var, pixel.cpp:720-737): AC energy of the luma block and the two 4:2:0 chroma blocks,
 * plus the frame's running sums Lowres::wp_sum / wp_ssd that weighted prediction analysis reads.  One wavefront per group; HBM-bound (every
 * source sample is read once: algorithmic bytes = 1.5 x width x height x sizeof(pixel)).  The double-precision mapping of the energies to QP
 * offsets (calcAdaptiveQuantFrame) is host code and not part of this entry point. */
__global__ __launch_bounds__(64) void k_aq_energy(const pixel* y, const pixel* u, const pixel* v, long stride, long cstride, int blocksW, int numBlocks, int qg,
                                                  uint32_t* energy, unsigned long long* wp)
{
    const int lane = xa_lane(), blk = blockIdx.x;
    if (blk >= numBlocks) return;
    const int bx = (blk % blocksW) * qg, by = (blk / blocksW) * qg;
    uint32_t total = 0;
    for (int plane = 0; plane < 3; plane++)
    {
        const int n = plane ? qg >> 1 : qg, log2n = plane ? (qg == 16 ? 3 : 2) : (qg == 16 ? 4 : 3);
        const pixel* src = plane == 0 ? y + (long)by * stride + bx : (plane == 1 ? u : v) + (long)(by >> 1) * cstride + (bx >> 1);
        const long st = plane ? cstride : stride;
        uint32_t sum = 0, sqr = 0;
        for (int i = lane; i < n * n; i += XA_WAVE)
        {
            const uint32_t p = src[(long)(i >> log2n) * st + (i & (n - 1))];
            sum += p; sqr += p * p;
        }
        sum = xa_wave_sum(sum); sqr = xa_wave_sum(sqr);
        total += sqr - (uint32_t)(((unsigned long long)sum * sum) >> (2 * log2n));
        /* (partial sums, added up by k_aq_reduce: six atomics per block on six words were half a millisecond per 1080p picture -- and a flood of atomics beside the encoder) */
        if (lane == 0) { wp[(size_t)blk * 6 + plane] = (unsigned long long)sum; wp[(size_t)blk * 6 + 3 + plane] = (unsigned long long)sqr; }
    }
    if (lane == 0) energy[blk] = total;
}

__global__ __launch_bounds__(256) void k_aq_reduce(const unsigned long long* part, int numBlocks, unsigned long long* wp)
{
    __shared__ unsigned long long t[256];
    unsigned long long a = 0;
    for (int i = threadIdx.x; i < numBlocks; i += 256) a += part[(size_t)i * 6 + blockIdx.x];
    t[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) t[threadIdx.x] += t[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) wp[blockIdx.x] = t[0];
}
extern "C" int x265amd_aq_energy(void* stream, const uint64_t planes[3], intptr_t stride, intptr_t cstride, int width, int height, int qg_size,
                                 uint32_t* d_energy, uint64_t* d_wp)
{
    if (!planes || !planes[0] || !planes[1] || !planes[2] || !d_energy || !d_wp || width <= 0 || height <= 0 || (qg_size != 16 && qg_size != 8))
        return xa_fail(X265AMD_EINVAL, "x265amd_aq_energy: bad arguments");
    const int bw = (width + qg_size - 1) / qg_size, bh = (height + qg_size - 1) / qg_size;
    hipStream_t st = (hipStream_t)stream;
    /* the blocks' partial sums: a block of the calling thread's, kept (it grows with the picture size) */
    static thread_local void* dPart = nullptr; static thread_local size_t partBytes = 0;
    const size_t need = (size_t)bw * bh * 6 * sizeof(unsigned long long);
    if (need > partBytes)
    {
        if (dPart) { (void)hipStreamSynchronize(st); xa_scratch_free(dPart); dPart = nullptr; partBytes = 0; }
        if (xa_scratch_alloc(&dPart, need) != hipSuccess) return xa_fail(X265AMD_EHIP, "x265amd_aq_energy: device allocation");
        partBytes = need;
    }
    hipLaunchKernelGGL(k_aq_energy, dim3(bw * bh), dim3(64), 0, st, (const pixel*)(uintptr_t)planes[0], (const pixel*)(uintptr_t)planes[1], (const pixel*)(uintptr_t)planes[2],
                       (long)stride, (long)cstride, bw, bw * bh, qg_size, d_energy, (unsigned long long*)dPart);
    hipLaunchKernelGGL(k_aq_reduce, dim3(6), dim3(256), 0, st, (const unsigned long long*)dPart, bw * bh, (unsigned long long*)d_wp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
