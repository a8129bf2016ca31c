/* Sample adaptive offset over a whole picture (include/x265amd.h: x265amd_sao_stats, x265amd_sao_apply).
 *
 * Device restatement of SAO::calcSaoStatsCTU with saoCuStatsBO/E0..E3 (reference: source/encoder/sao.cpp:735-917, :1762-1925;
 * sao-non-deblock off) and of SAO::generateLumaOffsets / generateChromaOffsets / applyPixelOffsets (sao.cpp:274-733), 4:2:0.
 * The reference walks CTUs in order and keeps the not-yet-offset samples it still needs in m_tmpU / m_tmpL; here the offset picture
 * is written to separate planes, so every sample is classified on the deblocked input directly and all samples are independent.
 * Statistics: one workgroup per (CTU, plane); every thread keeps the 4 x 5 edge-class sums and counts of its samples in registers
 * and the 32 band classes go through LDS atomics; one flush per thread at the end.  HBM-bound: the picture is read once (plus the
 * source picture for the statistics) and written once.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

struct SaoPlanes { const pixel* rec[3]; const pixel* fenc[3]; pixel* dst[3]; long stride, cstride; int width, height; int ctuRow0, ctuRows; int ctuCol0, ctuCols; int x0, x1; };      /* the CTU rows (and columns: statistics; luma sample columns x0 .. x1 - 1: offsets) of this launch */

XA_DEV int sao_sgn(int v) { return (v > 0) - (v < 0); }
XA_DEV int sao_class(int v, int a, int b)            /* SAO::s_eoTable[sign + sign + 2] (sao.cpp:65-72): {1, 2, 0, 3, 4} */
{
    const int e = sao_sgn(v - a) + sao_sgn(v - b) + 2;
    return e == 2 ? 0 : (e < 2 ? e + 1 : e);
}

/* One workgroup per (CTU, plane).  The CTU's deblocked samples with a halo of one sample and its source samples are staged in LDS first -- sixteen samples per lane
 * and load, rows aligned to sixteen (a CTU starts at a multiple of 32 samples in its plane) -- so that the nine neighbours of a sample are LDS reads instead of nine
 * one-sample loads from the picture (the first form of this kernel: 95 us for a luma CTU, all of it load latency; this one: see profiles/).  Then a lane per sample as
 * before: the four edge classes' sums and counts in registers, the band class into a histogram per wavefront. */
constexpr int kSaoTileW = 64 + 32;              /* 16 samples of margin on either side keep every row's loads aligned; only one of each is read */
__global__ __launch_bounds__(256) void k_sao_stats(SaoPlanes P, int32_t* count, int32_t* offsetOrg)
{
    __shared__ int sCnt[5 * 32], sOrg[5 * 32];
    __shared__ int sBand[4][2][32];
    __shared__ __attribute__((aligned(16))) pixel sRec[66 * kSaoTileW];
    __shared__ __attribute__((aligned(16))) pixel sSrc[64 * 64];
    const int ctuW = (P.width + 63) >> 6;
    const int plane = blockIdx.y, ctu = (P.ctuRow0 + (int)blockIdx.x / P.ctuCols) * ctuW + P.ctuCol0 + (int)blockIdx.x % P.ctuCols;
    const int cx = ctu % ctuW, cy = ctu / ctuW;
    const int sh = plane ? 1 : 0, po = plane ? 2 : 0;
    const long st = plane ? P.cstride : P.stride;
    const int picW = P.width >> sh, picH = P.height >> sh, lpelx = (cx * 64) >> sh, tpely = (cy * 64) >> sh;
    const int rpelx = min(lpelx + (64 >> sh), picW), bpely = min(tpely + (64 >> sh), picH);
    const int cw = rpelx - lpelx, ch = bpely - tpely;
    const pixel* r0 = P.rec[plane] + (long)tpely * st + lpelx;
    const pixel* f0 = P.fenc[plane] + (long)tpely * st + lpelx;
    for (int i = threadIdx.x; i < 5 * 32; i += blockDim.x) { sCnt[i] = 0; sOrg[i] = 0; }
    (&sBand[0][0][0])[threadIdx.x] = 0;            /* 4 x 2 x 32 = the workgroup's 256 lanes */
    {
        /* rows tpely - 1 .. tpely + ch of the deblocked plane, samples lpelx - 16 .. lpelx + cw + 15 as far as they lie inside the picture (what lies outside is never
         * classified, and a caller's planes need no margins), and the CTU's source samples: chunks of sixteen samples.  Whether a chunk is sixteen-byte aligned in memory
         * depends on the plane's origin: the access type says "unaligned", the hardware takes either */
        constexpr int kChunk = 16;
        struct __attribute__((packed, aligned(1))) Chunk { pixel v[kChunk]; };
        const int chunksPerRow = (cw + 2 * kChunk) / kChunk, rows = ch + 2;
        for (int i = threadIdx.x; i < chunksPerRow * rows; i += blockDim.x)
        {
            const int y = i / chunksPerRow, c = i - y * chunksPerRow;
            const int gy = tpely + y - 1, gx0 = lpelx + (c - 1) * kChunk;
            if (gy < 0 || gy >= picH) continue;                     /* a row outside the picture is never looked at (aboveUnavail, endYedge) */
            const pixel* src = r0 + (long)(y - 1) * st + (c - 1) * kChunk;
            if (gx0 >= 0 && gx0 + kChunk <= picW) *reinterpret_cast<Chunk*>(&sRec[y * kSaoTileW + c * kChunk]) = *reinterpret_cast<const Chunk*>(src);
            else
                for (int k = 0; k < kChunk; k++) if (gx0 + k >= 0 && gx0 + k < picW) sRec[y * kSaoTileW + c * kChunk + k] = src[k];      /* nor is a column outside it (x0e, endXedge) */
        }
        const int srcChunks = cw / kChunk;          /* cw is a multiple of 4 (widths are multiples of 8); the tail below */
        for (int i = threadIdx.x; i < srcChunks * ch; i += blockDim.x)
        {
            const int y = i / srcChunks, c = i - y * srcChunks;
            *reinterpret_cast<Chunk*>(&sSrc[y * 64 + c * kChunk]) = *reinterpret_cast<const Chunk*>(f0 + (long)y * st + c * kChunk);
        }
        const int tail = cw - srcChunks * kChunk;
        for (int i = threadIdx.x; i < tail * ch; i += blockDim.x)
        {
            const int y = i / tail, x = srcChunks * kChunk + (i - y * tail);
            sSrc[y * 64 + x] = f0[(long)y * st + x];
        }
    }
    __syncthreads();
    const bool atRight = rpelx == picW, atBottom = bpely == picH;
    const int aboveUnavail = !tpely;
    const int endXfull = atRight ? cw : cw - 5 + po, endXedge = atRight ? cw - 1 : cw - 5 + po;
    const int endYfull = atBottom ? ch : ch - 4 + po, endYedge = atBottom ? ch - 1 : ch - 4 + po;
    const int x0e = !lpelx;
    int cnt[4][5], org[4][5];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int c = 0; c < 5; c++) { cnt[t][c] = 0; org[t][c] = 0; }
    const int total = cw * ch, rounds = (total + (int)blockDim.x - 1) / (int)blockDim.x;
    constexpr int TS = kSaoTileW;
    for (int it = 0; it < rounds; it++)         /* uniform trip count: the band step uses wave-wide operations */
    {
        const int i = it * blockDim.x + threadIdx.x;
        const bool live = i < total;
        const int y = live ? i / cw : 0, x = live ? i - y * cw : cw;      /* x = cw lies outside every region */
        const pixel* r = &sRec[(y + 1) * TS + 16 + (live ? x : 0)];
        const int v = r[0], d = live ? (int)sSrc[y * 64 + x] - v : 0;
        /* band offset: a histogram per wavefront in LDS, a pair of LDS atomics per sample (lanes of a wavefront that hit the same band are serialised by the LDS, a
         * hundred cycles at worst; the loop over the wave's distinct bands this replaces cost six microseconds a round on noisy content) */
        if (x < endXfull && y < endYfull)
        {
            const int band = v >> (XA_DEPTH - 5);
            atomicAdd(&sBand[threadIdx.x >> 6][0][band], 1);
            atomicAdd(&sBand[threadIdx.x >> 6][1][band], d);
        }
        const bool inXe = x >= x0e && x < endXedge, inYe = y >= aboveUnavail && y < endYedge;
        int cls;
#define ACC(t, c) { cls = (c); _Pragma("unroll") for (int k = 0; k < 5; k++) if (cls == k) { cnt[t][k]++; org[t][k] += d; } }
        if (inXe && y < ch - 4 + po) ACC(0, sao_class(v, r[-1], r[1]))
        if (x < endXfull && inYe) ACC(1, sao_class(v, r[-TS], r[TS]))
        if (inXe && inYe)
        {
            ACC(2, sao_class(v, r[-TS - 1], r[TS + 1]))
            ACC(3, sao_class(v, r[-TS + 1], r[TS - 1]))
        }
#undef ACC
    }
    /* the edge classes: summed over the wavefront first (256 lanes adding to the same twenty words one by one is twenty microseconds of LDS serialisation) */
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int c = 0; c < 5; c++)
        {
            const int cs = xa_wave_sum(cnt[t][c]), os = xa_wave_sum(org[t][c]);
            if ((threadIdx.x & 63) == 0 && cs) { atomicAdd(&sCnt[t * 32 + c], cs); atomicAdd(&sOrg[t * 32 + c], os); }
        }
    __syncthreads();
    if (threadIdx.x < 32)
    {
        sCnt[4 * 32 + threadIdx.x] = sBand[0][0][threadIdx.x] + sBand[1][0][threadIdx.x] + sBand[2][0][threadIdx.x] + sBand[3][0][threadIdx.x];
        sOrg[4 * 32 + threadIdx.x] = sBand[0][1][threadIdx.x] + sBand[1][1][threadIdx.x] + sBand[2][1][threadIdx.x] + sBand[3][1][threadIdx.x];
    }
    __syncthreads();
    const size_t base = ((size_t)ctu * 3 + plane) * 5 * 32;
    for (int i = threadIdx.x; i < 5 * 32; i += blockDim.x) { count[base + i] = sCnt[i]; offsetOrg[base + i] = sOrg[i]; }
}

__global__ __launch_bounds__(256) void k_sao_apply(SaoPlanes P, const x265amd_sao_ctu* params)
{
    const int plane = blockIdx.z;
    const int sh = plane ? 1 : 0;
    const long st = plane ? P.cstride : P.stride;
    const int picW = P.width >> sh, picH = P.height >> sh;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = ((P.ctuRow0 * 64) >> sh) + blockIdx.y;
    if (x >= picW || y >= picH || y >= (((P.ctuRow0 + P.ctuRows) * 64) >> sh) || x < (P.x0 >> sh) || x >= (P.x1 >> sh)) return;
    const int ctuW = (P.width + 63) >> 6;
    const x265amd_sao_ctu p = params[((y << sh) >> 6) * ctuW + ((x << sh) >> 6)];
    const int type = p.type[plane ? 1 : 0];
    const pixel* s = P.rec[plane] + (long)y * st + x;
    const int v = s[0];
    int out = v;
    if (type == 4)
    {
        const int k = ((v >> (XA_DEPTH - 5)) - p.band_pos[plane]) & 31;
        if (k < 4) out = v + p.offset[plane][k];
    }
    else if (type >= 0)
    {
        const int dx = type == 1 ? 0 : (type == 3 ? 1 : -1), dy = type == 0 ? 0 : -1;
        const bool edge = (dx && (x == 0 || x == picW - 1)) || (dy && (y == 0 || y == picH - 1));
        if (!edge)
        {
            const int cls = sao_class(v, s[dy * st + dx], s[-dy * st - dx]);
            if (cls) out = v + p.offset[plane][cls - 1];
        }
    }
    P.dst[plane][(long)y * st + x] = xa_clip_pixel(out);
}

static int sao_fill(SaoPlanes& P, const uint64_t* rec, const uint64_t* fenc, const uint64_t* dst, intptr_t stride, intptr_t cstride, int width, int height)
{
    for (int c = 0; c < 3; c++)
    {
        P.rec[c] = (const pixel*)(uintptr_t)rec[c];
        P.fenc[c] = fenc ? (const pixel*)(uintptr_t)fenc[c] : nullptr;
        P.dst[c] = dst ? (pixel*)(uintptr_t)dst[c] : nullptr;
    }
    P.stride = (long)stride; P.cstride = (long)cstride; P.width = width; P.height = height; P.ctuRow0 = 0; P.ctuRows = (height + 63) >> 6;
    P.ctuCol0 = 0; P.ctuCols = (width + 63) >> 6; P.x0 = 0; P.x1 = width;
    return 0;
}

extern "C" int x265amd_sao_stats(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                                 int width, int height, int32_t* d_count, int32_t* d_offset_org)
{
    return x265amd_sao_stats_rows(stream, rec_planes, fenc_planes, stride, cstride, width, height, d_count, d_offset_org, 0, (height + 63) >> 6);
}

/* the statistics of CTU rows ctu_row_begin .. ctu_row_end - 1 (same arrays, indexed by the CTU's address in the picture).  A CTU's statistics leave out
 * the samples the deblocking of the CTUs to its right and below still changes (sao.cpp:760-776), so a row can be measured as soon as it is deblocked itself. */
extern "C" int x265amd_sao_stats_rows(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                                      int width, int height, int32_t* d_count, int32_t* d_offset_org, int ctu_row_begin, int ctu_row_end)
{
    return x265amd_sao_stats_rows_cols(stream, rec_planes, fenc_planes, stride, cstride, width, height, d_count, d_offset_org, ctu_row_begin, ctu_row_end, 0, (width + 63) >> 6);
}

/* ... of the CTU columns ctu_col_begin .. ctu_col_end - 1 of those rows (a CTU's statistics read nothing right of its own columns but what the deblocking of the
 * CTU to its right leaves alone) */
extern "C" int x265amd_sao_stats_rows_cols(void* stream, const uint64_t rec_planes[3], const uint64_t fenc_planes[3], intptr_t stride, intptr_t cstride,
                                           int width, int height, int32_t* d_count, int32_t* d_offset_org, int ctu_row_begin, int ctu_row_end, int ctu_col_begin, int ctu_col_end)
{
    if (ctu_col_begin < 0 || ctu_col_begin >= ctu_col_end || ctu_col_end > ((width + 63) >> 6)) return xa_fail(X265AMD_EINVAL, "x265amd_sao_stats: column range");
    if (!rec_planes || !fenc_planes || !d_count || !d_offset_org || width <= 0 || height <= 0 || (width & 7) || (height & 7) || ctu_row_begin < 0 || ctu_row_begin >= ctu_row_end ||
        ctu_row_end > ((height + 63) >> 6))
        return xa_fail(X265AMD_EINVAL, "x265amd_sao_stats: bad arguments");
    SaoPlanes P;
    sao_fill(P, rec_planes, fenc_planes, nullptr, stride, cstride, width, height);
    P.ctuRow0 = ctu_row_begin; P.ctuRows = ctu_row_end - ctu_row_begin; P.ctuCol0 = ctu_col_begin; P.ctuCols = ctu_col_end - ctu_col_begin;
    const int nctu = P.ctuCols * P.ctuRows;
    hipLaunchKernelGGL(k_sao_stats, dim3(nctu, 3), dim3(256), 0, (hipStream_t)stream, P, d_count, d_offset_org);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_sao_apply(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                                 int width, int height, const x265amd_sao_ctu* d_params)
{
    return x265amd_sao_apply_rows(stream, src_planes, dst_planes, stride, cstride, width, height, d_params, 0, (height + 63) >> 6);
}

/* the offset samples of CTU rows ctu_row_begin .. ctu_row_end - 1; the rows' samples are classified on src, whose row below must be deblocked already */
extern "C" int x265amd_sao_apply_rows(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                                      int width, int height, const x265amd_sao_ctu* d_params, int ctu_row_begin, int ctu_row_end)
{
    return x265amd_sao_apply_rows_cols(stream, src_planes, dst_planes, stride, cstride, width, height, d_params, ctu_row_begin, ctu_row_end, 0, width);
}

/* ... restricted to the luma sample columns x_begin .. x_end - 1 (even; chroma: half): a sample is classified on itself and its two neighbours along the class
 * direction, so the deblocked input must be final one sample beyond either end */
extern "C" int x265amd_sao_apply_rows_cols(void* stream, const uint64_t src_planes[3], const uint64_t dst_planes[3], intptr_t stride, intptr_t cstride,
                                           int width, int height, const x265amd_sao_ctu* d_params, int ctu_row_begin, int ctu_row_end, int x_begin, int x_end)
{
    if (x_begin < 0 || x_begin >= x_end || x_end > width || (x_begin & 1) || (x_end & 1)) return xa_fail(X265AMD_EINVAL, "x265amd_sao_apply: column range");
    if (!src_planes || !dst_planes || !d_params || width <= 0 || height <= 0 || (width & 7) || (height & 7) || ctu_row_begin < 0 || ctu_row_begin >= ctu_row_end ||
        ctu_row_end > ((height + 63) >> 6))
        return xa_fail(X265AMD_EINVAL, "x265amd_sao_apply: bad arguments");
    SaoPlanes P;
    sao_fill(P, src_planes, nullptr, dst_planes, stride, cstride, width, height);
    P.ctuRow0 = ctu_row_begin; P.ctuRows = ctu_row_end - ctu_row_begin; P.x0 = x_begin; P.x1 = x_end;
    const int lines = (height < ctu_row_end * 64 ? height : ctu_row_end * 64) - ctu_row_begin * 64;
    hipLaunchKernelGGL(k_sao_apply, dim3((width + 255) / 256, lines, 3), dim3(256), 0, (hipStream_t)stream, P, d_params);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
