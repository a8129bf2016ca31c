/* In-loop deblocking filter of a whole picture (include/x265amd.h: x265amd_deblock_picture).
 *
 * Device restatement of Deblock::deblockCU / getBoundaryStrength / edgeFilterLuma / edgeFilterChroma (reference:
 * source/common/deblock.cpp:72-495) and the sample filters pelFilterLuma (deblock.cpp:268-310), pelFilterLumaStrong_c,
 * pelFilterChroma_c (source/common/loopfilter.cpp:140-180), 4:2:0.  The reference recurses per CTU over its z-ordered CUData;
 * here the picture is described by one 12-byte record per 4x4 unit in raster order and filtered in the standard's two
 * picture-wide passes (all vertical edges, then all horizontal edges), one thread per 4-sample edge segment of the 8x8 grid:
 * edges touch at most three samples on either side, so the segments of one pass are independent.  Threads of a wavefront
 * take neighbouring segments along a sample row, so the rows they read and write are contiguous (HBM-bound kernel: the
 * picture is read once and written once per pass).
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

__device__ const uint8_t db_tc[54] = {      /* H.265 table 8-12 (deblock.cpp:497-509) */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2,
    2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__device__ const uint8_t db_beta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
    18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };

struct DbParams
{
    pixel* planes[3];
    long stride, cstride;
    int width, height;
    const x265amd_deblock_unit* units;
    int betaOffset, tcOffset, cqpOffset[2], bypassEnabled;
    int y4Begin, y4End;         /* the unit rows whose edges this launch filters (a band of CTU rows; the whole picture: 0 .. height / 4) */
    int xvBegin, xvEnd;         /* vertical edges: the unit columns x4 (edge between x4 - 1 and x4) with xvBegin <= x4 < xvEnd */
    int xhBegin, xhEnd;         /* horizontal edges: the unit columns xhBegin <= x4 < xhEnd */
};

__device__ const uint8_t db_cqp[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };     /* g_chromaScale[30..43], 4:2:0 (constants.cpp:346-350) */
XA_DEV int db_chroma_qp(int qp) { return qp < 30 ? qp : (qp < 44 ? (int)db_cqp[qp - 30] : min(qp - 6, 51)); }

XA_DEV bool db_far(int ax, int ay, int bx, int by) { return abs(ax - bx) >= 4 || abs(ay - by) >= 4; }

/* getBoundaryStrength (deblock.cpp:183-241) on the edge marks of deblockCU (:72-107) */
XA_DEV int db_strength(const x265amd_deblock_unit& q, const x265amd_deblock_unit& p, int tuEdge, int puEdge)
{
    const int bs = tuEdge ? 2 : (puEdge ? 1 : 0);
    if (!bs) return 0;
    if ((p.flags | q.flags) & X265AMD_DB_INTRA) return 2;
    if (bs > 1 && ((p.flags | q.flags) & X265AMD_DB_CBF)) return 1;
    const int rp0 = p.ref[0], rq0 = q.ref[0], rp1 = p.ref[1], rq1 = q.ref[1];
    const int p0x = rp0 >= 0 ? p.mv[0][0] : 0, p0y = rp0 >= 0 ? p.mv[0][1] : 0, q0x = rq0 >= 0 ? q.mv[0][0] : 0, q0y = rq0 >= 0 ? q.mv[0][1] : 0;
    const int p1x = rp1 >= 0 ? p.mv[1][0] : 0, p1y = rp1 >= 0 ? p.mv[1][1] : 0, q1x = rq1 >= 0 ? q.mv[1][0] : 0, q1y = rq1 >= 0 ? q.mv[1][1] : 0;
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))
    {
        if (rp0 != rp1)
        {
            if (rp0 == rq0) return (db_far(q0x, q0y, p0x, p0y) || db_far(q1x, q1y, p1x, p1y)) ? 1 : 0;
            return (db_far(q1x, q1y, p0x, p0y) || db_far(q0x, q0y, p1x, p1y)) ? 1 : 0;
        }
        return ((db_far(q0x, q0y, p0x, p0y) || db_far(q1x, q1y, p1x, p1y)) && (db_far(q1x, q1y, p0x, p0y) || db_far(q0x, q0y, p1x, p1y))) ? 1 : 0;
    }
    return 1;
}

/* one luma segment; m[i][k]: line i (0..3 along the edge), k = 0..7 across it (k = 3 | 4 straddle the edge) */
XA_DEV bool db_luma_segment(int m[4][8], int bs, int qp, int betaOffset, int tcOffset, int maskP, int maskQ)
{
    const int shift = XA_DEPTH - 8;
    const int beta = db_beta[xa_clip3(0, 51, qp + betaOffset)] << shift;
    const int dp0 = abs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = abs(m[0][4] - 2 * m[0][5] + m[0][6]);
    const int dp3 = abs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = abs(m[3][4] - 2 * m[3][5] + m[3][6]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return false;
    const int tc = db_tc[xa_clip3(0, 53, qp + 2 * (bs - 1) + tcOffset)] << shift;
    const bool s0 = (abs(m[0][0] - m[0][3]) + abs(m[0][7] - m[0][4]) < (beta >> 3)) && (abs(m[0][3] - m[0][4]) < ((tc * 5 + 1) >> 1));
    const bool s3 = (abs(m[3][0] - m[3][3]) + abs(m[3][7] - m[3][4]) < (beta >> 3)) && (abs(m[3][3] - m[3][4]) < ((tc * 5 + 1) >> 1));
    if (2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && s0 && s3)
    {
        const int tcP = (2 * tc) & maskP, tcQ = (2 * tc) & maskQ;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int m0 = m[i][0], m1 = m[i][1], m2 = m[i][2], m3 = m[i][3], m4 = m[i][4], m5 = m[i][5], m6 = m[i][6], m7 = m[i][7];
            m[i][1] = xa_clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1;
            m[i][2] = xa_clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2;
            m[i][3] = xa_clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3;
            m[i][4] = xa_clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4;
            m[i][5] = xa_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5;
            m[i][6] = xa_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6;
        }
        return true;
    }
    const int side = (beta + (beta >> 1)) >> 3;
    const int maskP1 = ((dp0 + dp3) < side ? -1 : 0) & maskP, maskQ1 = ((dq0 + dq3) < side ? -1 : 0) & maskQ;
    const int thrCut = tc * 10, tc2 = tc >> 1;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int m1 = m[i][1], m2 = m[i][2], m3 = m[i][3], m4 = m[i][4], m5 = m[i][5], m6 = m[i][6];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) < thrCut)
        {
            delta = xa_clip3(-tc, tc, delta);
            m[i][3] = xa_clip3(0, XA_PIXEL_MAX, m3 + (delta & maskP));
            m[i][4] = xa_clip3(0, XA_PIXEL_MAX, m4 - (delta & maskQ));
            if (maskP1) m[i][2] = xa_clip3(0, XA_PIXEL_MAX, m2 + xa_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
            if (maskQ1) m[i][5] = xa_clip3(0, XA_PIXEL_MAX, m5 + xa_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
        }
    }
    return true;
}

template<int DIR> XA_DEV void db_edge(const DbParams& P, int x4, int y4);

template<int DIR>
__global__ __launch_bounds__(256) void k_deblock(DbParams P)
{
    /* DIR 0: thread = (unit row y4, edge column ex), x4 = 2 ex; DIR 1: thread = (edge row ey, unit column x4), y4 = 2 ey */
    const int w4 = P.width >> 2;
    const int nx = DIR == 0 ? (w4 >> 1) : w4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = t / nx, ix = t - iy * nx;
    db_edge<DIR>(P, DIR == 0 ? 2 * ix : ix, P.y4Begin + (DIR == 0 ? iy : 2 * iy));
}

/* A unit of a few CTUs (what the filter thread of a picture coded in parallel launches per sweep: csrc/encoder_api.hip, filterRowsCols) by ONE workgroup: its vertical
 * edges, a barrier, its horizontal edges -- one launch instead of two, and no ordering between workgroups to wait for.  The same edges in the same order as the two
 * kernels above. */
__global__ __launch_bounds__(256) void k_deblock_unit(DbParams P)
{
    const int rows = P.y4End - P.y4Begin;
    const int vx0 = (P.xvBegin + 1) & ~1, nvx = P.xvEnd > vx0 ? (P.xvEnd - vx0 + 1) >> 1 : 0;         /* even unit columns in [xvBegin, xvEnd) */
    for (int t = threadIdx.x; t < nvx * rows; t += blockDim.x) { const int iy = t / nvx, ix = t - iy * nvx; db_edge<0>(P, vx0 + 2 * ix, P.y4Begin + iy); }
    __threadfence_block();
    __syncthreads();
    const int nhx = P.xhEnd - P.xhBegin, hrows = rows >> 1;
    for (int t = threadIdx.x; t < nhx * hrows; t += blockDim.x) { const int iy = t / nhx, ix = t - iy * nhx; db_edge<1>(P, P.xhBegin + ix, P.y4Begin + 2 * iy); }
}

template<int DIR> XA_DEV void db_edge(const DbParams& P, int x4, int y4)
{
    const int w4 = P.width >> 2, h4 = P.height >> 2;
    if (y4 >= h4 || y4 >= P.y4End || (DIR == 0 ? x4 == 0 : y4 == 0)) return;
    if (DIR == 0 ? (x4 < P.xvBegin || x4 >= P.xvEnd || x4 >= w4) : (x4 < P.xhBegin || x4 >= P.xhEnd || x4 >= w4)) return;
    const x265amd_deblock_unit q = P.units[y4 * w4 + x4];
    const x265amd_deblock_unit p = P.units[DIR == 0 ? y4 * w4 + x4 - 1 : (y4 - 1) * w4 + x4];
    const int bs = db_strength(q, p, q.flags & (DIR ? X265AMD_DB_TU_TOP : X265AMD_DB_TU_LEFT), q.flags & (DIR ? X265AMD_DB_PU_TOP : X265AMD_DB_PU_LEFT));
    if (!bs) return;
    int maskP = -1, maskQ = -1;
    if (P.bypassEnabled)
    {
        maskP = (p.flags & X265AMD_DB_BYPASS) ? 0 : -1; maskQ = (q.flags & X265AMD_DB_BYPASS) ? 0 : -1;
        if (!(maskP | maskQ)) return;
    }
    const int qp = (p.qp + q.qp + 1) >> 1;
    {
        pixel* src = P.planes[0] + (long)(4 * y4) * P.stride + 4 * x4;
        const long step = DIR ? 1 : P.stride, offset = DIR ? P.stride : 1;
        int m[4][8];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int k = 0; k < 8; k++) m[i][k] = src[i * step + (k - 4) * offset];
        if (db_luma_segment(m, bs, qp, P.betaOffset, P.tcOffset, maskP, maskQ))
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int k = 1; k < 7; k++) src[i * step + (k - 4) * offset] = (pixel)m[i][k];
        }
    }
    /* chroma: edges of the 8-sample chroma grid, one 4-sample segment per two luma units, strength of the first (deblock.cpp:455-493) */
    const int across = DIR ? y4 : x4, along = DIR ? x4 : y4;
    if (bs > 1 && !(across & 3) && !(along & 1))
    {
        const long step = DIR ? 1 : P.cstride, offset = DIR ? P.cstride : 1;
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            const int cq = db_chroma_qp(qp + P.cqpOffset[c]);
            const int tc = db_tc[xa_clip3(0, 53, cq + 2 + P.tcOffset)] << (XA_DEPTH - 8);
            pixel* src = P.planes[1 + c] + (long)(2 * y4) * P.cstride + 2 * x4;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int m2 = src[i * step - 2 * offset], m3 = src[i * step - offset], m4 = src[i * step], m5 = src[i * step + offset];
                const int delta = xa_clip3(-tc, tc, ((((m4 - m3) * 4) + m2 - m5 + 4) >> 3));
                src[i * step - offset] = (pixel)xa_clip3(0, XA_PIXEL_MAX, m3 + (delta & maskP));
                src[i * step] = (pixel)xa_clip3(0, XA_PIXEL_MAX, m4 - (delta & maskQ));
            }
        }
    }
}

extern "C" int x265amd_deblock_picture(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                                       int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                                       int cbQpOffset, int crQpOffset, int bypassEnabled, int passes)
{
    return x265amd_deblock_rows(stream, d_y, d_u, d_v, stride, cstride, width, height, d_units, betaOffsetDiv2, tcOffsetDiv2, cbQpOffset, crQpOffset, bypassEnabled, passes,
                                0, height >> 2);
}

/* The edges of the unit rows y4_begin .. y4_end - 1 (multiples of 2; a CTU row is 16 unit rows): what FrameFilter::processRow does for one CTU row
 * (reference: source/encoder/framefilter.cpp:559-590), as the two passes over that band.  Filtering band after band in row order gives the picture-wide
 * result: the horizontal edges of a band read what the vertical edges of this band and of the band above have left, and touch nothing below. */
extern "C" int x265amd_deblock_rows(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                                    int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                                    int cbQpOffset, int crQpOffset, int bypassEnabled, int passes, int y4_begin, int y4_end)
{
    return x265amd_deblock_rows_cols(stream, d_y, d_u, d_v, stride, cstride, width, height, d_units, betaOffsetDiv2, tcOffsetDiv2, cbQpOffset, crQpOffset, bypassEnabled, passes,
                                     y4_begin, y4_end, 0, (width + 63) >> 6);
}

/* The same band restricted to the CTU columns ctu_col_begin .. ctu_col_end - 1, for a CTU row that is filtered while its analysis still advances (the last rows
 * of a picture coded in parallel with the pictures that reference it, csrc/encoder_api.hip): the vertical edges right of the first column's left boundary up to
 * and INCLUDING the right boundary of the last column (the CTU to the right must be analysed), then the horizontal edges inside the columns.  Vertical edges
 * lie eight samples apart and touch three on either side, horizontal edges only read their own columns: chunk after chunk in column order gives the band's
 * result (the boundary edge of a chunk is filtered before the horizontal edges on either side of it read its samples). */
extern "C" int x265amd_deblock_rows_cols(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride,
                                         int width, int height, const x265amd_deblock_unit* d_units, int betaOffsetDiv2, int tcOffsetDiv2,
                                         int cbQpOffset, int crQpOffset, int bypassEnabled, int passes, int y4_begin, int y4_end, int ctu_col_begin, int ctu_col_end)
{
    if (ctu_col_begin < 0 || ctu_col_begin >= ctu_col_end || ctu_col_end > ((width + 63) >> 6)) return xa_fail(X265AMD_EINVAL, "x265amd_deblock_rows_cols: column range");
    if (!d_y || !d_u || !d_v || !d_units || width <= 0 || height <= 0 || (width & 7) || (height & 7) || y4_begin < 0 || y4_end > (height >> 2) || y4_begin >= y4_end || (y4_begin & 1) || (y4_end & 1))
        return xa_fail(X265AMD_EINVAL, "x265amd_deblock_picture: bad arguments (picture dimensions must be multiples of 8)");
    DbParams P;
    P.planes[0] = (pixel*)d_y; P.planes[1] = (pixel*)d_u; P.planes[2] = (pixel*)d_v;
    P.stride = (long)stride; P.cstride = (long)cstride; P.width = width; P.height = height; P.units = d_units;
    P.betaOffset = betaOffsetDiv2 * 2; P.tcOffset = tcOffsetDiv2 * 2; P.cqpOffset[0] = cbQpOffset; P.cqpOffset[1] = crQpOffset;
    P.bypassEnabled = bypassEnabled; P.y4Begin = y4_begin; P.y4End = y4_end;
    P.xvBegin = 16 * ctu_col_begin + 1; P.xvEnd = 16 * ctu_col_end + 1; P.xhBegin = 16 * ctu_col_begin; P.xhEnd = 16 * ctu_col_end;
    const int w4 = width >> 2, h4 = y4_end - y4_begin;
    static const bool fuse = !(getenv("X265AMD_DEBLOCK_FUSED") && atoi(getenv("X265AMD_DEBLOCK_FUSED")) == 0);
    if (fuse && passes == 3 && ctu_col_end - ctu_col_begin <= 8)
    {
        hipLaunchKernelGGL(k_deblock_unit, dim3(1), dim3(256), 0, (hipStream_t)stream, P);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
        return X265AMD_OK;
    }
    if (passes & 1)
    {
        const int n = (w4 >> 1) * h4;
        hipLaunchKernelGGL(k_deblock<0>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P);
    }
    if (passes & 2)
    {
        const int n = w4 * (h4 >> 1);
        hipLaunchKernelGGL(k_deblock<1>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
