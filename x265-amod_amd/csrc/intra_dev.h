/* Device code of the intra mode scan (see intra_kernels.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_INTRA_DEV_H
#define X265AMD_INTRA_DEV_H
#include "x265amd_dev.h"

#define IN_WAVES 4

struct IntraLds
{
    pixel ref[136], flt[136], refSw[136], fltSw[136];   /* neighbour sets and their left/above mirrored copies (horizontal modes) */
    int acc[35][4];                                     /* raw Hadamard sums per (mode, 16x16 group) */
    int16_t diff[64][64];                               /* per-lane 8x8 difference tile: diff[y*8+x][lane].  Rows are produced by a
                                                           rolled loop (runtime y), which in registers would turn into scratch memory
                                                           (measured: 2.4 GB of scratch traffic per 1080p launch) */
};


/* one prediction sample of `mode` at (y, x): intrapred.cpp:54-209 */
XA_DEV int in_pred_sample(const IntraLds& s, int mode, int N, int log2N, int dc, int y, int x)
{
    const int N2 = 2 * N;
    if (mode == 0)
    {
        const pixel* nb = (N >= 8) ? s.flt : s.ref;
        const pixel* above = nb + 1; const pixel* left = nb + N2 + 1;
        return ((N - 1 - x) * left[y] + (N - 1 - y) * above[x] + (x + 1) * above[N] + (y + 1) * left[N] + N) >> (log2N + 1);
    }
    if (mode == 1)
    {
        const pixel* above = s.ref + 1; const pixel* left = s.ref + N2 + 1;
        if (N <= 16)
        {
            if (x == 0 && y == 0) return (above[0] + left[0] + 2 * dc + 2) >> 2;
            if (y == 0) return (above[x] + 3 * dc + 2) >> 2;
            if (x == 0) return (left[y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    const bool filt = (xa_intra_filter_flags(mode) & N) != 0;
    const bool hor = mode < 18;
    const pixel* nb = hor ? (filt ? s.fltSw : s.refSw) : (filt ? s.flt : s.ref);
    int angOff = hor ? 10 - mode : mode - 26;
    int angle = xa_tbl.angle[8 + angOff];
    int invAngle = angle < 0 ? xa_tbl.invAngle[-angOff - 1] : 0;
    return hor ? ang_sample(nb, N, angle, invAngle, N <= 16, x, y) : ang_sample(nb, N, angle, invAngle, N <= 16, y, x);
}

/* one block of a job list on one wavefront: `s` is this wave's LDS */
XA_DEV void wave_intra_scan_job(const x265amd_intra_job* jobs, int ji, int32_t* out, pixel* nbOut, IntraLds& s, int lane)
{
    const x265amd_intra_job j = jobs[ji];
    const int log2N = j.log2_tr_size, N = 1 << log2N, N2 = 2 * N, units = N >> 2, L = 2 * units;
    const pixel* recon = reinterpret_cast<const pixel*>(j.recon);
    const pixel* fenc = reinterpret_cast<const pixel*>(j.fenc);
    const long rs = j.recon_stride;

    wave_intra_neighbours(recon, rs, j.avail, log2N, j.strong_smoothing != 0, N >= 8, s.ref, s.flt, lane);
    /* mirrored copies for the horizontal modes (intrapred.cpp:114-124) */
    for (int i = lane; i < N2; i += XA_WAVE)
    {
        s.refSw[1 + i] = s.ref[N2 + 1 + i]; s.refSw[N2 + 1 + i] = s.ref[1 + i];
        s.fltSw[1 + i] = s.flt[N2 + 1 + i]; s.fltSw[N2 + 1 + i] = s.flt[1 + i];
    }
    if (lane == 0) { s.refSw[0] = s.ref[0]; s.fltSw[0] = s.flt[0]; }
    for (int i = lane; i < 35 * 4; i += XA_WAVE) (&s.acc[0][0])[i] = 0;
    xa_wave_sync();
    if (nbOut)
    {
        pixel* o = nbOut + (size_t)ji * 2 * 129;
        for (int i = lane; i <= 4 * N; i += XA_WAVE) { o[i] = s.ref[i]; o[129 + i] = N >= 8 ? s.flt[i] : (pixel)0; }
    }
    /* DC value (intrapred.cpp:66-71) */
    int part = lane < N ? (int)s.ref[1 + lane] + (int)s.ref[N2 + 1 + lane] : 0;
    const int dc = (xa_wave_sum(part) + N) / N2;

    int32_t* res = out + (size_t)ji * 35;
    if (N == 4)     /* cu[4x4].sa8d = satd_4x4 (pixel.cpp:1171) */
    {
        if (lane < 35)
        {
            int d[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++)
                    d[y][x] = (int)fenc[y * j.fenc_stride + x] - in_pred_sample(s, lane, N, log2N, dc, y, x);
            int t[4][4], sum = 0;
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int s01 = d[y][0] + d[y][1], e01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], e23 = d[y][2] - d[y][3];
                t[y][0] = s01 + s23; t[y][1] = s01 - s23; t[y][2] = e01 + e23; t[y][3] = e01 - e23;
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int s01 = t[0][x] + t[1][x], e01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], e23 = t[2][x] - t[3][x];
                sum += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
            }
            res[lane] = sum >> 1;
        }
        return;
    }
    const int tpr = N >> 3, nt = tpr * tpr, items = 35 * nt, lgt = 2 * (log2N - 3);
    for (int it = lane; it < items; it += XA_WAVE)
    {
        int mode = it >> lgt, tile = it & (nt - 1);
        int ty = tile / tpr, tx = tile - ty * tpr;
#pragma unroll 1
        for (int i = 0; i < 64; i++)
        {
            int y = i >> 3, x = i & 7;
            s.diff[i][lane] = (int16_t)((int)fenc[(8 * ty + y) * j.fenc_stride + 8 * tx + x] - in_pred_sample(s, mode, N, log2N, dc, 8 * ty + y, 8 * tx + x));
        }
        int m[8][8];
#pragma unroll
        for (int y = 0; y < 8; y++)
#pragma unroll
            for (int x = 0; x < 8; x++) m[y][x] = s.diff[y * 8 + x][lane];
        int raw = xa_had8_abs_regs(m);
        if (N == 8) res[mode] = (raw + 2) >> 2;                              /* sa8d_8x8 (pixel.cpp:342-345) */
        else atomicAdd(&s.acc[mode][(ty >> 1) * (N >> 4) + (tx >> 1)], raw);  /* sa8d_16x16 groups (pixel.cpp:347-384) */
    }
    if (N == 8) return;
    xa_wave_sync();
    if (lane < 35)
    {
        int g = (N >> 4) * (N >> 4), tot = 0;
        for (int k = 0; k < g; k++) tot += (s.acc[lane][k] + 2) >> 2;
        res[lane] = tot;
    }
}

#endif
