/* Device code of the intra mode scan (see intra_kernels.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_INTRA_DEV_H
#define X265AMD_INTRA_DEV_H
#include "x265amd_dev.h"
#include <stddef.h>

#define IN_WAVES 4
#define IN_WG_WAVES 8          /* workgroup of the one-block-per-workgroup form */

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
    const x265amd_intra_job j = xa_ld_record(jobs + ji);
    const int log2N = j.log2_tr_size, N = 1 << log2N, N2 = 2 * N, units = N >> 2, L = 2 * units;
    if (lane == 0) XA_BYTES((4 * N + 1 + N * N) * sizeof(pixel) + 35 * 4);
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

/* =========================================================================================================
 * The same scan for ONE block by a whole workgroup: the form the device job queues use, where a command carries one block (the next CU of a
 * CTU row) and its latency is what counts.  One wavefront per (mode, 8x8 tile), one lane per sample: the mode is uniform across the wave (no
 * divergence between planar / DC / the two angular orientations), the 8x8 Hadamard runs across the lanes (six butterfly stages), the 35 modes of
 * an 8x8 block are five rounds of eight wavefronts.  4x4 blocks: sixteen lanes per mode.
 * ======================================================================================================= */
struct IntraScanLds
{
    pixel ref[136], flt[136], refSw[136], fltSw[136];
    int acc[35][4];
    int dc;
    pixel fenc[32 * 32];
};

XA_DEV void block_intra_scan_job(const x265amd_intra_job& j, int32_t* res, pixel* nbo, IntraScanLds& s, int tid, int nthr)
{
    static_assert(offsetof(IntraScanLds, acc) == offsetof(IntraLds, acc), "in_pred_sample reads the neighbour arrays at the head of either layout");
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    const int log2N = j.log2_tr_size, N = 1 << log2N, N2 = 2 * N;
    if (tid == 0) XA_BYTES((4 * N + 1 + N * N) * sizeof(pixel) + 35 * 4);
    const pixel* recon = reinterpret_cast<const pixel*>(j.recon);
    const pixel* fenc = reinterpret_cast<const pixel*>(j.fenc);
    __syncthreads();            /* the previous block's readers are done with the LDS */
    for (int i = tid; i < N * N; i += nthr) s.fenc[i] = fenc[(i >> log2N) * j.fenc_stride + (i & (N - 1))];
    if (wv == 0)
    {
        wave_intra_neighbours(recon, j.recon_stride, j.avail, log2N, j.strong_smoothing != 0, N >= 8, s.ref, s.flt, lane);
        for (int i = lane; i < N2; i += XA_WAVE)
        {
            s.refSw[1 + i] = s.ref[N2 + 1 + i]; s.refSw[N2 + 1 + i] = s.ref[1 + i];
            s.fltSw[1 + i] = s.flt[N2 + 1 + i]; s.fltSw[N2 + 1 + i] = s.flt[1 + i];
        }
        if (lane == 0) { s.refSw[0] = s.ref[0]; s.fltSw[0] = s.flt[0]; }
        for (int i = lane; i < 35 * 4; i += XA_WAVE) (&s.acc[0][0])[i] = 0;
        const int part = lane < N ? (int)s.ref[1 + lane] + (int)s.ref[N2 + 1 + lane] : 0;
        const int dcv = (xa_wave_sum(part) + N) / N2;
        if (lane == 0) s.dc = dcv;
        if (nbo)
            for (int i = lane; i <= 4 * N; i += XA_WAVE) { nbo[i] = s.ref[i]; nbo[129 + i] = N >= 8 ? s.flt[i] : (pixel)0; }
    }
    __syncthreads();
    const int dc = s.dc;
    /* the neighbour arrays of in_pred_sample live at the head of both LDS layouts */
    const IntraLds& nb = *reinterpret_cast<const IntraLds*>(&s);
    if (N == 4)         /* cu[4x4].sa8d = satd_4x4 (pixel.cpp:1171): sixteen lanes per mode */
    {
        const int l = tid & 15, x = l & 3, y = l >> 2, groups = nthr >> 4;
        for (int base = 0; base < 35; base += groups)
        {
            const int mode = base + (tid >> 4), m = mode < 35 ? mode : 34;
            int v = (int)s.fenc[y * 4 + x] - in_pred_sample(nb, m, 4, 2, dc, y, x);
            v = xa_butterfly<1>(v, l); v = xa_butterfly<2>(v, l); v = xa_butterfly<4>(v, l); v = xa_butterfly<8>(v, l);
            v = xa_row16_sum(abs(v));
            if (l == 0 && mode < 35) res[mode] = v >> 1;
        }
        return;
    }
    const int tpr = N >> 3, nt = tpr * tpr, tasks = 35 * nt, lgt = 2 * (log2N - 3);
    const int ly = lane >> 3, lx = lane & 7;
    for (int t0 = wv; t0 < tasks; t0 += 2 * nwv)       /* two tiles in flight per wavefront: their butterfly chains interleave */
    {
        const int t1 = t0 + nwv;
        const bool has1 = t1 < tasks;
        const int ta = t0, tb = has1 ? t1 : t0;
        const int modeA = ta >> lgt, tileA = ta & (nt - 1), modeB = tb >> lgt, tileB = tb & (nt - 1);
        const int yA = 8 * (tileA / tpr) + ly, xA = 8 * (tileA % tpr) + lx, yB = 8 * (tileB / tpr) + ly, xB = 8 * (tileB % tpr) + lx;
        int a = (int)s.fenc[yA * N + xA] - in_pred_sample(nb, modeA, N, log2N, dc, yA, xA);
        int b = (int)s.fenc[yB * N + xB] - in_pred_sample(nb, modeB, N, log2N, dc, yB, xB);
        a = xa_butterfly<1>(a, lane); b = xa_butterfly<1>(b, lane);
        a = xa_butterfly<2>(a, lane); b = xa_butterfly<2>(b, lane);
        a = xa_butterfly<4>(a, lane); b = xa_butterfly<4>(b, lane);
        a = xa_butterfly<8>(a, lane); b = xa_butterfly<8>(b, lane);
        a = xa_butterfly<16>(a, lane); b = xa_butterfly<16>(b, lane);
        a = xa_butterfly<32>(a, lane); b = xa_butterfly<32>(b, lane);
        const int rawA = xa_wave_sum(abs(a)), rawB = xa_wave_sum(abs(b));
        if (lane == 0)
        {
            if (N == 8) { res[modeA] = (rawA + 2) >> 2; if (has1) res[modeB] = (rawB + 2) >> 2; }         /* sa8d_8x8 (pixel.cpp:342-345) */
            else
            {
                atomicAdd(&s.acc[modeA][((tileA / tpr) >> 1) * (N >> 4) + ((tileA % tpr) >> 1)], rawA);      /* sa8d_16x16 groups (pixel.cpp:347-384) */
                if (has1) atomicAdd(&s.acc[modeB][((tileB / tpr) >> 1) * (N >> 4) + ((tileB % tpr) >> 1)], rawB);
            }
        }
    }
    if (N == 8) return;
    __syncthreads();
    if (tid < 35)
    {
        const int g = (N >> 4) * (N >> 4);
        int tot = 0;
        for (int k = 0; k < g; k++) tot += (s.acc[tid][k] + 2) >> 2;
        res[tid] = tot;
    }
}

#endif
