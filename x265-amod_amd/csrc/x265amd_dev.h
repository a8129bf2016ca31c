/* Device-side building blocks shared by every kernel of libx265amd (gfx950 only, wave64).
 *
 * Conventions
 *   - one 64-lane wavefront cooperates on one block-level operation; `lane` is 0..63
 *   - all arithmetic is the integer arithmetic of the reference's C primitives (bit-exact results);
 *     the citations name the reference function whose results each routine reproduces
 *     (paths relative to /root/reference/source)
 */
#ifndef X265AMD_DEV_H
#define X265AMD_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/x265amd.h"

/* algorithmic bytes of the running command: added by ONE lane per unit of work from the sizes in its job record.  Counted by the job server (device_queue.hip
 * defines XA_SERVER_BYTES before this header: its roofline counters); nothing in the ordinary kernels */
#ifdef XA_SERVER_BYTES
__shared__ unsigned long long xa_bytes_acc;
#define XA_BYTES(v) atomicAdd(&xa_bytes_acc, (unsigned long long)(v))
#else
#define XA_BYTES(v)
#endif


typedef x265amd_pixel pixel;

#define XA_DEPTH X265AMD_DEPTH
#define XA_PIXEL_MAX ((1 << XA_DEPTH) - 1)
#define XA_FENC_STRIDE 64
#define XA_IF_INTERNAL_PREC 14      /* common/constants.h:66-70 */
#define XA_IF_FILTER_PREC 6
#define XA_IF_INTERNAL_OFFS (1 << (XA_IF_INTERNAL_PREC - 1))
#define XA_WAVE 64

#define XA_DEV __device__ __forceinline__

/* ---- constant tables (generated at compile time from their defining rules; validated against the reference's
 *      arrays in common/constants.cpp:250-344 through the oracle, tests/test_oracle_vs_ref.py::test_tables) ---- */
struct XaTables
{
    int16_t dct[4][32 * 32];    /* dct[log2N-2][k*N+n]: HEVC core transform matrices (g_t4..g_t32) */
    int16_t dst4[16];           /* DST-VII 4x4 */
    int16_t lumaFilter[4][8];   /* g_lumaFilter */
    int16_t chromaFilter[8][4]; /* g_chromaFilter */
    int8_t angle[17];           /* intra angle table (intrapred.cpp:131) */
    int16_t invAngle[8];        /* intrapred.cpp:132 */
    uint8_t puW[25], puH[25];   /* LumaPU geometry (primitives.h:41-55) */
};

constexpr int xa_dct_coef32(int k, int n)
{
    constexpr int mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    if (k == 0) return 64;
    int a = ((2 * n + 1) * k) & 127;
    int sgn = 1;
    if (a > 64) a = 128 - a;
    if (a > 32) { a = 64 - a; sgn = -1; }
    return sgn * mag[a];
}

constexpr XaTables xa_make_tables()
{
    XaTables t = {};
    for (int l = 0; l < 4; l++)
    {
        int N = 4 << l, step = 32 / N;
        for (int k = 0; k < N; k++)
            for (int n = 0; n < N; n++)
                t.dct[l][k * N + n] = (int16_t)xa_dct_coef32(k * step, n);
    }
    constexpr int dst[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
    for (int i = 0; i < 16; i++) t.dst4[i] = (int16_t)dst[i];
    constexpr int lf[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    constexpr int cf[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                               { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) t.lumaFilter[i][j] = (int16_t)lf[i][j];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) t.chromaFilter[i][j] = (int16_t)cf[i][j];
    constexpr int ang[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    constexpr int inv[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
    for (int i = 0; i < 17; i++) t.angle[i] = (int8_t)ang[i];
    for (int i = 0; i < 8; i++) t.invAngle[i] = (int16_t)inv[i];
    constexpr int pw[25] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 12, 16, 4, 32, 24, 32, 8, 64, 48, 64, 16 };
    constexpr int ph[25] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 12, 16, 4, 16, 24, 32, 8, 32, 48, 64, 16, 64 };
    for (int i = 0; i < 25; i++) { t.puW[i] = (uint8_t)pw[i]; t.puH[i] = (uint8_t)ph[i]; }
    return t;
}

__device__ const XaTables xa_tbl = xa_make_tables();
static const XaTables xa_tbl_host = xa_make_tables();

/* constants.cpp:560-567 (g_intraFilterFlags[mode] & size) from its defining rule */
XA_DEV int xa_intra_filter_flags(int mode)
{
    if (mode == 1) return 0;
    if (mode == 0) return 8 | 16 | 32;
    int d1 = abs(mode - 26), d2 = abs(mode - 10);
    int d = d1 < d2 ? d1 : d2;
    return (d > 7 ? 8 : 0) | (d > 1 ? 16 : 0) | (d > 0 ? 32 : 0);
}

/* ---- wave helpers ---- */
XA_DEV int xa_lane() { return threadIdx.x & 63; }

/* orders LDS traffic of one wavefront: lanes of a wave run in lockstep and the DS unit is in-order per wave, so a
 * compiler-level fence is all that is needed between a wave's own writes and its cross-lane reads */
XA_DEV void xa_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template<class T> XA_DEV T xa_wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}

/* 32-bit sums use DPP row operations (one VALU instruction per step, no LDS round trips):
 * quad_perm / row_ror leave every lane with the sum of its 16-lane row, row_bcast:15 / :31 carry the row sums to lane 63 */
XA_DEV int xa_row16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);      /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);      /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, true);     /* row_ror:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);     /* row_ror:8 */
    return v;
}
template<> XA_DEV int xa_wave_sum<int>(int v)
{
    v = xa_row16_sum(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    /* row_bcast:31 into rows 2 and 3 */
    return __builtin_amdgcn_readlane(v, 63);
}
template<> XA_DEV uint32_t xa_wave_sum<uint32_t>(uint32_t v) { return (uint32_t)xa_wave_sum<int>((int)v); }



/* A job record as the host last wrote it: system-scope loads (sc0 sc1), which no cache level serves.  The records of the device job queues are
 * rewritten in place by the host between commands while the reading workgroup stays resident with its caches; in an ordinary kernel the loads are
 * merely uncached.  T: a multiple of 8 bytes, 8-byte aligned. */
template<class T> XA_DEV T xa_ld_record(const T* p)
{
    static_assert(sizeof(T) % 8 == 0 && alignof(T) >= 8, "job records are sequences of 64-bit words");
    union { T v; uint64_t w[sizeof(T) / 8]; } u;
    const uint64_t* s = reinterpret_cast<const uint64_t*>(p);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; i++) u.w[i] = __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return u.v;
}


/* Results the host reads (pinned host memory): system-scope stores, written through every cache level.  A resident workgroup has no kernel boundary
 * to flush for it, and its L2 otherwise keeps the lines it has stored to host memory. */
template<class T> XA_DEV void xa_st_result(T* p, const T& v)
{
    if constexpr (sizeof(T) % 8 == 0 && alignof(T) >= 8)
    {
        union { T v; uint64_t w[sizeof(T) / 8]; } u;
        u.v = v;
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 8; i++) __hip_atomic_store(reinterpret_cast<uint64_t*>(p) + i, u.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    else
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ---- Hadamard transforms ACROSS the lanes of a wavefront (one sample per lane): the workgroup-per-block forms of the kernels, where the time of one
 * block counts.  A butterfly stage exchanges with the lane whose index differs in bit M: DPP where one instruction does it, the LDS crossbar
 * otherwise.  The stages over bits 1, 2, 4 transform rows of 8, the stages over 8, 16, 32 the columns: all six give the 8x8 Hadamard of the wave's
 * 64 samples (one coefficient per lane, some order), four the 4x4 Hadamard of each group of 16 lanes.  Sums of absolute coefficients do not depend on
 * the order. ---- */
template<int M> XA_DEV int xa_lane_xor(int v)
{
    if (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);     /* quad_perm [1,0,3,2] */
    if (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     /* quad_perm [2,3,0,1] */
    if (M == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);    /* row_ror:8 */
    return __shfl_xor(v, M, 64);
}
template<int M> XA_DEV int xa_butterfly(int v, int lane) { const int o = xa_lane_xor<M>(v); return (lane & M) ? o - v : v + o; }
XA_DEV int xa_lane_had8x8(int v, int lane)
{
    v = xa_butterfly<1>(v, lane); v = xa_butterfly<2>(v, lane); v = xa_butterfly<4>(v, lane);
    v = xa_butterfly<8>(v, lane); v = xa_butterfly<16>(v, lane); v = xa_butterfly<32>(v, lane);
    return v;
}
XA_DEV int xa_lane_had4x4(int v, int lane)
{
    v = xa_butterfly<1>(v, lane); v = xa_butterfly<2>(v, lane); v = xa_butterfly<4>(v, lane); v = xa_butterfly<8>(v, lane);
    return v;
}

XA_DEV int xa_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
XA_DEV pixel xa_clip_pixel(int v) { return (pixel)xa_clip3(0, XA_PIXEL_MAX, v); }

/* ---- Hadamard tiles (pixel.cpp:189-340); results are the plain integer sums the reference's SWAR code yields ---- */
template<bool HASB>
XA_DEV int xa_had4_abs(const pixel* a, int sa, const pixel* b, int sb)
{
    int t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int d0 = a[y * sa + 0], d1 = a[y * sa + 1], d2 = a[y * sa + 2], d3 = a[y * sa + 3];
        if (HASB) { d0 -= b[y * sb + 0]; d1 -= b[y * sb + 1]; d2 -= b[y * sb + 2]; d3 -= b[y * sb + 3]; }
        int s01 = d0 + d1, e01 = d0 - d1, s23 = d2 + d3, e23 = d2 - d3;
        t[y][0] = s01 + s23; t[y][1] = s01 - s23; t[y][2] = e01 + e23; t[y][3] = e01 - e23;
    }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        int s01 = t[0][x] + t[1][x], e01 = t[0][x] - t[1][x], s23 = t[2][x] + t[3][x], e23 = t[2][x] - t[3][x];
        sum += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
    }
    return sum;
}

template<bool HASB>
XA_DEV int xa_had8_abs(const pixel* a, int sa, const pixel* b, int sb)
{
    int m[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
        int r[8];
#pragma unroll
        for (int x = 0; x < 8; x++)
            r[x] = HASB ? (int)a[y * sa + x] - (int)b[y * sb + x] : (int)a[y * sa + x];
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i += span * 2)
#pragma unroll
                for (int j = i; j < i + span; j++)
                {
                    int u = r[j], v = r[j + span];
                    r[j] = u + v; r[j + span] = u - v;
                }
#pragma unroll
        for (int x = 0; x < 8; x++) m[y][x] = r[x];
    }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
    {
        int r[8];
#pragma unroll
        for (int y = 0; y < 8; y++) r[y] = m[y][x];
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i += span * 2)
#pragma unroll
                for (int j = i; j < i + span; j++)
                {
                    int u = r[j], v = r[j + span];
                    r[j] = u + v; r[j + span] = u - v;
                }
#pragma unroll
        for (int y = 0; y < 8; y++) sum += abs(r[y]);
    }
    return sum;
}

/* sum of |Hadamard8x8(m)| for a difference block already in registers (same transform as xa_had8_abs) */
XA_DEV int xa_had8_abs_regs(int m[8][8])
{
#pragma unroll
    for (int y = 0; y < 8; y++)
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i += span * 2)
#pragma unroll
                for (int j = i; j < i + span; j++)
                {
                    int u = m[y][j], v = m[y][j + span];
                    m[y][j] = u + v; m[y][j + span] = u - v;
                }
    int sum = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
    {
#pragma unroll
        for (int span = 1; span < 8; span <<= 1)
#pragma unroll
            for (int i = 0; i < 8; i += span * 2)
#pragma unroll
                for (int j = i; j < i + span; j++)
                {
                    int u = m[j][x], v = m[j + span][x];
                    m[j][x] = u + v; m[j + span][x] = u - v;
                }
#pragma unroll
        for (int y = 0; y < 8; y++) sum += abs(m[y][x]);
    }
    return sum;
}

/* SAD of a w x h block, lanes strided over samples (pixel.cpp:40-54) */
XA_DEV int xa_wave_sad(const pixel* a, int sa, const pixel* b, int sb, int w, int h, int lane)
{
    int sum = 0, n = w * h;
    for (int i = lane; i < n; i += XA_WAVE)
    {
        int y = i / w, x = i - y * w;
        sum += abs((int)a[y * sa + x] - (int)b[y * sb + x]);
    }
    return xa_wave_sum(sum);
}

/* SATD of a w x h block: one lane per 4x4 tile, halved per tile (pixel.cpp:210-297) */
XA_DEV int xa_wave_satd(const pixel* a, int sa, const pixel* b, int sb, int w, int h, int lane)
{
    int tw = w >> 2, nt = tw * (h >> 2), sum = 0;
    for (int t = lane; t < nt; t += XA_WAVE)
    {
        int ty = t / tw, tx = t - ty * tw;
        sum += xa_had4_abs<true>(a + 4 * ty * sa + 4 * tx, sa, b + 4 * ty * sb + 4 * tx, sb) >> 1;
    }
    return xa_wave_sum(sum);
}

/* SA8D of a size x size block (size 8..64): one lane per 8x8 tile; 16x16 groups are rounded once
 * (pixel.cpp:342-384: sa8d_8x8, sa8d_16x16, sa8d16<w,h>).  size 4 -> satd_4x4 (pixel.cpp:1171). */
XA_DEV int xa_wave_sa8d(const pixel* a, int sa, const pixel* b, int sb, int size, int lane)
{
    if (size == 4)
    {
        int v = lane == 0 ? xa_had4_abs<true>(a, sa, b, sb) >> 1 : 0;
        return __shfl(v, 0, 64);
    }
    if (size == 8)
    {
        int v = lane == 0 ? (xa_had8_abs<true>(a, sa, b, sb) + 2) >> 2 : 0;
        return __shfl(v, 0, 64);
    }
    int g = size >> 4;              /* 16x16 groups per row */
    int raw = 0;
    if (lane < 4 * g * g)
    {
        int grp = lane >> 2, sub = lane & 3;
        int gy = grp / g, gx = grp - gy * g;
        int x = gx * 16 + (sub & 1) * 8, y = gy * 16 + (sub >> 1) * 8;
        raw = xa_had8_abs<true>(a + y * sa + x, sa, b + y * sb + x, sb);
    }
    raw += __shfl_xor(raw, 1, 64);
    raw += __shfl_xor(raw, 2, 64);
    int v = (lane & 3) == 0 && lane < 4 * g * g ? (raw + 2) >> 2 : 0;
    return xa_wave_sum(v);
}

/* ---- interpolation of one sample, shared by the MC kernel and the chroma-SATD part of the ME kernel ---- */
/* one interpolated sample.  SHORT = false: the pixel path (copy / hpp / vpp / hps(rowExt)+vsp: ipfilter.cpp:79-120, :169-210,
 * :250-292); SHORT = true: the 14-bit intermediate path (p2s / hps / vps / hps(rowExt)+vss: ipfilter.cpp:39-56, :122-167,
 * :212-248, :294-324).  TAPS 8: luma, fractions in quarter samples; TAPS 4: chroma, eighth samples. */
template<int TAPS, bool SHORT> XA_DEV int mc_sample(const pixel* src, long stride, int xf, int yf)
{
    const int half = TAPS / 2 - 1;
    const int headRoom = XA_IF_INTERNAL_PREC - XA_DEPTH;
    const int shiftH = XA_IF_FILTER_PREC - headRoom, offH = (int)((unsigned)-XA_IF_INTERNAL_OFFS << shiftH);
    const int16_t* cx = TAPS == 8 ? xa_tbl.lumaFilter[xf] : xa_tbl.chromaFilter[xf];
    const int16_t* cy = TAPS == 8 ? xa_tbl.lumaFilter[yf] : xa_tbl.chromaFilter[yf];
    if (!(xf | yf))
        return SHORT ? (int)(int16_t)((int16_t)(src[0] << headRoom) - (int16_t)XA_IF_INTERNAL_OFFS) : (int)src[0];
    if (!yf || !xf)
    {
        int sum = 0;
#pragma unroll
        for (int t = 0; t < TAPS; t++)
            sum += yf ? (int)src[(long)(t - half) * stride] * cy[t] : (int)src[t - half] * cx[t];
        if (SHORT) return (int)(int16_t)((sum + offH) >> shiftH);
        return xa_clip3(0, XA_PIXEL_MAX, (int)(int16_t)((sum + (1 << (XA_IF_FILTER_PREC - 1))) >> XA_IF_FILTER_PREC));
    }
    int sum = 0;
#pragma unroll 1
    for (int r = 0; r < TAPS; r++)
    {
        const pixel* row = src + (long)(r - half) * stride - half;
        int hs = 0;
#pragma unroll
        for (int t = 0; t < TAPS; t++) hs += (int)row[t] * cx[t];
        sum += (int)(int16_t)((hs + offH) >> shiftH) * cy[r];
    }
    if (SHORT) return (int)(int16_t)(sum >> XA_IF_FILTER_PREC);
    const int shiftV = XA_IF_FILTER_PREC + headRoom, offV = (1 << (shiftV - 1)) + (XA_IF_INTERNAL_OFFS << XA_IF_FILTER_PREC);
    return xa_clip3(0, XA_PIXEL_MAX, (int)(int16_t)((sum + offV) >> shiftV));
}


/* ---- intra building blocks shared by the job-list and the fused intra kernels ---- */
/* one angular prediction sample in "vertical orientation" on (possibly swapped) neighbours: intrapred.cpp:106-196 */
XA_DEV pixel ang_sample(const pixel* s, int N, int angle, int invAngle, int bFilter, int y, int x)
{
    int N2 = 2 * N;
    if (!angle)
    {
        if (bFilter && x == 0)
            return xa_clip_pixel((int16_t)(s[1] + ((s[N2 + 1 + y] - s[0]) >> 1)));
        return s[1 + x];
    }
    int angSum = (y + 1) * angle;
    int off = angSum >> 5, frac = angSum & 31;
    /* ref[i]: i >= -1 -> s[i+1]; i <= -2 (negative angles only) -> projected left neighbour */
    int i0 = off + x, i1 = off + x + 1;
    int r0, r1;
    if (i0 >= -1) r0 = s[i0 + 1];
    else r0 = s[N2 + ((128 + (-1 - i0) * invAngle) >> 8)];
    if (!frac) return (pixel)r0;
    if (i1 >= -1) r1 = s[i1 + 1];
    else r1 = s[N2 + ((128 + (-1 - i1) * invAngle) >> 8)];
    return (pixel)(((32 - frac) * r0 + frac * r1 + 16) >> 5);
}

/* intrapred.cpp:30-52 */
XA_DEV void wave_intra_filter(const pixel* s, pixel* f, int N, int lane)
{
    int N2 = 2 * N;
    for (int i = lane; i <= 2 * N2; i += XA_WAVE)
    {
        int v;
        if (i == 0) v = (2 * s[0] + s[1] + s[N2 + 1] + 2) >> 2;
        else if (i == N2 || i == 2 * N2) v = s[i];
        else if (i == N2 + 1) v = (2 * s[N2 + 1] + s[0] + s[N2 + 2] + 2) >> 2;
        else v = (2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2;
        f[i] = (pixel)v;
    }
}


/* ---- intra neighbour set: Predict::fillReferenceSamples + the smoothing of initAdiPattern (predict.cpp:600-649, :736-877).
 * recon: the block's top-left sample in the reconstructed plane; availMask bit u: neighbour unit u (4 samples) available, order
 * below-left (bottom-most first) ... left, above-left, above ... above-right.  ref / flt: [0] above-left, [1..2N] above +
 * above-right, [2N+1..4N] left + below-left (LDS, one wave).  flt is produced when wantFilter (8x8..32x32 luma). ---- */
XA_DEV int in_first_of_unit(int u, int L, int N2) { return u < L ? 4 * u : (u == L ? N2 : N2 + 1 + 4 * (u - L - 1)); }
XA_DEV int in_last_of_unit(int u, int L, int N2) { return u < L ? 4 * u + 3 : (u == L ? N2 : N2 + 1 + 4 * (u - L - 1) + 3); }
XA_DEV void wave_intra_neighbours(const pixel* recon, long rs, uint64_t availMask, int log2N, bool strongSmoothing, bool wantFilter,
                                  pixel* ref, pixel* flt, int lane)
{
    const int N = 1 << log2N, N2 = 2 * N, units = N >> 2, L = 2 * units;
    const uint64_t avail = availMask & ((units == 8) ? 0x1ffffffffull : ((1ull << (4 * units + 1)) - 1));
    /* ---- fillReferenceSamples: substitution order index i: 0 = bottom-most below-left ... 2N = above-left ... 4N ---- */
    for (int i = lane; i <= 4 * N; i += XA_WAVE)
    {
        int u = i < N2 ? i >> 2 : (i == N2 ? L : L + 1 + ((i - N2 - 1) >> 2));
        int src = i;
        if (!((avail >> u) & 1))
        {
            uint64_t before = avail & ((1ull << u) - 1);
            if (before) src = in_last_of_unit(63 - __clzll((long long)before), L, N2);
            else if (avail) src = in_first_of_unit(__ffsll((long long)avail) - 1, L, N2);
            else src = -1;
        }
        int v;
        if (src < 0) v = 1 << (XA_DEPTH - 1);
        else if (src < N2) v = recon[(long)(N2 - 1 - src) * rs - 1];
        else if (src == N2) v = recon[-rs - 1];
        else v = recon[-rs + (src - N2 - 1)];
        int d = i == N2 ? 0 : (i > N2 ? i - N2 : N2 + (N2 - i));      /* destination index in the neighbour-buffer layout */
        ref[d] = (pixel)v;
    }
    xa_wave_sync();
    /* ---- initAdiPattern(ALL_IDX): smoothing for 8x8 .. 32x32 ---- */
    if (wantFilter && N >= 8)
    {
        bool strong = false;
        if (strongSmoothing && N == 32)
        {
            const int threshold = 1 << (XA_DEPTH - 5);
            int topLeft = ref[0], topLast = ref[N2], leftLast = ref[2 * N2];
            strong = abs(topLeft + topLast - 2 * ref[32]) < threshold && abs(topLeft + leftLast - 2 * ref[N2 + 32]) < threshold;
            if (strong)
            {
                int init = (topLeft << 6) + N, deltaL = leftLast - topLeft, deltaR = topLast - topLeft;
                for (int i = lane; i <= 2 * N2; i += XA_WAVE)
                {
                    int v;
                    if (i == 0) v = topLeft;
                    else if (i == N2) v = topLast;
                    else if (i == 2 * N2) v = leftLast;
                    else if (i < N2) v = (init + deltaR * i) >> 6;
                    else v = (init + deltaL * (i - N2)) >> 6;
                    flt[i] = (pixel)v;
                }
            }
        }
        if (!strong) wave_intra_filter(ref, flt, N, lane);
    }
    xa_wave_sync();
}

/* writes one N x N prediction; s = neighbours in LDS.  keepTransposed: all-angles layout (intrapred.cpp:211-241) */
XA_DEV void wave_intra_pred(const pixel* s, pixel* swapped, int cu, int mode, int bFilter, pixel* dst, int ds, bool keepTransposed, int lane)
{
    int log2N = cu + 2, N = 1 << log2N, N2 = 2 * N;
    if (mode == 0)      /* planar: intrapred.cpp:90-104 */
    {
        const pixel* above = s + 1; const pixel* left = s + N2 + 1;
        int topRight = above[N], bottomLeft = left[N];
        for (int i = lane; i < N * N; i += XA_WAVE)
        {
            int y = i >> log2N, x = i & (N - 1);
            dst[y * ds + x] = (pixel)(((N - 1 - x) * left[y] + (N - 1 - y) * above[x] + (x + 1) * topRight + (y + 1) * bottomLeft + N) >> (log2N + 1));
        }
        return;
    }
    if (mode == 1)      /* DC: intrapred.cpp:54-88 */
    {
        const pixel* above = s + 1; const pixel* left = s + N2 + 1;
        int part = lane < N ? above[lane] + left[lane] : 0;
        int dc = (xa_wave_sum(part) + N) / (2 * N);
        for (int i = lane; i < N * N; i += XA_WAVE)
        {
            int y = i >> log2N, x = i & (N - 1);
            int v = dc;
            if (bFilter)
            {
                if (x == 0 && y == 0) v = (above[0] + left[0] + 2 * dc + 2) >> 2;
                else if (y == 0) v = (above[x] + 3 * dc + 2) >> 2;
                else if (x == 0) v = (left[y] + 3 * dc + 2) >> 2;
            }
            dst[y * ds + x] = (pixel)v;
        }
        return;
    }
    bool hor = mode < 18;
    const pixel* nb = s;
    if (hor)            /* mirror the neighbours: intrapred.cpp:114-124 */
    {
        for (int i = lane; i < N2; i += XA_WAVE)
        {
            swapped[1 + i] = s[N2 + 1 + i];
            swapped[N2 + 1 + i] = s[1 + i];
        }
        if (lane == 0) swapped[0] = s[0];
        xa_wave_sync();
        nb = swapped;
    }
    int angOff = hor ? 10 - mode : mode - 26;
    int angle = xa_tbl.angle[8 + angOff];
    int invAngle = angle < 0 ? xa_tbl.invAngle[-angOff - 1] : 0;
    bool flip = hor && !keepTransposed;
    for (int i = lane; i < N * N; i += XA_WAVE)
    {
        int y = i >> log2N, x = i & (N - 1);
        dst[y * ds + x] = flip ? ang_sample(nb, N, angle, invAngle, bFilter, x, y) : ang_sample(nb, N, angle, invAngle, bFilter, y, x);
    }
    xa_wave_sync();
}


/* ---- block metrics shared by the job-list and the fused TU kernels ---- */

/* ---- thread groups: the transform-chain helpers run on one wavefront (XaWave: a batch of many blocks, one per wavefront) or on a whole workgroup
 * (XaBlock: the device job queues, where a command carries a few large blocks and the time of one counts).  A group strides over samples with
 * idx / step(), orders its LDS traffic with sync() and reduces with sum() / maxv(). ---- */
struct XaWave
{
    int idx;
    XA_DEV int step() const { return XA_WAVE; }
    XA_DEV int wave() const { return 0; }
    XA_DEV int waves() const { return 1; }
    XA_DEV void sync() const { xa_wave_sync(); }
    XA_DEV int sum(int v) const { return xa_wave_sum(v); }
    XA_DEV uint32_t sum(uint32_t v) const { return xa_wave_sum(v); }
    XA_DEV uint64_t sum(uint64_t v) const { return xa_wave_sum(v); }
    XA_DEV int maxv(int v) const
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        return v;
    }
};
struct XaBlock
{
    int idx, nthr;
    unsigned long long* red;        /* LDS: nthr / 64 words */
    XA_DEV int step() const { return nthr; }
    XA_DEV int wave() const { return idx >> 6; }
    XA_DEV int waves() const { return nthr >> 6; }
    XA_DEV void sync() const { __syncthreads(); }
    XA_DEV uint64_t sum(uint64_t v) const
    {
        v = xa_wave_sum(v);
        if ((idx & 63) == 0) red[idx >> 6] = v;
        __syncthreads();
        uint64_t t = 0;
        for (int w = 0; w < (nthr >> 6); w++) t += red[w];
        __syncthreads();
        return t;
    }
    XA_DEV int sum(int v) const { return (int)(int64_t)sum((uint64_t)(int64_t)v); }
    XA_DEV uint32_t sum(uint32_t v) const { return (uint32_t)sum((uint64_t)v); }
    XA_DEV int maxv(int v) const
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        if ((idx & 63) == 0) red[idx >> 6] = (unsigned long long)(long long)v;
        __syncthreads();
        int t = (int)(long long)red[0];
        for (int w = 1; w < (nthr >> 6); w++) t = max(t, (int)(long long)red[w]);
        __syncthreads();
        return t;
    }
};

template<class G> XA_DEV uint64_t grp_sse_pp(const pixel* a, int sa, const pixel* b, int sb, int size, const G& g)     /* pixel.cpp:167-186 */
{
    uint64_t sum = 0;
    int n = size * size, sh = 31 - __clz(size);
    for (int i = g.idx; i < n; i += g.step())
    {
        int y = i >> sh, x = i & (size - 1);
        int t = (int)a[y * sa + x] - (int)b[y * sb + x];
        sum += (uint64_t)(uint32_t)(t * t);
    }
    sum = g.sum(sum);
#if XA_DEPTH <= 8
    sum = (uint32_t)sum;        /* sse_t is uint32_t below 10 bits (common/common.h:142-146) */
#endif
    return sum;
}
XA_DEV uint64_t wave_sse_pp(const pixel* a, int sa, const pixel* b, int sb, int size, int lane) { return grp_sse_pp(a, sa, b, sb, size, XaWave{ lane }); }

/* pixel.cpp:744-775 */
/* psyCost_pp (pixel.cpp:744-775): per 8x8 tile |(sa8d(src) - sad(src) / 4) - (sa8d(rec) - sad(rec) / 4)| with the "sa8d" and "sad" of a block against
 * zeros, i.e. the sum of its absolute Hadamard coefficients and the sum of its samples (4x4 blocks: satd).  One wavefront per tile, one lane per sample:
 * the Hadamard runs across the lanes, and its coefficient 0 (lane 0) IS the sum of the samples. */
template<class G> XA_DEV int grp_psy_cost(const pixel* src, int ss, const pixel* rec, int rs, int cu, const G& g)
{
    const int lane = g.idx & 63;
    int acc = 0;
    if (cu == 0)
    {
        if (g.wave() == 0)
        {
            const int l = lane & 15, y = l >> 2, x = l & 3;
            const int hs = xa_lane_had4x4((int)src[y * ss + x], lane), hr = xa_lane_had4x4((int)rec[y * rs + x], lane);
            const int sumS = xa_row16_sum(abs(hs)), sumR = xa_row16_sum(abs(hr));
            const int sadS = __builtin_amdgcn_readfirstlane(hs), sadR = __builtin_amdgcn_readfirstlane(hr);
            const int se = (__builtin_amdgcn_readfirstlane(sumS) >> 1) - (sadS >> 2), re = (__builtin_amdgcn_readfirstlane(sumR) >> 1) - (sadR >> 2);
            acc = abs(se - re);
        }
    }
    else
    {
        const int tiles = 1 << (cu - 1), nt = tiles * tiles, ly = lane >> 3, lx = lane & 7;
        for (int t = g.wave(); t < nt; t += g.waves())
        {
            const int ty = t / tiles, tx = t - ty * tiles;
            const int hs = xa_lane_had8x8((int)src[(8 * ty + ly) * ss + 8 * tx + lx], lane), hr = xa_lane_had8x8((int)rec[(8 * ty + ly) * rs + 8 * tx + lx], lane);
            const int sumS = xa_wave_sum(abs(hs)), sumR = xa_wave_sum(abs(hr));
            const int sadS = __builtin_amdgcn_readfirstlane(hs), sadR = __builtin_amdgcn_readfirstlane(hr);
            acc += abs((((sumS + 2) >> 2) - (sadS >> 2)) - (((sumR + 2) >> 2) - (sadR >> 2)));
        }
    }
    return g.sum(lane == 0 ? acc : 0);
}
XA_DEV int wave_psy_cost(const pixel* src, int ss, const pixel* rec, int rs, int cu, int lane) { return grp_psy_cost(src, ss, rec, rs, cu, XaWave{ lane }); }


/* ---- transform passes (dct.cpp:83-440 partial butterflies == exact integer matrix products) ---- */
template<class G> XA_DEV void grp_fwd_pass(const int16_t* T, int log2N, const int16_t* src, int16_t* dst, int shift, const G& g)
{
    int N = 1 << log2N, add = 1 << (shift - 1);
    for (int i = g.idx; i < N * N; i += g.step())
    {
        int k = i >> log2N, jj = i & (N - 1);   /* consecutive lanes: consecutive j (dst row k contiguous) */
        int sum = 0;
        for (int nn = 0; nn < N; nn++)
            sum += T[k * N + nn] * src[jj * N + nn];
        dst[k * N + jj] = (int16_t)((sum + add) >> shift);
    }
}
XA_DEV void wave_fwd_pass(const int16_t* T, int log2N, const int16_t* src, int16_t* dst, int shift, int lane) { grp_fwd_pass(T, log2N, src, dst, shift, XaWave{ lane }); }

template<class G> XA_DEV void grp_inv_pass(const int16_t* T, int log2N, const int16_t* src, int16_t* dst, int dstStride, int shift, const G& g)
{
    int N = 1 << log2N, add = 1 << (shift - 1);
    for (int i = g.idx; i < N * N; i += g.step())
    {
        int jj = i >> log2N, nn = i & (N - 1);
        int sum = 0;
        for (int k = 0; k < N; k++)
            sum += T[k * N + nn] * src[k * N + jj];
        dst[jj * dstStride + nn] = (int16_t)xa_clip3(-32768, 32767, (sum + add) >> shift);
    }
}
XA_DEV void wave_inv_pass(const int16_t* T, int log2N, const int16_t* src, int16_t* dst, int dstStride, int shift, int lane) { grp_inv_pass(T, log2N, src, dst, dstStride, shift, XaWave{ lane }); }


#endif /* X265AMD_DEV_H */
