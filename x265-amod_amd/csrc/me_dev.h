/* Device code of the motion search (see me_kernels.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_ME_DEV_H
#define X265AMD_ME_DEV_H
#include "x265amd_dev.h"

/* Every search helper exists twice: its body (inlined) and a real-function wrapper.  The HEX/DIA kernel variant inlines
 * the bodies into the search (no call overhead and no callee-saved VGPR spills: 2.02 -> 1.74 ms per 1080p frame, at 85 KB
 * of code); the variants that carry the star search (dozens of call sites) and the deferred kernel call the wrappers. */
#define ME_HELPER __device__ __forceinline__
#ifndef ME_WAVES
#define ME_WAVES 8
#endif
#ifndef ME_MIN_WAVES_PER_EU
#define ME_MIN_WAVES_PER_EU 4
#endif
#define ME_DEFERRED ((int)0x80000000)   /* result marker: redo this job with the direct-from-HBM kernel */
#define ME_QP_COUNT 70              /* QP_MAX_MAX + 1 (reference: common/constants.h) */
#define ME_TBL_HALF 65536           /* 2 * BC_MAX_MV: table index range is [-65536, 65536] (bitcost.h:81) */
#define ME_TBL_LEN (2 * ME_TBL_HALF + 1)


struct MeParams
{
    const pixel* fenc;
    const uint64_t* refs;
    const uint64_t* chroma;     /* NULL, or [0],[1] source U,V origins and [2+2r],[3+2r] reference r U,V origins */
    int stride, cstride;
    const x265amd_me_group* groups;
    const x265amd_me_job* jobs;
    x265amd_me_result* out;
    const uint16_t* tables;
    int maxWinW, maxWinH;
};

/* ---- per-wavefront search state ---- */
struct MeState
{
    const pixel* win;   /* LDS window: sample (winX + i, winY + j) at win[j * winW + i] */
    int winX, winY, winW, winH;
    const pixel* fencT; /* LDS 64x64 source tile, PU at (fx, fy) */
    int fx, fy;
    const pixel* refG;  /* HBM: sample (0,0) of the reference plane */
    int stride;
    int px, py, w, h;   /* PU position and size */
    const uint16_t* cost;
    int mvpx, mvpy;
    int mnx, mny, mxx, mxy;
    /* chroma SATD (MotionEstimate::bChromaSATD, motion.cpp:234-237): 4:2:0 chroma blocks of the PU in HBM */
    int chroma, cstride;
    const pixel* fencC[2];
    const pixel* refC[2];
    int acc[16];        /* per-candidate accumulators of the batched sub-pel comparisons */
    int cand[16];       /* their quarter-pel MVs, packed (qy << 16) | (qx & 0xffff) */
    int oob;            /* set by the window-resident kernel when a candidate left the staged window: the job is redone
                           by the direct-from-HBM kernel (k_me_deferred) */
};

extern __shared__ __attribute__((aligned(16))) char me_smem[];
/* the per-wavefront state lives in LDS (one MeState per wave): the search helpers are real functions (not inlined
 * into the ~60 call sites of the search) and find it through its LDS byte offset */
#define ME_S(off) (*reinterpret_cast<const MeState*>(me_smem + (off)))
#define ME_OOB(off) (reinterpret_cast<MeState*>(me_smem + (off))->oob = 1)
#define ME_OOB_COST 0x3fffffff      /* never wins a comparison; the job's result is discarded anyway */

/* dispatchers: inline the body (INL) or call the real-function wrapper */
template<bool SLOW, bool INL> XA_DEV int me_sad_at_f(int sOff, int mx, int my);
template<bool SATD, bool SLOW, bool INL> XA_DEV int me_subpel_f(int sOff, int qx, int qy);
template<bool SLOW, bool INL> XA_DEV int me_cost_multi_f(int sOff, int n, int m0, int m1, int m2, int m3);
template<bool SATD, bool SLOW, bool INL> XA_DEV int me_subpel_list_f(int sOff, int n);

XA_DEV int me_mvcost(const MeState& s, int qx, int qy) { return (uint16_t)(s.cost[qx - s.mvpx] + s.cost[qy - s.mvpy]); }

template<bool INWIN> XA_DEV int me_ref(const MeState& s, int X, int Y)
{
    return INWIN ? (int)s.win[(Y - s.winY) * s.winW + (X - s.winX)] : (int)s.refG[(long)Y * s.stride + X];
}

XA_DEV bool me_inwin(const MeState& s, int X0, int Y0, int X1, int Y1)     /* [X0,X1) x [Y0,Y1) inside the staged window */
{
    return X0 >= s.winX && Y0 >= s.winY && X1 <= s.winX + s.winW && Y1 <= s.winY + s.winH;
}

/* SAD of the PU against the reference block whose top-left sample is (X, Y): pixel.cpp:40-54 */
template<bool INWIN> XA_DEV int me_sad_fpel(const MeState& s, int X, int Y)
{
    int gpr = s.w >> 2, total = gpr * s.h, inv = ((1 << 20) + gpr - 1) / gpr, sum = 0;
    for (int gi = xa_lane(); gi < total; gi += XA_WAVE)
    {
        int y = (gi * inv) >> 20, x = (gi - y * gpr) << 2;
        const pixel* f = s.fencT + (s.fy + y) * 64 + s.fx + x;
#if XA_DEPTH == 8
        if (INWIN)
        {
            uint32_t fv = *reinterpret_cast<const uint32_t*>(f);
            int o = (Y + y - s.winY) * s.winW + (X + x - s.winX);
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(s.win) + (o >> 2);
            uint32_t rv = __builtin_amdgcn_alignbyte(wp[1], wp[0], o & 3);
            sum = __builtin_amdgcn_sad_u8(fv, rv, sum);
            continue;
        }
#endif
#pragma unroll
        for (int k = 0; k < 4; k++)
            sum += abs((int)f[k] - me_ref<INWIN>(s, X + x + k, Y + y));
    }
    return xa_wave_sum(sum);
}

/* chroma part of subpelCompare (motion.cpp:1625-1686): SATD of both 4:2:0 chroma blocks at the luma quarter-pel MV
 * (= chroma eighth-pel MV), prediction as predInterChromaPixel does it; one lane per 4x4 chroma tile, samples from HBM/L2 */
__device__ __noinline__ int me_chroma_satd(int sOff, int qx, int qy)
{
    const MeState& s = ME_S(sOff);
    const int lane = xa_lane();
    const int cw = s.w >> 1, ch = s.h >> 1, tw = cw >> 2, nt = tw * (ch >> 2);
    const int xf = qx & 7, yf = qy & 7;
    const long off = (long)(qy >> 3) * s.cstride + (qx >> 3);
    int sum = 0;
    for (int it = lane; it < 2 * nt; it += XA_WAVE)
    {
        int c = it >= nt, t = c ? it - nt : it;
        int ty = t / tw, tx = t - ty * tw;
        const pixel* ref = s.refC[c] + off + (long)(4 * ty) * s.cstride + 4 * tx;
        const pixel* f = s.fencC[c] + (long)(4 * ty) * s.cstride + 4 * tx;
        int d[4][4];
#pragma unroll 1
        for (int y = 0; y < 4; y++)
        {
            int r0 = (int)f[(long)y * s.cstride + 0] - mc_sample<4, false>(ref + (long)y * s.cstride + 0, s.cstride, xf, yf);
            int r1 = (int)f[(long)y * s.cstride + 1] - mc_sample<4, false>(ref + (long)y * s.cstride + 1, s.cstride, xf, yf);
            int r2 = (int)f[(long)y * s.cstride + 2] - mc_sample<4, false>(ref + (long)y * s.cstride + 2, s.cstride, xf, yf);
            int r3 = (int)f[(long)y * s.cstride + 3] - mc_sample<4, false>(ref + (long)y * s.cstride + 3, s.cstride, xf, yf);
            int s01 = r0 + r1, e01 = r0 - r1, s23 = r2 + r3, e23 = r2 - r3;
            int a0 = s01 + s23, a1 = s01 - s23, a2 = e01 + e23, a3 = e01 - e23;
            if (y == 0) { d[0][0] = a0; d[0][1] = a1; d[0][2] = a2; d[0][3] = a3; }
            else if (y == 1) { d[1][0] = a0; d[1][1] = a1; d[1][2] = a2; d[1][3] = a3; }
            else if (y == 2) { d[2][0] = a0; d[2][1] = a1; d[2][2] = a2; d[2][3] = a3; }
            else { d[3][0] = a0; d[3][1] = a1; d[3][2] = a2; d[3][3] = a3; }
        }
        int ts = 0;
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            int s01 = d[0][x] + d[1][x], e01 = d[0][x] - d[1][x], s23 = d[2][x] + d[3][x], e23 = d[2][x] - d[3][x];
            ts += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
        }
        sum += ts >> 1;
    }
    return xa_wave_sum(sum);
}

template<bool SLOW, bool INL> ME_HELPER int me_sad_at_b(int sOff, int mx, int my)     /* full-pel MV (mx,my) */
{
    const MeState& s = ME_S(sOff);
    int X = s.px + mx, Y = s.py + my;
    if (SLOW) return me_sad_fpel<false>(s, X, Y);
    /* the dword reads of the LDS path touch up to 3 samples past the block's right edge */
    if (me_inwin(s, X, Y, X + s.w + 4, Y + s.h)) return me_sad_fpel<true>(s, X, Y);
    ME_OOB(sOff);
    return ME_OOB_COST;
}

/* one interpolated luma sample at integer position (X,Y) + fraction (xf,yf)/4:
 * luma_hpp / luma_vpp / luma_hvpp (ipfilter.cpp:79-120, :169-210, :370-378 = hps(rowExt) + vsp) */
template<bool INWIN> XA_DEV int me_pred(const MeState& s, int X, int Y, int xf, int yf)
{
    if (!(xf | yf)) return me_ref<INWIN>(s, X, Y);
    const int16_t* cx = xa_tbl.lumaFilter[xf];
    const int16_t* cy = xa_tbl.lumaFilter[yf];
    if (!yf || !xf)
    {
        int sum = 0;
#pragma unroll 1
        for (int t = 0; t < 8; t++)
            sum += (yf ? me_ref<INWIN>(s, X, Y - 3 + t) * cy[t] : me_ref<INWIN>(s, X - 3 + t, Y) * cx[t]);
        int16_t val = (int16_t)((sum + (1 << (XA_IF_FILTER_PREC - 1))) >> XA_IF_FILTER_PREC);
        return xa_clip3(0, XA_PIXEL_MAX, val);
    }
    const int headRoom = XA_IF_INTERNAL_PREC - XA_DEPTH;
    const int shiftH = XA_IF_FILTER_PREC - headRoom, offH = (int)((unsigned)-XA_IF_INTERNAL_OFFS << shiftH);
    const int shiftV = XA_IF_FILTER_PREC + headRoom, offV = (1 << (shiftV - 1)) + (XA_IF_INTERNAL_OFFS << XA_IF_FILTER_PREC);
    int sum = 0;
#pragma unroll 1
    for (int r = 0; r < 8; r++)
    {
        int hs = 0;
#pragma unroll 1
        for (int t = 0; t < 8; t++)
            hs += me_ref<INWIN>(s, X - 3 + t, Y - 3 + r) * cx[t];
        sum += (int)(int16_t)((hs + offH) >> shiftH) * cy[r];
    }
    int16_t val = (int16_t)((sum + offV) >> shiftV);
    return xa_clip3(0, XA_PIXEL_MAX, val);
}

/* subpelCompare (motion.cpp:1596-1623) with cmp = sad or satd, quarter-pel MV (qx,qy) */
template<bool INWIN, bool SATD> __device__ __noinline__ int me_subpel_cmp(int sOff, int qx, int qy)
{
    const MeState& s = ME_S(sOff);
    int X0 = s.px + (qx >> 2), Y0 = s.py + (qy >> 2), xf = qx & 3, yf = qy & 3, sum = 0;
    if (SATD)       /* one lane per 4x4 tile: pixel.cpp:210-297 */
    {
        int tw = s.w >> 2, nt = tw * (s.h >> 2), inv = ((1 << 20) + tw - 1) / tw;
        for (int t = xa_lane(); t < nt; t += XA_WAVE)
        {
            int ty = (t * inv) >> 20, tx = t - ty * tw;
            int d[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int row[4];
#pragma unroll 1
                for (int x = 0; x < 4; x++)
                    row[x] = (int)s.fencT[(s.fy + 4 * ty + y) * 64 + s.fx + 4 * tx + x] - me_pred<INWIN>(s, X0 + 4 * tx + x, Y0 + 4 * ty + y, xf, yf);
                d[y][0] = row[0]; d[y][1] = row[1]; d[y][2] = row[2]; d[y][3] = row[3];
            }
            int tt[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int s01 = d[y][0] + d[y][1], e01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], e23 = d[y][2] - d[y][3];
                tt[y][0] = s01 + s23; tt[y][1] = s01 - s23; tt[y][2] = e01 + e23; tt[y][3] = e01 - e23;
            }
            int ts = 0;
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int s01 = tt[0][x] + tt[1][x], e01 = tt[0][x] - tt[1][x], s23 = tt[2][x] + tt[3][x], e23 = tt[2][x] - tt[3][x];
                ts += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
            }
            sum += ts >> 1;
        }
    }
    else
    {
        int n = s.w * s.h, inv = ((1 << 20) + s.w - 1) / s.w;
        for (int i = xa_lane(); i < n; i += XA_WAVE)
        {
            int y = (i * inv) >> 20, x = i - y * s.w;
            sum += abs((int)s.fencT[(s.fy + y) * 64 + s.fx + x] - me_pred<INWIN>(s, X0 + x, Y0 + y, xf, yf));
        }
    }
    return xa_wave_sum(sum);
}

#if XA_DEPTH == 8
/* ---- 8-bit fast paths: the staged window is read as dwords (4 samples), v_alignbyte realigns, v_dot4_u32_u8 does
 *      the 8-tap horizontal filter as (positive taps) - (negative taps) ---- */
XA_DEV uint32_t me_win_dword(const uint32_t* wd, int o)     /* samples o..o+3 of the window (o = byte offset) */
{
    const uint32_t* p = wd + (o >> 2);
    return __builtin_amdgcn_alignbyte(p[1], p[0], o & 3);
}

struct MeTaps { uint32_t pos[4][2], neg[4][2]; };
constexpr MeTaps me_make_taps()
{
    MeTaps t = {};
    constexpr int lf[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    for (int f = 0; f < 4; f++)
        for (int k = 0; k < 8; k++)
        {
            int c = lf[f][k];
            if (c > 0) t.pos[f][k >> 2] |= (uint32_t)c << (8 * (k & 3));
            if (c < 0) t.neg[f][k >> 2] |= (uint32_t)(-c) << (8 * (k & 3));
        }
    return t;
}
__device__ const MeTaps me_taps = me_make_taps();

/* vertical taps arranged by SOURCE row: c[yf][rr] packs, for output rows r = 0..3 of a 4x4 tile, the tap
 * g_lumaFilter[yf][rr - r] (0 when rr - r is outside 0..7) that source row rr (= by-3+rr) contributes */
struct MeVTaps { uint32_t c[11][4]; };    /* [source row][yf]: the row's four dwords are one uniform (scalar) load */
constexpr MeVTaps me_make_vtaps()
{
    MeVTaps t = {};
    constexpr int lf[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    for (int f = 0; f < 4; f++)
        for (int rr = 0; rr < 11; rr++)
            for (int r = 0; r < 4; r++)
            {
                int k = rr - r;
                int c = (k >= 0 && k < 8) ? lf[f][k] : 0;
                t.c[rr][f] |= (uint32_t)(uint8_t)(int8_t)c << (8 * r);
            }
    return t;
}
__device__ const MeVTaps me_vtaps = me_make_vtaps();

/* 8-tap horizontal filter sums of 4 consecutive outputs; d0..d2 hold samples x-3 .. x+8 of the row */
XA_DEV void me_hfilt4(uint32_t d0, uint32_t d1, uint32_t d2, int xf, int out[4])
{
    /* xf may differ from lane to lane (batched candidates): the packed taps are compile-time constants picked with
     * v_cndmask, not gathered from memory */
    constexpr MeTaps T = me_make_taps();
    const uint32_t pl = xf == 1 ? T.pos[1][0] : (xf == 2 ? T.pos[2][0] : T.pos[3][0]);
    const uint32_t ph = xf == 1 ? T.pos[1][1] : (xf == 2 ? T.pos[2][1] : T.pos[3][1]);
    const uint32_t nl = xf == 1 ? T.neg[1][0] : (xf == 2 ? T.neg[2][0] : T.neg[3][0]);
    const uint32_t nh = xf == 1 ? T.neg[1][1] : (xf == 2 ? T.neg[2][1] : T.neg[3][1]);
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        uint32_t a = j ? __builtin_amdgcn_alignbyte(d1, d0, j) : d0;
        uint32_t b = j ? __builtin_amdgcn_alignbyte(d2, d1, j) : d1;
        uint32_t pos = __builtin_amdgcn_udot4(a, pl, __builtin_amdgcn_udot4(b, ph, 0u, false), false);
        uint32_t neg = __builtin_amdgcn_udot4(a, nl, __builtin_amdgcn_udot4(b, nh, 0u, false), false);
        out[j] = (int)pos - (int)neg;
    }
}

/* prediction of one 4x4 tile whose top-left integer sample is window position (bx,by), fraction (xf,yf)/4:
 * luma_hpp / luma_vpp / luma_hvpp (ipfilter.cpp:79-120, :169-210, :370-378).  The vertical pass is streamed: each of
 * the 11 source rows is filtered horizontally once and immediately accumulated into the (up to 4) output rows it
 * contributes to, so only the 16 accumulators stay live. */
XA_DEV void me_pred_tile(const uint32_t* wd, int winW, int bx, int by, int xf, int yf, int pred[4][4])
{
    if (!yf)
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            if (!xf)
            {
                uint32_t d = me_win_dword(wd, (by + r) * winW + bx);
#pragma unroll
                for (int x = 0; x < 4; x++) pred[r][x] = (int)__builtin_amdgcn_ubfe(d, 8 * x, 8);
            }
            else
            {
                int o = (by + r) * winW + bx - 3;
                int hs[4];
                me_hfilt4(me_win_dword(wd, o), me_win_dword(wd, o + 4), me_win_dword(wd, o + 8), xf, hs);
#pragma unroll
                for (int x = 0; x < 4; x++)
                    pred[r][x] = xa_clip3(0, 255, (int)(int16_t)((hs[x] + 32) >> 6));
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int x = 0; x < 4; x++) pred[r][x] = 0;
    /* not unrolled on purpose: one source row in flight keeps the routine near 40 VGPRs (occupancy) */
#pragma unroll 1
    for (int rr = 0; rr < 11; rr++)     /* source rows by-3 .. by+7 */
    {
        int iv[4];
        if (!xf)
        {
            uint32_t d = me_win_dword(wd, (by - 3 + rr) * winW + bx);
#pragma unroll
            for (int x = 0; x < 4; x++) iv[x] = (int)__builtin_amdgcn_ubfe(d, 8 * x, 8);
        }
        else
        {
            /* hps with row extension: shift 0, offset -8192 at 8 bits (ipfilter.cpp:122-167) */
            int o = (by - 3 + rr) * winW + bx - 3;
            me_hfilt4(me_win_dword(wd, o), me_win_dword(wd, o + 4), me_win_dword(wd, o + 8), xf, iv);
#pragma unroll
            for (int x = 0; x < 4; x++) iv[x] = (int)(int16_t)(iv[x] - XA_IF_INTERNAL_OFFS);
        }
        /* taps of source row rr for output rows 0..3, one signed byte each.  rr is wave-uniform (scalar load of the row);
         * yf may differ from lane to lane (batched candidates): selected with two v_cndmask instead of a per-lane gather */
        const uint32_t c1 = me_vtaps.c[rr][1], c2 = me_vtaps.c[rr][2], c3 = me_vtaps.c[rr][3];
        const uint32_t cw = yf == 1 ? c1 : (yf == 2 ? c2 : c3);
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int c = (int)(int8_t)(cw >> (8 * r));
#pragma unroll
            for (int x = 0; x < 4; x++) pred[r][x] += iv[x] * c;
        }
    }
    /* vpp: (sum + 32) >> 6 ; vsp after hps: (sum + 2048 + (8192 << 6)) >> 12 */
    const int off = xf ? (1 << 11) + (XA_IF_INTERNAL_OFFS << XA_IF_FILTER_PREC) : 32, sh = xf ? 12 : 6;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int x = 0; x < 4; x++)
            pred[r][x] = xa_clip3(0, 255, (int)(int16_t)((pred[r][x] + off) >> sh));
}

/* subpelCompare (motion.cpp:1596-1623) on the LDS window, one lane per 4x4 tile */
template<bool SATD> XA_DEV int me_subpel_cmp_fast(const MeState& s, int qx, int qy)
{
    const uint32_t* wd = reinterpret_cast<const uint32_t*>(s.win);
    const uint32_t* fd = reinterpret_cast<const uint32_t*>(s.fencT);
    int X0 = s.px + (qx >> 2) - s.winX, Y0 = s.py + (qy >> 2) - s.winY, xf = qx & 3, yf = qy & 3, sum = 0;
    int tw = s.w >> 2, nt = tw * (s.h >> 2), inv = ((1 << 20) + tw - 1) / tw;
    for (int t = xa_lane(); t < nt; t += XA_WAVE)
    {
        int ty = (t * inv) >> 20, tx = t - ty * tw;
        int pred[4][4];
        me_pred_tile(wd, s.winW, X0 + 4 * tx, Y0 + 4 * ty, xf, yf, pred);
        int d[4][4];
#pragma unroll
        for (int y = 0; y < 4; y++)
        {
            uint32_t f = fd[(s.fy + 4 * ty + y) * 16 + (s.fx >> 2) + tx];
#pragma unroll
            for (int x = 0; x < 4; x++)
                d[y][x] = (int)__builtin_amdgcn_ubfe(f, 8 * x, 8) - pred[y][x];
        }
        if (SATD)
        {
            int tt[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                int s01 = d[y][0] + d[y][1], e01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], e23 = d[y][2] - d[y][3];
                tt[y][0] = s01 + s23; tt[y][1] = s01 - s23; tt[y][2] = e01 + e23; tt[y][3] = e01 - e23;
            }
            int ts = 0;
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                int s01 = tt[0][x] + tt[1][x], e01 = tt[0][x] - tt[1][x], s23 = tt[2][x] + tt[3][x], e23 = tt[2][x] - tt[3][x];
                ts += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
            }
            sum += ts >> 1;
        }
        else
        {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) sum += abs(d[y][x]);
        }
    }
    return xa_wave_sum(sum);
}
#endif /* XA_DEPTH == 8 */

template<bool SATD, bool SLOW, bool INL> ME_HELPER int me_subpel_b(int sOff, int qx, int qy)
{
    const MeState& s = ME_S(sOff);
    int chromaCost = 0;
    if constexpr (!INL) { if (s.chroma) chromaCost = me_chroma_satd(sOff, qx, qy); }      /* the inlined variant never sees chroma jobs */
    if (!SATD && !((qx | qy) & 3)) return me_sad_at_f<SLOW, INL>(sOff, qx >> 2, qy >> 2) + chromaCost;
    if (SLOW) return me_subpel_cmp<false, SATD>(sOff, qx, qy) + chromaCost;
    int X0 = s.px + (qx >> 2), Y0 = s.py + (qy >> 2);
    /* +8 on the right: the dword reads of the fast path touch samples up to x+8 of the last tile */
    if (me_inwin(s, X0 - 3, Y0 - 3, X0 + s.w + 8, Y0 + s.h + 4))
    {
#if XA_DEPTH == 8
        return me_subpel_cmp_fast<SATD>(s, qx, qy) + chromaCost;
#else
        return me_subpel_cmp<true, SATD>(sOff, qx, qy) + chromaCost;
#endif
    }
    ME_OOB(sOff);
    return ME_OOB_COST;
}

__device__ const int8_t me_hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
__device__ const uint8_t me_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
__device__ const int8_t me_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
__device__ const int8_t me_offsets[16][2] = { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
                                              { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } };
/* motion.cpp:48-58: hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd */
__device__ const uint8_t me_workload[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                               { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };

/* SAD + MV cost of up to four full-pel candidates in one call (the reference's sad_x3 / sad_x4 batches,
 * motion.cpp:271-360).  Candidates are packed (my << 16) | (mx & 0xffff).  Returns, in lane k, the cost of candidate k.
 * Small PUs run several candidates side by side in one pass: a PU with <= 16 four-sample groups (8x8) evaluates
 * four candidates at once in the four 16-lane quarters of the wavefront. */
#define ME_PK(mx, my) ((int)(((uint32_t)(my) << 16) | ((uint32_t)(mx) & 0xffffu)))
template<bool SLOW, bool INL> ME_HELPER int me_cost_multi_b(int sOff, int n, int m0, int m1, int m2, int m3)
{
    const MeState& s = ME_S(sOff);
    const int lane = xa_lane();
    int res = 0x7fffffff;
#if XA_DEPTH == 8
    const int w = s.w, h = s.h, gpr = w >> 2, ng = gpr * h;
    bool fast = !SLOW && (w & (w - 1)) == 0;
    for (int k = 0; k < n; k++)
    {
        int mk = k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : m3;
        int X = s.px + (int)(int16_t)(mk & 0xffff), Y = s.py + (mk >> 16);
        fast = fast && me_inwin(s, X, Y, X + w + 4, Y + h);
    }
    if (fast)
    {
        const uint32_t* wd = reinterpret_cast<const uint32_t*>(s.win);
        const uint32_t* fd = reinterpret_cast<const uint32_t*>(s.fencT);
        const int lg = 31 - __clz(gpr);
        const int lgseg = ng <= 16 ? 4 : ng <= 32 ? 5 : 6, seg = 1 << lgseg, cpp = 64 >> lgseg;
        for (int c0 = 0; c0 < n; c0 += cpp)
        {
            int cl = c0 + (lane >> lgseg);
            int mk = cl == 0 ? m0 : cl == 1 ? m1 : cl == 2 ? m2 : m3;
            if (cl >= n) mk = m0;
            int mx = (int)(int16_t)(mk & 0xffff), my = mk >> 16;
            int X = s.px + mx - s.winX, Y = s.py + my - s.winY, sum = 0;
            for (int gi = lane & (seg - 1); gi < ng; gi += seg)
            {
                int y = gi >> lg, x4 = gi & (gpr - 1);
                uint32_t fv = fd[(s.fy + y) * 16 + (s.fx >> 2) + x4];
                uint32_t rv = me_win_dword(wd, (Y + y) * s.winW + X + 4 * x4);
                sum = __builtin_amdgcn_sad_u8(fv, rv, sum);
            }
            sum = xa_row16_sum(sum);                    /* every lane: sum of its 16-lane row (DPP) */
            if (lgseg >= 5) sum += __shfl_xor(sum, 16, 64);
            if (lgseg >= 6) sum += __shfl_xor(sum, 32, 64);
            int cost = sum + me_mvcost(s, mx * 4, my * 4);
            for (int jj = 0; jj < cpp && c0 + jj < n; jj++)
            {
                int v = __shfl(cost, jj << lgseg, 64);
                if (lane == c0 + jj) res = v;
            }
        }
        return res;
    }
#endif
    for (int k = 0; k < n; k++)
    {
        int mk = k == 0 ? m0 : k == 1 ? m1 : k == 2 ? m2 : m3;
        int mx = (int)(int16_t)(mk & 0xffff), my = mk >> 16;
        int c = me_sad_at_f<SLOW, INL>(sOff, mx, my) + me_mvcost(s, mx * 4, my * 4);
        if (lane == k) res = c;
    }
    return res;
}
XA_DEV int ME_LANE(int v, int k) { return __builtin_amdgcn_readlane(v, k); }

/* Sub-pel comparison (subpelCompare, motion.cpp:1596-1623) of up to 16 quarter-pel candidates in one call: the
 * candidates are read from MeState::cand[0..n), lane k returns the SAD/SATD of candidate k (MV cost not included).  All
 * (candidate, 4x4 tile) pairs are spread over the lanes -- an 8x8 PU with 4 candidates keeps 16 lanes busy instead of 4, a
 * 16x16 PU all 64 -- and summed per candidate with LDS atomics.  Evaluation has no side effects, so callers may evaluate
 * candidates the reference would skip and apply the reference's tests and selection order afterwards. */
template<bool SATD, bool SLOW, bool INL> ME_HELPER int me_subpel_list_b(int sOff, int n)
{
    const MeState& s = ME_S(sOff);
    const int lane = xa_lane();
#if XA_DEPTH == 8
    bool fast = !SLOW;
    for (int k = 0; k < n && fast; k++)
    {
        int qx = (int)(int16_t)(s.cand[k] & 0xffff), qy = s.cand[k] >> 16;
        int X0 = s.px + (qx >> 2), Y0 = s.py + (qy >> 2);
        fast = me_inwin(s, X0 - 3, Y0 - 3, X0 + s.w + 8, Y0 + s.h + 4);
    }
    if (fast)
    {
        int* acc = reinterpret_cast<int*>(me_smem + sOff + offsetof(MeState, acc));
        if (lane < 16) acc[lane] = 0;
        xa_wave_sync();
        const uint32_t* wd = reinterpret_cast<const uint32_t*>(s.win);
        const uint32_t* fd = reinterpret_cast<const uint32_t*>(s.fencT);
        const int tw = s.w >> 2, nt = tw * (s.h >> 2), invT = ((1 << 20) + tw - 1) / tw, invN = ((1 << 20) + nt - 1) / nt;
        for (int item = lane; item < nt * n; item += XA_WAVE)
        {
            int c = (item * invN) >> 20, t = item - c * nt;
            int ty = (t * invT) >> 20, tx = t - ty * tw;
            int qx = (int)(int16_t)(s.cand[c] & 0xffff), qy = s.cand[c] >> 16;
            int pred[4][4];
            me_pred_tile(wd, s.winW, s.px + (qx >> 2) - s.winX + 4 * tx, s.py + (qy >> 2) - s.winY + 4 * ty, qx & 3, qy & 3, pred);
            int d[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                uint32_t f = fd[(s.fy + 4 * ty + y) * 16 + (s.fx >> 2) + tx];
#pragma unroll
                for (int x = 0; x < 4; x++)
                    d[y][x] = (int)__builtin_amdgcn_ubfe(f, 8 * x, 8) - pred[y][x];
            }
            int cost = 0;
            if (SATD)
            {
                int tt[4][4];
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    int s01 = d[y][0] + d[y][1], e01 = d[y][0] - d[y][1], s23 = d[y][2] + d[y][3], e23 = d[y][2] - d[y][3];
                    tt[y][0] = s01 + s23; tt[y][1] = s01 - s23; tt[y][2] = e01 + e23; tt[y][3] = e01 - e23;
                }
#pragma unroll
                for (int x = 0; x < 4; x++)
                {
                    int s01 = tt[0][x] + tt[1][x], e01 = tt[0][x] - tt[1][x], s23 = tt[2][x] + tt[3][x], e23 = tt[2][x] - tt[3][x];
                    cost += abs(s01 + s23) + abs(s01 - s23) + abs(e01 + e23) + abs(e01 - e23);
                }
                cost >>= 1;
            }
            else
            {
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) cost += abs(d[y][x]);
            }
            atomicAdd(&acc[c], cost);
        }
        xa_wave_sync();
        int res = lane < n ? acc[lane] : ME_OOB_COST;
        xa_wave_sync();
        if constexpr (!INL)
        {
            if (s.chroma)
                for (int k = 0; k < n; k++)
                {
                    int cc = me_chroma_satd(sOff, (int)(int16_t)(s.cand[k] & 0xffff), s.cand[k] >> 16);
                    if (lane == k) res += cc;
                }
        }
        return res;
    }
#endif
    int res = ME_OOB_COST;
    for (int k = 0; k < n; k++)
    {
        int c = me_subpel_f<SATD, SLOW, INL>(sOff, (int)(int16_t)(s.cand[k] & 0xffff), s.cand[k] >> 16);
        if (lane == k) res = c;
    }
    return res;
}

/* real-function wrappers and the dispatchers */
template<bool SLOW> __device__ __noinline__ int me_sad_at_w(int sOff, int mx, int my) { return me_sad_at_b<SLOW, false>(sOff, mx, my); }
template<bool SATD, bool SLOW> __device__ __noinline__ int me_subpel_w(int sOff, int qx, int qy) { return me_subpel_b<SATD, SLOW, false>(sOff, qx, qy); }
template<bool SLOW> __device__ __noinline__ int me_cost_multi_w(int sOff, int n, int m0, int m1, int m2, int m3) { return me_cost_multi_b<SLOW, false>(sOff, n, m0, m1, m2, m3); }
template<bool SATD, bool SLOW> __device__ __noinline__ int me_subpel_list_w(int sOff, int n) { return me_subpel_list_b<SATD, SLOW, false>(sOff, n); }
template<bool SLOW, bool INL> XA_DEV int me_sad_at_f(int sOff, int mx, int my)
{ if constexpr (INL) return me_sad_at_b<SLOW, true>(sOff, mx, my); else return me_sad_at_w<SLOW>(sOff, mx, my); }
template<bool SATD, bool SLOW, bool INL> XA_DEV int me_subpel_f(int sOff, int qx, int qy)
{ if constexpr (INL) return me_subpel_b<SATD, SLOW, true>(sOff, qx, qy); else return me_subpel_w<SATD, SLOW>(sOff, qx, qy); }
template<bool SLOW, bool INL> XA_DEV int me_cost_multi_f(int sOff, int n, int m0, int m1, int m2, int m3)
{ if constexpr (INL) return me_cost_multi_b<SLOW, true>(sOff, n, m0, m1, m2, m3); else return me_cost_multi_w<SLOW>(sOff, n, m0, m1, m2, m3); }
template<bool SATD, bool SLOW, bool INL> XA_DEV int me_subpel_list_f(int sOff, int n)
{ if constexpr (INL) return me_subpel_list_b<SATD, SLOW, true>(sOff, n); else return me_subpel_list_w<SATD, SLOW>(sOff, n); }

/* One refinement round of motion.cpp:1535-1586: costs (comparison + MV cost) of the `dirs` (4 or 8) neighbours
 * square1[1..dirs] * step of the quarter-pel MV (bqx,bqy); lane i-1 returns the cost of direction i. */
template<bool SATD, bool SLOW, bool INL> XA_DEV int me_subpel_dirs_f(int sOff, int bqx, int bqy, int step, int dirs)
{
    const MeState& s = ME_S(sOff);
    const int lane = xa_lane();
    const int qx = bqx + (lane < dirs ? me_square1[lane + 1][0] * step : 0), qy = bqy + (lane < dirs ? me_square1[lane + 1][1] * step : 0);
    int* cand = reinterpret_cast<int*>(me_smem + sOff + offsetof(MeState, cand));
    if (lane < dirs) cand[lane] = ME_PK(qx, qy);
    xa_wave_sync();
    int v = me_subpel_list_f<SATD, SLOW, INL>(sOff, dirs);
    return lane < dirs ? v + me_mvcost(s, qx, qy) : ME_OOB_COST;
}

#define me_subpel_sad(qx, qy) me_subpel_f<false, SLOW, INL>(sOff, qx, qy)
#define me_subpel_satd(qx, qy) me_subpel_f<true, SLOW, INL>(sOff, qx, qy)
XA_DEV bool me_in_range(const MeState& s, int x, int y) { return x >= s.mnx && x <= s.mxx && y >= s.mny && y <= s.mxy; }

#define ME_COST(mx, my) ME_LANE((me_cost_multi_f<SLOW, INL>(sOff, 1, ME_PK(mx, my), 0, 0, 0)), 0)
#define me_sad_at(s, mx, my) me_sad_at_f<SLOW, INL>(sOff, mx, my)
#define ME_COST_MV(mx, my) do { int c_ = ME_COST(mx, my); if (c_ < bcost) { bcost = c_; bx = (mx); by = (my); } } while (0)
#define ME_COST_PT(mx, my, point, dist) do { int c_ = ME_COST(mx, my); if (c_ < bcost) { bcost = c_; bx = (mx); by = (my); bPointNr = (point); bDistance = (dist); } } while (0)

/* StarPatternSearch: motion.cpp:387-629 (the x4 batches evaluate the same points in the same order) */
template<bool SLOW> __device__ __noinline__ void me_star_pattern(int sOff, int& bx, int& by, int& bcost, int& bPointNr, int& bDistance, int earlyExitIters, int merange)
{
    constexpr bool INL = false;
    const MeState& s = ME_S(sOff);
    const int ox = bx, oy = by;
    int saved = bcost, rounds = 0;
    {
        const int dist = 1;
        int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        bool all = top >= s.mny && left >= s.mnx && right <= s.mxx && bottom <= s.mxy;
        if (all || top >= s.mny) ME_COST_PT(ox, top, 2, dist);
        if (all || left >= s.mnx) ME_COST_PT(left, oy, 4, dist);
        if (all || right <= s.mxx) ME_COST_PT(right, oy, 5, dist);
        if (all || bottom <= s.mxy) ME_COST_PT(ox, bottom, 7, dist);
        if (bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        int top2 = oy - (dist >> 1), bottom2 = oy + (dist >> 1), left2 = ox - (dist >> 1), right2 = ox + (dist >> 1);
        saved = bcost;
        bool all = top >= s.mny && left >= s.mnx && right <= s.mxx && bottom <= s.mxy;
        if (all)
        {
            ME_COST_PT(ox, top, 2, dist);
            ME_COST_PT(left2, top2, 1, dist >> 1);
            ME_COST_PT(right2, top2, 3, dist >> 1);
            ME_COST_PT(left, oy, 4, dist);
            ME_COST_PT(right, oy, 5, dist);
            ME_COST_PT(left2, bottom2, 6, dist >> 1);
            ME_COST_PT(right2, bottom2, 8, dist >> 1);
            ME_COST_PT(ox, bottom, 7, dist);
        }
        else
        {
            if (top >= s.mny) ME_COST_PT(ox, top, 2, dist);
            if (top2 >= s.mny)
            {
                if (left2 >= s.mnx) ME_COST_PT(left2, top2, 1, (dist >> 1));
                if (right2 <= s.mxx) ME_COST_PT(right2, top2, 3, (dist >> 1));
            }
            if (left >= s.mnx) ME_COST_PT(left, oy, 4, dist);
            if (right <= s.mxx) ME_COST_PT(right, oy, 5, dist);
            if (bottom2 <= s.mxy)
            {
                if (left2 >= s.mnx) ME_COST_PT(left2, bottom2, 6, (dist >> 1));
                if (right2 <= s.mxx) ME_COST_PT(right2, bottom2, 8, (dist >> 1));
            }
            if (bottom <= s.mxy) ME_COST_PT(ox, bottom, 7, dist);
        }
        if (bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {
        int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        saved = bcost;
        bool all = top >= s.mny && left >= s.mnx && right <= s.mxx && bottom <= s.mxy;
        if (all || top >= s.mny) ME_COST_PT(ox, top, 0, dist);
        if (all || left >= s.mnx) ME_COST_PT(left, oy, 0, dist);
        if (all || right <= s.mxx) ME_COST_PT(right, oy, 0, dist);
        if (all || bottom <= s.mxy) ME_COST_PT(ox, bottom, 0, dist);
        for (int index = 1; index < 4; index++)
        {
            int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
            int posXL = ox - ((dist >> 2) * index), posXR = ox + ((dist >> 2) * index);
            if (all || posYT >= s.mny)
            {
                if (all || posXL >= s.mnx) ME_COST_PT(posXL, posYT, 0, dist);
                if (all || posXR <= s.mxx) ME_COST_PT(posXR, posYT, 0, dist);
            }
            if (all || posYB <= s.mxy)
            {
                if (all || posXL >= s.mnx) ME_COST_PT(posXL, posYB, 0, dist);
                if (all || posXR <= s.mxx) ME_COST_PT(posXR, posYB, 0, dist);
            }
        }
        if (bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}

/* MotionEstimate::motionEstimate: motion.cpp:764-1594 (full-resolution reference, one slice, luma only) */
/* `jp` points at the job record in HBM: its fields are wave-uniform scalar loads (a by-value copy would live in scratch
 * because mvc[] is indexed dynamically -- measured as 457 MB of scratch writes per 1080p launch) */
template<bool SLOW, bool STAR> __device__ void me_search(int sOff, const x265amd_me_job* __restrict__ jp, x265amd_me_result* out)
{
    constexpr bool INL = !SLOW && !STAR;    /* the HEX/DIA window-resident variant inlines the helper bodies */
    if constexpr (INL)
    {
        /* that variant carries neither the star search nor chroma SATD: such jobs are redone by k_me_deferred */
        if (ME_S(sOff).chroma)
        {
            if (xa_lane() == 0) { out->mv[0] = 0; out->mv[1] = 0; out->cost = ME_DEFERRED; }
            return;
        }
    }
    const MeState& s = ME_S(sOff);
    const int qminx = s.mnx * 4, qminy = s.mny * 4, qmaxx = s.mxx * 4, qmaxy = s.mxy * 4;
    const int merange = jp->merange, numCand = jp->num_cand, method = jp->method & 0x7f, subme = jp->subme;
    /* motion.cpp:797-846: predictor, zero MV, candidates */
    int pmx = xa_clip3(qminx, qmaxx, s.mvpx), pmy = xa_clip3(qminy, qmaxy, s.mvpy);
    int bestprex = pmx, bestprey = pmy;
    /* the clipped predictor and every clipped candidate are compared in ONE batched call (evaluation has no side effects;
     * the reference's skip tests and update order are applied below on the returned values) */
    {
        int* cand = reinterpret_cast<int*>(me_smem + sOff + offsetof(MeState, cand));
        const int lane = xa_lane();
        if (lane == 0) cand[0] = ME_PK(pmx, pmy);
        if (lane >= 1 && lane <= numCand)
            cand[lane] = ME_PK(xa_clip3(qminx, qmaxx, jp->mvc[lane - 1][0]), xa_clip3(qminy, qmaxy, jp->mvc[lane - 1][1]));
        xa_wave_sync();
    }
    const int preSads = me_subpel_list_f<false, SLOW, INL>(sOff, 1 + numCand);
    int bprecost = __shfl(preSads, 0, 64);
    int bx = (pmx + 2) >> 2, by = (pmy + 2) >> 2;
    int bcost = bprecost;
    if ((pmx | pmy) & 3)
        bcost = ME_COST(bx, by);
    if (pmx | pmy)
    {
        int cost = ME_COST(0, 0);
        if (cost < bcost)
        {
            bcost = cost;
            bx = 0;
            by = max(min(0, s.mxy), s.mny);
        }
    }
    for (int i = 0; i < numCand; i++)
    {
        int cx = xa_clip3(qminx, qmaxx, jp->mvc[i][0]), cy = xa_clip3(qminy, qmaxy, jp->mvc[i][1]);
        if ((cx | cy) && (cx != pmx || cy != pmy) && (cx != bestprex || cy != bestprey))
        {
            int cost = __shfl(preSads, i + 1, 64) + me_mvcost(s, cx, cy);
            if (cost < bprecost) { bprecost = cost; bestprex = cx; bestprey = cy; }
        }
    }

    switch (method)
    {
    case X265AMD_ME_DIA:    /* motion.cpp:855-877 */
    {
        int i = merange;
        do
        {
            int cv = me_cost_multi_f<SLOW, INL>(sOff, 4, ME_PK(bx, by - 1), ME_PK(bx, by + 1), ME_PK(bx - 1, by), ME_PK(bx + 1, by));
            int c0 = ME_LANE(cv, 0), c1 = ME_LANE(cv, 1), c2 = ME_LANE(cv, 2), c3 = ME_LANE(cv, 3);
            int packed = bcost << 4;
            if (by - 1 >= s.mny && by - 1 <= s.mxy && (c0 << 4) + 1 < packed) packed = (c0 << 4) + 1;
            if (by + 1 >= s.mny && by + 1 <= s.mxy && (c1 << 4) + 3 < packed) packed = (c1 << 4) + 3;
            if ((c2 << 4) + 4 < packed) packed = (c2 << 4) + 4;
            if ((c3 << 4) + 12 < packed) packed = (c3 << 4) + 12;
            bcost = packed >> 4;
            if (!(packed & 15)) break;
            bx -= (int)((uint32_t)packed << 28) >> 30;
            by -= (int)((uint32_t)packed << 30) >> 30;
        }
        while (--i && me_in_range(s, bx, by));
        break;
    }
    case X265AMD_ME_HEX:    /* motion.cpp:879-987 */
    {
        int c0, c1, c2, c3, packed, dir, cv;
        cv = me_cost_multi_f<SLOW, INL>(sOff, 3, ME_PK(bx - 2, by), ME_PK(bx - 1, by + 2), ME_PK(bx + 1, by + 2), 0);
        c0 = ME_LANE(cv, 0); c1 = ME_LANE(cv, 1); c2 = ME_LANE(cv, 2);
        packed = bcost << 3;
        if (by >= s.mny && by <= s.mxy && (c0 << 3) + 2 < packed) packed = (c0 << 3) + 2;
        if (by + 2 >= s.mny && by + 2 <= s.mxy)
        {
            if ((c1 << 3) + 3 < packed) packed = (c1 << 3) + 3;
            if ((c2 << 3) + 4 < packed) packed = (c2 << 3) + 4;
        }
        cv = me_cost_multi_f<SLOW, INL>(sOff, 3, ME_PK(bx + 2, by), ME_PK(bx + 1, by - 2), ME_PK(bx - 1, by - 2), 0);
        c0 = ME_LANE(cv, 0); c1 = ME_LANE(cv, 1); c2 = ME_LANE(cv, 2);
        if (by >= s.mny && by <= s.mxy && (c0 << 3) + 5 < packed) packed = (c0 << 3) + 5;
        if (by - 2 >= s.mny && by - 2 <= s.mxy)
        {
            if ((c1 << 3) + 6 < packed) packed = (c1 << 3) + 6;
            if ((c2 << 3) + 7 < packed) packed = (c2 << 3) + 7;
        }
        if (packed & 7)
        {
            dir = (packed & 7) - 2;
            if (by + me_hex2[dir + 1][1] >= s.mny && by + me_hex2[dir + 1][1] <= s.mxy)
            {
                bx += me_hex2[dir + 1][0]; by += me_hex2[dir + 1][1];
                for (int i = (merange >> 1) - 1; i > 0 && me_in_range(s, bx, by); i--)
                {
                    cv = me_cost_multi_f<SLOW, INL>(sOff, 3, ME_PK(bx + me_hex2[dir][0], by + me_hex2[dir][1]), ME_PK(bx + me_hex2[dir + 1][0], by + me_hex2[dir + 1][1]),
                                         ME_PK(bx + me_hex2[dir + 2][0], by + me_hex2[dir + 2][1]), 0);
                    int cc[3] = { ME_LANE(cv, 0), ME_LANE(cv, 1), ME_LANE(cv, 2) };
                    packed &= ~7;
                    for (int k = 0; k < 3; k++)
                        if (by + me_hex2[dir + k][1] >= s.mny && by + me_hex2[dir + k][1] <= s.mxy && (cc[k] << 3) + k + 1 < packed)
                            packed = (cc[k] << 3) + k + 1;
                    if (!(packed & 7)) break;
                    dir += (packed & 7) - 2;
                    dir = me_mod6m1[dir + 1];
                    bx += me_hex2[dir + 1][0]; by += me_hex2[dir + 1][1];
                }
            }
        }
        bcost = packed >> 3;
        /* square refine */
        dir = 0;
        cv = me_cost_multi_f<SLOW, INL>(sOff, 4, ME_PK(bx, by - 1), ME_PK(bx, by + 1), ME_PK(bx - 1, by), ME_PK(bx + 1, by));
        c0 = ME_LANE(cv, 0); c1 = ME_LANE(cv, 1); c2 = ME_LANE(cv, 2); c3 = ME_LANE(cv, 3);
        bool upOk = by - 1 >= s.mny && by - 1 <= s.mxy, dnOk = by + 1 >= s.mny && by + 1 <= s.mxy;
        if (upOk && c0 < bcost) { bcost = c0; dir = 1; }
        if (dnOk && c1 < bcost) { bcost = c1; dir = 2; }
        if (c2 < bcost) { bcost = c2; dir = 3; }
        if (c3 < bcost) { bcost = c3; dir = 4; }
        cv = me_cost_multi_f<SLOW, INL>(sOff, 4, ME_PK(bx - 1, by - 1), ME_PK(bx - 1, by + 1), ME_PK(bx + 1, by - 1), ME_PK(bx + 1, by + 1));
        c0 = ME_LANE(cv, 0); c1 = ME_LANE(cv, 1); c2 = ME_LANE(cv, 2); c3 = ME_LANE(cv, 3);
        if (upOk && c0 < bcost) { bcost = c0; dir = 5; }
        if (dnOk && c1 < bcost) { bcost = c1; dir = 6; }
        if (upOk && c2 < bcost) { bcost = c2; dir = 7; }
        if (dnOk && c3 < bcost) { bcost = c3; dir = 8; }
        bx += me_square1[dir][0]; by += me_square1[dir][1];
        break;
    }
    case X265AMD_ME_STAR:   /* motion.cpp:1156-1265 */
    if constexpr (STAR)
    {
        int bPointNr = 0, bDistance = 0;
        bool stop = false;
        me_star_pattern<SLOW>(sOff, bx, by, bcost, bPointNr, bDistance, 3, merange);
        if (bDistance == 1)
        {
            if (!bPointNr) stop = true;
            else
            {
                int saved = bcost;
                int x1 = bx + me_offsets[(bPointNr - 1) * 2][0], y1 = by + me_offsets[(bPointNr - 1) * 2][1];
                int x2 = bx + me_offsets[(bPointNr - 1) * 2 + 1][0], y2 = by + me_offsets[(bPointNr - 1) * 2 + 1][1];
                if (me_in_range(s, x1, y1)) ME_COST_MV(x1, y1);
                if (me_in_range(s, x2, y2)) ME_COST_MV(x2, y2);
                if (bcost == saved) stop = true;
            }
        }
        if (stop) break;
        const int RasterDistance = 5;
        if (bDistance > RasterDistance)
        {
            for (int ty = s.mny; ty <= s.mxy; ty += RasterDistance)
                for (int tx = s.mnx; tx <= s.mxx; tx += RasterDistance)
                {
                    if (tx + RasterDistance * 3 <= s.mxx)
                    {
                        int c[4];
                        for (int k = 0; k < 4; k++) c[k] = me_sad_at(s, tx + RasterDistance * k, ty);
                        c[0] += me_mvcost(s, tx * 4, ty * 4);
                        if (c[0] < bcost) { bcost = c[0]; bx = tx; by = ty; }
                        tx += RasterDistance;
                        c[1] += me_mvcost(s, tx * 4, ty * 4);
                        if (c[1] < bcost) { bcost = c[1]; bx = tx; by = ty; }
                        tx += RasterDistance;
                        c[2] += me_mvcost(s, tx * 4, ty * 4);
                        if (c[2] < bcost) { bcost = c[2]; bx = tx; by = ty; }
                        tx += RasterDistance;
                        c[3] += me_mvcost(s, tx * 8, ty * 8);       /* sic: the reference shifts this one by 3 (motion.cpp:1219) */
                        if (c[3] < bcost) { bcost = c[3]; bx = tx; by = ty; }
                    }
                    else
                        ME_COST_MV(tx, ty);
                }
        }
        while (bDistance > 0)
        {
            bDistance = 0; bPointNr = 0;
            me_star_pattern<SLOW>(sOff, bx, by, bcost, bPointNr, bDistance, 32, merange);
            if (bDistance == 1)
            {
                if (!bPointNr) break;
                int x1 = bx + me_offsets[(bPointNr - 1) * 2][0], y1 = by + me_offsets[(bPointNr - 1) * 2][1];
                int x2 = bx + me_offsets[(bPointNr - 1) * 2 + 1][0], y2 = by + me_offsets[(bPointNr - 1) * 2 + 1][1];
                if (me_in_range(s, x1, y1)) ME_COST_MV(x1, y1);
                if (me_in_range(s, x2, y2)) ME_COST_MV(x2, y2);
                break;
            }
        }
        break;
    }
    else
    {
        ME_OOB(sOff);       /* this kernel variant was built without the star search: redo in k_me_deferred */
        break;
    }
    default:    /* UMH / SEA / FULL are not implemented: flagged, never silently replaced */
        if (xa_lane() == 0) { out->mv[0] = 0; out->mv[1] = 0; out->cost = -1; }
        return;
    }

    /* motion.cpp:1473-1594 */
    if (bprecost < bcost) { bx = bestprex; by = bestprey; bcost = bprecost; }
    else { bx *= 4; by *= 4; }
    const uint8_t* wl = me_workload[subme];
    if (!bcost)
        bcost = me_mvcost(s, bx, by);
    else
    {
        bool hsatd = wl[4] != 0;
        if (hsatd)
            bcost = me_subpel_satd(bx, by) + me_mvcost(s, bx, by);
        for (int iter = 0; iter < wl[0]; iter++)
        {
            int bdir = 0;
            int cv = hsatd ? me_subpel_dirs_f<true, SLOW, INL>(sOff, bx, by, 2, wl[1]) : me_subpel_dirs_f<false, SLOW, INL>(sOff, bx, by, 2, wl[1]);
            for (int i = 1; i <= wl[1]; i++)
            {
                int qy = by + me_square1[i][1] * 2;
                if (qy < qminy || qy > qmaxy) continue;
                int cost = __shfl(cv, i - 1, 64);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += me_square1[bdir][0] * 2; by += me_square1[bdir][1] * 2; }
            else break;
        }
        if (!hsatd)
            bcost = me_subpel_satd(bx, by) + me_mvcost(s, bx, by);
        for (int iter = 0; iter < wl[2]; iter++)
        {
            int bdir = 0;
            int cv = me_subpel_dirs_f<true, SLOW, INL>(sOff, bx, by, 1, wl[3]);
            for (int i = 1; i <= wl[3]; i++)
            {
                int qy = by + me_square1[i][1];
                if (qy < qminy || qy > qmaxy) continue;
                int cost = __shfl(cv, i - 1, 64);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bx += me_square1[bdir][0]; by += me_square1[bdir][1]; }
            else break;
        }
    }
    if (!SLOW && s.oob) { bx = 0; by = 0; bcost = ME_DEFERRED; }
    if (xa_lane() == 0) { out->mv[0] = (int16_t)bx; out->mv[1] = (int16_t)by; out->cost = bcost; }
}

/* fills the per-wave state for one job (lane 0) */
XA_DEV void me_set_job(MeState& s, const x265amd_me_job& j, const x265amd_me_group& g, const MeParams& p, const pixel* win, int winW, int winH,
                       const pixel* fencT, const pixel* refG)
{
    s.win = win; s.winX = g.win_x; s.winY = g.win_y; s.winW = winW; s.winH = winH;
    s.fencT = fencT; s.refG = refG; s.stride = p.stride;
    s.px = j.x; s.py = j.y; s.w = j.w; s.h = j.h;
    s.fx = j.x - g.fenc_x; s.fy = j.y - g.fenc_y;
    s.cost = p.tables + (size_t)j.qp * ME_TBL_LEN + ME_TBL_HALF;
    s.mvpx = j.mvp[0]; s.mvpy = j.mvp[1];
    s.mnx = j.mvmin[0]; s.mny = j.mvmin[1]; s.mxx = j.mvmax[0]; s.mxy = j.mvmax[1];
    s.oob = 0;
    /* bChromaSATD = requested && subpelRefine > 2 && chroma PU a multiple of 4x4 (NULL chromaSatd otherwise) */
    s.chroma = p.chroma && (j.method & X265AMD_ME_CHROMA_SATD) && j.subme > 2 && ((((j.w >> 1) | (j.h >> 1)) & 3) == 0);
    s.cstride = p.cstride;
    if (s.chroma)
    {
        const long coff = (long)(j.y >> 1) * p.cstride + (j.x >> 1);
        for (int c = 0; c < 2; c++)
        {
            s.fencC[c] = reinterpret_cast<const pixel*>(p.chroma[c]) + coff;
            s.refC[c] = reinterpret_cast<const pixel*>(p.chroma[2 + 2 * g.ref + c]) + coff;
        }
    }
}

/* window-resident kernel: one workgroup per group, window + source tile in LDS, one wavefront per job */
/* body of one workgroup of NT threads (a multiple of 64) serving group `vb`; the ordinary kernel and the device job server share it */
template<bool STAR> XA_DEV void block_me_search(const MeParams& p, int vb, int tid, int nthr)
{
    char* smem = me_smem;
    pixel* win = reinterpret_cast<pixel*>(smem);
    pixel* fencT = win + p.maxWinW * p.maxWinH + 16;     /* +16: the dword reads may run past the last window sample */
    int* counter = reinterpret_cast<int*>(fencT + 64 * 64);
    const int sOff = (int)(reinterpret_cast<char*>(counter + 4) - smem) + (tid >> 6) * (int)sizeof(MeState);
    MeState& s = *reinterpret_cast<MeState*>(smem + sOff);

    const x265amd_me_group g = p.groups[vb];
    const pixel* refG = reinterpret_cast<const pixel*>(p.refs[g.ref]);
    if (tid == 0) XA_BYTES(((unsigned long long)g.win_w * g.win_h + 64 * 64) * sizeof(pixel) + (unsigned long long)g.num_jobs * (sizeof(x265amd_me_job) + 8));     /* window + source tile + job / result records */

    /* stage the reference window: rows of win_w samples, 4 samples per lane, coalesced along the row */
    {
        int gpr = g.win_w >> 2, total = gpr * g.win_h;
        for (int i = tid; i < total; i += nthr)
        {
            int y = i / gpr, x = (i - y * gpr) << 2;
            const pixel* src = refG + (long)(g.win_y + y) * p.stride + g.win_x + x;
            pixel v[4];
            __builtin_memcpy(v, src, sizeof(v));
            __builtin_memcpy(win + y * g.win_w + x, v, sizeof(v));
        }
        for (int i = tid; i < 16 * 64; i += nthr)
        {
            int y = i >> 4, x = (i & 15) << 2;
            pixel v[4];
            __builtin_memcpy(v, p.fenc + (long)(g.fenc_y + y) * p.stride + g.fenc_x + x, sizeof(v));
            __builtin_memcpy(fencT + y * 64 + x, v, sizeof(v));
        }
        if (tid == 0) *counter = 0;
    }
    __syncthreads();

    const int lane = xa_lane();
    for (;;)
    {
        int ji = 0;
        if (lane == 0) ji = atomicAdd(counter, 1);
        ji = __shfl(ji, 0, 64);
        if (ji >= g.num_jobs) break;
        const x265amd_me_job* jp = p.jobs + g.first_job + ji;
        if (lane == 0) me_set_job(s, *jp, g, p, win, g.win_w, g.win_h, fencT, refG);
        xa_wave_sync();
        me_search<false, STAR>(sOff, jp, p.out + g.first_job + ji);
        xa_wave_sync();
    }
}

/* The groups of one command side by side (device job server: the searches of ONE prediction unit in its reference pictures arrive as groups of one job each, and
 * a workgroup that takes them one after the other keeps seven of its eight wavefronts idle): every group gets a window and a source tile of its own in LDS, all
 * are staged together, then wavefront k runs the job of group k.  The caller has checked that the groups hold one job each, that there are no more of them than
 * wavefronts and that the LDS holds them (me_multi_fits). */
XA_DEV size_t me_multi_region_bytes(const MeParams& p) { return (((size_t)p.maxWinW * p.maxWinH + 16 + 64 * 64) * sizeof(pixel) + 15) & ~(size_t)15; }
XA_DEV bool me_multi_fits(const MeParams& p, int groups, int waves, size_t ldsBytes)
{
    return groups > 1 && groups <= waves && (size_t)groups * me_multi_region_bytes(p) + 16 + (size_t)waves * sizeof(MeState) <= ldsBytes;
}
template<bool STAR> XA_DEV void block_me_search_multi(const MeParams& p, int groups, int tid, int nthr)
{
    char* smem = me_smem;
    const size_t region = me_multi_region_bytes(p);
    const int wv = tid >> 6, lane = tid & 63;
    const int sOff = (int)((size_t)groups * region + 16) + wv * (int)sizeof(MeState);
    MeState& s = *reinterpret_cast<MeState*>(smem + sOff);
    for (int gi = 0; gi < groups; gi++)
    {
        const x265amd_me_group g = p.groups[gi];
        const pixel* refG = reinterpret_cast<const pixel*>(p.refs[g.ref]);
        if (tid == 0) XA_BYTES(((unsigned long long)g.win_w * g.win_h + 64 * 64) * sizeof(pixel) + (unsigned long long)g.num_jobs * (sizeof(x265amd_me_job) + 8));
        pixel* win = reinterpret_cast<pixel*>(smem + (size_t)gi * region);
        pixel* fencT = win + p.maxWinW * p.maxWinH + 16;
        const int gpr = g.win_w >> 2, total = gpr * g.win_h;
        for (int i = tid; i < total; i += nthr)
        {
            const int y = i / gpr, x = (i - y * gpr) << 2;
            pixel v[4];
            __builtin_memcpy(v, refG + (long)(g.win_y + y) * p.stride + g.win_x + x, sizeof(v));
            __builtin_memcpy(win + y * g.win_w + x, v, sizeof(v));
        }
        for (int i = tid; i < 16 * 64; i += nthr)
        {
            const int y = i >> 4, x = (i & 15) << 2;
            pixel v[4];
            __builtin_memcpy(v, p.fenc + (long)(g.fenc_y + y) * p.stride + g.fenc_x + x, sizeof(v));
            __builtin_memcpy(fencT + y * 64 + x, v, sizeof(v));
        }
    }
    __syncthreads();
    if (wv < groups)
    {
        const x265amd_me_group g = p.groups[wv];
        const pixel* refG = reinterpret_cast<const pixel*>(p.refs[g.ref]);
        pixel* win = reinterpret_cast<pixel*>(smem + (size_t)wv * region);
        pixel* fencT = win + p.maxWinW * p.maxWinH + 16;
        const x265amd_me_job* jp = p.jobs + g.first_job;
        if (lane == 0) me_set_job(s, *jp, g, p, win, g.win_w, g.win_h, fencT, refG);
        xa_wave_sync();
        me_search<false, STAR>(sOff, jp, p.out + g.first_job);
        xa_wave_sync();
    }
}

/* direct-from-HBM kernel: redoes the jobs the window-resident kernel marked ME_DEFERRED (a candidate left the staged
 * window, or the search method was not compiled into the fast variant).  Same arithmetic, reference samples read
 * from HBM/L2; only the 64x64 source tile is staged. */
XA_DEV void block_me_deferred(const MeParams& p, int vb, int tid, int nthr)
{
    char* smem = me_smem;
    pixel* fencT = reinterpret_cast<pixel*>(smem);
    int* flag = reinterpret_cast<int*>(fencT + 64 * 64);
    const int sOff = (int)(reinterpret_cast<char*>(flag + 4) - smem) + (tid >> 6) * (int)sizeof(MeState);
    MeState& s = *reinterpret_cast<MeState*>(smem + sOff);
    const x265amd_me_group g = p.groups[vb];
    const pixel* refG = reinterpret_cast<const pixel*>(p.refs[g.ref]);
    const int lane = xa_lane(), wv = tid >> 6, nwv = nthr >> 6;

    /* anything to redo in this group? */
    if (tid == 0) *flag = 0;
    __syncthreads();
    int any = 0;
    for (int i = tid; i < g.num_jobs; i += nthr) any |= p.out[g.first_job + i].cost == ME_DEFERRED;
    if (any) atomicOr(flag, 1);
    __syncthreads();
    if (!*flag) return;
    for (int i = tid; i < 16 * 64; i += nthr)
    {
        int y = i >> 4, x = (i & 15) << 2;
        pixel v[4];
        __builtin_memcpy(v, p.fenc + (long)(g.fenc_y + y) * p.stride + g.fenc_x + x, sizeof(v));
        __builtin_memcpy(fencT + y * 64 + x, v, sizeof(v));
    }
    __syncthreads();
    for (int ji = wv; ji < g.num_jobs; ji += nwv)
    {
        if (p.out[g.first_job + ji].cost != ME_DEFERRED) continue;
        const x265amd_me_job* jp = p.jobs + g.first_job + ji;
        if (lane == 0) me_set_job(s, *jp, g, p, nullptr, 0, 0, fencT, refG);
        xa_wave_sync();
        me_search<true, true>(sOff, jp, p.out + g.first_job + ji);
        xa_wave_sync();
    }
}

#endif
