/* Device code of motion compensation (see mc_kernels.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_MC_DEV_H
#define X265AMD_MC_DEV_H
#include "x265amd_dev.h"

#define MC_WAVES 4

struct McPlane { const pixel* src[2]; long stride; int xf[2], yf[2]; int w, h; pixel* dst; int dstStride; int c; };

/* one prediction sample of a plane */
template<int TAPS> XA_DEV int mc_one(const McPlane& p, const x265amd_mc_job& j, int mode, int lsel, int x, int y)
{
    /* mode 0: pixel path from list lsel; 1: weighted uni from list lsel; 2: bi average; 3: weighted bi;
     * 4: pixel average of the two pixel-path predictions (pixelavg_pp of two predInterLumaPixel, search.cpp:2499-2511) */
    const int c = p.c;
    const int shiftNum = XA_IF_INTERNAL_PREC - XA_DEPTH;
    int v;
    if (mode == 0)
        v = mc_sample<TAPS, false>(p.src[lsel] + (long)y * p.stride + x, p.stride, p.xf[lsel], p.yf[lsel]);
    else if (mode == 4)
    {
        int a = mc_sample<TAPS, false>(p.src[0] + (long)y * p.stride + x, p.stride, p.xf[0], p.yf[0]);
        int b = mc_sample<TAPS, false>(p.src[1] + (long)y * p.stride + x, p.stride, p.xf[1], p.yf[1]);
        v = (a + b + 1) >> 1;       /* pixelavg_pp (pixel.cpp:880-893) */
    }
    else if (mode == 1)
    {
        /* addWeightUni -> weight_sp_c (predict.cpp:520-577, pixel.cpp:493-517) */
        int s0 = mc_sample<TAPS, true>(p.src[lsel] + (long)y * p.stride + x, p.stride, p.xf[lsel], p.yf[lsel]);
        int shift = j.wp[lsel][c].denom + shiftNum, round = shift ? 1 << (shift - 1) : 0;
        int off = j.wp[lsel][c].o * (1 << (XA_DEPTH - 8));
        v = xa_clip3(0, XA_PIXEL_MAX, ((j.wp[lsel][c].w * (s0 + XA_IF_INTERNAL_OFFS) + round) >> shift) + off);
    }
    else
    {
        int s0 = mc_sample<TAPS, true>(p.src[0] + (long)y * p.stride + x, p.stride, p.xf[0], p.yf[0]);
        int s1 = mc_sample<TAPS, true>(p.src[1] + (long)y * p.stride + x, p.stride, p.xf[1], p.yf[1]);
        if (mode == 2)      /* addAvg (pixel.cpp:860-879) */
        {
            const int shift = shiftNum + 1, offset = (1 << (shift - 1)) + 2 * XA_IF_INTERNAL_OFFS;
            v = xa_clip3(0, XA_PIXEL_MAX, (s0 + s1 + offset) >> shift);
        }
        else                /* addWeightBi / weightBidir (predict.cpp:52-55, :411-518) */
        {
            int shift = j.wp[0][c].denom + shiftNum + 1, round = shift ? 1 << (shift - 1) : 0;
            int offset = (j.wp[0][c].o + j.wp[1][c].o) * (1 << (XA_DEPTH - 8));
            v = xa_clip3(0, XA_PIXEL_MAX, (j.wp[0][c].w * (s0 + XA_IF_INTERNAL_OFFS) + j.wp[1][c].w * (s1 + XA_IF_INTERNAL_OFFS) + round + (offset * (1 << (shift - 1)))) >> shift);
        }
    }
    return v;
}
/* the samples idx, idx + step, ... of the block: a wavefront takes (lane, 64), several wavefronts sharing a block (sub * 64 + lane, waves * 64) */
template<int TAPS> XA_DEV void mc_plane(const McPlane& p, const x265amd_mc_job& j, int mode, int lsel, int idx, int step)
{
    int inv = ((1 << 20) + p.w - 1) / p.w;
    for (int i = idx; i < p.w * p.h; i += step)
    {
        int y = (i * inv) >> 20, x = i - y * p.w;
        p.dst[(long)y * p.dstStride + x] = (pixel)mc_one<TAPS>(p, j, mode, lsel, x, y);
    }
}

/* what a job's prediction is made of: the clipped vectors and the combination rule (CUData::clipMv; predict.cpp:82-243) */
struct McSetup { int mv[2][2]; int refs[2]; int mode, lsel; bool doChroma; };
XA_DEV McSetup mc_setup(const x265amd_mc_job& j, int picW, int picH)
{
    McSetup s;
    s.refs[0] = j.ref0; s.refs[1] = j.ref1;
    s.mv[0][0] = j.mv0[0]; s.mv[0][1] = j.mv0[1]; s.mv[1][0] = j.mv1[0]; s.mv[1][1] = j.mv1[1];
    {
        const int maxCU = 64, offset = 8;
        int xmax = (picW + offset - j.cu_x - 1) << 2, xmin = -((maxCU + offset + j.cu_x - 1) << 2);
        int ymax = (picH + offset - j.cu_y - 1) << 2, ymin = -((maxCU + offset + j.cu_y - 1) << 2);
        for (int l = 0; l < 2; l++)
        {
            s.mv[l][0] = min(xmax, max(xmin, s.mv[l][0]));
            s.mv[l][1] = min(ymax, max(ymin, s.mv[l][1]));
        }
    }
    s.lsel = 0;
    if (j.slice_type)
        s.mode = ((j.flags & 4) && j.wp[0][0].present) ? 1 : 0;
    else
    {
        bool wb = (j.flags & 8) != 0;
        if (s.refs[0] >= 0 && s.refs[1] >= 0)
            s.mode = (wb && (j.wp[0][0].present || j.wp[1][0].present)) ? 3 : 2;
        else
        {
            s.lsel = s.refs[0] >= 0 ? 0 : 1;
            s.mode = (wb && j.wp[s.lsel][0].present) ? 1 : 0;
        }
    }
    if (j.flags & 16) s.mode = 4;
    s.doChroma = (j.flags & 2) && s.mode != 4;
    return s;
}
/* the plane description of plane c (0 luma, 1 / 2 chroma) of the job's block for the samples from (ox, oy) on (block coordinates of that plane) */
XA_DEV McPlane mc_plane_of(const x265amd_mc_job& j, const McSetup& s, const uint64_t* planes, long stride, long cstride, int c)
{
    McPlane p;
    if (c == 0)
    {
        for (int l = 0; l < 2; l++)
        {
            p.src[l] = s.refs[l] >= 0 ? reinterpret_cast<const pixel*>(planes[3 * s.refs[l]]) + (long)(j.y + (s.mv[l][1] >> 2)) * stride + j.x + (s.mv[l][0] >> 2) : nullptr;
            p.xf[l] = s.mv[l][0] & 3; p.yf[l] = s.mv[l][1] & 3;
        }
        p.stride = stride; p.w = j.w; p.h = j.h; p.dst = reinterpret_cast<pixel*>(j.dst_y); p.dstStride = j.dst_stride; p.c = 0;
    }
    else
    {
        for (int l = 0; l < 2; l++)
        {
            p.src[l] = s.refs[l] >= 0 ? reinterpret_cast<const pixel*>(planes[3 * s.refs[l] + c]) + (long)((j.y >> 1) + (s.mv[l][1] >> 3)) * cstride + (j.x >> 1) + (s.mv[l][0] >> 3) : nullptr;
            p.xf[l] = s.mv[l][0] & 7; p.yf[l] = s.mv[l][1] & 7;
        }
        p.stride = cstride; p.w = j.w >> 1; p.h = j.h >> 1; p.dst = reinterpret_cast<pixel*>(c == 1 ? j.dst_u : j.dst_v); p.dstStride = j.dst_cstride; p.c = c;
    }
    return p;
}

/* COST: after the prediction has been written, its distortion against the source picture (x265amd_inter_cost) */
struct XaArgsMc
{
    const uint64_t* planes; long stride, cstride; int picW, picH; const x265amd_mc_job* jobs; int n;
    const uint64_t* fencPlanes; long fstride, fcstride; uint32_t* cost;
};

/* one job of a list on one wavefront (idx = lane, step = 64), or -- prediction only -- on the `step / 64` wavefronts that share it */
template<bool COST>
XA_DEV void wave_mc_rec(const XaArgsMc& a, const x265amd_mc_job& j, int ji, int idx, int step = XA_WAVE)
{
    const int lane = idx & 63;
    const uint64_t* planes = a.planes; const long stride = a.stride, cstride = a.cstride; const int picW = a.picW, picH = a.picH;
    const uint64_t* fencPlanes = a.fencPlanes; const long fstride = a.fstride, fcstride = a.fcstride; uint32_t* cost = a.cost;
    if (idx == 0)       /* per direction: the luma block with its 8-tap border and both chroma blocks with their 4-tap border; the prediction written (+ the source read when costs are taken) */
        XA_BYTES((unsigned long long)((j.ref0 >= 0) + (j.ref1 >= 0)) * ((unsigned)(j.w + 7) * (j.h + 7) + 2u * (unsigned)(j.w / 2 + 3) * (j.h / 2 + 3)) * sizeof(pixel) +
                 3ull * j.w * j.h / 2 * sizeof(pixel) * (COST ? 2 : 1));
    const McSetup su = mc_setup(j, picW, picH);
    const bool doChroma = su.doChroma;
    if (j.flags & 1)
    {
        const McPlane p = mc_plane_of(j, su, planes, stride, cstride, 0);
        mc_plane<8>(p, j, su.mode, su.lsel, idx, step);
    }
    if (doChroma)
        for (int c = 1; c < 3; c++)
        {
            const McPlane p = mc_plane_of(j, su, planes, stride, cstride, c);
            mc_plane<4>(p, j, su.mode, su.lsel, idx, step);
        }
    if constexpr (COST)
    {
        /* the wave reads back what it has just written */
        __threadfence_block();
        xa_wave_sync();
        const int metric = j.metric;
        uint32_t lumaCost = 0, chromaCost = 0;
        int cu = 0;
        while ((4 << cu) < j.w) cu++;
        if (j.flags & 1)
        {
            const pixel* f = reinterpret_cast<const pixel*>(fencPlanes[0]) + (long)j.y * fstride + j.x;
            const pixel* d = reinterpret_cast<const pixel*>(j.dst_y);
            lumaCost = metric == 1 ? (uint32_t)xa_wave_sad(f, (int)fstride, d, j.dst_stride, j.w, j.h, lane)
                     : metric == 2 ? (uint32_t)xa_wave_satd(f, (int)fstride, d, j.dst_stride, j.w, j.h, lane)
                     : metric == 3 ? (uint32_t)xa_wave_sa8d(f, (int)fstride, d, j.dst_stride, 4 << cu, lane) : 0;
        }
        if (doChroma && j.chroma_cost && metric >= 2)
            for (int c = 1; c < 3; c++)
            {
                const pixel* f = reinterpret_cast<const pixel*>(fencPlanes[c]) + (long)(j.y >> 1) * fcstride + (j.x >> 1);
                const pixel* d = reinterpret_cast<const pixel*>(c == 1 ? j.dst_u : j.dst_v);
                chromaCost += metric == 2 ? (uint32_t)xa_wave_satd(f, (int)fcstride, d, j.dst_cstride, j.w >> 1, j.h >> 1, lane)
                                          : (uint32_t)xa_wave_sa8d(f, (int)fcstride, d, j.dst_cstride, 2 << cu, lane);
            }
        if (lane == 0) { cost[2 * ji] = lumaCost; cost[2 * ji + 1] = chromaCost; }
    }
}
/* the job as the host last wrote it */
template<bool COST>
XA_DEV void wave_mc_job(const XaArgsMc& a, int ji, int idx, int step = XA_WAVE)
{
    const x265amd_mc_job j = xa_ld_record(a.jobs + ji);
    wave_mc_rec<COST>(a, j, ji, idx, step);
}

#endif
