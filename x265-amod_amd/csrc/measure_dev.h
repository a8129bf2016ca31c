/* Device code of the CU assembly + measurement step (see inter_rd.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_MEASURE_DEV_H
#define X265AMD_MEASURE_DEV_H
#include "x265amd_dev.h"

/* ---------------- device: CU assembly + measurement ---------------- */
#define RD_LUMA_ELEMS 4096
#define RD_CHROMA_ELEMS 1024
#define RD_SCRATCH_ELEMS (4 * RD_LUMA_ELEMS + 3 * 2 * RD_CHROMA_ELEMS)      /* luma layers 4..32, chroma layers 4..16 x {U, V} */
#define RD_SEL_BYTES 384                                                     /* 256 luma units (row length 16) + 2 x 64 chroma units (row length 8) */

struct CuMeasureJob
{
    uint64_t fenc[3];               /* source block top-left per plane */
    uint64_t pred, recon;           /* tiles: 64x64 luma (stride 64), 32x32 U, 32x32 V (stride 32) */
    uint64_t resi;                  /* the CU's residual scratch (int16, RD_SCRATCH_ELEMS) */
    uint64_t sel;                   /* per unit: transform layer (log2 size) whose residual block covers it, 0xFF = none */
    int32_t fenc_stride, fenc_cstride, log2_size, assemble;
};
typedef x265amd_cu_measure CuMeasure;

XA_DEV size_t rd_layer_offset(int plane, int layer)
{
    return plane ? (size_t)4 * RD_LUMA_ELEMS + (size_t)((layer - 2) * 2 + (plane - 1)) * RD_CHROMA_ELEMS : (size_t)(layer - 2) * RD_LUMA_ELEMS;
}

/* one wavefront per CU: reconYuv = predYuv (+ clipped residual where a block was kept: Yuv::addClip, yuv.cpp:158-183), then
 * sse_pp per plane and the luma psyCost against the source (search.cpp:2937-2958 / :2869-2889) */
XA_DEV void wave_cu_measure_job(const CuMeasureJob* jobs, int ji, CuMeasure* out, pixel* tile /* this wave's 64 x 64 LDS tile */, int lane)
{
    const CuMeasureJob j = xa_ld_record(jobs + ji);
    if (lane == 0) XA_BYTES((9ull << (2 * j.log2_size)) / 2 * sizeof(pixel));         /* source, prediction tile read; reconstruction tile written: 3 x 1.5 N^2 samples */
    const uint8_t* sel = reinterpret_cast<const uint8_t*>(j.sel);
    const int16_t* resi = reinterpret_cast<const int16_t*>(j.resi);
    CuMeasure m;
    m.psy = 0; m.sa8d = 0; m.sa8d_luma = 0; m.src_mean = 0; m.src_homo = 0; m.reserved = 0;
    for (int plane = 0; plane < 3; plane++)
    {
        const int log2S = plane ? j.log2_size - 1 : j.log2_size, s = 1 << log2S, ts = plane ? 32 : 64;
        const size_t tileOff = plane ? 4096 + (size_t)(plane - 1) * 1024 : 0;
        const pixel* pred = reinterpret_cast<const pixel*>(j.pred) + tileOff;
        pixel* recon = reinterpret_cast<pixel*>(j.recon) + tileOff;
        for (int i = lane; i < s * s; i += XA_WAVE)
        {
            const int y = i >> log2S, x = i & (s - 1);
            int v = pred[y * ts + x];
            if (j.assemble)
            {
                const int layer = plane ? sel[256 + (plane - 1) * 64 + (y >> 2) * 8 + (x >> 2)] : sel[(y >> 2) * 16 + (x >> 2)];
                if (layer != 0xFF) v = xa_clip_pixel(v + (int)resi[rd_layer_offset(plane, layer) + y * ts + x]);
            }
            tile[y * ts + x] = (pixel)v;
            recon[y * ts + x] = (pixel)v;
        }
        xa_wave_sync();
        const pixel* f = reinterpret_cast<const pixel*>(j.fenc[plane]);
        const int fs = plane ? j.fenc_cstride : j.fenc_stride;
        m.sse[plane] = wave_sse_pp(f, fs, tile, ts, s, lane);
        if (!plane) m.psy = (uint32_t)wave_psy_cost(f, fs, tile, ts, j.log2_size - 2, lane);
        m.sa8d += (uint32_t)xa_wave_sa8d(f, fs, tile, ts, s, lane);
        if (!plane)
        {
            m.sa8d_luma = m.sa8d;
            uint32_t sum = 0;                                   /* complexityCheckCU (analysis.cpp:3538-3559): mean, then mean |sample - mean| */
            for (int i = lane; i < s * s; i += XA_WAVE) sum += f[(i >> log2S) * fs + (i & (s - 1))];
            const uint32_t mean = (uint32_t)xa_wave_sum((int)sum) / (uint32_t)(s * s);
            uint32_t dev = 0;
            for (int i = lane; i < s * s; i += XA_WAVE) { const int v = (int)f[(i >> log2S) * fs + (i & (s - 1))] - (int)mean; dev += (uint32_t)(v < 0 ? -v : v); }
            m.src_mean = mean; m.src_homo = (uint32_t)xa_wave_sum((int)dev) / (uint32_t)(s * s); m.reserved = 0;
        }
        xa_wave_sync();
    }
    if (lane == 0) out[ji] = m;
}


/* =========================================================================================================
 * The same step for ONE CU by a whole workgroup (device job queues: the command carries the one candidate the row is waiting for).  All samples of
 * the three planes are assembled at once, then one wavefront per 8x8 tile takes the three Hadamards a luma tile needs (difference for SA8D, source and
 * reconstruction for the psy energy; chroma tiles: the difference) across its lanes.
 * ======================================================================================================= */
struct CuMeasureLds
{
    pixel tile[64 * 64 + 2 * 32 * 32];      /* the assembled CU: Y (stride 64), U, V (stride 32) */
    pixel src[64 * 64 + 2 * 32 * 32];       /* the source CU, same layout */
    unsigned long long sse[3];
    unsigned int accY[16], accC[2][4], psy, sumSrc, devSrc;
};

XA_DEV void block_cu_measure_job(const CuMeasureJob& j, CuMeasure* out, CuMeasureLds& s, int tid, int nthr)
{
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    if (tid == 0) XA_BYTES((9ull << (2 * j.log2_size)) / 2 * sizeof(pixel));
    const uint8_t* sel = reinterpret_cast<const uint8_t*>(j.sel);
    const int16_t* resi = reinterpret_cast<const int16_t*>(j.resi);
    const int log2S = j.log2_size, S = 1 << log2S, C = S >> 1;
    __syncthreads();            /* the previous CU's readers are done with the LDS */
    if (tid < 3) s.sse[tid] = 0;
    if (tid < 16) s.accY[tid] = 0;
    if (tid < 8) (&s.accC[0][0])[tid] = 0;
    if (tid == 0) { s.psy = 0; s.sumSrc = 0; s.devSrc = 0; }
    __syncthreads();
    /* ---- assemble: reconYuv = predYuv (+ clipped residual where a block was kept), source alongside; squared differences and the source sum on the way ---- */
    {
        unsigned long long sq = 0;
        unsigned int sum = 0;
        const pixel* pred = reinterpret_cast<const pixel*>(j.pred);
        pixel* recon = reinterpret_cast<pixel*>(j.recon);
        const pixel* f = reinterpret_cast<const pixel*>(j.fenc[0]);
        for (int i = tid; i < S * S; i += nthr)
        {
            const int y = i >> log2S, x = i & (S - 1);
            int v = pred[y * 64 + x];
            if (j.assemble)
            {
                const int layer = sel[(y >> 2) * 16 + (x >> 2)];
                if (layer != 0xFF) v = xa_clip_pixel(v + (int)resi[rd_layer_offset(0, layer) + y * 64 + x]);
            }
            const int sv = f[y * j.fenc_stride + x];
            s.tile[y * 64 + x] = (pixel)v; recon[y * 64 + x] = (pixel)v; s.src[y * 64 + x] = (pixel)sv;
            const int t = sv - v;
            sq += (unsigned long long)(unsigned int)(t * t); sum += (unsigned int)sv;
        }
        sq = xa_wave_sum(sq); sum = xa_wave_sum(sum);
        if (lane == 0 && (sq | sum)) { atomicAdd(&s.sse[0], sq); atomicAdd(&s.sumSrc, sum); }
        for (int plane = 1; plane < 3; plane++)
        {
            const size_t off = 4096 + (size_t)(plane - 1) * 1024;
            const pixel* fc = reinterpret_cast<const pixel*>(j.fenc[plane]);
            unsigned long long sqc = 0;
            for (int i = tid; i < C * C; i += nthr)
            {
                const int y = i >> (log2S - 1), x = i & (C - 1);
                int v = pred[off + y * 32 + x];
                if (j.assemble)
                {
                    const int layer = sel[256 + (plane - 1) * 64 + (y >> 2) * 8 + (x >> 2)];
                    if (layer != 0xFF) v = xa_clip_pixel(v + (int)resi[rd_layer_offset(plane, layer) + y * 32 + x]);
                }
                const int sv = fc[y * j.fenc_cstride + x];
                s.tile[off + y * 32 + x] = (pixel)v; recon[off + y * 32 + x] = (pixel)v; s.src[off + y * 32 + x] = (pixel)sv;
                const int t = sv - v;
                sqc += (unsigned long long)(unsigned int)(t * t);
            }
            sqc = xa_wave_sum(sqc);
            if (lane == 0 && sqc) atomicAdd(&s.sse[plane], sqc);
        }
    }
    __syncthreads();
    /* ---- mean absolute deviation of the source (Analysis::complexityCheckCU, analysis.cpp:3538-3559) ---- */
    {
        const unsigned int mean = s.sumSrc / (unsigned int)(S * S);
        unsigned int dev = 0;
        for (int i = tid; i < S * S; i += nthr) { const int v = (int)s.src[(i >> log2S) * 64 + (i & (S - 1))] - (int)mean; dev += (unsigned int)(v < 0 ? -v : v); }
        dev = xa_wave_sum(dev);
        if (lane == 0 && dev) atomicAdd(&s.devSrc, dev);
    }
    /* ---- Hadamard tiles: luma 8x8 tiles (three transforms each), then the chroma tiles of both planes ---- */
    const int tY = S >> 3, nY = tY * tY;                    /* luma tiles per row / in all */
    const int tC = C >> 3, nC = tC * tC;                    /* chroma 8x8 tiles per row / per plane (0 when the chroma block is 4x4) */
    const int ly = lane >> 3, lx = lane & 7;
    for (int t = wv; t < nY + 2 * nC; t += nwv)
    {
        if (t < nY)
        {
            const int ty = t / tY, tx = t - ty * tY, p = (8 * ty + ly) * 64 + 8 * tx + lx;
            const int sv = s.src[p], rv = s.tile[p];
            const int hd = xa_lane_had8x8(sv - rv, lane), hs = xa_lane_had8x8(sv, lane), hr = xa_lane_had8x8(rv, lane);
            const int rawD = xa_wave_sum(abs(hd)), rawS = xa_wave_sum(abs(hs)), rawR = xa_wave_sum(abs(hr));
            const int sadS = __builtin_amdgcn_readfirstlane(hs), sadR = __builtin_amdgcn_readfirstlane(hr);     /* coefficient 0 = the sum of the samples */
            if (lane == 0)
            {
                const int se = ((rawS + 2) >> 2) - (sadS >> 2), re = ((rawR + 2) >> 2) - (sadR >> 2);      /* psyCost_pp 8x8 (pixel.cpp:744-775) */
                atomicAdd(&s.psy, (unsigned int)abs(se - re));
                atomicAdd(&s.accY[S == 8 ? 0 : (ty >> 1) * (S >> 4) + (tx >> 1)], (unsigned int)rawD);
            }
        }
        else
        {
            const int u = t - nY, plane = u / nC, k = u - plane * nC, ty = k / tC, tx = k - ty * tC;
            const int p = 4096 + plane * 1024 + (8 * ty + ly) * 32 + 8 * tx + lx;
            const int rawD = xa_wave_sum(abs(xa_lane_had8x8((int)s.src[p] - (int)s.tile[p], lane)));
            if (lane == 0) atomicAdd(&s.accC[plane][C == 8 ? 0 : (ty >> 1) * (C >> 4) + (tx >> 1)], (unsigned int)rawD);
        }
    }
    if (C == 4 && wv == nwv - 1)        /* 4x4 chroma blocks (8x8 CUs): satd_4x4 of U in lanes 0-15, of V in lanes 16-31 */
    {
        const int plane = (lane >> 4) & 1, l = lane & 15, p = 4096 + plane * 1024 + (l >> 2) * 32 + (l & 3);
        int v = lane < 32 ? (int)s.src[p] - (int)s.tile[p] : 0;
        v = xa_row16_sum(abs(xa_lane_had4x4(v, lane)));
        if (lane < 32 && l == 0) s.accC[plane][0] = (unsigned int)v;
    }
    __syncthreads();
    if (tid == 0)
    {
        CuMeasure m;
        m.sse[0] = s.sse[0]; m.sse[1] = s.sse[1]; m.sse[2] = s.sse[2];
#if XA_DEPTH <= 8
        m.sse[0] = (uint32_t)m.sse[0]; m.sse[1] = (uint32_t)m.sse[1]; m.sse[2] = (uint32_t)m.sse[2];      /* sse_t is uint32_t below 10 bits (common/common.h:142-146) */
#endif
        m.psy = s.psy;
        unsigned int sy = 0, sc = 0;
        const int gY = S == 8 ? 1 : (S >> 4) * (S >> 4);
        for (int k = 0; k < gY; k++) sy += (s.accY[k] + 2) >> 2;                   /* sa8d_8x8 / sa8d_16x16 groups (pixel.cpp:342-384) */
        for (int plane = 0; plane < 2; plane++)
        {
            if (C == 4) sc += s.accC[plane][0] >> 1;                              /* cu[4x4].sa8d = satd_4x4 (pixel.cpp:1171) */
            else { const int gC = C == 8 ? 1 : (C >> 4) * (C >> 4); for (int k = 0; k < gC; k++) sc += (s.accC[plane][k] + 2) >> 2; }
        }
        m.sa8d_luma = sy; m.sa8d = sy + sc;
        m.src_mean = s.sumSrc / (unsigned int)(S * S); m.src_homo = s.devSrc / (unsigned int)(S * S); m.reserved = 0;
        *out = m;
    }
}

#endif
