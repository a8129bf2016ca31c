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
    const CuMeasureJob j = jobs[ji];
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

#endif
