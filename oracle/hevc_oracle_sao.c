/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the data-parallel halves of sample adaptive offset for a whole picture:
 *   statistics  SAO::calcSaoStatsCTU with saoCuStatsBO/E0..E3 (reference: source/encoder/sao.cpp:735-917, :1762-1925), per CTU and
 *               plane, sao-non-deblock off (the right / bottom strips whose deblocking was not finished when the reference
 *               gathered a CTU's statistics stay excluded, exactly as there);
 *   application SAO::generateLumaOffsets / generateChromaOffsets / applyPixelOffsets (sao.cpp:274-733) for given per-CTU parameters:
 *               every sample is classified on the deblocked, not yet offset samples (what m_tmpU / m_tmpL preserve), 4:2:0.
 * The rate-distortion choice of the parameters (rdoSaoUnitCu, sao.cpp:1225-1760) is host control flow and not part of this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file; the product never does.
 * Pinned against the reference's own SAO class by tests/test_sao.py (oracle/_ref) and a golden digest.
 */
#include <stdint.h>
#include <string.h>

#ifndef ORC_DEPTH
#define ORC_DEPTH 8
#endif
#if ORC_DEPTH > 8
typedef uint16_t pixel;
#else
typedef uint8_t pixel;
#endif

typedef struct { int8_t type[2]; uint8_t bandPos[3]; int8_t offset[3][4]; uint8_t pad[3]; } OrcSaoCtu;   /* type: -1 off, 0..3 EO_0..3, 4 BO; [0] luma, [1] chroma */

static int sgn(int v) { return (v > 0) - (v < 0); }
static const int k_eoClass[5] = { 1, 2, 0, 3, 4 };      /* SAO::s_eoTable (sao.cpp:65-72) */
/* neighbour pair of each edge type: (dx, dy) of a; b is the opposite sample */
static const int k_eoDx[4] = { -1, 0, -1, 1 }, k_eoDy[4] = { 0, -1, -1, -1 };

/* count / offsetOrg: [ctu][plane][type 0..4][class 0..31] int32 */
void orc_sao_stats_picture(const pixel* const* rec, const pixel* const* fenc, intptr_t stride, intptr_t cstride, int width, int height,
                           int32_t* count, int32_t* offsetOrg)
{
    const int ctuW = (width + 63) / 64, ctuH = (height + 63) / 64;
    memset(count, 0, sizeof(int32_t) * ctuW * ctuH * 3 * 5 * 32);
    memset(offsetOrg, 0, sizeof(int32_t) * ctuW * ctuH * 3 * 5 * 32);
    for (int cy = 0; cy < ctuH; cy++)
        for (int cx = 0; cx < ctuW; cx++)
            for (int plane = 0; plane < 3; plane++)
            {
                const int sh = plane ? 1 : 0, po = plane ? 2 : 0;
                const intptr_t st = plane ? cstride : stride;
                const int picW = width >> sh, picH = height >> sh, lpelx = (cx * 64) >> sh, tpely = (cy * 64) >> sh;
                const int rpelx = lpelx + (64 >> sh) < picW ? lpelx + (64 >> sh) : picW, bpely = tpely + (64 >> sh) < picH ? tpely + (64 >> sh) : picH;
                const int cw = rpelx - lpelx, ch = bpely - tpely;
                const pixel* r0 = rec[plane] + (intptr_t)tpely * st + lpelx;
                const pixel* f0 = fenc[plane] + (intptr_t)tpely * st + lpelx;
                int32_t* cnt = count + ((size_t)(cy * ctuW + cx) * 3 + plane) * 5 * 32;
                int32_t* org = offsetOrg + ((size_t)(cy * ctuW + cx) * 3 + plane) * 5 * 32;
                const int atRight = rpelx == picW, atBottom = bpely == picH, aboveUnavail = !tpely;
                const int endXfull = atRight ? cw : cw - 5 + po, endXedge = atRight ? cw - 1 : cw - 5 + po;
                const int endYfull = atBottom ? ch : ch - 4 + po, endYedge = atBottom ? ch - 1 : ch - 4 + po;
                for (int y = 0; y < ch; y++)
                    for (int x = 0; x < cw; x++)
                    {
                        const int v = r0[y * st + x], d = (int)f0[y * st + x] - v;
                        if (x < endXfull && y < endYfull)       /* band offset */
                        {
                            cnt[4 * 32 + (v >> (ORC_DEPTH - 5))]++; org[4 * 32 + (v >> (ORC_DEPTH - 5))] += d;
                        }
                        for (int t = 0; t < 4; t++)
                        {
                            const int x0 = t == 1 ? 0 : !lpelx, x1 = t == 1 ? endXfull : endXedge;
                            const int y0 = t == 0 ? 0 : aboveUnavail, y1 = t == 0 ? ch - 4 + po : endYedge;
                            if (x < x0 || x >= x1 || y < y0 || y >= y1) continue;
                            const int a = r0[(y + k_eoDy[t]) * st + x + k_eoDx[t]], b = r0[(y - k_eoDy[t]) * st + x - k_eoDx[t]];
                            const int cls = k_eoClass[sgn(v - a) + sgn(v - b) + 2];
                            cnt[t * 32 + cls]++; org[t * 32 + cls] += d;
                        }
                    }
            }
}

/* src: deblocked planes; dst: planes receiving the offset picture (the picture area only); params: one record per CTU, raster order */
void orc_sao_apply_picture(const pixel* const* src, pixel* const* dst, intptr_t stride, intptr_t cstride, int width, int height, const OrcSaoCtu* params)
{
    const int ctuW = (width + 63) / 64;
    const int pmax = (1 << ORC_DEPTH) - 1;
    for (int plane = 0; plane < 3; plane++)
    {
        const int sh = plane ? 1 : 0;
        const intptr_t st = plane ? cstride : stride;
        const int picW = width >> sh, picH = height >> sh;
        for (int y = 0; y < picH; y++)
            for (int x = 0; x < picW; x++)
            {
                const OrcSaoCtu* p = params + ((y << sh) >> 6) * ctuW + ((x << sh) >> 6);
                const int type = p->type[plane ? 1 : 0];
                const int v = src[plane][y * st + x];
                int out = v;
                if (type == 4)
                {
                    const int k = ((v >> (ORC_DEPTH - 5)) - p->bandPos[plane]) & 31;
                    if (k < 4) out = v + p->offset[plane][k];
                }
                else if (type >= 0)
                {
                    const int dx = k_eoDx[type], dy = k_eoDy[type];
                    const int edgeX = dx && (x == 0 || x == picW - 1), edgeY = dy && (y == 0 || y == picH - 1);
                    if (!edgeX && !edgeY)
                    {
                        const int a = src[plane][(y + dy) * st + x + dx], b = src[plane][(y - dy) * st + x - dx];
                        const int cls = k_eoClass[sgn(v - a) + sgn(v - b) + 2];
                        if (cls) out = v + p->offset[plane][cls - 1];
                    }
                }
                dst[plane][y * st + x] = (pixel)(out < 0 ? 0 : (out > pmax ? pmax : out));
            }
    }
}
