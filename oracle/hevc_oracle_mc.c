/* TEST INFRASTRUCTURE ONLY -- CPU oracle for inter prediction.
 *
 * Restates Predict::motionCompensation (reference: source/common/predict.cpp:77-243) with predInterLuma/Chroma
 * Pixel/Short (:245-408), addWeightBi/Uni (:411-577), Yuv::addAvg (yuv.cpp:189-211) and CUData::clipMv
 * (cudata.cpp:1915-1928), 4:2:0, composed from the oracle's interpolation primitives (hevc_oracle.c).
 * Pinned against the reference's own Predict::motionCompensation through oracle/_ref/librefprims*.so
 * (tests/test_mc_oracle_vs_ref.py) and golden vectors (tests/golden/mc_golden.npz).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_DEPTH
#define ORC_DEPTH 8
#endif
#if ORC_DEPTH > 8
typedef uint16_t pixel;
#else
typedef uint8_t pixel;
#endif
#define IF_INTERNAL_PREC 14
#define IF_INTERNAL_OFFS (1 << (IF_INTERNAL_PREC - 1))
#define CSP420 1

int orc_partition_from_sizes(int w, int h);
void orc_luma_hpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx);
void orc_luma_hps(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int rowExt);
void orc_luma_vpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx);
void orc_luma_vps(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx);
void orc_luma_vss(int part, const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx);
void orc_luma_hvpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int ix, int iy);
void orc_luma_p2s(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds);
void orc_chroma_hpp(int csp, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx);
void orc_chroma_hps(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int rowExt);
void orc_chroma_vpp(int csp, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx);
void orc_chroma_vps(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx);
void orc_chroma_vsp(int csp, int part, const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx);
void orc_chroma_vss(int csp, int part, const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx);
void orc_chroma_p2s(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds);
void orc_addAvg(int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds);
void orc_chroma_addAvg(int csp, int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds);
void orc_weight_sp(const int16_t* src, pixel* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset);

typedef struct
{
    uint64_t dstY, dstU, dstV; int32_t dstStride, dstCStride;
    int16_t x, y, cuX, cuY; uint8_t w, h; int8_t ref0, ref1; int16_t mv0[2], mv1[2];
    uint8_t sliceType, flags;
    struct { int16_t w, o; uint8_t denom, present; } wp[2][3];
    uint8_t reserved[2];
} PackedMcJob;

typedef struct { int w, o, offset, shift, round; } WV;

static inline pixel clip_pix(int v) { return (pixel)(v < 0 ? 0 : (v > ((1 << ORC_DEPTH) - 1) ? ((1 << ORC_DEPTH) - 1) : v)); }

/* cudata.cpp:1915-1928 */
static void clip_mv(int* mvx, int* mvy, int cuX, int cuY, int picW, int picH)
{
    const int maxCU = 64, offset = 8;
    int xmax = (picW + offset - cuX - 1) << 2, xmin = -((maxCU + offset + cuX - 1) << 2);
    int ymax = (picH + offset - cuY - 1) << 2, ymin = -((maxCU + offset + cuY - 1) << 2);
    *mvx = *mvx < xmin ? xmin : (*mvx > xmax ? xmax : *mvx);
    *mvy = *mvy < ymin ? ymin : (*mvy > ymax ? ymax : *mvy);
}

/* predInterLumaPixel / predInterChromaPixel (predict.cpp:245-265, :305-352) */
static void pred_pixel(int part, int w, const pixel* const pl[3], intptr_t stride, intptr_t cstride, int mvx, int mvy, pixel* dY, pixel* dU, pixel* dV, int ds, int dcs, int luma, int chroma)
{
    if (luma)
    {
        const pixel* src = pl[0] + (mvx >> 2) + (mvy >> 2) * stride;
        int xf = mvx & 3, yf = mvy & 3;
        if (!(xf | yf)) orc_luma_hpp(part, src, stride, dY, ds, 0);     /* coefficient set 0 is the identity: same samples as copy_pp */
        else if (!yf) orc_luma_hpp(part, src, stride, dY, ds, xf);
        else if (!xf) orc_luma_vpp(part, src, stride, dY, ds, yf);
        else orc_luma_hvpp(part, src, stride, dY, ds, xf, yf);
    }
    if (chroma)
    {
        intptr_t off = (mvx >> 3) + (mvy >> 3) * cstride;
        int xf = mvx & 7, yf = mvy & 7;
        pixel* d[2] = { dU, dV };
        for (int c = 0; c < 2; c++)
        {
            const pixel* ref = pl[1 + c] + off;
            if (!(xf | yf)) orc_chroma_hpp(CSP420, part, ref, cstride, d[c], dcs, 0);
            else if (!yf) orc_chroma_hpp(CSP420, part, ref, cstride, d[c], dcs, xf);
            else if (!xf) orc_chroma_vpp(CSP420, part, ref, cstride, d[c], dcs, yf);
            else
            {
                int16_t immed[32 * (32 + 3)];
                int cw = w >> 1;
                orc_chroma_hps(CSP420, part, ref, cstride, immed, cw, xf, 1);
                orc_chroma_vsp(CSP420, part, immed + cw, cw, d[c], dcs, yf);
            }
        }
    }
}

/* predInterLumaShort / predInterChromaShort (predict.cpp:267-303, :354-408): int16 at stride 64 / 32 */
static void pred_short(int part, int w, const pixel* const pl[3], intptr_t stride, intptr_t cstride, int mvx, int mvy, int16_t* sY, int16_t* sU, int16_t* sV, int luma, int chroma)
{
    if (luma)
    {
        const pixel* src = pl[0] + (mvx >> 2) + (mvy >> 2) * stride;
        int xf = mvx & 3, yf = mvy & 3;
        if (!(xf | yf)) orc_luma_p2s(part, src, stride, sY, 64);
        else if (!yf) orc_luma_hps(part, src, stride, sY, 64, xf, 0);
        else if (!xf) orc_luma_vps(part, src, stride, sY, 64, yf);
        else
        {
            int16_t immed[64 * (64 + 7)];
            orc_luma_hps(part, src, stride, immed, w, xf, 1);
            orc_luma_vss(part, immed + 3 * w, w, sY, 64, yf);
        }
    }
    if (chroma)
    {
        intptr_t off = (mvx >> 3) + (mvy >> 3) * cstride;
        int xf = mvx & 7, yf = mvy & 7, cw = w >> 1;
        int16_t* d[2] = { sU, sV };
        for (int c = 0; c < 2; c++)
        {
            const pixel* ref = pl[1 + c] + off;
            if (!(xf | yf)) orc_chroma_p2s(CSP420, part, ref, cstride, d[c], 32);
            else if (!yf) orc_chroma_hps(CSP420, part, ref, cstride, d[c], 32, xf, 0);
            else if (!xf) orc_chroma_vps(CSP420, part, ref, cstride, d[c], 32, yf);
            else
            {
                int16_t immed[32 * (32 + 3)];
                orc_chroma_hps(CSP420, part, ref, cstride, immed, cw, xf, 1);
                orc_chroma_vss(CSP420, part, immed + cw, cw, d[c], 32, yf);
            }
        }
    }
}

int orc_motion_compensation_batch(const uint64_t* planes, intptr_t stride, intptr_t cstride, int picW, int picH, const PackedMcJob* jobs, int n)
{
    static int16_t sh[2][3][64 * 64];
    for (int i = 0; i < n; i++)
    {
        const PackedMcJob* j = &jobs[i];
        int part = orc_partition_from_sizes(j->w, j->h), luma = j->flags & 1, chroma = (j->flags >> 1) & 1;
        pixel* dY = (pixel*)j->dstY; pixel* dU = (pixel*)j->dstU; pixel* dV = (pixel*)j->dstV;
        const int refs[2] = { j->ref0, j->ref1 };
        const pixel* pl[2][3];
        int mv[2][2] = { { j->mv0[0], j->mv0[1] }, { j->mv1[0], j->mv1[1] } };
        for (int l = 0; l < 2; l++)
            if (refs[l] >= 0)
            {
                pl[l][0] = (const pixel*)planes[3 * refs[l]] + (intptr_t)j->y * stride + j->x;
                pl[l][1] = (const pixel*)planes[3 * refs[l] + 1] + (intptr_t)(j->y >> 1) * cstride + (j->x >> 1);
                pl[l][2] = (const pixel*)planes[3 * refs[l] + 2] + (intptr_t)(j->y >> 1) * cstride + (j->x >> 1);
                clip_mv(&mv[l][0], &mv[l][1], j->cuX, j->cuY, picW, picH);
            }
        WV wv0[3];
        int planesN = chroma ? 3 : 1;
        if (j->sliceType)       /* P slice: predict.cpp:82-122 */
        {
            if ((j->flags & 4) && j->wp[0][0].present)
            {
                pred_short(part, j->w, pl[0], stride, cstride, mv[0][0], mv[0][1], sh[0][0], sh[0][1], sh[0][2], luma, chroma);
                for (int c = 0; c < planesN; c++)
                {
                    int denom = j->wp[0][c].denom, shift = denom + IF_INTERNAL_PREC - ORC_DEPTH;
                    (void)wv0;
                    /* addWeightUni (predict.cpp:520-577): round is derived from the combined shift */
                    int round = shift ? 1 << (shift - 1) : 0;
                    int off = j->wp[0][c].o * (1 << (ORC_DEPTH - 8));
                    if (c == 0) { if (luma) orc_weight_sp(sh[0][0], dY, 64, j->dstStride, j->w, j->h, j->wp[0][0].w, round, shift, off); }
                    else orc_weight_sp(sh[0][c], c == 1 ? dU : dV, 32, j->dstCStride, j->w >> 1, j->h >> 1, j->wp[0][c].w, round, shift, off);
                }
            }
            else
                pred_pixel(part, j->w, pl[0], stride, cstride, mv[0][0], mv[0][1], dY, dU, dV, j->dstStride, j->dstCStride, luma, chroma);
            continue;
        }
        /* B slice: predict.cpp:123-243 */
        int biW = 0, uniW = 0;
        if (j->flags & 8)
        {
            int p0 = refs[0] >= 0, p1 = refs[1] >= 0;
            if (p0 && p1 && (j->wp[0][0].present || j->wp[1][0].present)) biW = 1;
            else uniW = 1;      /* weights of the list that is used (list 0 first) land in wv0 */
        }
        if (refs[0] >= 0 && refs[1] >= 0)
        {
            pred_short(part, j->w, pl[0], stride, cstride, mv[0][0], mv[0][1], sh[0][0], sh[0][1], sh[0][2], luma, chroma);
            pred_short(part, j->w, pl[1], stride, cstride, mv[1][0], mv[1][1], sh[1][0], sh[1][1], sh[1][2], luma, chroma);
            if (biW)
            {
                for (int c = 0; c < planesN; c++)   /* addWeightBi (predict.cpp:411-518) */
                {
                    if (c == 0 && !luma) continue;
                    int w0 = j->wp[0][c].w, w1 = j->wp[1][c].w;
                    int offset = (j->wp[0][c].o + j->wp[1][c].o) * (1 << (ORC_DEPTH - 8));
                    int shift = j->wp[0][c].denom + (IF_INTERNAL_PREC - ORC_DEPTH) + 1;
                    int round = shift ? (1 << (shift - 1)) : 0;
                    int cw = c ? j->w >> 1 : j->w, ch = c ? j->h >> 1 : j->h, ss = c ? 32 : 64, ds = c ? j->dstCStride : j->dstStride;
                    pixel* d = c == 0 ? dY : (c == 1 ? dU : dV);
                    for (int y = 0; y < ch; y++)
                        for (int x = 0; x < cw; x++)
                            d[y * ds + x] = clip_pix((w0 * (sh[0][c][y * ss + x] + IF_INTERNAL_OFFS) + w1 * (sh[1][c][y * ss + x] + IF_INTERNAL_OFFS) + round + (offset * (1 << (shift - 1)))) >> shift);
                }
            }
            else
            {
                if (luma) orc_addAvg(part, sh[0][0], sh[1][0], dY, 64, 64, j->dstStride);
                if (chroma)
                {
                    orc_chroma_addAvg(CSP420, part, sh[0][1], sh[1][1], dU, 32, 32, j->dstCStride);
                    orc_chroma_addAvg(CSP420, part, sh[0][2], sh[1][2], dV, 32, 32, j->dstCStride);
                }
            }
        }
        else
        {
            int l = refs[0] >= 0 ? 0 : 1;
            /* uni-prediction weights always come from wv0 = the used list's table when !biW (predict.cpp:155-166) */
            if ((j->flags & 8) && j->wp[l][0].present && uniW)
            {
                pred_short(part, j->w, pl[l], stride, cstride, mv[l][0], mv[l][1], sh[0][0], sh[0][1], sh[0][2], luma, chroma);
                for (int c = 0; c < planesN; c++)
                {
                    int shift = j->wp[l][c].denom + IF_INTERNAL_PREC - ORC_DEPTH;
                    int round = shift ? 1 << (shift - 1) : 0;
                    int off = j->wp[l][c].o * (1 << (ORC_DEPTH - 8));
                    if (c == 0) { if (luma) orc_weight_sp(sh[0][0], dY, 64, j->dstStride, j->w, j->h, j->wp[l][0].w, round, shift, off); }
                    else orc_weight_sp(sh[0][c], c == 1 ? dU : dV, 32, j->dstCStride, j->w >> 1, j->h >> 1, j->wp[l][c].w, round, shift, off);
                }
            }
            else
                pred_pixel(part, j->w, pl[l], stride, cstride, mv[l][0], mv[l][1], dY, dU, dV, j->dstStride, j->dstCStride, luma, chroma);
        }
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------
 * distortion of inter prediction candidates: motion compensation of the candidate followed by the metric the
 * reference's decision uses at that point --
 *   SAD   (metric 1): Search::selectMVP, search.cpp:1992-2018 (m_me.bufSAD of predInterLumaPixel);
 *   SATD  (metric 2): Search::mergeEstimation search.cpp:1891-1966 and the bi-prediction tries of predInterSearch
 *                     search.cpp:2473-2576 (m_me.bufSATD [+ bufChromaSATD]);
 *   SA8D  (metric 3): the merge scan of Analysis::checkMerge2Nx2N_rd0_4, analysis.cpp:2750-2880 (cu[].sa8d [+ chroma sa8d]).
 * flags & 16: the prediction is the pixel average of the two pixel-path predictions (pixelavg_pp of two
 * predInterLumaPixel results, search.cpp:2499-2511) instead of motionCompensation's addAvg.
 * reserved[0] = metric, reserved[1] = 1 to add the chroma metric.  out[i] = { luma, chroma (U + V) }.
 * ------------------------------------------------------------------------------------------------------- */
int orc_sad(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_satd(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_chroma_satd(int csp, int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_chroma_sa8d(int csp, int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
void orc_pixelavg_pp(int part, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0, const pixel* s1, intptr_t ss1);
void orc_chroma_pixelavg_pp(int csp, int part, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0, const pixel* s1, intptr_t ss1);

int orc_inter_cost_batch(const uint64_t* planes, intptr_t stride, intptr_t cstride, int picW, int picH, const PackedMcJob* jobs, int n,
                         const uint64_t* fencPlanes, intptr_t fstride, intptr_t fcstride, uint32_t* out)
{
    static pixel tmp[2][3][64 * 64];
    for (int i = 0; i < n; i++)
    {
        const PackedMcJob* j = &jobs[i];
        /* the pixel-average form exists for luma only (with chroma SATD on, the reference runs motionCompensation instead) */
        const int part = orc_partition_from_sizes(j->w, j->h), luma = j->flags & 1, chroma = ((j->flags >> 1) & 1) && !(j->flags & 16);
        pixel* dY = (pixel*)j->dstY; pixel* dU = (pixel*)j->dstU; pixel* dV = (pixel*)j->dstV;
        if (j->flags & 16)
        {
            const int refs[2] = { j->ref0, j->ref1 };
            for (int l = 0; l < 2; l++)
            {
                const pixel* pl[3];
                int mvx = l ? j->mv1[0] : j->mv0[0], mvy = l ? j->mv1[1] : j->mv0[1];
                pl[0] = (const pixel*)planes[3 * refs[l]] + (intptr_t)j->y * stride + j->x;
                pl[1] = (const pixel*)planes[3 * refs[l] + 1] + (intptr_t)(j->y >> 1) * cstride + (j->x >> 1);
                pl[2] = (const pixel*)planes[3 * refs[l] + 2] + (intptr_t)(j->y >> 1) * cstride + (j->x >> 1);
                clip_mv(&mvx, &mvy, j->cuX, j->cuY, picW, picH);
                pred_pixel(part, j->w, pl, stride, cstride, mvx, mvy, tmp[l][0], tmp[l][1], tmp[l][2], 64, 32, luma, chroma);
            }
            if (luma)
                for (int y = 0; y < j->h; y++)
                    for (int x = 0; x < j->w; x++) dY[y * j->dstStride + x] = (pixel)((tmp[0][0][y * 64 + x] + tmp[1][0][y * 64 + x] + 1) >> 1);    /* pixelavg_pp, pixel.cpp:880-893 */
            if (chroma)
                for (int y = 0; y < j->h / 2; y++)
                    for (int x = 0; x < j->w / 2; x++)
                    {
                        dU[y * j->dstCStride + x] = (pixel)((tmp[0][1][y * 32 + x] + tmp[1][1][y * 32 + x] + 1) >> 1);
                        dV[y * j->dstCStride + x] = (pixel)((tmp[0][2][y * 32 + x] + tmp[1][2][y * 32 + x] + 1) >> 1);
                    }
        }
        else
            orc_motion_compensation_batch(planes, stride, cstride, picW, picH, j, 1);
        const pixel* fY = (const pixel*)fencPlanes[0] + (intptr_t)j->y * fstride + j->x;
        const pixel* fU = (const pixel*)fencPlanes[1] + (intptr_t)(j->y >> 1) * fcstride + (j->x >> 1);
        const pixel* fV = (const pixel*)fencPlanes[2] + (intptr_t)(j->y >> 1) * fcstride + (j->x >> 1);
        const int metric = j->reserved[0], addChroma = j->reserved[1];
        int cu = 0;
        while ((4 << cu) < j->w) cu++;
        uint32_t l = 0, c = 0;
        if (luma)
            l = metric == 1 ? (uint32_t)orc_sad(part, fY, fstride, dY, j->dstStride) : metric == 2 ? (uint32_t)orc_satd(part, fY, fstride, dY, j->dstStride)
              : metric == 3 ? (uint32_t)orc_sa8d(cu, fY, fstride, dY, j->dstStride) : 0;
        if (chroma && addChroma)
        {
            if (metric == 2) c = (uint32_t)(orc_chroma_satd(CSP420, part, fU, fcstride, dU, j->dstCStride) + orc_chroma_satd(CSP420, part, fV, fcstride, dV, j->dstCStride));
            else if (metric == 3) c = (uint32_t)(orc_chroma_sa8d(CSP420, cu, fU, fcstride, dU, j->dstCStride) + orc_chroma_sa8d(CSP420, cu, fV, fcstride, dV, j->dstCStride));
        }
        out[2 * i] = l; out[2 * i + 1] = c;
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------
 * reference-plane production: extendPicBorder (pixel.cpp:1044-1058) and MotionReference::applyWeight over all rows
 * (reference.cpp:109-185; weight_pp_c pixel.cpp:519-538)
 * ------------------------------------------------------------------------------------------------------- */
void orc_extend_pic_border(pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < marginX; x++)
        {
            pic[y * stride - marginX + x] = pic[y * stride];
            pic[y * stride + width + x] = pic[y * stride + width - 1];
        }
    for (int y = 0; y < marginY; y++)
    {
        memcpy(pic - marginX - (y + 1) * stride, pic - marginX, (size_t)(width + 2 * marginX) * sizeof(pixel));
        memcpy(pic - marginX + (height + y) * stride, pic - marginX + (height - 1) * stride, (size_t)(width + 2 * marginX) * sizeof(pixel));
    }
}

void orc_weight_plane(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int marginX, int marginY, int inputWeight, int inputOffset, int log2Denom)
{
    const int correction = IF_INTERNAL_PREC - ORC_DEPTH;
    const int offset = inputOffset * (1 << (ORC_DEPTH - 8));
    const int round = (log2Denom ? 1 << (log2Denom - 1) : 0) << correction, shift = log2Denom + correction;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++)
            dst[y * stride + x] = clip_pix(((inputWeight * ((int)src[y * stride + x] << correction) + round) >> shift) + offset);
    orc_extend_pic_border(dst, stride, width, height, marginX, marginY);
}
