/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the in-loop deblocking filter of a whole picture:
 * Deblock::deblockCTU / deblockCU / setEdgefilter* / getBoundaryStrength / edgeFilterLuma / edgeFilterChroma (reference:
 * source/common/deblock.cpp:37-510) with the sample filters pelFilterLuma (deblock.cpp:268-310), pelFilterLumaStrong_c and
 * pelFilterChroma_c (source/common/loopfilter.cpp:140-180), 4:2:0.
 *
 * The reference walks CTUs in raster order (vertical edges of a CTU, then horizontal edges of the previous one); the result is
 * that of the standard's two picture-wide passes -- all vertical edges, then all horizontal edges on the vertically filtered
 * samples -- because an edge only touches three samples on either side and edges lie on the 8x8 grid.  Here the picture is
 * described by one record per 4x4 unit in raster order (what the per-CTU CUData arrays hold in z-order):
 *   flags bit0 intra, bit1 coded luma coefficients in the unit's TU, bit2 cu_transquant_bypass,
 *         bit3 / bit4: the unit's LEFT border is a TU (or CU) edge / a PU edge, bit5 / bit6: same for its TOP border,
 *   qp, ref[2] (picture identity per list, -1 = unused; equality stands for the reference's Frame* comparison), mv[2].
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file; the product never does.
 * Pinned against the reference's own Deblock class on CUData fixtures by tests/test_deblock.py (oracle/_ref) and a golden digest.
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

typedef struct { uint8_t flags; int8_t qp; int8_t ref[2]; int16_t mv[2][2]; } OrcDbUnit;
enum { DB_INTRA = 1, DB_CBF = 2, DB_BYPASS = 4, DB_TU_LEFT = 8, DB_PU_LEFT = 16, DB_TU_TOP = 32, DB_PU_TOP = 64 };

/* H.265 table 8-12 (deblock.cpp:497-509) */
static const uint8_t k_tc[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2,
    2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static const uint8_t k_beta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,
    18, 20, 22, 24, 26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static pixel clip_pel(int v) { return (pixel)clip3(0, (1 << ORC_DEPTH) - 1, v); }
static int chroma_qp(int qp)        /* g_chromaScale for 4:2:0 (constants.cpp:346-350) by rule */
{
    static const uint8_t mid[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
    if (qp < 30) return qp;
    if (qp < 44) return mid[qp - 30];
    return qp - 6 > 51 ? 51 : qp - 6;
}

/* getBoundaryStrength (deblock.cpp:183-241) after the edge marking of deblockCU (:72-107) */
static int boundary_strength(const OrcDbUnit* q, const OrcDbUnit* p, int tuEdge, int puEdge)
{
    int bs = tuEdge ? 2 : (puEdge ? 1 : 0);
    if (!bs) return 0;
    if ((p->flags & DB_INTRA) || (q->flags & DB_INTRA)) return 2;
    if (bs > 1 && ((q->flags & DB_CBF) || (p->flags & DB_CBF))) return 1;
    const int rp0 = p->ref[0], rq0 = q->ref[0], rp1 = p->ref[1], rq1 = q->ref[1];
    int mp0[2] = { rp0 >= 0 ? p->mv[0][0] : 0, rp0 >= 0 ? p->mv[0][1] : 0 }, mq0[2] = { rq0 >= 0 ? q->mv[0][0] : 0, rq0 >= 0 ? q->mv[0][1] : 0 };
    int mp1[2] = { rp1 >= 0 ? p->mv[1][0] : 0, rp1 >= 0 ? p->mv[1][1] : 0 }, mq1[2] = { rq1 >= 0 ? q->mv[1][0] : 0, rq1 >= 0 ? q->mv[1][1] : 0 };
#define FAR(a, b) (abs((a)[0] - (b)[0]) >= 4 || abs((a)[1] - (b)[1]) >= 4)
    /* -1 on both sides compares equal, like two NULL Frame pointers; a P slice (no list 1) reduces to its own rule (:206-210) */
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))
    {
        if (rp0 != rp1)
        {
            if (rp0 == rq0) return (FAR(mq0, mp0) || FAR(mq1, mp1)) ? 1 : 0;
            return (FAR(mq1, mp0) || FAR(mq0, mp1)) ? 1 : 0;
        }
        return ((FAR(mq0, mp0) || FAR(mq1, mp1)) && (FAR(mq1, mp0) || FAR(mq0, mp1))) ? 1 : 0;
    }
    return 1;
#undef FAR
}

/* one 4-sample luma segment: src = first sample on the Q side, offset across the edge, step along it (edgeFilterLuma :312-419) */
static void luma_segment(pixel* src, intptr_t step, intptr_t offset, int bs, int qp, int betaOffset, int tcOffset, int maskP, int maskQ)
{
    const int shift = ORC_DEPTH - 8;
    const int beta = k_beta[clip3(0, 51, qp + betaOffset)] << shift;
#define DP(s) abs((int)(s)[-offset * 3] - 2 * (int)(s)[-offset * 2] + (int)(s)[-offset])
#define DQ(s) abs((int)(s)[0] - 2 * (int)(s)[offset] + (int)(s)[offset * 2])
    const int dp0 = DP(src), dq0 = DQ(src), dp3 = DP(src + step * 3), dq3 = DQ(src + step * 3);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int tc = k_tc[clip3(0, 53, qp + 2 * (bs - 1) + tcOffset)] << shift;
#define STRONG(s) ((abs((int)(s)[-offset * 4] - (int)(s)[-offset]) + abs((int)(s)[offset * 3] - (int)(s)[0]) < (beta >> 3)) && \
                   (abs((int)(s)[-offset] - (int)(s)[0]) < ((tc * 5 + 1) >> 1)))
    const int sw = 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) && STRONG(src) && STRONG(src + step * 3);
    if (sw)
    {
        const int tcP = (2 * tc) & maskP, tcQ = (2 * tc) & maskQ;
        for (int i = 0; i < 4; i++, src += step)
        {
            const int m0 = src[-offset * 4], m1 = src[-offset * 3], m2 = src[-offset * 2], m3 = src[-offset];
            const int m4 = src[0], m5 = src[offset], m6 = src[offset * 2], m7 = src[offset * 3];
            src[-offset * 3] = (pixel)(clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
            src[-offset * 2] = (pixel)(clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
            src[-offset] = (pixel)(clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
            src[0] = (pixel)(clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
            src[offset] = (pixel)(clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
            src[offset * 2] = (pixel)(clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
        }
        return;
    }
    const int side = (beta + (beta >> 1)) >> 3;
    const int maskP1 = ((dp0 + dp3) < side ? -1 : 0) & maskP, maskQ1 = ((dq0 + dq3) < side ? -1 : 0) & maskQ;
    const int thrCut = tc * 10, tc2 = tc >> 1;
    for (int i = 0; i < 4; i++, src += step)
    {
        const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) < thrCut)
        {
            delta = clip3(-tc, tc, delta);
            src[-offset] = clip_pel(m3 + (delta & maskP));
            src[0] = clip_pel(m4 - (delta & maskQ));
            if (maskP1)
            {
                const int m1 = src[-offset * 3];
                src[-offset * 2] = clip_pel(m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
            }
            if (maskQ1)
            {
                const int m6 = src[offset * 2];
                src[offset] = clip_pel(m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
            }
        }
    }
#undef DP
#undef DQ
#undef STRONG
}

/* one 4-sample chroma segment (edgeFilterChroma :421-495, pelFilterChroma_c) */
static void chroma_segment(pixel* src, intptr_t step, intptr_t offset, int qpA, int cqpOffset, int tcOffset, int maskP, int maskQ)
{
    const int qp = chroma_qp(qpA + cqpOffset);
    const int tc = k_tc[clip3(0, 53, qp + 2 + tcOffset)] << (ORC_DEPTH - 8);
    for (int i = 0; i < 4; i++, src += step)
    {
        const int m4 = src[0], m3 = src[-offset], m5 = src[offset], m2 = src[-offset * 2];
        const int delta = clip3(-tc, tc, ((((m4 - m3) * 4) + m2 - m5 + 4) >> 3));
        src[-offset] = clip_pel(m3 + (delta & maskP));
        src[0] = clip_pel(m4 - (delta & maskQ));
    }
}

/* planes: sample (0,0) of Y, U, V; width / height multiples of 8; units: (width/4) x (height/4) records, raster order.
 * pass bit0: vertical edges, bit1: horizontal edges (3 = the whole filter). */
void orc_deblock_picture(pixel* const* planes, intptr_t stride, intptr_t cstride, int width, int height, const OrcDbUnit* units,
                         int betaOffsetDiv2, int tcOffsetDiv2, int cbQpOffset, int crQpOffset, int bypassEnabled, int pass)
{
    const int w4 = width >> 2, h4 = height >> 2;
    const int betaOffset = betaOffsetDiv2 * 2, tcOffset = tcOffsetDiv2 * 2;
    for (int dir = 0; dir < 2; dir++)
    {
        if (!((pass >> dir) & 1)) continue;
        for (int y4 = 0; y4 < h4; y4++)
            for (int x4 = 0; x4 < w4; x4++)
            {
                const int along = dir ? x4 : y4, across = dir ? y4 : x4;
                if ((across & 1) || across == 0) continue;                      /* 8x8 grid, not the picture border */
                const OrcDbUnit* q = units + y4 * w4 + x4;
                const OrcDbUnit* p = dir ? q - w4 : q - 1;
                const int bs = boundary_strength(q, p, q->flags & (dir ? DB_TU_TOP : DB_TU_LEFT), q->flags & (dir ? DB_PU_TOP : DB_PU_LEFT));
                if (!bs) continue;
                int maskP = -1, maskQ = -1;
                if (bypassEnabled)
                {
                    maskP = (p->flags & DB_BYPASS) ? 0 : -1; maskQ = (q->flags & DB_BYPASS) ? 0 : -1;
                    if (!(maskP | maskQ)) continue;
                }
                const int qp = (p->qp + q->qp + 1) >> 1;
                pixel* src = planes[0] + (intptr_t)(4 * y4) * stride + 4 * x4;
                luma_segment(src, dir ? 1 : stride, dir ? stride : 1, bs, qp, betaOffset, tcOffset, maskP, maskQ);
                /* chroma: edges on the 8-sample chroma grid, one segment per two luma units, strength of the first of them */
                if (bs > 1 && !(across & 3) && !(along & 1))
                    for (int c = 1; c < 3; c++)
                    {
                        pixel* sc = planes[c] + (intptr_t)(2 * y4) * cstride + 2 * x4;
                        chroma_segment(sc, dir ? 1 : cstride, dir ? cstride : 1, qp, c == 1 ? cbQpOffset : crQpOffset, tcOffset, maskP, maskQ);
                    }
            }
    }
}
