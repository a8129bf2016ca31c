/* Host helpers shared by the inter search (inter_search.hip) and the CTU analysis (ctu_analysis.hip): motion vector arithmetic, the
 * reference's bit-cost formula, PU geometry, MV clipping. */
#ifndef X265AMD_INTER_COMMON_H
#define X265AMD_INTER_COMMON_H
#include "x265amd_host.h"
#include <math.h>
#include <stdlib.h>

namespace xa_inter {

struct Mv { int x, y; };
inline bool operator==(Mv a, Mv b) { return a.x == b.x && a.y == b.y; }

inline double is_lambda(int qp)           /* x265_lambda_tab (constants.cpp:34-52) by rule */
{
    double v = pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265AMD_DEPTH - 8));
    return floor(v * 10000.0 + 0.5) / 10000.0;
}
/* BitCost::s_bitsizes (bitcost.cpp:95-109), evaluated as the reference build does */
inline float is_bitsize(int d)
{
    const int i = abs(d);
    const double log2_2 = (double)(float)(2.0 / log(2.0));
    return i ? (float)(log((double)(float)(i + 1)) * log2_2 + (double)1.718f) : 0.718f;
}
inline uint32_t is_bitcost(Mv mv, Mv mvp) { return (uint32_t)(is_bitsize(mv.x - mvp.x) + is_bitsize(mv.y - mvp.y) + 0.5f); }

struct Geo { int x, y, w, h; };
inline Geo pu_geo(int cuX, int cuY, int size, int part, int idx)
{
    static const uint8_t rects[8][4][4] = {
        { { 0, 0, 4, 4 } }, { { 0, 0, 4, 2 }, { 0, 2, 4, 2 } }, { { 0, 0, 2, 4 }, { 2, 0, 2, 4 } }, { { 0, 0, 2, 2 }, { 2, 0, 2, 2 }, { 0, 2, 2, 2 }, { 2, 2, 2, 2 } },
        { { 0, 0, 4, 1 }, { 0, 1, 4, 3 } }, { { 0, 0, 4, 3 }, { 0, 3, 4, 1 } }, { { 0, 0, 1, 4 }, { 1, 0, 3, 4 } }, { { 0, 0, 3, 4 }, { 3, 0, 1, 4 } } };
    const uint8_t* r = rects[part][idx];
    const int q = size / 4;
    return Geo{ cuX + r[0] * q, cuY + r[1] * q, r[2] * q, r[3] * q };
}


/* cu.clipMv (cudata.cpp:1915-1928) */
inline void clip_mv(Mv& mv, int cuX, int cuY, int picW, int picH)
{
    const int maxCU = 64, offset = 8;
    const int xmax = (picW + offset - cuX - 1) << 2, xmin = -((maxCU + offset + cuX - 1) << 2);
    const int ymax = (picH + offset - cuY - 1) << 2, ymin = -((maxCU + offset + cuY - 1) << 2);
    mv.x = mv.x < xmin ? xmin : (mv.x > xmax ? xmax : mv.x);
    mv.y = mv.y < ymin ? ymin : (mv.y > ymax ? ymax : mv.y);
}
/* Search::setSearchRange without slice / intra-refresh restrictions (search.cpp:2724-2768); lag = Search::m_refLagPixels: the picture height, or
 * param.searchRange when pictures are coded in parallel (search.cpp:92) */
inline void search_range(Mv mvp, int merange, int cuX, int cuY, int picW, int picH, int lag, Mv& mn, Mv& mx)
{
    mn = Mv{ mvp.x - (merange << 2), mvp.y - (merange << 2) }; mx = Mv{ mvp.x + (merange << 2), mvp.y + (merange << 2) };
    clip_mv(mn, cuX, cuY, picW, picH); clip_mv(mx, cuX, cuY, picW, picH);
    const int maxLen = (1 << 15) - 1;
    mn.x = mn.x < -maxLen ? -maxLen : mn.x; mn.y = mn.y < -maxLen ? -maxLen : mn.y;
    mx.x = mx.x > maxLen ? maxLen : mx.x; mx.y = mx.y > maxLen ? maxLen : mx.y;
    mn.x >>= 2; mn.y >>= 2; mx.x >>= 2; mx.y >>= 2;
    mn.y = mn.y < lag ? mn.y : lag; mx.y = mx.y < lag ? mx.y : lag;
    mx.y = mx.y > mn.y ? mx.y : mn.y;
}

/* merge and AMVP candidates a picture coded in parallel with its references must not use (search.cpp:1934-1936, :2009-2010; analysis.cpp:2803, :2933) */
inline bool below_lag(int mvY, int searchRange) { return mvY >= (searchRange + 1) * 4; }

} // namespace xa_inter
#endif
