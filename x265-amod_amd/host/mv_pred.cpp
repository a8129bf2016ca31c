/* Motion vector prediction (include/x265amd.h: x265amd_merge_candidates, x265amd_amvp_candidates): host C++, part of SURVEY row a3.
 *
 * Restatement of CUData::getInterMergeCandidates (reference: source/common/cudata.cpp:1458-1712), getNeighbourMV / getPMV (:1715-1875),
 * getDirectPMV / getIndirectPMV / getColMVP / getCollocatedMV / scaleMvByPOCDist (:1931-2045) and the neighbour look-ups getPULeft /
 * Above / AboveLeft / AboveRight / BelowLeft (:605-760) on a picture-wide raster map of 4x4 units instead of z-ordered per-CTU arrays.
 * "Available" means what it means there: inside the picture and already coded when the current CU is reached in z-order inside its
 * CTU, CTUs in raster order.  The current CU's earlier PUs must already be written to the map when a later PU is predicted (the
 * reference reads them from the CU object under analysis).
 */
#include "x265amd.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct MV2 { int x, y; };
inline bool mvEq(const int16_t* a, const int16_t* b) { return a[0] == b[0] && a[1] == b[1]; }

struct Ctx
{
    const x265amd_mvpred_info* I;
    const x265amd_mv_unit* cur;
    const x265amd_mv_unit* col;
    int w4, h4;
    const x265amd_mv_unit* at(int x, int y) const { return (x < 0 || y < 0 || x >= I->pic_width || y >= I->pic_height) ? nullptr : cur + (y >> 2) * w4 + (x >> 2); }
    static unsigned z(int x, int y)         /* z-order of the 4x4 unit holding sample (x, y) inside its CTU */
    {
        unsigned r = 0;
        for (int b = 0; b < 4; b++) r |= (((unsigned)(x >> 2) >> b) & 1u) << (2 * b) | (((unsigned)(y >> 2) >> b) & 1u) << (2 * b + 1);
        return r;
    }
    /* the five spatial neighbours of the corner units (cudata.cpp:605-760).  (cx, cy): the unit the look-up starts from */
    const x265amd_mv_unit* left(int cx, int cy) const { return at(cx - 4, cy); }
    const x265amd_mv_unit* above(int cx, int cy) const { return at(cx, cy - 4); }
    const x265amd_mv_unit* aboveLeft(int cx, int cy) const { return at(cx - 4, cy - 4); }
    const x265amd_mv_unit* aboveRight(int cx, int cy) const
    {
        if (cx + 4 >= I->pic_width) return nullptr;
        const int ux = cx & 63, uy = cy & 63;
        if (ux < 60)
        {
            if (uy) return z(ux, uy) > z(ux + 4, uy - 4) ? at(cx + 4, cy - 4) : nullptr;
            return at(cx + 4, cy - 4);          /* CTU above */
        }
        if (uy) return nullptr;
        return at(cx + 4, cy - 4);              /* CTU above-right */
    }
    const x265amd_mv_unit* belowLeft(int cx, int cy) const
    {
        if (cy + 4 >= I->pic_height) return nullptr;
        const int ux = cx & 63, uy = cy & 63;
        if (uy < 60)
        {
            if (ux) return z(ux, uy) > z(ux - 4, uy + 4) ? at(cx - 4, cy + 4) : nullptr;
            return at(cx - 4, cy + 4);          /* CTU to the left */
        }
        return nullptr;
    }
    static bool inter(const x265amd_mv_unit* u) { return u && (u->pred_mode == X265AMD_MODE_INTER || u->pred_mode == X265AMD_MODE_SKIP); }
    static bool sameMotion(const x265amd_mv_unit* a, const x265amd_mv_unit* b)        /* hasEqualMotion (:1439-1456) */
    {
        if (a->inter_dir != b->inter_dir) return false;
        for (int l = 0; l < 2; l++)
            if ((a->inter_dir & (1 << l)) && (!mvEq(a->mv[l], b->mv[l]) || a->ref_idx[l] != b->ref_idx[l])) return false;
        return true;
    }
    static MV2 scale(MV2 mv, int curPOC, int curRefPOC, int colPOC, int colRefPOC)     /* scaleMvByPOCDist + scaleMv (:2030-2045, :104-110) */
    {
        const int d = colPOC - colRefPOC, b = curPOC - curRefPOC;
        if (d == b) return mv;
        const int tdb = b < -128 ? -128 : (b > 127 ? 127 : b), tdd = d < -128 ? -128 : (d > 127 ? 127 : d);
        const int x = (0x4000 + abs(tdd / 2)) / tdd;
        int s = (tdb * x + 32) >> 6;
        s = s < -4096 ? -4096 : (s > 4095 ? 4095 : s);
        int mx = (s * mv.x + 127 + (s * mv.x < 0)) >> 8, my = (s * mv.y + 127 + (s * mv.y < 0)) >> 8;
        mx = mx < -32768 ? -32768 : (mx > 32767 ? 32767 : mx); my = my < -32768 ? -32768 : (my > 32767 ? 32767 : my);
        return MV2{ mx, my };
    }
    /* position of the temporal candidate H (right-bottom) or -1 (:1628-1652 / :1834-1858) */
    bool rightBottom(int px, int py, int pw, int ph, int& hx, int& hy) const
    {
        const int rbx = px + pw - 4, rby = py + ph - 4;
        if (!(rbx + 4 < I->pic_width && rby + 4 < I->pic_height)) return false;
        const bool notLastCol = (rbx & 63) < 60, notLastRow = (rby & 63) < 60;
        if (notLastCol && notLastRow) { hx = rbx + 4; hy = rby + 4; return true; }
        if (notLastRow) { hx = rbx + 4; hy = rby + 4; return true; }       /* first column of the CTU to the right */
        return false;                                                        /* below the CTU row: never used */
    }
    const x265amd_mv_unit* colUnit(int x, int y) const { return col + (y >> 2) * w4 + (x >> 2); }
    /* getColMVP (:1968-2001) */
    bool colMVP(MV2& out, int refIdx, int list, int x, int y) const
    {
        const x265amd_mv_unit* u = colUnit(x, y);
        const x265amd_mv_unit* c = colUnit(x & ~15, y & ~15);
        if (u->pred_mode == X265AMD_MODE_NONE || c->pred_mode == X265AMD_MODE_INTRA) return false;
        int cl = I->check_ldc ? list : I->col_from_l0;
        int ci = c->ref_idx[cl];
        if (ci < 0) { cl = !cl; ci = c->ref_idx[cl]; if (ci < 0) return false; }
        out = scale(MV2{ c->mv[cl][0], c->mv[cl][1] }, I->poc, I->ref_poc[list][refIdx], I->col_poc, I->col_ref_poc[cl][ci]);
        return true;
    }
};

struct PuGeom { int x, y, w, h; };
PuGeom puGeom(int cuX, int cuY, int size, int part, int idx)
{
    static const uint8_t rects[8][4][4] = {     /* x, y, w, h in quarters of the CU (partTable, cudata.cpp) */
        { { 0, 0, 4, 4 } }, { { 0, 0, 4, 2 }, { 0, 2, 4, 2 } }, { { 0, 0, 2, 4 }, { 2, 0, 2, 4 } }, { { 0, 0, 2, 2 }, { 2, 0, 2, 2 }, { 0, 2, 2, 2 }, { 2, 2, 2, 2 } },
        { { 0, 0, 4, 1 }, { 0, 1, 4, 3 } }, { { 0, 0, 4, 3 }, { 0, 3, 4, 1 } }, { { 0, 0, 1, 4 }, { 1, 0, 3, 4 } }, { { 0, 0, 3, 4 }, { 3, 0, 1, 4 } } };
    const uint8_t* r = rects[part][idx];
    const int q = size / 4;
    return PuGeom{ cuX + r[0] * q, cuY + r[1] * q, r[2] * q, r[3] * q };
}

} // namespace

extern "C" {

int x265amd_merge_candidates(const x265amd_mvpred_info* I, const x265amd_mv_unit* cur, const x265amd_mv_unit* col, int cuX, int cuY, int log2CU,
                             int partSize, int puIdx, x265amd_merge_cand* out)
{
    Ctx C{ I, cur, col, I->pic_width >> 2, I->pic_height >> 2 };
    const int maxCand = I->max_num_merge_cand, isB = I->is_inter_b;
    for (int i = 0; i < maxCand; i++) { memset(&out[i], 0, sizeof(out[i])); out[i].ref_idx[0] = out[i].ref_idx[1] = -1; }
    const PuGeom p = puGeom(cuX, cuY, 1 << log2CU, partSize, puIdx);
    const int ltx = p.x, lty = p.y, rtx = p.x + p.w - 4, lby = p.y + p.h - 4;
    int count = 0;
    auto take = [&](const x265amd_mv_unit* u) {
        out[count].dir = u->inter_dir;
        out[count].mv[0][0] = u->mv[0][0]; out[count].mv[0][1] = u->mv[0][1]; out[count].ref_idx[0] = u->ref_idx[0];
        if (isB) { out[count].mv[1][0] = u->mv[1][0]; out[count].mv[1][1] = u->mv[1][1]; out[count].ref_idx[1] = u->ref_idx[1]; }
        return ++count == maxCand;
    };
    /* (isDiffMER is always true for neighbours outside the PU at the default parallel merge level 2) */
    const x265amd_mv_unit* a1 = C.left(ltx, lby);
    const bool availA1 = a1 && !(puIdx == 1 && (partSize == 2 || partSize == 6 || partSize == 7)) && Ctx::inter(a1);
    if (availA1 && take(a1)) return maxCand;
    const x265amd_mv_unit* b1 = C.above(rtx, lty);
    const bool availB1 = b1 && !(puIdx == 1 && (partSize == 1 || partSize == 4 || partSize == 5)) && Ctx::inter(b1);
    if (availB1 && (!availA1 || !Ctx::sameMotion(a1, b1)) && take(b1)) return maxCand;
    const x265amd_mv_unit* b0 = C.aboveRight(rtx, lty);
    const bool availB0 = Ctx::inter(b0);
    if (availB0 && (!availB1 || !Ctx::sameMotion(b1, b0)) && take(b0)) return maxCand;
    const x265amd_mv_unit* a0 = C.belowLeft(ltx, lby);
    const bool availA0 = Ctx::inter(a0);
    if (availA0 && (!availA1 || !Ctx::sameMotion(a1, a0)) && take(a0)) return maxCand;
    if (count < 4)
    {
        const x265amd_mv_unit* b2 = C.aboveLeft(ltx, lty);
        if (Ctx::inter(b2) && (!availA1 || !Ctx::sameMotion(a1, b2)) && (!availB1 || !Ctx::sameMotion(b1, b2)) && take(b2)) return maxCand;
    }
    if (I->temporal_mvp)
    {
        int hx = 0, hy = 0;
        const bool haveH = C.rightBottom(p.x, p.y, p.w, p.h, hx, hy);
        int dir = 0;
        for (int list = 0; list < (isB ? 2 : 1); list++)
        {
            MV2 mv;
            bool ok = haveH && C.colMVP(mv, 0, list, hx, hy);
            if (!ok) ok = C.colMVP(mv, 0, list, p.x + p.w / 2, p.y + p.h / 2);
            if (ok) { dir |= 1 << list; out[count].mv[list][0] = (int16_t)mv.x; out[count].mv[list][1] = (int16_t)mv.y; out[count].ref_idx[list] = 0; }
        }
        if (dir) { out[count].dir = (uint8_t)dir; if (++count == maxCand) return maxCand; }
    }
    if (isB)
    {
        const unsigned cutoff = (unsigned)(count * (count - 1));
        unsigned pl0 = 0xEDC984, pl1 = 0xB73621;
        for (unsigned k = 0; k < cutoff; k++, pl0 >>= 2, pl1 >>= 2)
        {
            const int i = pl0 & 3, j = pl1 & 3;
            if ((out[i].dir & 1) && (out[j].dir & 2))
            {
                const int r0 = out[i].ref_idx[0], r1 = out[j].ref_idx[1];
                if (!(I->ref_poc[0][r0] == I->ref_poc[1][r1] && mvEq(out[i].mv[0], out[j].mv[1])))
                {
                    out[count].mv[0][0] = out[i].mv[0][0]; out[count].mv[0][1] = out[i].mv[0][1]; out[count].ref_idx[0] = (int8_t)r0;
                    out[count].mv[1][0] = out[j].mv[1][0]; out[count].mv[1][1] = out[j].mv[1][1]; out[count].ref_idx[1] = (int8_t)r1;
                    out[count].dir = 3;
                    if (++count == maxCand) return maxCand;
                }
            }
        }
    }
    const int numRef = isB ? (I->num_ref_idx[0] < I->num_ref_idx[1] ? I->num_ref_idx[0] : I->num_ref_idx[1]) : I->num_ref_idx[0];
    int r = 0, refcnt = 0;
    while (count < maxCand)
    {
        out[count].dir = isB ? 3 : 1;
        out[count].mv[0][0] = out[count].mv[0][1] = 0; out[count].ref_idx[0] = (int8_t)r;
        if (isB) { out[count].mv[1][0] = out[count].mv[1][1] = 0; out[count].ref_idx[1] = (int8_t)r; }
        count++;
        if (refcnt == numRef - 1) r = 0;
        else { ++r; ++refcnt; }
    }
    return count;
}

int x265amd_amvp_candidates(const x265amd_mvpred_info* I, const x265amd_mv_unit* cur, const x265amd_mv_unit* col, int cuX, int cuY, int log2CU,
                            int partSize, int puIdx, int list, int refIdx, int16_t amvp[2][2], int16_t mvc[12][2])
{
    Ctx C{ I, cur, col, I->pic_width >> 2, I->pic_height >> 2 };
    const PuGeom p = puGeom(cuX, cuY, 1 << log2CU, partSize, puIdx);
    const int ltx = p.x, lty = p.y, rtx = p.x + p.w - 4, lby = p.y + p.h - 4;
    /* MVP_DIR order: LEFT, ABOVE, ABOVE_RIGHT, BELOW_LEFT, ABOVE_LEFT (cudata.h:63-71) */
    const x265amd_mv_unit* nb[5] = { C.left(ltx, lby), C.above(rtx, lty), C.aboveRight(rtx, lty), C.belowLeft(ltx, lby), C.aboveLeft(ltx, lty) };
    MV2 direct[5], indirect[5];
    bool vd[5], vi[5];
    const int curRefPOC = I->ref_poc[list][refIdx];
    for (int d = 0; d < 5; d++)
    {
        vd[d] = vi[d] = false;
        const x265amd_mv_unit* u = nb[d];
        if (!u) continue;
        /* getDirectPMV / getIndirectPMV (:1931-1966): the list asked for first, then the other one.  A neighbour that is not inter carries
         * ref_idx -1 in both lists (what CUData holds for intra blocks) */
        for (int k = 0, l = list; k < 2; k++, l = !l)
        {
            const int ri = u->ref_idx[l];
            if (ri >= 0 && curRefPOC == I->ref_poc[l][ri]) { direct[d] = MV2{ u->mv[l][0], u->mv[l][1] }; vd[d] = true; break; }
        }
        for (int k = 0, l = list; k < 2; k++, l = !l)
        {
            const int ri = u->ref_idx[l];
            if (ri >= 0) { indirect[d] = Ctx::scale(MV2{ u->mv[l][0], u->mv[l][1] }, I->poc, curRefPOC, I->poc, I->ref_poc[l][ri]); vi[d] = true; break; }
        }
    }
    enum { L = 0, A = 1, AR = 2, BL = 3, AL = 4 };
    MV2 cand[3]; int num = 0;
    if (vd[BL]) cand[num++] = direct[BL];
    else if (vd[L]) cand[num++] = direct[L];
    else if (vi[BL]) cand[num++] = indirect[BL];
    else if (vi[L]) cand[num++] = indirect[L];
    const bool addedSmvp = num > 0;
    if (vd[AR]) cand[num++] = direct[AR];
    else if (vd[A]) cand[num++] = direct[A];
    else if (vd[AL]) cand[num++] = direct[AL];
    if (!addedSmvp)
    {
        if (vi[AR]) cand[num++] = indirect[AR];
        else if (vi[A]) cand[num++] = indirect[A];
        else if (vi[AL]) cand[num++] = indirect[AL];
    }
    int numMvc = 0;
    for (int d = 0; d < 5; d++)
    {
        if (vd[d] && (direct[d].x || direct[d].y)) { mvc[numMvc][0] = (int16_t)direct[d].x; mvc[numMvc++][1] = (int16_t)direct[d].y; }
        if (vi[d] && (indirect[d].x || indirect[d].y)) { mvc[numMvc][0] = (int16_t)indirect[d].x; mvc[numMvc++][1] = (int16_t)indirect[d].y; }
    }
    if (num == 2) num -= cand[0].x == cand[1].x && cand[0].y == cand[1].y;
    if (I->temporal_mvp && num < 2)
    {
        /* getCollocatedMV (:2003-2028) at H, else at the centre; then the scaling of getPMV (:1789-1803) */
        int hx = 0, hy = 0;
        const x265amd_mv_unit* c = nullptr; bool have = false;
        auto tryAt = [&](int x, int y) {
            const x265amd_mv_unit* u = C.colUnit(x, y);
            const x265amd_mv_unit* cc = C.colUnit(x & ~15, y & ~15);
            if (u->pred_mode == X265AMD_MODE_NONE || cc->pred_mode == X265AMD_MODE_INTRA) return false;
            c = cc;
            /* "neighbour->unifiedRef != -1": both packed entries are -1 only when neither list of the co-located block holds a reference */
            int packed[2];
            for (int l = 0; l < 2; l++)
            {
                int cl = I->check_ldc ? l : I->col_from_l0;
                if (cc->ref_idx[cl] < 0) cl = !cl;
                packed[l] = (int16_t)(cc->ref_idx[cl] | (cl << 4));
            }
            return !(packed[0] == -1 && packed[1] == -1);
        };
        if (C.rightBottom(p.x, p.y, p.w, p.h, hx, hy)) have = tryAt(hx, hy);
        if (!have) have = tryAt(p.x + p.w / 2, p.y + p.h / 2);
        if (have)
        {
            int cl = I->check_ldc ? list : I->col_from_l0;
            if (c->ref_idx[cl] < 0) cl = !cl;
            const int packed = (int16_t)(c->ref_idx[cl] | (cl << 4));
            if (packed != -1)
            {
                const MV2 s = Ctx::scale(MV2{ c->mv[cl][0], c->mv[cl][1] }, I->poc, curRefPOC, I->col_poc, I->col_ref_poc[packed >> 4][packed & 15]);
                mvc[numMvc][0] = (int16_t)s.x; mvc[numMvc++][1] = (int16_t)s.y;
                cand[num++] = s;
            }
        }
    }
    while (num < 2) cand[num++] = MV2{ 0, 0 };
    amvp[0][0] = (int16_t)cand[0].x; amvp[0][1] = (int16_t)cand[0].y; amvp[1][0] = (int16_t)cand[1].x; amvp[1][1] = (int16_t)cand[1].y;
    return numMvc;
}

} // extern "C"
