/* SAO parameter decision of a picture (include/x265amd.h: x265amd_sao_rdo): host C++, the last piece of SURVEY section 8f rank 2.
 *
 * Restatement of SAO::startSlice (reference: source/encoder/sao.cpp:227-272: which planes are filtered at all, from the share of unfiltered
 * CTUs in earlier pictures of the same depth), rdoSaoUnitCu (:1225-1374), saoStatsInitialOffset (:1378-1433), estIterOffset (:1449-1478),
 * saoLumaComponentParamDist / saoChromaComponentParamDist (:1479-1761) and rdoSaoUnitRowEnd (:1207-1223) over the statistics
 * x265amd_sao_stats delivers for all CTUs at once.  The entropy state restarts with every CTU row, as every row owns its SAO object in
 * the reference (framefilter.cpp:239).  --limit-sao and sao-non-deblock are not supported. */
#include "cabac_coder.h"
#include <math.h>

namespace {

struct Snap { uint8_t ctx[X265AMD_CTX_STRIDE]; uint64_t frac; };
typedef int32_t PerPlane[3][5][32];
const int kOffsetThresh = 1 << ((X265AMD_DEPTH - 5) < 5 ? (X265AMD_DEPTH - 5) : 5);

inline int32_t roundIBDI(int32_t num, int32_t den) { return num >= 0 ? ((num * 2 + den) / (den * 2)) : -((-num * 2 + den) / (den * 2)); }
inline int64_t estSaoDist(int32_t count, int32_t offset, int32_t offsetOrg) { return (count * offset - offsetOrg * 2) * offset; }
inline int64_t rdoCost(int64_t dist, uint32_t bits, int64_t lambda) { return dist + ((bits * lambda + 128) >> 8); }

double lambda2(int qp)              /* x265_lambda2_tab by rule, as csrc/tu_kernels.hip */
{
    double v = floor(0.038 * exp(0.234 * (double)qp) * 10000.0) / 10000.0;
    return v * (double)(1 << (2 * (X265AMD_DEPTH - 8)));
}

struct Rdo
{
    x265amd_cabac* c;
    Snap cur, temp;
    PerPlane count, offsetOrg, offset;
    uint32_t bits() const { return (uint32_t)(c->fracBits >> 15); }
    void resetBits() { c->fracBits &= 32767; }
    void store(Snap& s) const { memcpy(s.ctx, c->ctx, X265AMD_CTX_STRIDE); s.frac = c->fracBits; }
    void load(const Snap& s) { memcpy(c->ctx, s.ctx, X265AMD_CTX_STRIDE); c->fracBits = s.frac; }

    void initialOffset(int p0, int p1)
    {
        for (int plane = p0; plane <= p1; plane++)
        {
            for (int t = 0; t < 4; t++)
                for (int cl = 1; cl < 5; cl++)
                    if (count[plane][t][cl])
                    {
                        int32_t o = roundIBDI(offsetOrg[plane][t][cl], count[plane][t][cl]);
                        o = o < -kOffsetThresh + 1 ? -kOffsetThresh + 1 : (o > kOffsetThresh - 1 ? kOffsetThresh - 1 : o);
                        offset[plane][t][cl] = cl < 3 ? (o > 0 ? o : 0) : (o < 0 ? o : 0);
                    }
            for (int cl = 0; cl < 32; cl++)
                if (count[plane][4][cl])
                {
                    int32_t o = roundIBDI(offsetOrg[plane][4][cl], count[plane][4][cl]);
                    offset[plane][4][cl] = o < -kOffsetThresh + 1 ? -kOffsetThresh + 1 : (o > kOffsetThresh - 1 ? kOffsetThresh - 1 : o);
                }
        }
    }
    void estIterOffset(int typeIdx, int64_t lambda, int32_t cnt, int32_t org, int32_t& off, int32_t& distClasses, int64_t& costClasses)
    {
        int bestOffset = 0;
        distClasses = 0;
        int64_t bestCost = rdoCost(0, 1, lambda);
        while (off != 0)
        {
            uint32_t rate = typeIdx == 4 ? (uint32_t)abs(off) + 2 : (uint32_t)abs(off) + 1;
            if (abs(off) == kOffsetThresh - 1) rate--;
            const int64_t dist = estSaoDist(cnt, off, org);
            const int64_t cost = rdoCost(dist, rate, lambda);
            if (cost < bestCost) { bestCost = cost; bestOffset = off; distClasses = (int)dist; }
            off = off > 0 ? off - 1 : off + 1;
        }
        costClasses = bestCost;
        off = bestOffset;
    }
    void codeEO(const int32_t* off, int typeIdx, int plane)
    {
        const uint32_t th = (uint32_t)kOffsetThresh - 1;
        if (plane != 2) { c->bin(1, C_SAO_TYPE); c->binEP(1); }
        c->saoMaxUvlc((uint32_t)off[0], th); c->saoMaxUvlc((uint32_t)off[1], th); c->saoMaxUvlc((uint32_t)-off[2], th); c->saoMaxUvlc((uint32_t)-off[3], th);
        if (plane != 2) c->binsEP((uint32_t)typeIdx, 2);
    }
    void codeBO(const int32_t* off, int bandPos, int plane)
    {
        const uint32_t th = (uint32_t)kOffsetThresh - 1;
        if (plane != 2) { c->bin(1, C_SAO_TYPE); c->binEP(0); }
        for (int i = 0; i < 4; i++) c->saoMaxUvlc((uint32_t)abs(off[i]), th);
        for (int i = 0; i < 4; i++) if (off[i]) c->binEP(off[i] < 0);
        c->binsEP((uint32_t)bandPos, 5);
    }
    void luma(x265amd_sao_ctu& p, int64_t& rateDist, const int64_t* lambda)
    {
        int64_t bestDist = 0;
        int bestType = -1;
        int32_t distClasses[32]; int64_t costClasses[32];
        load(temp); resetBits();
        c->bin(0, C_SAO_TYPE);
        int64_t costPartBest = rdoCost(0, bits(), lambda[0]);
        for (int t = 0; t < 4; t++)
        {
            int64_t estDist = 0;
            for (int cl = 1; cl < 5; cl++)
            {
                estIterOffset(t, lambda[0], count[0][t][cl], offsetOrg[0][t][cl], offset[0][t][cl], distClasses[cl], costClasses[cl]);
                estDist += distClasses[cl];
            }
            load(temp); resetBits();
            codeEO(offset[0][t] + 1, t, 0);
            const int64_t cost = rdoCost(estDist, bits(), lambda[0]);
            if (cost < costPartBest) { costPartBest = cost; bestDist = estDist; bestType = t; }
        }
        if (bestType != -1)
        {
            p.reserved[0] = 0; p.type[0] = (int8_t)bestType; p.band_pos[0] = 0;
            for (int i = 0; i < 4; i++) p.offset[0][i] = (int8_t)offset[0][bestType][i + 1];
        }
        for (int cl = 0; cl < 32; cl++) estIterOffset(4, lambda[0], count[0][4][cl], offsetOrg[0][4][cl], offset[0][4][cl], distClasses[cl], costClasses[cl]);
        int32_t bestClassBO = 0;
        int64_t currentRDCost = costClasses[0] + costClasses[1] + costClasses[2] + costClasses[3];
        int64_t bestRDCostBO = currentRDCost;
        for (int i = 1; i < 32 - 4 + 1; i++)
        {
            currentRDCost -= costClasses[i - 1];
            currentRDCost += costClasses[i + 3];
            if (currentRDCost < bestRDCostBO) { bestRDCostBO = currentRDCost; bestClassBO = i; }
        }
        int64_t estDist = 0;
        for (int cl = bestClassBO; cl < bestClassBO + 4; cl++) estDist += distClasses[cl];
        load(temp); resetBits();
        codeBO(offset[0][4] + bestClassBO, bestClassBO, 0);
        const int64_t cost = rdoCost(estDist, bits(), lambda[0]);
        if (cost < costPartBest)
        {
            costPartBest = cost; bestDist = estDist;
            p.reserved[0] = 0; p.type[0] = 4; p.band_pos[0] = (uint8_t)bestClassBO;
            for (int i = 0; i < 4; i++) p.offset[0][i] = (int8_t)offset[0][4][i + bestClassBO];
        }
        rateDist = (bestDist << 8) / lambda[0];
        load(temp);
        c->saoOffset(p.type[0], p.band_pos[0], p.offset[0], 0);
        store(temp);
    }
    void chroma(x265amd_sao_ctu& p, int64_t& rateDist, const int64_t* lambda, int64_t& bestCost)
    {
        int64_t bestDist = 0;
        int bestType = -1;
        int64_t costClasses[32]; int32_t distClasses[32];
        int32_t bestClassBO[2] = { 0, 0 };
        load(temp); resetBits();
        c->bin(0, C_SAO_TYPE);
        int64_t costPartBest = rdoCost(0, bits(), lambda[1]);
        for (int t = 0; t < 4; t++)
        {
            int64_t estDist[2] = { 0, 0 };
            for (int comp = 1; comp < 3; comp++)
                for (int cl = 1; cl < 5; cl++)
                {
                    estIterOffset(t, lambda[1], count[comp][t][cl], offsetOrg[comp][t][cl], offset[comp][t][cl], distClasses[cl], costClasses[cl]);
                    estDist[comp - 1] += distClasses[cl];
                }
            load(temp); resetBits();
            for (int comp = 0; comp < 2; comp++) codeEO(offset[comp + 1][t] + 1, t, comp + 1);
            const int64_t cost = rdoCost(estDist[0] + estDist[1], bits(), lambda[1]);
            if (cost < costPartBest) { costPartBest = cost; bestDist = estDist[0] + estDist[1]; bestType = t; }
        }
        if (bestType != -1)
        {
            p.type[1] = (int8_t)bestType;
            for (int comp = 0; comp < 2; comp++)
            {
                p.band_pos[comp + 1] = 0;
                for (int i = 0; i < 4; i++) p.offset[comp + 1][i] = (int8_t)offset[comp + 1][bestType][i + 1];
            }
        }
        int64_t estDist[2];
        for (int comp = 1; comp < 3; comp++)
        {
            int64_t bestRDCostBO = 0x7FFFFFFFFFFFFFFFLL;
            for (int cl = 0; cl < 32; cl++) estIterOffset(4, lambda[1], count[comp][4][cl], offsetOrg[comp][4][cl], offset[comp][4][cl], distClasses[cl], costClasses[cl]);
            for (int i = 0; i < 32 - 4 + 1; i++)
            {
                int64_t cur = 0;
                for (int j = i; j < i + 4; j++) cur += costClasses[j];
                if (cur < bestRDCostBO) { bestRDCostBO = cur; bestClassBO[comp - 1] = i; }
            }
            estDist[comp - 1] = 0;
            for (int cl = bestClassBO[comp - 1]; cl < bestClassBO[comp - 1] + 4; cl++) estDist[comp - 1] += distClasses[cl];
        }
        load(temp); resetBits();
        for (int comp = 0; comp < 2; comp++) codeBO(offset[comp + 1][4] + bestClassBO[comp], bestClassBO[comp], comp + 1);
        const int64_t cost = rdoCost(estDist[0] + estDist[1], bits(), lambda[1]);
        if (cost < costPartBest)
        {
            costPartBest = cost; bestDist = estDist[0] + estDist[1];
            p.type[1] = 4;
            for (int comp = 0; comp < 2; comp++)
            {
                p.band_pos[comp + 1] = (uint8_t)bestClassBO[comp];
                for (int i = 0; i < 4; i++) p.offset[comp + 1][i] = (int8_t)offset[comp + 1][4][i + bestClassBO[comp]];
            }
        }
        rateDist += (bestDist << 8) / lambda[1];
        load(temp);
        c->saoOffset(p.type[1], p.band_pos[1], p.offset[1], 1);
        c->saoOffset(p.type[1], p.band_pos[2], p.offset[2], 2);
        store(temp);
        bestCost = rateDist + bits();
    }
};

} // namespace

extern "C" int x265amd_sao_rdo(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                               const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags)
{
    if (!si) return X265AMD_EINVAL;
    return x265amd_sao_rdo_rows(si, referenced, frame_threads, qp_min, qp_max, units, count, offset_org, depth_sao_rate, params, sao_flags, 0, (si->pic_height + 63) >> 6);
}

/* CTU rows ctu_row_begin .. ctu_row_end - 1 of the same decision: every CTU row owns its SAO object and entropy state in the reference (framefilter.cpp:239), and
 * a CTU only looks at the parameters of its left and upper neighbours, so rows can be decided one by one in order.  The share of unfiltered CTUs
 * (depth_sao_rate, rdoSaoUnitRowEnd) is only kept when the call covers the whole picture; the reference reads it with one frame thread only (sao.cpp:264). */
static int sao_rdo_range(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                         const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags,
                         int ctu_row_begin, int ctu_row_end, int colBegin, int colEnd, uint8_t* carry);

extern "C" int x265amd_sao_rdo_rows(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                                    const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags,
                                    int ctu_row_begin, int ctu_row_end)
{
    if (!si) return X265AMD_EINVAL;
    return sao_rdo_range(si, referenced, frame_threads, qp_min, qp_max, units, count, offset_org, depth_sao_rate, params, sao_flags, ctu_row_begin, ctu_row_end, 0,
                         (si->pic_width + 63) >> 6, nullptr);
}

extern "C" int x265amd_sao_rdo_cols(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                                    const int32_t* count, const int32_t* offset_org, x265amd_sao_ctu* params, int32_t* sao_flags,
                                    int ctu_row, int ctu_col_begin, int ctu_col_end, uint8_t* carry)
{
    if (!si || !carry || frame_threads <= 1 || ctu_col_begin < 0 || ctu_col_begin >= ctu_col_end || ctu_col_end > ((si->pic_width + 63) >> 6)) return X265AMD_EINVAL;
    double unused[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    return sao_rdo_range(si, referenced, frame_threads, qp_min, qp_max, units, count, offset_org, unused, params, sao_flags, ctu_row, ctu_row + 1, ctu_col_begin, ctu_col_end, carry);
}

static int sao_rdo_range(const x265amd_slice_info* si, int referenced, int frame_threads, int qp_min, int qp_max, x265amd_cu_unit* units,
                         const int32_t* count, const int32_t* offset_org, double* depth_sao_rate, x265amd_sao_ctu* params, int32_t* sao_flags,
                         int ctu_row_begin, int ctu_row_end, int colBegin, int colEnd, uint8_t* carry)
{
    if (!si || !units || !count || !offset_org || !depth_sao_rate || !params || !sao_flags) return X265AMD_EINVAL;
    const int ctuW = (si->pic_width + 63) >> 6, ctuH = (si->pic_height + 63) >> 6, numCtu = ctuW * ctuH, w4 = si->pic_width >> 2;
    if (ctu_row_begin < 0 || ctu_row_begin >= ctu_row_end || ctu_row_end > ctuH) return X265AMD_EINVAL;
    const bool whole = ctu_row_begin == 0 && ctu_row_end == ctuH;
    if (!whole && frame_threads == 1) return X265AMD_EINVAL;
    /* SAO::startSlice */
    const int refDepth = si->slice_type == 2 ? 0 : (si->slice_type == 1 ? 1 : 2 + !referenced);
    sao_flags[0] = 1; sao_flags[1] = 1;
    if (frame_threads == 1)
    {
        if (refDepth > 0 && depth_sao_rate[refDepth - 1] > 0.75) sao_flags[0] = 0;
        if (refDepth > 0 && depth_sao_rate[4 + refDepth - 1] > 0.5) sao_flags[1] = 0;
    }
    int numNoSao[2] = { 0, 0 };
    static const uint8_t chromaScale[70] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33, 34, 34,
                                             35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51, 51 };
    Rdo* R = new Rdo;
    R->c = x265amd_cabac_open(si, units, 1);
    if (!R->c) { delete R; return X265AMD_EINVAL; }
    Snap init;
    R->store(init);
    static_assert(sizeof(Snap) <= X265AMD_CTX_STRIDE + 8, "the caller's carry buffer holds a Snap");
    if (carry && colBegin) memcpy(&R->cur, carry, sizeof(Snap));          /* the row's state as the previous call left it */
    for (int addr = ctu_row_begin * ctuW; addr < ctu_row_end * ctuW; addr++)
    {
        const int idxX = addr % ctuW, row = addr / ctuW;
        if (idxX < colBegin || idxX >= colEnd) continue;
        if (!idxX) R->cur = init;                   /* every CTU row owns its SAO object and entropy state */
        x265amd_sao_ctu& p = params[addr];
        memset(&p, 0, sizeof(p));
        p.type[0] = p.type[1] = -1;
        if (!sao_flags[0] && !sao_flags[1]) continue;
        const x265amd_cu_unit& u0 = units[(row * 16) * w4 + idxX * 16];
        const int qp = u0.qp;
        int qpCb = qp < 0 ? 0 : (qp > 69 ? 69 : qp);
        qpCb = chromaScale[qpCb];
        qpCb = qpCb < qp_min ? qp_min : (qpCb > qp_max ? qp_max : qpCb);
        const int64_t lambda[2] = { (int64_t)floor(256.0 * lambda2(qp)), (int64_t)floor(256.0 * lambda2(qpCb)) };
        const bool allowMerge[2] = { idxX != 0, row != 0 };
        const int addrMerge[2] = { idxX ? addr - 1 : -1, row ? addr - ctuW : -1 };
        memset(R->count, 0, sizeof(PerPlane)); memset(R->offsetOrg, 0, sizeof(PerPlane)); memset(R->offset, 0, sizeof(PerPlane));
        for (int plane = 0; plane < 3; plane++)
        {
            if (!sao_flags[plane > 0]) continue;
            memcpy(R->count[plane], count + ((size_t)addr * 3 + plane) * 5 * 32, sizeof(int32_t) * 5 * 32);
            memcpy(R->offsetOrg[plane], offset_org + ((size_t)addr * 3 + plane) * 5 * 32, sizeof(int32_t) * 5 * 32);
        }
        R->load(R->cur); R->resetBits();
        if (allowMerge[0]) R->c->bin(0, C_SAO_MERGE);
        if (allowMerge[1]) R->c->bin(0, C_SAO_MERGE);
        R->store(R->temp);
        int64_t bestCost = 0, rateDist = 0;
        if (sao_flags[0]) { R->initialOffset(0, 0); R->luma(p, rateDist, lambda); }
        if (sao_flags[1]) { R->initialOffset(1, 2); R->chroma(p, rateDist, lambda, bestCost); }
        for (int mergeIdx = 0; mergeIdx < 2; mergeIdx++)
        {
            if (!allowMerge[mergeIdx]) continue;
            const x265amd_sao_ctu& src = params[addrMerge[mergeIdx]];
            int64_t mergeDist = 0;
            for (int plane = 0; plane < 3; plane++)
            {
                int64_t estDist = 0;
                const int typeIdx = src.type[plane > 0];
                if (typeIdx >= 0)
                {
                    const int bandPos = typeIdx == 4 ? src.band_pos[plane] : 1;
                    for (int cl = 0; cl < 4; cl++) estDist += estSaoDist(R->count[plane][typeIdx][cl + bandPos], src.offset[plane][cl], R->offsetOrg[plane][typeIdx][cl + bandPos]);
                }
                mergeDist += (estDist << 8) / lambda[!!plane];
            }
            R->load(R->cur); R->resetBits();
            if (allowMerge[0]) R->c->bin((uint32_t)(1 - mergeIdx), C_SAO_MERGE);
            if (allowMerge[1] && mergeIdx == 1) R->c->bin(1, C_SAO_MERGE);
            const int64_t mergeCost = mergeDist + R->bits();
            if (mergeCost < bestCost)
            {
                bestCost = mergeCost;
                R->store(R->temp);
                p.reserved[0] = (uint8_t)(mergeIdx ? 2 : 1);
                if (sao_flags[0]) { p.type[0] = src.type[0]; p.band_pos[0] = src.band_pos[0]; memcpy(p.offset[0], src.offset[0], 4); }
                if (sao_flags[1]) { p.type[1] = src.type[1]; for (int pl = 1; pl < 3; pl++) { p.band_pos[pl] = src.band_pos[pl]; memcpy(p.offset[pl], src.offset[pl], 4); } }
            }
        }
        if (p.type[0] < 0) numNoSao[0]++;
        if (p.type[1] < 0) numNoSao[1]++;
        R->load(R->temp);
        R->store(R->cur);
    }
    if (carry) memcpy(carry, &R->cur, sizeof(Snap));
    /* rdoSaoUnitRowEnd -- which the reference never reaches in a picture of one slice, with wavefronts or without: FrameFilter::processRow asks whether EVERY row's
     * reconstruction flag is set before it has set the last row's own (framefilter.cpp:622-647 against :650-664; without wavefronts the rows are filtered one after the
     * other by the same function, frameencoder.cpp:928-960), so the rates stay at their initial zero and SAO is never switched off by the picture before (seen in the
     * reference's own objects: all eight rates 0.000 after every picture of a --frame-threads 1 encode; the fixtures rc_ft1/ and cli_nowpp_ft1/ pin it -- rounds 2 to 6
     * had the update for pictures without wavefronts, which only a --no-wpp --frame-threads 1 encode ever showed).  X265AMD_SAO_RATES=1: the update as SAO::rdoSaoUnitRowEnd has it. */
    static const bool rates = getenv("X265AMD_SAO_RATES") && atoi(getenv("X265AMD_SAO_RATES")) != 0;
    if (whole && rates)
    {
        depth_sao_rate[refDepth] = sao_flags[0] ? numNoSao[0] / (double)numCtu : 1.0;
        depth_sao_rate[4 + refDepth] = sao_flags[1] ? numNoSao[1] / (double)numCtu : 1.0;
    }
    x265amd_cabac_close(R->c);
    delete R;
    return X265AMD_OK;
}
