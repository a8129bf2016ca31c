/* Residual RD of inter CUs (include/x265amd.h: x265amd_inter_residual_rd): SURVEY row a8.
 *
 * Restatement of Search::encodeResAndCalcRdInterCU (reference: source/encoder/search.cpp:2822-2975) with estimateResidualQT
 * (:3178-3857), splitTU (:3126-3176), estimateNullCbfCost (:3114-3124), codeInterSubdivCbfQT (:3859-3887), saveResidualQTData
 * (:3889-3972), updateModeCost (search.h:437-439) and checkDQP (search.cpp:3974-4003) for a batch of independent candidate CUs.
 *
 * The reference evaluates one TU at a time: transform + quantise, count the coefficient bits with the RD entropy coder, inverse
 * transform, measure, compare with the zero-residual alternative, then recurse into the four sub-TUs and compare again.  With plain
 * quantisation (rdoqLevel 0) none of the block arithmetic depends on the entropy state, so here
 *   1. every node of every CU's residual quad-tree (all allowed transform sizes, luma + both chroma planes) runs as ONE
 *      x265amd_tu_chain launch, and one k_cu_measure launch prices the no-residual alternative of every CU;
 *   2. the host walks each tree in the reference's order with the bit-counting CABAC coder (host/cabac_coder.h), reading the
 *      per-node measurements and levels, and takes the reference's decisions (coded block flags, transform splits, root flag, skip);
 *   3. a second k_cu_measure launch assembles the chosen residual blocks, reconstructs every CU and measures the final distortion /
 *      psy energy.
 * The walk keeps the reference's quirks: the fractional bit carry of m_fracBits through resetBits(), coded block flags of parent depths
 * living only in a node's first unit, split energy added to the cost of a rejected split.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"
#include "../host/cabac_coder.h"
#include <string.h>
#include <vector>
#include <functional>

#include "measure_dev.h"
#include "xa_queue.h"

__global__ __launch_bounds__(64) void k_cu_measure(const CuMeasureJob* jobs, int n, CuMeasure* out)
{
    __shared__ pixel tile[64 * 64];
    if ((int)blockIdx.x >= n) return;
    wave_cu_measure_job(jobs, blockIdx.x, out, tile, xa_lane());
}

/* one CU per workgroup (block_cu_measure_job): the form for a handful of CUs, where the time of one counts */
#define MEASURE_WG_WAVES 8
__global__ __launch_bounds__(64 * MEASURE_WG_WAVES) void k_cu_measure_wg(const CuMeasureJob* jobs, int n, CuMeasure* out)
{
    __shared__ CuMeasureLds lds;
    if ((int)blockIdx.x >= n) return;
    const CuMeasureJob j = xa_ld_record(jobs + blockIdx.x);
    block_cu_measure_job(j, out + blockIdx.x, lds, threadIdx.x, 64 * MEASURE_WG_WAVES);
}
static bool measure_use_wg(int n)
{
    static const int wgMax = getenv("X265AMD_MEASURE_WG_MAX") ? atoi(getenv("X265AMD_MEASURE_WG_MAX")) : 64;
    return n <= wgMax;
}

/* ---------------- host: the reference's walk ---------------- */
namespace {

#if X265AMD_DEPTH < 10
typedef uint32_t sse_t;             /* common/common.h:142-146 */
#else
typedef uint64_t sse_t;
#endif

struct Snap { uint8_t ctx[X265AMD_CTX_STRIDE]; uint64_t frac; };
struct Cost { uint64_t rdcost; uint32_t bits; sse_t distortion; uint32_t energy; };       /* search.h:  struct Cost */
const uint64_t kMaxCost = 0x7FFFFFFFFFFFFFFFULL;
const uint32_t kCtxCbf[3][5] = { { 1, 0, 0, 0, 0 }, { 2, 3, 4, 5, 6 }, { 2, 3, 4, 5, 6 } };          /* contexts.h:120 */

struct CuPlan
{
    int x, y, log2, size, depth, qp, part;
    int range[2];
    int lumaRes[6];                 /* first result index of luma layer L (log2 size), -1 when the layer is not evaluated */
    int chromaRes[5][3];            /* the same of chroma layer C and plane 1 / 2 */
};

inline uint32_t zorder_in_cu(int x4, int y4)
{
    uint32_t z = 0;
    for (int b = 0; b < 4; b++) z |= (((uint32_t)x4 >> b) & 1) << (2 * b) | (((uint32_t)y4 >> b) & 1) << (2 * b + 1);
    return z;
}
inline size_t host_layer_offset(int plane, int layer)
{
    return plane ? (size_t)4 * RD_LUMA_ELEMS + (size_t)((layer - 2) * 2 + (plane - 1)) * RD_CHROMA_ELEMS : (size_t)(layer - 2) * RD_LUMA_ELEMS;
}

struct Walker
{
    x265amd_cabac& c;
    const CuPlan& P;
    const x265amd_tu_result* res;   /* all results of the launch */
    const int16_t* levels;          /* this CU's level scratch (host copy), RD_SCRATCH_ELEMS */
    uint64_t lambda2, lambda; uint32_t psyRd;
    Snap rqtRoot[6], rqtTest[6];

    /* RDOQ: the transform units are not quantised ahead of the walk; `demand(first job, count, luma?, size, tuDepth)` runs them when the walk reaches them,
     * under the entropy state it has then (it fills res / levels for those jobs) */
    std::function<int(int, int, bool, int, int, const uint8_t*)> demand;
    /* ... and a node's three units at once: demandNode(luma job, U job, V job, luma size, chroma size, tuDepth, contexts) -- the three run side by side on the device and the
     * walk waits once.  Sound because the chroma units' RDOQ tables are made from CHROMA contexts only (Entropy::estBit with bIsLuma false: coded block flags, significance,
     * greater-than, last position of the chroma sets), and nothing between the node's start and the point where the reference makes them (search.cpp:3397, behind the luma
     * unit's coefficients) moves one of those: the luma unit codes luma contexts.  The walk still compares the contexts when it gets there (chromaEstSame) and asks again
     * if they ever differ. */
    std::function<int(int, int, int, int, int, int, const uint8_t*)> demandNode;
    int limitTU = 0;            /* param.limitTU (1, the breadth-first form with its cache, is not built) */
    int maxTUDepth = -1;        /* Search::m_maxTUDepth during this CU's walk */
    /* the contexts Entropy::estBit reads for a chroma unit (entropy.cpp:2236-2350 with bIsLuma false; device form: wave_est_bit) */
    static bool chromaEstSame(const uint8_t* a, const uint8_t* b)
    {
        return !memcmp(a + C_QT_CBF, b + C_QT_CBF, 7) && a[C_QT_ROOT_CBF] == b[C_QT_ROOT_CBF] && !memcmp(a + C_SIG_CG + 2, b + C_SIG_CG + 2, 2) &&
               !memcmp(a + C_SIG + 27, b + C_SIG + 27, 15) && !memcmp(a + C_ONE + 16, b + C_ONE + 16, 8) && !memcmp(a + C_ABS + 4, b + C_ABS + 4, 2) &&
               !memcmp(a + C_LAST_X + 15, b + C_LAST_X + 15, 3) && !memcmp(a + C_LAST_X + 18 + 15, b + C_LAST_X + 18 + 15, 3);
    }
    int err = 0;

    Walker(x265amd_cabac& coder, const CuPlan& plan, const x265amd_tu_result* r, const int16_t* lv) : c(coder), P(plan), res(r), levels(lv) {}
    int nodeJob(int plane, int layer, int x, int y) const
    {
        const int sh = plane ? 1 : 0, n = 1 << layer, nt = (P.size >> sh) >> layer;
        const int tx = ((x - P.x) >> sh) / n, ty = ((y - P.y) >> sh) / n;
        return (plane ? P.chromaRes[layer][plane] : P.lumaRes[layer]) + ty * nt + tx;
    }

    /* Entropy: getNumberOfWrittenBits / resetBits / load / store / bitsCodeBin (entropy.h:120-140, :219-224; entropy.cpp:2445-2455) */
    uint32_t bits() const { return (uint32_t)(c.fracBits >> 15); }
    void resetBits() { c.fracBits &= 32767; }
    void store(Snap& s) const { memcpy(s.ctx, c.ctx, X265AMD_CTX_STRIDE); s.frac = c.fracBits; }
    void load(const Snap& s) { memcpy(c.ctx, s.ctx, X265AMD_CTX_STRIDE); c.fracBits = s.frac; }
    uint32_t bitsCodeBin(uint32_t bin, int ctxIdx) const { return (uint32_t)(((c.fracBits & 32767) + k_bits[c.ctx[ctxIdx] ^ bin]) >> 15); }
    uint32_t estimateCbfBits(uint32_t cbf, int ttype, int tuDepth) const { return bitsCodeBin(cbf, C_QT_CBF + (int)kCtxCbf[ttype][tuDepth]); }

    /* RDCost (rdcost.h:99-153) */
    uint64_t calcRdCost(sse_t dist, uint32_t b) const { return dist + (((uint64_t)b * lambda2 + 128) >> 8); }
    uint64_t calcPsyRdCost(sse_t dist, uint32_t b, uint32_t energy) const { return dist + ((lambda * psyRd * energy) >> 24) + (((uint64_t)b * lambda2) >> 8); }
    uint64_t cost(sse_t dist, uint32_t b, uint32_t energy) const { return psyRd ? calcPsyRdCost(dist, b, energy) : calcRdCost(dist, b); }
    uint64_t estimateNullCbfCost(sse_t dist, uint32_t energy, int tuDepth, int ttype) const { return cost(dist, estimateCbfBits(0, ttype, tuDepth), energy); }

    x265amd_cu_unit& U(int x, int y) { return c.U(x >> 2, y >> 2); }
    bool cbfBit(int x, int y, int plane, int d) { return (U(x, y).cbf[plane] >> d) & 1; }
    void setTuDepth(int x, int y, int size, int d)
    {
        for (int yy = y; yy < y + size; yy += 4) for (int xx = x; xx < x + size; xx += 4) U(xx, yy).tu_depth = (uint8_t)d;
    }
    void setCbf(int plane, int x, int y, int size, int v)           /* setCbfSubParts / setCbfPartRange: plain assignment */
    {
        for (int yy = y; yy < y + size; yy += 4) for (int xx = x; xx < x + size; xx += 4) U(xx, yy).cbf[plane] = (uint8_t)v;
    }

    const x265amd_tu_result& nodeResult(int plane, int layer, int x, int y) const
    {
        const int sh = plane ? 1 : 0, n = 1 << layer, nt = (P.size >> sh) >> layer;
        const int tx = ((x - P.x) >> sh) / n, ty = ((y - P.y) >> sh) / n;
        return res[(plane ? P.chromaRes[layer][plane] : P.lumaRes[layer]) + ty * nt + tx];
    }
    const int16_t* nodeLevels(int plane, int layer, int x, int y) const
    {
        const int sh = plane ? 1 : 0, n = 1 << layer, nt = (P.size >> sh) >> layer;
        const int tx = ((x - P.x) >> sh) / n, ty = ((y - P.y) >> sh) / n;
        return levels + host_layer_offset(plane, layer) + (size_t)(ty * nt + tx) * n * n;
    }

    void estimateResidualQT(int x, int y, int tuDepth, Cost& outCosts)
    {
        const int log2TrSize = P.log2 - tuDepth, depth = P.depth + tuDepth, trSize = 1 << log2TrSize;
        bool bCheckSplit = log2TrSize > P.range[0];
        bool bCheckFull = log2TrSize <= P.range[1];
        /* --limit-tu 2 / 3 / 4 (search.cpp:3209-3216): no transform units below the depth the first quarter of the CU settled on (depth first) or the neighbourhood suggests */
        if (limitTU >= 2 && bCheckSplit && maxTUDepth >= 0) bCheckSplit = log2TrSize > P.log2 - maxTUDepth;
        const bool bSplitPresentFlag = bCheckSplit && bCheckFull;
        if (P.part != 0 && !tuDepth && bCheckSplit) bCheckFull = false;

        int log2TrSizeC = log2TrSize - 1;
        bool codeChroma = true;
        if (log2TrSizeC < 2)
        {
            log2TrSizeC = 2;
            codeChroma = !((x & 4) || (y & 4));             /* !(absPartIdx & 3): the quartet's first 4x4 carries the chroma block */
        }
        const int chromaArea = trSize < 8 ? 8 : trSize;     /* luma extent of the units the chroma block covers */

        Cost fullCost = { kMaxCost, 0, 0, 0 };
        uint8_t fullChromaCtx[X265AMD_CTX_STRIDE];
        uint32_t cbfFlag[3] = { 0, 0, 0 }, singleBits[3] = { 0, 0, 0 }, singleEnergy[3] = { 0, 0, 0 };
        sse_t singleDist[3] = { 0, 0, 0 };

        store(rqtRoot[depth]);

        uint8_t nodeStartCtx[X265AMD_CTX_STRIDE];
        bool chromaAhead = false;
        if (bCheckFull)
        {
            setTuDepth(x, y, trSize, tuDepth);
            if (demand && !err)
            {
                if (codeChroma && demandNode)
                {
                    memcpy(nodeStartCtx, c.ctx, X265AMD_CTX_COUNT);
                    err = demandNode(nodeJob(0, log2TrSize, x, y), nodeJob(1, log2TrSizeC, x, y), nodeJob(2, log2TrSizeC, x, y), log2TrSize, log2TrSizeC, tuDepth, c.ctx);
                    chromaAhead = !err;
                }
                else err = demand(nodeJob(0, log2TrSize, x, y), 1, true, log2TrSize, tuDepth, c.ctx);
            }
            {
                const x265amd_tu_result& r = nodeResult(0, log2TrSize, x, y);
                cbfFlag[0] = r.num_sig != 0;
                resetBits();
                if (bSplitPresentFlag && log2TrSize > P.range[0]) c.bin(0, C_TRANS_SUBDIV + 5 - log2TrSize);
                if (cbfFlag[0]) c.coeffNxN(nodeLevels(0, log2TrSize, x, y), log2TrSize, 0, U(x, y));
                singleBits[0] = bits();
                const sse_t zeroDist = (sse_t)r.zero_dist;
                const uint32_t zeroEnergy = psyRd ? r.zero_energy : 0;
                if (cbfFlag[0])
                {
                    const sse_t nzDist = (sse_t)r.nz_dist;
                    const uint32_t nzCbfBits = estimateCbfBits(1, 0, tuDepth);
                    const uint32_t nzEnergy = psyRd ? r.nz_energy : 0;
                    const uint64_t singleCost = cost(nzDist, nzCbfBits + singleBits[0], nzEnergy);
                    const uint64_t nullCost = estimateNullCbfCost(zeroDist, zeroEnergy, tuDepth, 0);
                    if (nullCost < singleCost)
                    {
                        cbfFlag[0] = 0; singleBits[0] = 0;
                        singleDist[0] = zeroDist; singleEnergy[0] = zeroEnergy;
                    }
                    else { singleDist[0] = nzDist; singleEnergy[0] = nzEnergy; }
                }
                else { singleDist[0] = zeroDist; singleBits[0] = 0; singleEnergy[0] = zeroEnergy; }
                setCbf(0, x, y, trSize, cbfFlag[0] << tuDepth);
            }
            if (codeChroma)
                for (int p = 1; p < 3; p++)
                {
                    /* one table for both chroma planes, taken before U is coded (search.cpp:3397) */
                    if (p == 1 && demand && !err)
                    {
                        memcpy(fullChromaCtx, c.ctx, X265AMD_CTX_COUNT);
                        if (!(chromaAhead && chromaEstSame(nodeStartCtx, c.ctx)))
                        {
                            err = demand(nodeJob(1, log2TrSizeC, x, y), 1, false, log2TrSizeC, tuDepth, c.ctx);
                            if (!err) err = demand(nodeJob(2, log2TrSizeC, x, y), 1, false, -log2TrSizeC, tuDepth, c.ctx);      /* negative size: keep the table */
                        }
                    }
                    const x265amd_tu_result& r = nodeResult(p, log2TrSizeC, x, y);
                    cbfFlag[p] = r.num_sig != 0;
                    const uint32_t latestBitCount = bits();
                    if (cbfFlag[p]) c.coeffNxN(nodeLevels(p, log2TrSizeC, x, y), log2TrSizeC, p, U(x, y));
                    singleBits[p] = bits() - latestBitCount;
                    const sse_t zeroDist = (sse_t)r.zero_dist;          /* scaleChromaDist: weight 256 for 4:2:0 (rdcost.h:80-90) */
                    const uint32_t zeroEnergy = psyRd ? r.zero_energy : 0;
                    if (cbfFlag[p])
                    {
                        const sse_t nzDist = (sse_t)r.nz_dist;
                        const uint32_t nzCbfBits = estimateCbfBits(1, p, tuDepth);
                        const uint32_t nzEnergy = psyRd ? r.nz_energy : 0;
                        const uint64_t singleCost = cost(nzDist, nzCbfBits + singleBits[p], nzEnergy);
                        const uint64_t nullCost = estimateNullCbfCost(zeroDist, zeroEnergy, tuDepth, p);
                        if (nullCost < singleCost)
                        {
                            cbfFlag[p] = 0; singleBits[p] = 0;
                            singleDist[p] = zeroDist; singleEnergy[p] = zeroEnergy;
                        }
                        else { singleDist[p] = nzDist; singleEnergy[p] = nzEnergy; }
                    }
                    else { singleBits[p] = 0; singleDist[p] = zeroDist; singleEnergy[p] = zeroEnergy; }
                    setCbf(p, x, y, chromaArea, cbfFlag[p] << tuDepth);
                }

            /* the flags are priced from the node's start state; the coefficient bits were collected above (:3652-3690) */
            load(rqtRoot[depth]);
            resetBits();
            if (codeChroma)
            {
                c.bin(cbfFlag[1], C_QT_CBF + 2 + tuDepth);
                c.bin(cbfFlag[2], C_QT_CBF + 2 + tuDepth);
            }
            c.bin(cbfFlag[0], C_QT_CBF + !tuDepth);
            const uint32_t cbfBits = bits();
            const uint32_t coeffBits = singleBits[0] + singleBits[1] + singleBits[2];
            fullCost.bits = bSplitPresentFlag ? cbfBits + coeffBits : coeffBits;
            fullCost.distortion += singleDist[0];
            fullCost.energy += singleEnergy[0];
            fullCost.distortion += singleDist[1];
            fullCost.distortion += singleDist[2];
            fullCost.rdcost = cost(fullCost.distortion, fullCost.bits, fullCost.energy);
            if (limitTU && bCheckSplit)
            {
                /* "stop recursion if the TU's energy level is minimal" (search.cpp:3713-3726): no luma level, or a few levels that are all +-1 */
                const uint32_t numCoeff = (uint32_t)trSize * trSize, numSigY = nodeResult(0, log2TrSize, x, y).num_sig;
                if (!cbfFlag[0]) bCheckSplit = false;
                else if (numSigY < numCoeff / 64)
                {
                    uint32_t energy = 0;
                    const int16_t* lv = nodeLevels(0, log2TrSize, x, y);
                    for (uint32_t i = 0; i < numCoeff; i++) energy += (uint32_t)abs(lv[i]);
                    if (energy == numSigY) bCheckSplit = false;
                }
            }
        }

        if (bCheckSplit)
        {
            if (bCheckFull)
            {
                store(rqtTest[depth]);
                load(rqtRoot[depth]);
            }
            Cost splitCost = { 0, 0, 0, 0 };
            if (bSplitPresentFlag && (log2TrSize <= P.range[1] && log2TrSize > P.range[0]))
            {
                resetBits();
                c.bin(1, C_TRANS_SUBDIV + 5 - log2TrSize);
                splitCost.bits = bits();
            }
            const bool yCbCrCbf = splitTU(x, y, tuDepth, splitCost);
            if (yCbCrCbf || !bCheckFull)
            {
                if (splitCost.rdcost < fullCost.rdcost)
                {
                    outCosts.distortion += splitCost.distortion;
                    outCosts.rdcost += splitCost.rdcost;
                    outCosts.bits += splitCost.bits;
                    outCosts.energy += splitCost.energy;
                    return;
                }
                else
                    outCosts.energy += splitCost.energy;
            }
            load(rqtTest[depth]);
            /* RDOQ: an 8x8 node and its 4x4 children own the SAME 4x4 chroma blocks (one slot in the scratch, where the reference has one buffer per
             * layer); the children's pass re-quantised them under its own entropy state.  The full node won: bring its version back */
            if (demand && !err && bCheckFull && codeChroma && log2TrSize == 3)
            {
                err = demand(nodeJob(1, log2TrSizeC, x, y), 1, false, log2TrSizeC, tuDepth, fullChromaCtx);
                if (!err) err = demand(nodeJob(2, log2TrSizeC, x, y), 1, false, -log2TrSizeC, tuDepth, fullChromaCtx);
            }
        }

        setTuDepth(x, y, trSize, tuDepth);
        setCbf(0, x, y, trSize, cbfFlag[0] << tuDepth);
        if (codeChroma)
        {
            setCbf(1, x, y, trSize, cbfFlag[1] << tuDepth);
            setCbf(2, x, y, trSize, cbfFlag[2] << tuDepth);
        }
        outCosts.distortion += fullCost.distortion;
        outCosts.rdcost += fullCost.rdcost;
        outCosts.bits += fullCost.bits;
        outCosts.energy += fullCost.energy;
    }

    bool splitTU(int x, int y, int tuDepth, Cost& splitCost)
    {
        const int depth = P.depth + tuDepth, log2TrSize = P.log2 - tuDepth, half = 1 << (log2TrSize - 1);
        uint32_t ycbf = 0, ucbf = 0, vcbf = 0;
        for (int q = 0; q < 4; q++)
        {
            const int qx = x + (q & 1) * half, qy = y + (q >> 1) * half;
            if ((limitTU == 2 || limitTU == 4) && tuDepth == 0 && q == 1)
            {
                /* depth first (search.cpp:3136-3142): the deepest transform unit of the CU's first quarter bounds the other three */
                maxTUDepth = 0;
                for (int yy = 0; yy < half; yy += 4) for (int xx = 0; xx < half; xx += 4) maxTUDepth = std::max(maxTUDepth, (int)U(P.x + xx, P.y + yy).tu_depth);
            }
            estimateResidualQT(qx, qy, tuDepth + 1, splitCost);
            ycbf |= cbfBit(qx, qy, 0, tuDepth + 1);
            ucbf |= cbfBit(qx, qy, 1, tuDepth + 1);
            vcbf |= cbfBit(qx, qy, 2, tuDepth + 1);
        }
        U(x, y).cbf[0] |= (uint8_t)(ycbf << tuDepth);
        U(x, y).cbf[1] |= (uint8_t)(ucbf << tuDepth);
        U(x, y).cbf[2] |= (uint8_t)(vcbf << tuDepth);

        load(rqtRoot[depth]);
        resetBits();
        codeInterSubdivCbfQT(x, y, tuDepth);
        splitCost.bits += bits();
        splitCost.rdcost = cost(splitCost.distortion, splitCost.bits, splitCost.energy);
        return ycbf || ucbf || vcbf;
    }

    void codeInterSubdivCbfQT(int x, int y, int tuDepth)
    {
        const bool bSubdiv = tuDepth < U(x, y).tu_depth;
        const int log2TrSize = P.log2 - tuDepth;
        if (!(log2TrSize - 1 < 2))
        {
            const int psz = 2 << log2TrSize;
            const int px = P.x + ((x - P.x) & ~(psz - 1)), py = P.y + ((y - P.y) & ~(psz - 1));
            for (int p = 1; p < 3; p++)
                if (!tuDepth || cbfBit(px, py, p, tuDepth - 1))
                {
                    /* Entropy::codeQtCbfChroma(cu, ...) (entropy.cpp:1758-1780) */
                    const bool canQuadSplit = log2TrSize - 1 > 2;
                    const int lowest = tuDepth + ((bSubdiv && !canQuadSplit) ? 1 : 0);
                    c.bin(cbfBit(x, y, p, lowest), C_QT_CBF + tuDepth + 2);
                }
        }
        if (!bSubdiv) c.bin(cbfBit(x, y, 0, tuDepth), C_QT_CBF + !tuDepth);
        else
        {
            const int half = 1 << (log2TrSize - 1);
            for (int q = 0; q < 4; q++) codeInterSubdivCbfQT(x + (q & 1) * half, y + (q >> 1) * half, tuDepth + 1);
        }
    }

    /* saveResidualQTData: which layer's blocks make up the final residual / levels */
    void collect(int x, int y, int tuDepth, uint8_t* sel, int16_t* coeffCu)
    {
        const int log2TrSize = P.log2 - tuDepth;
        if (tuDepth < U(x, y).tu_depth)
        {
            const int half = 1 << (log2TrSize - 1);
            for (int q = 0; q < 4; q++) collect(x + (q & 1) * half, y + (q >> 1) * half, tuDepth + 1, sel, coeffCu);
            return;
        }
        const int trSize = 1 << log2TrSize;
        const int ux = (x - P.x) >> 2, uy = (y - P.y) >> 2;
        const uint32_t z = zorder_in_cu(ux, uy);
        if (cbfBit(x, y, 0, tuDepth))
            for (int yy = 0; yy < trSize >> 2; yy++) for (int xx = 0; xx < trSize >> 2; xx++) sel[(uy + yy) * 16 + ux + xx] = (uint8_t)log2TrSize;
        if (coeffCu) memcpy(coeffCu + (z << 4), nodeLevels(0, log2TrSize, x, y), sizeof(int16_t) << (2 * log2TrSize));
        int log2TrSizeC = log2TrSize - 1;
        bool codeChroma = true;
        if (log2TrSizeC < 2) { log2TrSizeC = 2; codeChroma = !((x & 4) || (y & 4)); }
        if (!codeChroma) return;
        const int cs = 1 << log2TrSizeC;
        for (int p = 1; p < 3; p++)
        {
            if (cbfBit(x, y, p, tuDepth))
                for (int yy = 0; yy < cs >> 2; yy++) for (int xx = 0; xx < cs >> 2; xx++) sel[256 + (p - 1) * 64 + ((uy >> 1) + yy) * 8 + (ux >> 1) + xx] = (uint8_t)log2TrSizeC;
            if (coeffCu) memcpy(coeffCu + 4096 + (p - 1) * 1024 + ((z << 4) >> 2), nodeLevels(p, log2TrSizeC, x, y), sizeof(int16_t) << (2 * log2TrSizeC));
        }
    }
};

struct DevBuf
{
    void* p = nullptr;
    ~DevBuf() { xa_scratch_free(p); }
    hipError_t alloc(size_t bytes) { return xa_scratch_alloc(&p, bytes ? bytes : 16); }
};

} // namespace

/* the QP of a CU's lambdas (x265amd_rd_cu.reserved[0] when it differs from the quantiser's: QPs above 51) */
static inline int lambda_qp(const x265amd_rd_cu& cu) { return cu.reserved[0] ? (int)cu.reserved[0] : (int)cu.qp; }
/* RDCost::setQP for one CU */
static void rd_lambdas(const x265amd_slice_info* si, const x265amd_rd_params* rp, int qp, Walker& w)
{
    uint64_t rd[6];
    x265amd_rdcost(qp, si->slice_type, rp->psy_rd, 0, 0, 0, rd);
    w.lambda2 = rd[0]; w.lambda = rd[1]; w.psyRd = (uint32_t)rd[2];
}

/* the nodes of one CU's tree; jobs == NULL: only the plan */
static int make_plan(const x265amd_slice_info* si, const x265amd_rd_cu& cu, int part, CuPlan& P, int firstJob, const uint64_t* src, intptr_t stride, intptr_t cstride,
                     uint64_t tile, uint64_t scratch, std::vector<x265amd_tu_job>* jobs, uint64_t levels = 0)
{
    XA_HOSTPROF("rd.make_plan");
    static const uint8_t chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31,
                                             32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };       /* H.265 table 8-10 */
    P.x = cu.x; P.y = cu.y; P.log2 = cu.log2_size; P.size = 1 << P.log2; P.depth = 6 - P.log2; P.qp = cu.qp;
    if (P.log2 < 3 || P.log2 > 6 || (P.x & (P.size - 1)) || (P.y & (P.size - 1)) || P.x < 0 || P.y < 0 || P.x + P.size > si->pic_width || P.y + P.size > si->pic_height)
        return xa_fail(X265AMD_EINVAL, "inter_residual_rd: CU outside the picture or misaligned");
    P.part = part;
    /* CUData::getInterTUQtDepthRange (cudata.cpp:983-993) */
    const int splitFlag = si->tu_max_depth_inter == 1 && P.part != 0;
    const uint32_t lo = (uint32_t)P.log2 - (uint32_t)(si->tu_max_depth_inter - 1 + splitFlag);      /* unsigned as in the reference */
    P.range[0] = lo < (uint32_t)si->tu_log2_min ? si->tu_log2_min : (lo > (uint32_t)si->tu_log2_max ? si->tu_log2_max : (int)lo);
    P.range[1] = si->tu_log2_max;
    if (P.range[0] < 2 || P.range[1] > 5 || P.range[0] > P.range[1]) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: transform size range");
    /* Quant::setQPforQuant (quant.cpp:221-244), chroma QP offsets 0 */
    const int qpQuant = P.qp < 0 ? 0 : (P.qp > 51 ? 51 : P.qp);
    const int bd = 6 * (X265AMD_DEPTH - 8);
    int qpC = qpQuant < -bd ? -bd : (qpQuant > 57 ? 57 : qpQuant);
    if (qpC >= 30) qpC = chromaScale[qpC];
    const int hi = P.log2 < P.range[1] ? P.log2 : P.range[1];
    for (int L = 0; L < 6; L++) P.lumaRes[L] = -1;
    for (int C = 0; C < 5; C++) P.chromaRes[C][0] = P.chromaRes[C][1] = P.chromaRes[C][2] = -1;
    /* levels: where the chains leave the quantised levels when not at the head of the scratch -- pinned host memory, so that they reach the host with the
     * chains' own stores instead of a copy command behind them (the whole level area of a CU is 45 KB, a small CU uses a fraction) */
    const uint64_t levelBase = levels ? levels : scratch, resiBase = scratch + (uint64_t)RD_SCRATCH_ELEMS * 2, dumpBase = scratch + (uint64_t)RD_SCRATCH_ELEMS * 4;
    int count = firstJob;
    x265amd_tu_job j;
    memset(&j, 0, sizeof(j));
    j.slice_type = (uint8_t)si->slice_type; j.sign_hide = (uint8_t)(si->sign_hide != 0);
    for (int L = hi; L >= P.range[0]; L--)
    {
        const int N = 1 << L, nt = P.size >> L;
        P.lumaRes[L] = count;
        count += nt * nt;
        if (!jobs) continue;
        j.log2_tr_size = (uint8_t)L; j.ttype = 0; j.qp_scaled = (uint8_t)(qpQuant + bd);
        j.fenc_stride = (int32_t)stride; j.pred_stride = 64; j.resi_stride = 64; j.recon_stride = 64;
        for (int ty = 0; ty < nt; ty++)
            for (int tx = 0; tx < nt; tx++)
            {
                const size_t tileOff = (size_t)ty * N * 64 + (size_t)tx * N;
                j.fenc = src[0] + ((uint64_t)(P.y + ty * N) * stride + P.x + tx * N) * sizeof(pixel);
                j.pred = tile + tileOff * sizeof(pixel);
                j.coeff = levelBase + (host_layer_offset(0, L) + (size_t)(ty * nt + tx) * N * N) * 2;
                j.resi = resiBase + (host_layer_offset(0, L) + tileOff) * 2;
                j.recon = dumpBase + (host_layer_offset(0, L) + tileOff) * sizeof(pixel);
                jobs->push_back(j);
            }
    }
    for (int C = 4; C >= 2; C--)
    {
        const bool wanted = (C + 1 <= hi && C + 1 >= P.range[0]) || (C == 2 && P.range[0] == 2);
        if (!wanted) continue;
        const int N = 1 << C, nt = (P.size >> 1) >> C;
        for (int p = 1; p < 3; p++)
        {
            P.chromaRes[C][p] = count;
            count += nt * nt;
            if (!jobs) continue;
            j.log2_tr_size = (uint8_t)C; j.ttype = (uint8_t)p; j.qp_scaled = (uint8_t)(qpC + bd);
            j.fenc_stride = (int32_t)cstride; j.pred_stride = 32; j.resi_stride = 32; j.recon_stride = 32;
            for (int ty = 0; ty < nt; ty++)
                for (int tx = 0; tx < nt; tx++)
                {
                    const size_t tileOff = (size_t)ty * N * 32 + (size_t)tx * N;
                    j.fenc = src[p] + ((uint64_t)((P.y >> 1) + ty * N) * cstride + (P.x >> 1) + tx * N) * sizeof(pixel);
                    j.pred = tile + (4096 + (size_t)(p - 1) * 1024 + tileOff) * sizeof(pixel);
                    j.coeff = levelBase + (host_layer_offset(p, C) + (size_t)(ty * nt + tx) * N * N) * 2;
                    j.resi = resiBase + (host_layer_offset(p, C) + tileOff) * 2;
                    j.recon = dumpBase + (host_layer_offset(p, C) + tileOff) * sizeof(pixel);
                    jobs->push_back(j);
                }
        }
    }
    return count;
}

extern "C" size_t x265amd_inter_rd_scratch_bytes(void) { return (size_t)RD_SCRATCH_ELEMS * (2 + 2 + sizeof(pixel)); }

static int inter_rd_plan_levels(const x265amd_slice_info* si, const x265amd_rd_cu* cus, int n, const x265amd_cu_unit* cu_units, const uint64_t* src,
                                intptr_t stride, intptr_t cstride, uint64_t pred, size_t tile_bytes, uint64_t scratch, x265amd_tu_job* jobs_out, int cap, uint64_t levels);
extern "C" int x265amd_inter_rd_plan(const x265amd_slice_info* si, const x265amd_rd_cu* cus, int n, const x265amd_cu_unit* cu_units, const uint64_t* src,
                                     intptr_t stride, intptr_t cstride, uint64_t pred, size_t tile_bytes, uint64_t scratch, x265amd_tu_job* jobs_out, int cap)
{
    return inter_rd_plan_levels(si, cus, n, cu_units, src, stride, cstride, pred, tile_bytes, scratch, jobs_out, cap, 0);
}
/* levels != 0: CU i's levels go to levels + i * RD_SCRATCH_ELEMS * 2 (same layout as the head of its scratch) */
static int inter_rd_plan_levels(const x265amd_slice_info* si, const x265amd_rd_cu* cus, int n, const x265amd_cu_unit* cu_units, const uint64_t* src,
                                intptr_t stride, intptr_t cstride, uint64_t pred, size_t tile_bytes, uint64_t scratch, x265amd_tu_job* jobs_out, int cap, uint64_t levels)
{
    XA_HOSTPROF("rd.plan_levels");
    if (!si || !cus || !cu_units || !src || n < 0) return xa_fail(X265AMD_EINVAL, "inter_rd_plan: null argument");
    std::vector<x265amd_tu_job> jobs;
    const size_t perCu = x265amd_inter_rd_scratch_bytes();
    int count = 0;
    for (int i = 0; i < n; i++)
    {
        CuPlan P;
        count = make_plan(si, cus[i], cu_units[(size_t)i * 256].part_size, P, count, src, stride, cstride, pred + (uint64_t)tile_bytes * i, scratch + (uint64_t)perCu * i, &jobs,
                          levels ? levels + (uint64_t)RD_SCRATCH_ELEMS * 2 * i : 0);
        if (count < 0) return count;
    }
    if (jobs_out)
    {
        if (cap < count) return xa_fail(X265AMD_EINVAL, "inter_rd_plan: job array too small");
        memcpy(jobs_out, jobs.data(), sizeof(x265amd_tu_job) * jobs.size());
    }
    return count;
}

/* runs jobs [first, first + count) of CU i now (RDOQ): ctx = the walk's entropy state */
struct RdoqDemand
{
    std::function<int(int i, const uint8_t* ctx, int first, int count, bool luma, int log2TrSize, int tuDepth)> unit;
    std::function<int(int i, const uint8_t* ctx, int jobY, int jobU, int jobV, int log2TrSize, int log2TrSizeC, int tuDepth)> node;      /* may be empty */
};
static int inter_rd_walk_impl(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                              x265amd_cu_unit* cu_units, const x265amd_tu_result* res, const int16_t* levels, size_t levels_stride_bytes,
                              const x265amd_cu_measure* zero_meas, uint8_t* sel, x265amd_rd_result* out, int16_t* coeff_out, const RdoqDemand* demand);

extern "C" int x265amd_inter_rd_walk(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                                     x265amd_cu_unit* cu_units, const x265amd_tu_result* res, const int16_t* levels, size_t levels_stride_bytes,
                                     const x265amd_cu_measure* zero_meas, uint8_t* sel, x265amd_rd_result* out, int16_t* coeff_out)
{
    if (rp && rp->rdoq_level) return xa_fail(X265AMD_EINVAL, "inter_rd_walk: with RDOQ the transform units are quantised during the walk: use x265amd_inter_residual_rd");
    return inter_rd_walk_impl(si, rp, units, cus, n, cu_units, res, levels, levels_stride_bytes, zero_meas, sel, out, coeff_out, nullptr);
}

static int inter_rd_walk_impl(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                              x265amd_cu_unit* cu_units, const x265amd_tu_result* res, const int16_t* levels, size_t levels_stride_bytes,
                              const x265amd_cu_measure* zero_meas, uint8_t* sel, x265amd_rd_result* out, int16_t* coeff_out, const RdoqDemand* demand)
{
    XA_HOSTPROF("rd.walk_impl");
    if (!si || !rp || !units || !cus || !cu_units || !res || !levels || !zero_meas || !sel || !out || n < 0) return xa_fail(X265AMD_EINVAL, "inter_rd_walk: null argument");
    if (si->tq_bypass_enabled) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: lossless coding is not supported");
    const int w4 = si->pic_width >> 2;
    x265amd_cabac* coder = x265amd_cabac_open(si, units, 1);
    if (coder) coder->ctuInProgress = true;          /* the analysis of a CTU asks (cabac_coder.h: lastQP) */
    if (!coder) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: slice description");
    memset(sel, 0xFF, (size_t)RD_SEL_BYTES * n);
    std::vector<int16_t> coeffCu(4096 + 2048);
    std::vector<x265amd_cu_unit> saved(256);
    int firstJob = 0;
    for (int i = 0; i < n; i++)
    {
        CuPlan P;
        x265amd_cu_unit* mine = cu_units + (size_t)i * 256;
        const int next = make_plan(si, cus[i], mine[0].part_size, P, firstJob, nullptr, 0, 0, 0, 0, nullptr);
        if (next < 0) { x265amd_cabac_close(coder); return next; }
        firstJob = next;
        const x265amd_rd_cu& cu = cus[i];
        const int u4 = P.size >> 2;
        for (int yy = 0; yy < u4; yy++)
        {
            memcpy(&saved[yy * u4], &units[((P.y >> 2) + yy) * w4 + (P.x >> 2)], sizeof(x265amd_cu_unit) * u4);
            memcpy(&units[((P.y >> 2) + yy) * w4 + (P.x >> 2)], &mine[yy * u4], sizeof(x265amd_cu_unit) * u4);
        }
        Walker w(*coder, P, res, (const int16_t*)((const char*)levels + levels_stride_bytes * i));
        rd_lambdas(si, rp, lambda_qp(cu), w);
        for (int yy = 0; yy < u4; yy++)
            for (int xx = 0; xx < u4; xx++)
            {
                x265amd_cu_unit& u = units[((P.y >> 2) + yy) * w4 + (P.x >> 2) + xx];
                u.cbf[0] = u.cbf[1] = u.cbf[2] = 0; u.tu_depth = 0; u.depth = (uint8_t)P.depth;
            }
        Snap cur;
        memset(&cur, 0, sizeof(cur));
        memcpy(cur.ctx, cu.ctx, X265AMD_CTX_COUNT);
        cur.frac = cu.frac_bits;

        w.load(cur);
        Cost costs = { 0, 0, 0, 0 };
        w.limitTU = rp->limit_tu;
        if (rp->limit_tu >= 3)
        {
            /* Search::encodeResAndCalcRdInterCU (search.cpp:2851-2865): the neighbourhood's depth, kept inside what this CU's transform range allows */
            w.maxTUDepth = (int)cu.reserved[1] - 1;
            if (w.maxTUDepth != -1)
            {
                const int splitFlag = P.part != 0, minSize = P.range[0], maxSize = std::min(P.range[1], P.log2 - splitFlag);
                w.maxTUDepth = std::max(P.log2 - maxSize, std::min(P.log2 - minSize, w.maxTUDepth));
            }
        }
        if (demand)
        {
            w.demand = [&, i](int first, int count, bool luma, int log2TrSize, int tuDepth, const uint8_t* ctx) { return demand->unit(i, ctx, first, count, luma, log2TrSize, tuDepth); };
            if (demand->node)
                w.demandNode = [&, i](int jy, int ju, int jv, int l2, int l2c, int tuDepth, const uint8_t* ctx) { return demand->node(i, ctx, jy, ju, jv, l2, l2c, tuDepth); };
        }
        w.estimateResidualQT(P.x, P.y, 0, costs);
        if (w.err) { x265amd_cabac_close(coder); return w.err; }

        /* the RD cost of not signalling any residual (:2869-2895) */
        const x265amd_cu_measure& m0 = zero_meas[i];
        sse_t cbf0Dist = (sse_t)m0.sse[0];
        cbf0Dist += (sse_t)m0.sse[1];
        cbf0Dist += (sse_t)m0.sse[2];
        w.load(cur);
        w.resetBits();
        coder->bin(0, C_QT_ROOT_CBF);
        const uint32_t cbf0Bits = w.bits();
        const uint64_t cbf0Cost = w.cost(cbf0Dist, cbf0Bits, w.psyRd ? m0.psy : 0);
        if (cbf0Cost < costs.rdcost)
            for (int yy = 0; yy < u4; yy++)
                for (int xx = 0; xx < u4; xx++)
                {
                    x265amd_cu_unit& u = units[((P.y >> 2) + yy) * w4 + (P.x >> 2) + xx];
                    u.cbf[0] = u.cbf[1] = u.cbf[2] = 0; u.tu_depth = 0;
                }
        x265amd_cu_unit& u0 = units[(P.y >> 2) * w4 + (P.x >> 2)];
        const bool rootCbf = u0.cbf[0] || u0.cbf[1] || u0.cbf[2];
        std::fill(coeffCu.begin(), coeffCu.end(), 0);
        if (rootCbf) w.collect(P.x, P.y, 0, sel + (size_t)RD_SEL_BYTES * i, coeffCu.data());

        /* signal bits of the inter / merge / skip coded CU (:2900-2930) */
        w.load(cur);
        w.resetBits();
        uint32_t coeffBits, bits, mvBits;
        const x265amd_cu_unit* l = coder->at((P.x >> 2) - 1, P.y >> 2);
        const x265amd_cu_unit* a = coder->at(P.x >> 2, (P.y >> 2) - 1);
        const int skipCtx = (x265amd_cabac::coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (x265amd_cabac::coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
        if (u0.merge_flag && u0.part_size == 0 && !rootCbf)
        {
            for (int yy = 0; yy < u4; yy++)
                for (int xx = 0; xx < u4; xx++) units[((P.y >> 2) + yy) * w4 + (P.x >> 2) + xx].pred_mode = X265AMD_MODE_SKIP;
            coder->bin(1, C_SKIP + skipCtx);
            const uint32_t skipFlagBits = w.bits();
            coder->mergeIndex(u0);
            mvBits = w.bits() - skipFlagBits;
            coeffBits = 0;
            bits = mvBits + skipFlagBits;
        }
        else
        {
            coder->bin(0, C_SKIP + skipCtx);
            const uint32_t skipFlagBits = w.bits();
            coder->bin(0, C_PRED_MODE);
            coder->partSize(u0, P.depth, P.size);
            coder->predInfo(P.x, P.y, P.size, u0);
            mvBits = w.bits() - skipFlagBits;
            bool dqp = si->use_dqp != 0;
            /* Entropy::codeCoeff (entropy.cpp:1207-1222) */
            if (!(u0.merge_flag && u0.part_size == 0)) coder->bin(rootCbf, C_QT_ROOT_CBF);
            if (rootCbf)
            {
                coder->coeffCtu[0] = coeffCu.data(); coder->coeffCtu[1] = coeffCu.data() + 4096; coder->coeffCtu[2] = coeffCu.data() + 4096 + 1024;
                coder->ctuX0 = P.x; coder->ctuY0 = P.y;
                coder->transform(P.x, P.y, P.x, P.y, 0, P.log2, dqp, P.range);
            }
            bits = w.bits();
            coeffBits = bits - mvBits - skipFlagBits;
        }
        x265amd_rd_result& r = out[i];
        memset(&r, 0, sizeof(r));
        r.total_bits = bits; r.mv_bits = mvBits; r.coeff_bits = coeffBits; r.res_energy = (uint32_t)(sse_t)m0.sse[0];
        /* checkDQP's entropy side (:3974-4003); the cost follows in x265amd_inter_rd_finish once the distortion is known */
        if (si->use_dqp && P.depth <= si->max_cu_dqp_depth)
        {
            if (rootCbf)
            {
                if (rp->rd_level >= 3)
                {
                    w.resetBits();
                    coder->deltaQP(P.x, P.y);
                    r.total_bits += w.bits();
                }
                else if (rp->rd_level == 2) r.total_bits++;
            }
            else
            {
                const int8_t q = (int8_t)coder->refQP(P.x, P.y);
                for (int yy = 0; yy < u4; yy++)
                    for (int xx = 0; xx < u4; xx++) units[((P.y >> 2) + yy) * w4 + (P.x >> 2) + xx].qp = q;
            }
        }
        memcpy(r.ctx, coder->ctx, X265AMD_CTX_COUNT);
        r.frac_bits = coder->fracBits;
        if (coeff_out) memcpy(coeff_out + (size_t)i * (4096 + 2048), coeffCu.data(), sizeof(int16_t) * (4096 + 2048));
        for (int yy = 0; yy < u4; yy++)
        {
            memcpy(&mine[yy * u4], &units[((P.y >> 2) + yy) * w4 + (P.x >> 2)], sizeof(x265amd_cu_unit) * u4);
            memcpy(&units[((P.y >> 2) + yy) * w4 + (P.x >> 2)], &saved[yy * u4], sizeof(x265amd_cu_unit) * u4);
        }
    }
    x265amd_cabac_close(coder);
    return X265AMD_OK;
}

/* reconstruction-side results (:2932-2958), updateModeCost with the bits checkDQP may have added */
extern "C" void x265amd_inter_rd_finish(const x265amd_slice_info* si, const x265amd_rd_params* rp, const x265amd_rd_cu* cus, int n,
                                        const x265amd_cu_measure* final_meas, x265amd_rd_result* out)
{
    for (int i = 0; i < n; i++)
    {
        const x265amd_cu_measure& m1 = final_meas[i];
        x265amd_rd_result& r = out[i];
        uint64_t rd[6];
        x265amd_rdcost(lambda_qp(cus[i]), si->slice_type, rp->psy_rd, 0, 0, 0, rd);
        const sse_t bestLumaDist = (sse_t)m1.sse[0];
        sse_t bestChromaDist = (sse_t)m1.sse[1];
        bestChromaDist += (sse_t)m1.sse[2];
        const sse_t distortion = bestLumaDist + bestChromaDist;
        r.luma_distortion = (uint32_t)bestLumaDist; r.chroma_distortion = (uint32_t)bestChromaDist; r.distortion = distortion;
        r.psy_energy = rd[2] ? m1.psy : 0;
        x265amd_rdcost(lambda_qp(cus[i]), si->slice_type, rp->psy_rd, distortion, r.total_bits, r.psy_energy, rd);
        r.rd_cost = rd[2] ? rd[4] : rd[3];
    }
}

/* Search::encodeResAndCalcRdSkipCU (search.cpp:2770-2818): the entropy side and the cost, given prediction-vs-source measurements */
extern "C" int x265amd_skip_rd_host(const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const x265amd_rd_cu* cus, int n,
                                    x265amd_cu_unit* cu_units, const x265amd_cu_measure* meas, x265amd_rd_result* out)
{
    XA_HOSTPROF("rd.skip_rd_host");
    if (!si || !rp || !units || !cus || !cu_units || !meas || !out || n < 0) return xa_fail(X265AMD_EINVAL, "skip_rd: null argument");
    if (si->tq_bypass_enabled) return xa_fail(X265AMD_EINVAL, "skip_rd: lossless coding is not supported");
    const int w4 = si->pic_width >> 2;
    x265amd_cabac* coder = x265amd_cabac_open(si, units, 1);
    if (coder) coder->ctuInProgress = true;          /* the analysis of a CTU asks (cabac_coder.h: lastQP) */
    if (!coder) return xa_fail(X265AMD_EINVAL, "skip_rd: slice description");
    std::vector<x265amd_cu_unit> saved(256);
    for (int i = 0; i < n; i++)
    {
        const x265amd_rd_cu& cu = cus[i];
        const int log2 = cu.log2_size, size = 1 << log2, u4 = size >> 2, x = cu.x, y = cu.y;
        if (log2 < 3 || log2 > 6 || (x & (size - 1)) || (y & (size - 1)) || x < 0 || y < 0 || x + size > si->pic_width || y + size > si->pic_height)
        {
            x265amd_cabac_close(coder);
            return xa_fail(X265AMD_EINVAL, "skip_rd: CU outside the picture or misaligned");
        }
        x265amd_cu_unit* mine = cu_units + (size_t)i * 256;
        for (int yy = 0; yy < u4; yy++)
        {
            memcpy(&saved[yy * u4], &units[((y >> 2) + yy) * w4 + (x >> 2)], sizeof(x265amd_cu_unit) * u4);
            memcpy(&units[((y >> 2) + yy) * w4 + (x >> 2)], &mine[yy * u4], sizeof(x265amd_cu_unit) * u4);
        }
        for (int yy = 0; yy < u4; yy++)
            for (int xx = 0; xx < u4; xx++)
            {
                x265amd_cu_unit& u = units[((y >> 2) + yy) * w4 + (x >> 2) + xx];
                u.pred_mode = X265AMD_MODE_SKIP; u.cbf[0] = u.cbf[1] = u.cbf[2] = 0; u.tu_depth = 0; u.depth = (uint8_t)(6 - log2);
            }
        memcpy(coder->ctx, cu.ctx, X265AMD_CTX_COUNT);
        coder->fracBits = cu.frac_bits & 32767;             /* load(cur); resetBits() */
        const x265amd_cu_unit* l = coder->at((x >> 2) - 1, y >> 2);
        const x265amd_cu_unit* a = coder->at(x >> 2, (y >> 2) - 1);
        const int skipCtx = (x265amd_cabac::coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (x265amd_cabac::coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
        coder->bin(1, C_SKIP + skipCtx);
        const uint32_t skipFlagBits = (uint32_t)(coder->fracBits >> 15);
        coder->mergeIndex(units[(y >> 2) * w4 + (x >> 2)]);
        x265amd_rd_result& r = out[i];
        memset(&r, 0, sizeof(r));
        r.mv_bits = (uint32_t)(coder->fracBits >> 15) - skipFlagBits;
        r.coeff_bits = 0;
        r.total_bits = r.mv_bits + skipFlagBits;
        const x265amd_cu_measure& m = meas[i];
        const sse_t luma = (sse_t)m.sse[0];
        sse_t chroma = (sse_t)m.sse[1];
        chroma += (sse_t)m.sse[2];
        const sse_t distortion = luma + chroma;
        uint64_t rd[6];
        x265amd_rdcost(lambda_qp(cu), si->slice_type, rp->psy_rd, 0, 0, 0, rd);
        r.luma_distortion = (uint32_t)luma; r.chroma_distortion = (uint32_t)chroma; r.distortion = distortion; r.res_energy = (uint32_t)luma;
        r.psy_energy = rd[2] ? m.psy : 0;
        x265amd_rdcost(lambda_qp(cu), si->slice_type, rp->psy_rd, distortion, r.total_bits, r.psy_energy, rd);
        r.rd_cost = rd[2] ? rd[4] : rd[3];
        memcpy(r.ctx, coder->ctx, X265AMD_CTX_COUNT);
        r.frac_bits = coder->fracBits;
        for (int yy = 0; yy < u4; yy++)
        {
            memcpy(&mine[yy * u4], &units[((y >> 2) + yy) * w4 + (x >> 2)], sizeof(x265amd_cu_unit) * u4);
            memcpy(&units[((y >> 2) + yy) * w4 + (x >> 2)], &saved[yy * u4], sizeof(x265amd_cu_unit) * u4);
        }
    }
    x265amd_cabac_close(coder);
    return X265AMD_OK;
}

static void fill_measure_jobs(CuMeasureJob* mjobs, const x265amd_rd_cu* cus, int n, const uint64_t* h_src, intptr_t stride, intptr_t cstride,
                              uint64_t d_pred, uint64_t d_recon, size_t tile_bytes, const char* scratch, size_t perCuBytes, const char* dSel)
{
    for (int i = 0; i < n; i++)
    {
        const x265amd_rd_cu& cu = cus[i];
        CuMeasureJob& m = mjobs[i];
        m.fenc[0] = h_src[0] + ((uint64_t)cu.y * stride + cu.x) * sizeof(pixel);
        m.fenc[1] = h_src[1] + ((uint64_t)(cu.y >> 1) * cstride + (cu.x >> 1)) * sizeof(pixel);
        m.fenc[2] = h_src[2] + ((uint64_t)(cu.y >> 1) * cstride + (cu.x >> 1)) * sizeof(pixel);
        m.pred = d_pred + (uint64_t)tile_bytes * i; m.recon = d_recon + (uint64_t)tile_bytes * i;
        m.resi = scratch ? (uint64_t)(uintptr_t)(scratch + perCuBytes * i) + (uint64_t)RD_SCRATCH_ELEMS * 2 : 0;
        m.sel = dSel ? (uint64_t)(uintptr_t)(dSel + (size_t)RD_SEL_BYTES * i) : 0;
        m.fenc_stride = (int32_t)stride; m.fenc_cstride = (int32_t)cstride; m.log2_size = cu.log2_size; m.assemble = 0;
    }
}

extern "C" int x265amd_measure_tiles(void* stream_, const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                                     uint64_t d_tiles, size_t tile_bytes, x265amd_cu_measure* out)
{
    if (!h_src || !cus || !d_tiles || !out || n < 0) return xa_fail(X265AMD_EINVAL, "measure_tiles: null argument");
    if (n == 0) return X265AMD_OK;
    XaMapped mJobs; XaMappedOut mMeas;  /* job and result records live in host memory the kernel reads / writes in place */
    XA_HIP_CHECK(mJobs.alloc(sizeof(CuMeasureJob) * n));
    XA_HIP_CHECK(mMeas.alloc(sizeof(x265amd_cu_measure) * n));
    fill_measure_jobs((CuMeasureJob*)mJobs.p, cus, n, h_src, stride, cstride, d_tiles, d_tiles, tile_bytes, nullptr, 0, nullptr);       /* "recon" = the tile itself */
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mJobs.p), (uint64_t)(uintptr_t)((x265amd_cu_measure*)mMeas.p), 0, 0, n }; hipError_t le;
      if (measure_use_wg(n)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure_wg, dim3(n), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mJobs.p, n, (x265amd_cu_measure*)mMeas.p);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure, dim3(n), dim3(64), 0, (const CuMeasureJob*)mJobs.p, n, (x265amd_cu_measure*)mMeas.p);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));
    memcpy(out, mMeas.p, sizeof(x265amd_cu_measure) * n);
    return X265AMD_OK;
}

extern "C" int x265amd_measure_tile_list(void* stream_, const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                                         const uint64_t* tile_addrs, x265amd_cu_measure* out)
{
    XA_HOSTPROF("rd.measure_tile_list");
    if (!h_src || !cus || !tile_addrs || !out || n < 0) return xa_fail(X265AMD_EINVAL, "measure_tile_list: null argument");
    if (n == 0) return X265AMD_OK;
    XaMapped mJobs; XaMappedOut mMeas;
    XA_HIP_CHECK(mJobs.alloc(sizeof(CuMeasureJob) * n));
    XA_HIP_CHECK(mMeas.alloc(sizeof(x265amd_cu_measure) * n));
    CuMeasureJob* jobs = (CuMeasureJob*)mJobs.p;
    for (int i = 0; i < n; i++) fill_measure_jobs(jobs + i, cus + i, 1, h_src, stride, cstride, tile_addrs[i], tile_addrs[i], 0, nullptr, 0, nullptr);
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(jobs), (uint64_t)(uintptr_t)((x265amd_cu_measure*)mMeas.p), 0, 0, n }; hipError_t le;
      if (measure_use_wg(n)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure_wg, dim3(n), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)jobs, n, (x265amd_cu_measure*)mMeas.p);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure, dim3(n), dim3(64), 0, (const CuMeasureJob*)jobs, n, (x265amd_cu_measure*)mMeas.p);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));
    memcpy(out, mMeas.p, sizeof(x265amd_cu_measure) * n);
    return X265AMD_OK;
}

extern "C" int x265amd_skip_rd(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                               const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                               x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes, x265amd_rd_result* out)
{
    if (!si || !rp || !units || !h_src || !cus || !cu_units || !d_pred || !d_recon || !out || n < 0) return xa_fail(X265AMD_EINVAL, "skip_rd: null argument");
    if (tile_bytes < (size_t)(4096 + 2048) * sizeof(pixel)) return xa_fail(X265AMD_EINVAL, "skip_rd: tile too small");
    if (n == 0) return X265AMD_OK;
    XaMapped mJobs; XaMappedOut mMeas;
    XA_HIP_CHECK(mJobs.alloc(sizeof(CuMeasureJob) * n));
    XA_HIP_CHECK(mMeas.alloc(sizeof(x265amd_cu_measure) * n));
    fill_measure_jobs((CuMeasureJob*)mJobs.p, cus, n, h_src, stride, cstride, d_pred, d_recon, tile_bytes, nullptr, 0, nullptr);
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mJobs.p), (uint64_t)(uintptr_t)((x265amd_cu_measure*)mMeas.p), 0, 0, n }; hipError_t le;
      if (measure_use_wg(n)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure_wg, dim3(n), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mJobs.p, n, (x265amd_cu_measure*)mMeas.p);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure, dim3(n), dim3(64), 0, (const CuMeasureJob*)mJobs.p, n, (x265amd_cu_measure*)mMeas.p);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));
    std::vector<x265amd_cu_measure> meas((const x265amd_cu_measure*)mMeas.p, (const x265amd_cu_measure*)mMeas.p + n);
    return x265amd_skip_rd_host(si, rp, units, cus, n, cu_units, meas.data(), out);
}

static bool compose_final_measure(const x265amd_slice_info* si, const x265amd_rd_cu& cu, int part, const x265amd_tu_result* res, const uint8_t* sel, x265amd_cu_measure& m);
static int assemble_async(void* stream_, const CuMeasureJob& job, const uint8_t* sel);
static int inter_residual_rd_impl(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                                  const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                                  x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes,
                                  x265amd_rd_result* out, int16_t* coeff_out, bool lazyAssemble);
extern "C" int x265amd_inter_residual_rd(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                                         const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                                         x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes,
                                         x265amd_rd_result* out, int16_t* coeff_out)
{
    return inter_residual_rd_impl(stream_, si, rp, units, h_src, stride, cstride, cus, n, cu_units, d_pred, d_recon, tile_bytes, out, coeff_out, false);
}
/* the same for one CU on a device job queue, returning with the reconstruction tile still being assembled (the queue's later commands are ordered behind it) when
 * the results do not need the assembled samples (compose_final_measure) */
int xa_inter_residual_rd_lazy(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                              const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cu,
                              x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes, x265amd_rd_result* out, int16_t* coeff_out)
{
    return inter_residual_rd_impl(stream_, si, rp, units, h_src, stride, cstride, cu, 1, cu_units, d_pred, d_recon, tile_bytes, out, coeff_out, true);
}
static int inter_residual_rd_impl(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units,
                                  const uint64_t* h_src, intptr_t stride, intptr_t cstride, const x265amd_rd_cu* cus, int n,
                                  x265amd_cu_unit* cu_units, uint64_t d_pred, uint64_t d_recon, size_t tile_bytes,
                                  x265amd_rd_result* out, int16_t* coeff_out, bool lazyAssemble)
{
    XA_HOSTPROF("rd.inter_residual_rd_impl (all)");
    if (!si || !rp || !units || !h_src || !cus || !cu_units || !d_pred || !d_recon || !out || n < 0) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: null argument");
    if (tile_bytes < (size_t)(4096 + 2048) * sizeof(pixel)) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: tile too small");
    if (si->tq_bypass_enabled) return xa_fail(X265AMD_EINVAL, "inter_residual_rd: lossless coding is not supported");
    if (n == 0) return X265AMD_OK;

    /* ---- plan + launch 1: all transform chains and the no-residual measurement ---- */
    const size_t perCuBytes = x265amd_inter_rd_scratch_bytes();
    DevBuf dScratch;
    XaMapped mJobs, mMJobs, dSel; XaMappedOut mRes, mMeas, mLevels;       /* records the kernels touch once: host memory, read / written in place (dSel: host writes, the assembly reads) */
    XA_HIP_CHECK(dScratch.alloc(perCuBytes * n));
    char* scratch = (char*)dScratch.p;
    const int nJobs = x265amd_inter_rd_plan(si, cus, n, cu_units, h_src, stride, cstride, d_pred, tile_bytes, (uint64_t)(uintptr_t)scratch, nullptr, 0);
    if (nJobs < 0) return nJobs;
    XA_HIP_CHECK(mJobs.alloc(sizeof(x265amd_tu_job) * nJobs));
    XA_HIP_CHECK(mRes.alloc(sizeof(x265amd_tu_result) * nJobs));
    XA_HIP_CHECK(mMJobs.alloc(sizeof(CuMeasureJob) * n));
    XA_HIP_CHECK(mMeas.alloc(sizeof(x265amd_cu_measure) * n * 2));
    XA_HIP_CHECK(mLevels.alloc((size_t)RD_SCRATCH_ELEMS * 2 * n));
    XA_HIP_CHECK(dSel.alloc((size_t)RD_SEL_BYTES * n));
    const bool rdoq = rp->rdoq_level != 0;
    inter_rd_plan_levels(si, cus, n, cu_units, h_src, stride, cstride, d_pred, tile_bytes, (uint64_t)(uintptr_t)scratch, (x265amd_tu_job*)mJobs.p, nJobs,
                         rdoq ? 0 : (uint64_t)(uintptr_t)mLevels.p);        /* without RDOQ the chains write the levels where the host reads them */
    int rc = X265AMD_OK;
    if (!rdoq)
    {
        rc = x265amd_tu_chain(stream_, (const x265amd_tu_job*)mJobs.p, nJobs, (x265amd_tu_result*)mRes.p);
        if (rc != X265AMD_OK) return rc;
    }
    CuMeasureJob* mjobs = (CuMeasureJob*)mMJobs.p;
    x265amd_cu_measure* meas = (x265amd_cu_measure*)mMeas.p;
    fill_measure_jobs(mjobs, cus, n, h_src, stride, cstride, d_pred, d_recon, tile_bytes, scratch, perCuBytes, (const char*)dSel.p);
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mjobs), (uint64_t)(uintptr_t)(meas), 0, 0, n }; hipError_t le;
      if (measure_use_wg(n)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure_wg, dim3(n), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mjobs, n, meas);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure, dim3(n), dim3(64), 0, (const CuMeasureJob*)mjobs, n, meas);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));

    /* ---- the walk ---- */
    std::vector<uint8_t> sel((size_t)RD_SEL_BYTES * n);
    XaMapped mCtx, mEstJob, mRdoq, mNodeCtx, mNodeEst, mNodeRq, mNodeJobs;
    XaMappedOut mNodeRes;
    DevBuf dEst, dEst2;
    RdoqDemand demandFn;
    if (rdoq)
    {
        /* Quant::rdoQuant reads Entropy::m_estBitsSbac, refreshed by Entropy::estBit from the live contexts before each transform: k_est_bit on the
         * walk's contexts, then the unit's chain, in stream order; results and levels are read back before the walk prices the unit */
        XA_HIP_CHECK(mCtx.alloc(X265AMD_CTX_STRIDE)); XA_HIP_CHECK(mEstJob.alloc(sizeof(x265amd_est_job))); XA_HIP_CHECK(mRdoq.alloc(sizeof(x265amd_tu_rdoq)));
        XA_HIP_CHECK(dEst.alloc(sizeof(x265amd_est_bits)));
        XA_HIP_CHECK(xa_fill_async(stream_, dEst.p, 0, sizeof(x265amd_est_bits)));
        demandFn.unit = [&](int i, const uint8_t* ctx, int first, int count, bool luma, int log2TrSize, int tuDepth) -> int {
            const x265amd_tu_job* jobs = (const x265amd_tu_job*)mJobs.p;
            for (int k = first; k < first + count; k++)
            {
                if (log2TrSize > 0)
                {
                    memcpy(mCtx.p, ctx, X265AMD_CTX_COUNT);
                    x265amd_est_job* ej = (x265amd_est_job*)mEstJob.p;
                    memset(ej, 0, sizeof(*ej));
                    ej->ctx = (uint64_t)(uintptr_t)mCtx.p; ej->est = (uint64_t)(uintptr_t)dEst.p; ej->log2_tr_size = (uint8_t)log2TrSize; ej->is_luma = luma ? 1 : 0;
                    const int r = x265amd_est_bit(stream_, ej, 1);
                    if (r != X265AMD_OK) return r;
                }
                x265amd_tu_rdoq* rq = (x265amd_tu_rdoq*)mRdoq.p;
                memset(rq, 0, sizeof(*rq));
                rq->est_bits = (uint64_t)(uintptr_t)dEst.p;
                x265amd_rdoq_lambda(jobs[k].qp_scaled, &rq->lambda2, &rq->lambda);
                rq->psy_rdoq_scale = rp->psy_rdoq_scale; rq->rdoq_level = (uint8_t)rp->rdoq_level; rq->tu_depth = (uint8_t)tuDepth;
                const int r = x265amd_tu_chain_rdoq(stream_, jobs + k, rq, 1, (x265amd_tu_result*)mRes.p + k);
                if (r != X265AMD_OK) return r;
                const size_t nCoeff = (size_t)1 << (2 * jobs[k].log2_tr_size);
                const size_t off = (size_t)(jobs[k].coeff - (uint64_t)(uintptr_t)scratch);          /* inside CU i's scratch: i * perCuBytes + level offset */
                char* dst = (char*)mLevels.p + (size_t)RD_SCRATCH_ELEMS * 2 * i + (off - perCuBytes * i);
                if (xa_copy_async(stream_, dst, (const void*)(uintptr_t)jobs[k].coeff, nCoeff * 2, hipMemcpyDeviceToHost) != hipSuccess || xa_stream_sync(stream_) != hipSuccess)
                    return xa_fail(X265AMD_EHIP, "inter_residual_rd: RDOQ unit");
            }
            return X265AMD_OK;
        };
        /* a node's three units as two commands and one wait: both tables (luma, chroma) by two wavefronts of one est_bit command, the three chains by three wavefronts of
         * one tu_chain_rdoq command -- the job records are copied side by side for that (they lie apart in the plan), the results go back to their places */
        static const bool together = !(getenv("X265AMD_RDOQ_TOGETHER") && atoi(getenv("X265AMD_RDOQ_TOGETHER")) == 0);
        if (together)
        {
            XA_HIP_CHECK(mNodeCtx.alloc(X265AMD_CTX_STRIDE)); XA_HIP_CHECK(mNodeEst.alloc(2 * sizeof(x265amd_est_job))); XA_HIP_CHECK(mNodeRq.alloc(3 * sizeof(x265amd_tu_rdoq)));
            XA_HIP_CHECK(mNodeJobs.alloc(3 * sizeof(x265amd_tu_job))); XA_HIP_CHECK(mNodeRes.alloc(3 * sizeof(x265amd_tu_result)));
            XA_HIP_CHECK(dEst2.alloc(2 * sizeof(x265amd_est_bits)));
            XA_HIP_CHECK(xa_fill_async(stream_, dEst2.p, 0, 2 * sizeof(x265amd_est_bits)));
            demandFn.node = [&](int i, const uint8_t* ctx, int jy, int ju, int jv, int log2TrSize, int log2TrSizeC, int tuDepth) -> int {
                const x265amd_tu_job* jobs = (const x265amd_tu_job*)mJobs.p;
                const int idx[3] = { jy, ju, jv };
                memcpy(mNodeCtx.p, ctx, X265AMD_CTX_COUNT);
                x265amd_est_job* ej = (x265amd_est_job*)mNodeEst.p;
                memset(ej, 0, 2 * sizeof(*ej));
                char* tables = (char*)dEst2.p;
                for (int t = 0; t < 2; t++)
                {
                    ej[t].ctx = (uint64_t)(uintptr_t)mNodeCtx.p; ej[t].est = (uint64_t)(uintptr_t)(tables + (size_t)t * sizeof(x265amd_est_bits));
                    ej[t].log2_tr_size = (uint8_t)(t ? log2TrSizeC : log2TrSize); ej[t].is_luma = t ? 0 : 1;
                }
                int r = x265amd_est_bit(stream_, ej, 2);
                if (r != X265AMD_OK) return r;
                x265amd_tu_job* nj = (x265amd_tu_job*)mNodeJobs.p;
                x265amd_tu_rdoq* rq = (x265amd_tu_rdoq*)mNodeRq.p;
                memset(rq, 0, 3 * sizeof(*rq));
                for (int k = 0; k < 3; k++)
                {
                    nj[k] = jobs[idx[k]];
                    rq[k].est_bits = (uint64_t)(uintptr_t)(tables + (size_t)(k ? 1 : 0) * sizeof(x265amd_est_bits));
                    x265amd_rdoq_lambda(nj[k].qp_scaled, &rq[k].lambda2, &rq[k].lambda);
                    rq[k].psy_rdoq_scale = rp->psy_rdoq_scale; rq[k].rdoq_level = (uint8_t)rp->rdoq_level; rq[k].tu_depth = (uint8_t)tuDepth;
                }
                r = x265amd_tu_chain_rdoq(stream_, nj, rq, 3, (x265amd_tu_result*)mNodeRes.p);
                if (r != X265AMD_OK) return r;
                for (int k = 0; k < 3; k++)
                {
                    const size_t nCoeff = (size_t)1 << (2 * nj[k].log2_tr_size);
                    const size_t off = (size_t)(nj[k].coeff - (uint64_t)(uintptr_t)scratch);
                    char* dst = (char*)mLevels.p + (size_t)RD_SCRATCH_ELEMS * 2 * i + (off - perCuBytes * i);
                    if (xa_copy_async(stream_, dst, (const void*)(uintptr_t)nj[k].coeff, nCoeff * 2, hipMemcpyDeviceToHost) != hipSuccess) return xa_fail(X265AMD_EHIP, "inter_residual_rd: RDOQ node");
                }
                if (xa_stream_sync(stream_) != hipSuccess) return xa_fail(X265AMD_EHIP, "inter_residual_rd: RDOQ node");
                for (int k = 0; k < 3; k++) ((x265amd_tu_result*)mRes.p)[idx[k]] = ((const x265amd_tu_result*)mNodeRes.p)[k];
                return X265AMD_OK;
            };
        }
    }
    rc = inter_rd_walk_impl(si, rp, units, cus, n, cu_units, (const x265amd_tu_result*)mRes.p, (const int16_t*)mLevels.p, (size_t)RD_SCRATCH_ELEMS * 2, meas, sel.data(), out,
                            coeff_out, rdoq ? &demandFn : nullptr);
    if (rc != X265AMD_OK) return rc;

    /* ---- launch 2: assemble, reconstruct, measure ---- */
    for (int i = 0; i < n; i++) mjobs[i].assemble = 1;
    if (lazyAssemble && n == 1 && !rdoq && xa_is_queue(stream_))
    {
        static const bool composeOn = !(getenv("X265AMD_COMPOSE_MEASURE") && atoi(getenv("X265AMD_COMPOSE_MEASURE")) == 0);
        x265amd_cu_measure mc;
        if (composeOn && compose_final_measure(si, cus[0], cu_units[0].part_size, (const x265amd_tu_result*)mRes.p, sel.data(), mc))
        {
            rc = assemble_async(stream_, mjobs[0], sel.data());
            if (rc != X265AMD_OK) return rc;
            x265amd_inter_rd_finish(si, rp, cus, 1, &mc, out);
            return X265AMD_OK;
        }
    }
    memcpy(dSel.p, sel.data(), sel.size());
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mjobs), (uint64_t)(uintptr_t)(meas + n), 0, 0, n }; hipError_t le;
      if (measure_use_wg(n)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure_wg, dim3(n), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mjobs, n, meas + n);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, n, qa, k_cu_measure, dim3(n), dim3(64), 0, (const CuMeasureJob*)mjobs, n, meas + n);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));
    x265amd_inter_rd_finish(si, rp, cus, n, meas + n, out);
    return X265AMD_OK;
}

/* What the second measurement would report, put together from the transform units' own results: with ONE transform size per plane (no split to choose from:
 * tu-inter-depth 1, 2Nx2N) the CU's reconstruction is, unit by unit, either the unit's reconstruction (kept) or its prediction (dropped), the squared error adds
 * up over the units and so does the psy energy (psyCost_pp sums over 8x8 blocks, pixel.cpp:744-775; luma units are 8x8 or larger here).  false: not that case. */
static bool compose_final_measure(const x265amd_slice_info* si, const x265amd_rd_cu& cu, int part, const x265amd_tu_result* res, const uint8_t* sel, x265amd_cu_measure& m)
{
    CuPlan P;
    if (make_plan(si, cu, part, P, 0, nullptr, 0, 0, 0, 0, nullptr) < 0) return false;
    int lumaLayers = 0, L0 = -1;
    for (int L = 2; L <= 5; L++) if (P.lumaRes[L] >= 0) { lumaLayers++; L0 = L; }
    if (lumaLayers != 1 || L0 < 3) return false;
    int chromaLayers = 0, C0 = -1;
    for (int C = 2; C <= 4; C++) if (P.chromaRes[C][1] >= 0) { chromaLayers++; C0 = C; }
    if (chromaLayers != 1) return false;
    memset(&m, 0, sizeof(m));
    const int nt = P.size >> L0, n4 = 1 << (L0 - 2);
    for (int ty = 0; ty < nt; ty++)
        for (int tx = 0; tx < nt; tx++)
        {
            const x265amd_tu_result& r = res[P.lumaRes[L0] + ty * nt + tx];
            const bool kept = sel[(ty * n4) * 16 + tx * n4] == (uint8_t)L0;
            m.sse[0] += kept ? r.nz_dist : r.zero_dist;
            m.psy += kept ? r.nz_energy : r.zero_energy;
        }
    const int ntc = (P.size >> 1) >> C0, c4 = 1 << (C0 - 2);
    for (int p = 1; p < 3; p++)
        for (int ty = 0; ty < ntc; ty++)
            for (int tx = 0; tx < ntc; tx++)
            {
                const x265amd_tu_result& r = res[P.chromaRes[C0][p] + ty * ntc + tx];
                const bool kept = sel[256 + (p - 1) * 64 + (ty * c4) * 8 + tx * c4] == (uint8_t)C0;
                m.sse[p] += kept ? r.nz_dist : r.zero_dist;
            }
    return true;
}

/* the assembly of the kept residual blocks into the reconstruction tile without anybody waiting for it (its measurements are not needed: composed above): job record
 * and selection go through device scratch that the queue's later commands may reuse in order */
static int assemble_async(void* stream_, const CuMeasureJob& job, const uint8_t* sel)
{
    DevBuf dJob, dSel2, dOut;
    XA_HIP_CHECK(dJob.alloc(sizeof(CuMeasureJob)));
    XA_HIP_CHECK(dSel2.alloc(RD_SEL_BYTES));
    XA_HIP_CHECK(dOut.alloc(sizeof(x265amd_cu_measure)));
    CuMeasureJob j = job;
    j.assemble = 1; j.sel = (uint64_t)(uintptr_t)dSel2.p;
    XA_HIP_CHECK(xa_copy_async(stream_, dSel2.p, sel, RD_SEL_BYTES, hipMemcpyHostToDevice));
    XA_HIP_CHECK(xa_copy_async(stream_, dJob.p, &j, sizeof(j), hipMemcpyHostToDevice));
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)dJob.p, (uint64_t)(uintptr_t)dOut.p, 0, 0, 1 }; hipError_t le;
      if (measure_use_wg(1)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure_wg, dim3(1), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)dJob.p, 1, (x265amd_cu_measure*)dOut.p);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure, dim3(1), dim3(64), 0, (const CuMeasureJob*)dJob.p, 1, (x265amd_cu_measure*)dOut.p);
      XA_HIP_CHECK(le); }
    return X265AMD_OK;
}

/* checkMerge2Nx2N_rd0_4's two RD measurements of the chosen candidate as ONE device round trip where the reference's result allows it (analysis.cpp:2852-2880):
 * encodeResAndCalcRdSkipCU and the first half of encodeResAndCalcRdInterCU share the prediction-against-source measurement, so the transform chains and that
 * one measurement go out together (the measurement leaves the prediction in the skip mode's reconstruction tile).  When no transform unit keeps a level the
 * residual mode is the skip mode -- same bits, distortion, energy and cost, and `tempPred->rdCost < bestPred->rdCost` is false -- so the tree walk, the
 * assembly and the second measurement are left out; *merge_is_skip says so (the merge mode's units then equal the skip mode's, its reconstruction tile is not
 * written).  Otherwise the walk and the assembly run as in x265amd_inter_residual_rd.  Without RDOQ only (returns X265AMD_EINVAL with it). */
int xa_merge_rd(void* stream_, const x265amd_slice_info* si, const x265amd_rd_params* rp, x265amd_cu_unit* units, const uint64_t* h_src, intptr_t stride, intptr_t cstride,
                const x265amd_rd_cu* cu, x265amd_cu_unit* skip_units, x265amd_cu_unit* merge_units, uint64_t d_pred, uint64_t d_recon_skip, uint64_t d_recon_merge,
                x265amd_rd_result* out_skip, x265amd_rd_result* out_merge, int16_t* coeff_out, int* merge_is_skip)
{
    XA_HOSTPROF("rd.xa_merge_rd (all)");
    if (!si || !rp || !units || !h_src || !cu || !skip_units || !merge_units || !d_pred || !d_recon_skip || !d_recon_merge || !out_skip || !out_merge || !merge_is_skip)
        return xa_fail(X265AMD_EINVAL, "merge_rd: null argument");
    if (rp->rdoq_level || si->tq_bypass_enabled) return xa_fail(X265AMD_EINVAL, "merge_rd: RDOQ / lossless go through the separate entry points");
    const size_t tile_bytes = (size_t)(4096 + 2048) * sizeof(pixel);
    const size_t perCuBytes = x265amd_inter_rd_scratch_bytes();
    DevBuf dScratch;
    XaMapped mJobs, mMJobs, dSel; XaMappedOut mRes, mMeas, mLevels;
    XA_HIP_CHECK(dScratch.alloc(perCuBytes));
    char* scratch = (char*)dScratch.p;
    const int nJobs = x265amd_inter_rd_plan(si, cu, 1, merge_units, h_src, stride, cstride, d_pred, tile_bytes, (uint64_t)(uintptr_t)scratch, nullptr, 0);
    if (nJobs < 0) return nJobs;
    XA_HIP_CHECK(mJobs.alloc(sizeof(x265amd_tu_job) * nJobs));
    XA_HIP_CHECK(mRes.alloc(sizeof(x265amd_tu_result) * nJobs));
    XA_HIP_CHECK(mMJobs.alloc(sizeof(CuMeasureJob)));
    XA_HIP_CHECK(mMeas.alloc(sizeof(x265amd_cu_measure) * 2));
    XA_HIP_CHECK(mLevels.alloc((size_t)RD_SCRATCH_ELEMS * 2));
    XA_HIP_CHECK(dSel.alloc((size_t)RD_SEL_BYTES));
    inter_rd_plan_levels(si, cu, 1, merge_units, h_src, stride, cstride, d_pred, tile_bytes, (uint64_t)(uintptr_t)scratch, (x265amd_tu_job*)mJobs.p, nJobs, (uint64_t)(uintptr_t)mLevels.p);
    int rc = x265amd_tu_chain(stream_, (const x265amd_tu_job*)mJobs.p, nJobs, (x265amd_tu_result*)mRes.p);
    if (rc != X265AMD_OK) return rc;
    CuMeasureJob* mjobs = (CuMeasureJob*)mMJobs.p;
    x265amd_cu_measure* meas = (x265amd_cu_measure*)mMeas.p;
    fill_measure_jobs(mjobs, cu, 1, h_src, stride, cstride, d_pred, d_recon_skip, tile_bytes, scratch, perCuBytes, (const char*)dSel.p);
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mjobs), (uint64_t)(uintptr_t)(meas), 0, 0, 1 }; hipError_t le;
      if (measure_use_wg(1)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure_wg, dim3(1), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mjobs, 1, meas);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure, dim3(1), dim3(64), 0, (const CuMeasureJob*)mjobs, 1, meas);
      XA_HIP_CHECK(le); }
    xa_phase(XA_PH_RD_PLAN);
    XA_HIP_CHECK(xa_stream_sync(stream_));
    xa_phase(XA_PH_OTHER);

    const x265amd_cu_measure m0 = meas[0];
    rc = x265amd_skip_rd_host(si, rp, units, cu, 1, skip_units, &m0, out_skip);
    if (rc != X265AMD_OK) return rc;
    xa_phase(XA_PH_RD_SKIPHOST);
    bool anyLevel = si->use_dqp != 0;            /* checkDQP touches the units of a residual-free CU: the walk does that */
    const x265amd_tu_result* res = (const x265amd_tu_result*)mRes.p;
    for (int k = 0; k < nJobs; k++) anyLevel |= res[k].num_sig != 0;
    if (!anyLevel)
    {
        *merge_is_skip = 1;
        *out_merge = *out_skip;
        memcpy(merge_units, skip_units, sizeof(x265amd_cu_unit) * 256);
        if (coeff_out) memset(coeff_out, 0, sizeof(int16_t) * (4096 + 2048));
        return X265AMD_OK;
    }
    *merge_is_skip = 0;
    std::vector<uint8_t> sel((size_t)RD_SEL_BYTES);
    rc = inter_rd_walk_impl(si, rp, units, cu, 1, merge_units, res, (const int16_t*)mLevels.p, (size_t)RD_SCRATCH_ELEMS * 2, &m0, sel.data(), out_merge, coeff_out, nullptr);
    if (rc != X265AMD_OK) return rc;
    xa_phase(XA_PH_RD_WALK);
    mjobs[0].assemble = 1; mjobs[0].recon = d_recon_merge;
    {
        /* one transform size per plane: the final measurement follows from the units' results, the assembly runs without being waited for */
        static const bool composeOn = !(getenv("X265AMD_COMPOSE_MEASURE") && atoi(getenv("X265AMD_COMPOSE_MEASURE")) == 0);
        x265amd_cu_measure mc;
        if (composeOn && xa_is_queue(stream_) && compose_final_measure(si, *cu, merge_units[0].part_size, res, sel.data(), mc))
        {
            /* the scratch with the residual blocks must outlive this call: it goes back to the queue's own list and is reused behind the assembly, in order */
            rc = assemble_async(stream_, mjobs[0], sel.data());
            if (rc != X265AMD_OK) return rc;
            x265amd_inter_rd_finish(si, rp, cu, 1, &mc, out_merge);
            return X265AMD_OK;
        }
    }
    memcpy(dSel.p, sel.data(), sel.size());
    { const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)(mjobs), (uint64_t)(uintptr_t)(meas + 1), 0, 0, 1 }; hipError_t le;
      if (measure_use_wg(1)) XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure_wg, dim3(1), dim3(64 * MEASURE_WG_WAVES), 0, (const CuMeasureJob*)mjobs, 1, meas + 1);
      else XA_LAUNCH(le, stream_, XA_OP_CU_MEASURE, 1, qa, k_cu_measure, dim3(1), dim3(64), 0, (const CuMeasureJob*)mjobs, 1, meas + 1);
      XA_HIP_CHECK(le); }
    XA_HIP_CHECK(xa_stream_sync(stream_));
    x265amd_inter_rd_finish(si, rp, cu, 1, meas + 1, out_merge);
    return X265AMD_OK;
}
