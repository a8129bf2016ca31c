/* The host arithmetic of the reference's rate control as `--preset medium` comes: cuTree, the constant-rate-factor branch of RateControl, the QP of a CU
 * (include/x265amd_ratecontrol.h).  Restated from the reference's functions named below; double precision in their order and operand types.
 * COMPILED WITH THE REFERENCE'S FLOATING-POINT FLAGS (build.sh: files named fm_*.cpp get -O2 -ffast-math, as source/CMakeLists.txt:226-240 gives every file of
 * the reference): what is rounded to integers downstream -- propagate amounts, QP offsets, QPs -- has to come out bit for bit, and tests/test_ratecontrol.py
 * checks that it does against the reference's own Lookahead / RateControl objects. */
#include "x265amd.h"
#include "x265amd_ratecontrol.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace {

enum { TYPE_B = 5 };
enum { LOWRES_COST_MASK = (1 << 14) - 1, LOWRES_COST_SHIFT = 14 };      /* common/lowres.h */

inline double clipDuration(double f) { return f < 0.01 ? 0.01 : (f > 1.00 ? 1.00 : f); }       /* CLIP_DURATION (ratecontrol.h:40-48) */
const double kBaseFrameDuration = 0.04;

/* x265_exp2fix8 (source/common/common.cpp:96-103); the table is 256 * (2^(i/64) - 1) rounded (source/common/constants.cpp:552-558) */
int exp2fix8(double x)
{
    static int lut[64];
    static bool init = false;
    if (!init) { for (int i = 0; i < 64; i++) lut[i] = (int)floor(256.0 * (pow(2.0, i / 64.0) - 1.0) + 0.5); init = true; }
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

/* estimateCUPropagateCost (source/common/pixel.cpp:931-959) */
void propagateCostRow(int* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts, const int32_t* invQscales, const double* fpsFactor, int len)
{
    double fps = *fpsFactor / 256;
    for (int i = 0; i < len; i++)
    {
        int intraCost = intraCosts[i];
        int interCost = std::min(intraCosts[i], interCosts[i] & LOWRES_COST_MASK);
        double propagateIntra = intraCost * invQscales[i];
        double propagateAmount = (double)propagateIn[i] + propagateIntra * fps;
        double propagateNum = (double)(intraCost - interCost);
        double propagateDenom = (double)intraCost;
        dst[i] = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
    }
}

/* Lookahead::estimateCUPropagate (slicetype.cpp:3502-3608), qgSize 16 and above, no VBV */
void estimateCUPropagate(const x265amd_cutree_params& P, x265amd_cutree_frame* const* frames, std::vector<int>& scratch, double averageDuration, int p0, int p1, int b, int referenced,
                         const uint16_t* lowresCosts, const int16_t* mvsList[2])
{
    const int w8 = P.width8, h8 = P.height8;
    uint16_t* refCosts[2] = { frames[p0]->propagate_cost, frames[p1]->propagate_cost };
    int32_t distScaleFactor = (((b - p0) << 8) + ((p1 - p0) >> 1)) / (p1 - p0);
    int32_t bipredWeight = P.weighted_bipred ? 64 - (distScaleFactor >> 2) : 32;
    int32_t bipredWeights[2] = { bipredWeight, 64 - bipredWeight };
    memset(scratch.data(), 0, w8 * sizeof(int));
    uint16_t* propagateCost = frames[b]->propagate_cost;
    double fpsFactor = clipDuration((double)P.fps_denom / P.fps_num) / clipDuration(averageDuration);
    if (!referenced) memset(frames[b]->propagate_cost, 0, w8 * sizeof(uint16_t));
    int32_t strideInCU = w8;
    for (uint16_t blocky = 0; blocky < h8; blocky++)
    {
        int cuIndex = blocky * strideInCU;
        propagateCostRow(scratch.data(), propagateCost, frames[b]->intra_cost + cuIndex, lowresCosts + cuIndex, frames[b]->inv_qscale + cuIndex, &fpsFactor, w8);
        if (referenced) propagateCost += w8;
        for (uint16_t blockx = 0; blockx < w8; blockx++, cuIndex++)
        {
            int32_t propagate_amount = scratch[blockx];
            if (propagate_amount > 0)
            {
                int32_t lists_used = lowresCosts[cuIndex] >> LOWRES_COST_SHIFT;
                for (uint16_t list = 0; list < 2; list++)
                {
                    if ((lists_used >> list) & 1)
                    {
#define CLIP_ADD(s, x) (s) = (uint16_t)std::min((s) + (x), (1 << 16) - 1)
                        int32_t listamount = propagate_amount;
                        if (lists_used == 3) listamount = (listamount * bipredWeights[list] + 32) >> 6;
                        const int16_t* mvs = mvsList[list];
                        if (!mvs[2 * cuIndex] && !mvs[2 * cuIndex + 1])
                        {
                            CLIP_ADD(refCosts[list][cuIndex], listamount);
                            continue;
                        }
                        int32_t x = mvs[2 * cuIndex];
                        int32_t y = mvs[2 * cuIndex + 1];
                        int32_t cux = (x >> 5) + blockx;
                        int32_t cuy = (y >> 5) + blocky;
                        int32_t idx0 = cux + cuy * strideInCU;
                        int32_t idx1 = idx0 + 1;
                        int32_t idx2 = idx0 + strideInCU;
                        int32_t idx3 = idx0 + strideInCU + 1;
                        x &= 31;
                        y &= 31;
                        int32_t idx0weight = (32 - y) * (32 - x);
                        int32_t idx1weight = (32 - y) * x;
                        int32_t idx2weight = y * (32 - x);
                        int32_t idx3weight = y * x;
                        if (cux < w8 - 1 && cuy < h8 - 1 && cux >= 0 && cuy >= 0)
                        {
                            CLIP_ADD(refCosts[list][idx0], (listamount * idx0weight + 512) >> 10);
                            CLIP_ADD(refCosts[list][idx1], (listamount * idx1weight + 512) >> 10);
                            CLIP_ADD(refCosts[list][idx2], (listamount * idx2weight + 512) >> 10);
                            CLIP_ADD(refCosts[list][idx3], (listamount * idx3weight + 512) >> 10);
                        }
                        else
                        {
                            if (cux < w8 && cuy < h8 && cux >= 0 && cuy >= 0) CLIP_ADD(refCosts[list][idx0], (listamount * idx0weight + 512) >> 10);
                            if (cux + 1 < w8 && cuy < h8 && cux + 1 >= 0 && cuy >= 0) CLIP_ADD(refCosts[list][idx1], (listamount * idx1weight + 512) >> 10);
                            if (cux < w8 && cuy + 1 < h8 && cux >= 0 && cuy + 1 >= 0) CLIP_ADD(refCosts[list][idx2], (listamount * idx2weight + 512) >> 10);
                            if (cux + 1 < w8 && cuy + 1 < h8 && cux + 1 >= 0 && cuy + 1 >= 0) CLIP_ADD(refCosts[list][idx3], (listamount * idx3weight + 512) >> 10);
                        }
#undef CLIP_ADD
                    }
                }
            }
        }
    }
}

/* Lookahead::cuTreeFinish (slicetype.cpp:3750-3800), qgSize 16 and above, no hevc-aq */
void cuTreeFinish(const x265amd_cutree_params& P, x265amd_cutree_frame* frame, double averageDuration, int ref0Distance)
{
    int fpsFactor = (int)(clipDuration(averageDuration) / clipDuration((double)P.fps_denom / P.fps_num) * 256);
    double weightdelta = 0.0;
    if (ref0Distance && frame->weighted_cost_delta && frame->weighted_cost_delta[ref0Distance - 1] > 0)
        weightdelta = (1.0 - frame->weighted_cost_delta[ref0Distance - 1]);
    const int cuCount = P.width8 * P.height8;
    for (int cuIndex = 0; cuIndex < cuCount; cuIndex++)
    {
        int intracost = (frame->intra_cost[cuIndex] * frame->inv_qscale[cuIndex] + 128) >> 8;
        if (intracost)
        {
            int propagateCost = (frame->propagate_cost[cuIndex] * fpsFactor + 128) >> 8;
            double log2_ratio = log2(intracost + propagateCost) - log2(intracost) + weightdelta;
            frame->qp_cutree_offset[cuIndex] = frame->qp_aq_offset[cuIndex] - P.strength * log2_ratio;
        }
    }
}

}

/* Lookahead::cuTree (slicetype.cpp:3399-3500) */
extern "C" int x265amd_cutree(const x265amd_cutree_params* p, x265amd_cutree_frame* const* frames, int numframes, int b_intra, x265amd_cutree_estimate_fn estimate, void* ctx)
{
    if (!p || !frames || !estimate || numframes < 0 || p->width8 < 1 || p->height8 < 1 || !p->fps_num || !p->fps_denom || p->lookahead_depth <= 0) return X265AMD_EINVAL;
    const x265amd_cutree_params& P = *p;
    const size_t cuCount = (size_t)P.width8 * P.height8;
    int idx = !b_intra;
    int lastnonb, curnonb = 1;
    int bframes = 0;
    double totalDuration = 0.0;
    for (int j = 0; j <= numframes; j++) totalDuration += (double)P.fps_denom / P.fps_num;
    double averageDuration = totalDuration / (numframes + 1);
    int i = numframes;
    while (i > 0 && frames[i]->slice_type == TYPE_B) i--;
    lastnonb = i;
    if (lastnonb < idx) return X265AMD_OK;
    memset(frames[lastnonb]->propagate_cost, 0, cuCount * sizeof(uint16_t));
    std::vector<int> scratch((size_t)P.width8);
    int rc = 0;
    auto single = [&](int p0, int p1, int b, const uint16_t*& lc, const int16_t* mv[2]) -> int {
        lc = nullptr; mv[0] = mv[1] = nullptr;
        return estimate(ctx, p0, p1, b, &lc, &mv[0], &mv[1]);
    };
    auto propagate = [&](int p0, int p1, int b, int referenced) -> int {
        /* (the reference's estimateCUPropagate reads lowresCosts[b - p0][p1 - b] and the fields of those two distances: what the estimate just before it left) */
        const uint16_t* lc; const int16_t* mv[2];
        const int r = single(p0, p1, b, lc, mv);
        if (r) return r;
        if (!lc || (b > p0 && !mv[0]) || (p1 > b && !mv[1])) return X265AMD_EINVAL;
        static const int16_t none[2] = { 0, 0 };
        (void)none;
        estimateCUPropagate(P, frames, scratch, averageDuration, p0, p1, b, referenced, lc, mv);
        return 0;
    };
    while (i-- > idx)
    {
        curnonb = i;
        while (frames[curnonb]->slice_type == TYPE_B && curnonb > 0) curnonb--;
        if (curnonb < idx) break;
        {
            const uint16_t* lc; const int16_t* mv[2];
            if ((rc = single(curnonb, lastnonb, lastnonb, lc, mv)) != 0) return rc;
        }
        memset(frames[curnonb]->propagate_cost, 0, cuCount * sizeof(uint16_t));
        bframes = lastnonb - curnonb - 1;
        if (P.b_pyramid && bframes > 1)
        {
            int middle = (bframes + 1) / 2 + curnonb;
            {
                const uint16_t* lc; const int16_t* mv[2];
                if ((rc = single(curnonb, lastnonb, middle, lc, mv)) != 0) return rc;
            }
            memset(frames[middle]->propagate_cost, 0, cuCount * sizeof(uint16_t));
            while (i > curnonb)
            {
                int p0 = i > middle ? middle : curnonb;
                int p1 = i < middle ? middle : lastnonb;
                if (i != middle)
                {
                    if ((rc = propagate(p0, p1, i, 0)) != 0) return rc;
                }
                i--;
            }
            if ((rc = propagate(curnonb, lastnonb, middle, 1)) != 0) return rc;
        }
        else
        {
            while (i > curnonb)
            {
                if ((rc = propagate(curnonb, lastnonb, i, 0)) != 0) return rc;
                i--;
            }
        }
        if ((rc = propagate(curnonb, lastnonb, lastnonb, 1)) != 0) return rc;
        lastnonb = curnonb;
    }
    cuTreeFinish(P, frames[lastnonb], averageDuration, lastnonb);
    if (P.b_pyramid && bframes > 1) cuTreeFinish(P, frames[lastnonb + (bframes + 1) / 2], averageDuration, 0);
    return X265AMD_OK;
}

/* Lookahead::frameCostRecalculate (slicetype.cpp:3802-3880), qgSize 16 and above, no hevc-aq */
extern "C" int64_t x265amd_frame_cost_recalculate(const x265amd_cutree_params* p, const uint16_t* lowres_costs, const double* qp_offset)
{
    if (!p || !lowres_costs || !qp_offset) return -1;
    const int w8 = p->width8, h8 = p->height8;
    int64_t score = 0;
    for (int cuy = h8 - 1; cuy >= 0; cuy--)
    {
        for (int cux = w8 - 1; cux >= 0; cux--)
        {
            int cuxy = cux + cuy * w8;
            int cuCost = lowres_costs[cuxy] & LOWRES_COST_MASK;
            double qp_adj = qp_offset[cuxy];
            cuCost = (cuCost * exp2fix8(qp_adj) + 128) >> 8;
            if ((cuy > 0 && cuy < h8 - 1 && cux > 0 && cux < w8 - 1) || w8 <= 2 || h8 <= 2) score += cuCost;
        }
    }
    return score;
}

/* ---- RateControl, the constant-rate-factor branch ---- */
namespace {
enum { B_SLICE = 0, P_SLICE = 1, I_SLICE = 2 };
/* (out of line as in the reference, where they live in another translation unit: inlined, -ffast-math is free to fold one into the other or into the pow() of getQScale) */
__attribute__((noinline)) double qScale2qp(double qScale) { return 12.0 + 6.0 * (double)log2(qScale / 0.85); }       /* x265_qScale2qp (common.cpp:244-247) */
__attribute__((noinline)) double qp2qScale(double qp) { return 0.85 * pow(2.0, (qp - 12.0) / 6.0); }                 /* x265_qp2qScale (common.cpp:249-252) */
inline double clip3d(double lo, double hi, double v) { return v < lo ? lo : (v > hi ? hi : v); }
const int kAbrScenecutInitQpMin = 12;       /* ABR_SCENECUT_INIT_QP_MIN (ratecontrol.cpp:338) */
}

struct x265amd_rc
{
    x265amd_rc_params p;
    int ncu;
    double qCompress, rateFactorConstant, ipOffset, pbOffset, frameDuration;
    double lastQScaleFor[3], lmin[3], lmax[3];
    double accumPQp, accumPNorm, shortTermCplxSum, shortTermCplxCount, lastRceq;
    int lastNonBPictType, framesDone, qp;
    bool isSceneTransition;
    int64_t currentSatd;
};

/* RateControl::RateControl + init (ratecontrol.cpp:184-360, :436-500) */
extern "C" x265amd_rc* x265amd_rc_open(const x265amd_rc_params* pp)
{
    if (!pp || pp->width < 16 || pp->height < 16 || !pp->fps_num || !pp->fps_denom || pp->ip_factor <= 0 || pp->pb_factor <= 0) return nullptr;
    x265amd_rc* r = new x265amd_rc;
    r->p = *pp;
    const x265amd_rc_params& p = r->p;
    const int lowresCuWidth = ((p.width / 2) + 8 - 1) >> 3, lowresCuHeight = ((p.height / 2) + 8 - 1) >> 3;
    r->ncu = lowresCuWidth * lowresCuHeight;
    r->qCompress = p.cu_tree ? 1 : p.q_compress;
    {
        double baseCplx = r->ncu * (p.bframes ? 120 : 80);
        double mbtree_offset = p.cu_tree ? (1.0 - p.q_compress) * 13.5 : 0;
        r->rateFactorConstant = pow(baseCplx, 1 - r->qCompress) / qp2qScale(p.rf_constant + mbtree_offset);
    }
    r->frameDuration = (double)p.fps_denom / p.fps_num;
    r->qp = (int)p.rf_constant;
    r->lastRceq = 1;
    r->shortTermCplxSum = 0; r->shortTermCplxCount = 0;
    r->lastNonBPictType = I_SLICE;
    r->ipOffset = 6.0 * log2(p.ip_factor);
    r->pbOffset = 6.0 * log2(p.pb_factor);
    for (int i = 0; i < 3; i++)
    {
        r->lastQScaleFor[i] = qp2qScale((int)p.rf_constant);        /* CRF_INIT_QP */
        r->lmin[i] = qp2qScale(p.qp_min);
        r->lmax[i] = qp2qScale(p.qp_max);
    }
    r->framesDone = 0;
    r->accumPNorm = .01;
    r->accumPQp = (int)p.rf_constant * r->accumPNorm;
    r->isSceneTransition = false;
    r->currentSatd = 0;
    return r;
}
extern "C" void x265amd_rc_close(x265amd_rc* rc) { delete rc; }

/* RateControl::rateControlStart with rateEstimateQscale, getQScale, clipQscale's tail and accumPQpUpdate, for rc.rateControlMode == X265_RC_CRF in one pass
 * (ratecontrol.cpp:1334-1643, :1900-2375, :2931-2954, :2535-2694) */
extern "C" int x265amd_rc_start(x265amd_rc* r, const x265amd_rc_frame* f, double* avg_qp_rc)
{
    if (!r || !f || f->slice_type < 0 || f->slice_type > 2) return -1;
    const x265amd_rc_params& p = r->p;
    const int sliceType = f->slice_type;
    const bool isRefFrameScenecut = sliceType != I_SLICE && f->ref0_scenecut;
    if (f->scenecut) r->isSceneTransition = true;
    else if (sliceType != B_SLICE && !isRefFrameScenecut) r->isSceneTransition = false;
    r->currentSatd = f->satd_cost >> (X265AMD_DEPTH - 8);
    double q;
    if (sliceType == B_SLICE)
    {
        double q0 = f->ref_avg_qp_rc[0];
        double q1 = f->ref_avg_qp_rc[1];
        bool i0 = f->ref_slice_type[0] == I_SLICE;
        bool i1 = f->ref_slice_type[1] == I_SLICE;
        int dt0 = abs(f->poc - f->ref_poc[0]);
        int dt1 = abs(f->poc - f->ref_poc[1]);
        if (f->ref_slice_type[0] == B_SLICE && f->ref_is_referenced[0]) q0 -= r->pbOffset / 2;
        if (f->ref_slice_type[1] == B_SLICE && f->ref_is_referenced[1]) q1 -= r->pbOffset / 2;
        if (i0 && i1) q = (q0 + q1) / 2 + r->ipOffset;
        else if (i0) q = q1;
        else if (i1) q = q0;
        else q = (q0 * dt1 + q1 * dt0) / (dt0 + dt1);
        if (f->is_referenced) q += r->pbOffset / 2;
        else q += r->pbOffset;
        if (r->isSceneTransition)
        {
            q = std::max((double)kAbrScenecutInitQpMin, q);
            double minScenecutQscale = qp2qScale(kAbrScenecutInitQpMin);
            r->lastQScaleFor[P_SLICE] = std::max(minScenecutQscale, r->lastQScaleFor[P_SLICE]);
        }
        q = qp2qScale(q);
    }
    else
    {
        double lqmin = r->lmin[sliceType];
        double lqmax = r->lmax[sliceType];
        r->shortTermCplxSum *= 0.5;
        r->shortTermCplxCount *= 0.5;
        r->shortTermCplxSum += r->currentSatd / (clipDuration(r->frameDuration) / kBaseFrameDuration);
        r->shortTermCplxCount++;
        const int coeffBits = (int)r->currentSatd;
        const double blurredComplexity = r->shortTermCplxSum / r->shortTermCplxCount;
        {
            /* getQScale(rce, m_rateFactorConstant) */
            if (p.cu_tree)
            {
                double timescale = (double)p.fps_denom / (2 * p.fps_num);
                q = pow(kBaseFrameDuration / clipDuration(2 * timescale), 1 - p.q_compress);
            }
            else
                q = pow(blurredComplexity, 1 - p.q_compress);
            if (coeffBits == 0) q = r->lastQScaleFor[sliceType];
            else
            {
                r->lastRceq = q;
                q /= r->rateFactorConstant;
            }
        }
        if (sliceType == I_SLICE && p.keyframe_max > 1 && r->lastNonBPictType != I_SLICE)
        {
            q = qp2qScale(r->accumPQp / r->accumPNorm);
            q /= fabs(p.ip_factor);
        }
        else if (r->framesDone > 0) { }
        else if (r->qCompress != 1)
            q = qp2qScale((int)p.rf_constant) / fabs(p.ip_factor);
        q = clip3d(lqmin, lqmax, q);
        if (r->isSceneTransition)
        {
            double minScenecutQscale = qp2qScale(kAbrScenecutInitQpMin);
            q = std::max(minScenecutQscale, q);
            r->lastQScaleFor[P_SLICE] = std::max(minScenecutQscale, r->lastQScaleFor[P_SLICE]);
        }
        q = clip3d(r->lmin[sliceType], r->lmax[sliceType], q);           /* clipQscale without VBV */
        r->lastQScaleFor[sliceType] = q;
        if (f->poc == 0 || r->lastQScaleFor[P_SLICE] < q) r->lastQScaleFor[P_SLICE] = q * fabs(p.ip_factor);
    }
    double qd = qScale2qp(q);
    qd = clip3d((double)p.qp_min, (double)p.qp_max, qd);
    r->qp = int(qd + 0.5);
    if (avg_qp_rc) *avg_qp_rc = qd;
    /* accumPQpUpdate */
    r->accumPQp *= .95;
    r->accumPNorm *= .95;
    r->accumPNorm += 1;
    if (sliceType == I_SLICE) r->accumPQp += r->qp + r->ipOffset;
    else r->accumPQp += r->qp;
    if (sliceType != B_SLICE) r->lastNonBPictType = sliceType;
    r->framesDone++;
    return r->qp;
}

/* Analysis::calculateQpforCuSize (analysis.cpp:3634-3714), qgSize 16 and above, no hevc-aq */
extern "C" int x265amd_cu_qp(double base_qp, const double* qpoffs, int width, int height, int x, int y, int size, int qp_min, int qp_max)
{
    double qp = base_qp;
    if (qpoffs)
    {
        const int loopIncr = 16;
        uint32_t maxCols = (width + (loopIncr - 1)) / loopIncr;
        double dQpOffset = 0;
        uint32_t cnt = 0;
        for (uint32_t block_yy = y; block_yy < (uint32_t)(y + size) && block_yy < (uint32_t)height; block_yy += loopIncr)
        {
            for (uint32_t block_xx = x; block_xx < (uint32_t)(x + size) && block_xx < (uint32_t)width; block_xx += loopIncr)
            {
                uint32_t idx = ((block_yy / loopIncr) * (maxCols)) + (block_xx / loopIncr);
                dQpOffset += qpoffs[idx];
                cnt++;
            }
        }
        dQpOffset /= cnt;
        qp += dQpOffset;
    }
    const int q = (int)(qp + 0.5);
    return q < qp_min ? qp_min : (q > qp_max ? qp_max : q);
}
