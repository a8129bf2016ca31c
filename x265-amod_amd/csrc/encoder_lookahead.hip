/* The encoder object's lookahead (encoder_impl.h): what Lookahead / LookaheadTLD / CostEstimateGroup of the reference do for it -- Lowres::init, the adaptive quantisation
 * offsets, the cost estimates (on the device: lowres_kernels.hip), the lookahead's and the slice's weight analysis, scene-cut detection, the B-frame trellis, cuTree,
 * slicetypeDecide (reference: source/encoder/slicetype.cpp, source/common/lowres.cpp, source/encoder/weightPrediction.cpp). */
#include "encoder_impl.h"

/* ---- the lookahead's slice-type decision with scene-cut detection (param.scenecutThreshold > 0, bFrameAdaptive 0) ----
 * Lowres::init + LookaheadTLD::lowresIntraEstimate for every picture handed in (lowres.cpp:337-403, slicetype.cpp:715-824): x265amd_lowres_init,
 * x265amd_lowres_intra_costs; costEst[0][0] = the intra costs of the blocks that are not on the picture's edge. */
int x265amd_encoder::lowresInit(Pic& pic)
{
    if (xa_scratch_alloc((void**)&pic.dLowres, lowPlaneElems * 4 * sizeof(pixel)) != hipSuccess || xa_scratch_alloc((void**)&pic.dIntraCost, (size_t)lowCuW * lowCuH * 4 + (size_t)lowCuW * lowCuH) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if (hipMemsetAsync(pic.dLowres, 0, lowPlaneElems * 4 * sizeof(pixel), laStream) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    pixel* planes[4];
    for (int k = 0; k < 4; k++) planes[k] = pic.dLowres + (size_t)k * lowPlaneElems + lowOrg;
    int rc = x265amd_lowres_init(laStream, pic.dSrc + org[0], stride, lowW, lowH, planes, lowStride, marginX, marginY);
    if (rc != X265AMD_OK) return rc;
    const int lambda = X265AMD_DEPTH > 8 ? 16 : 1;          /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP], X265_LOOKAHEAD_QP = 12 + 6 * (depth - 8) (common.h:213) */
    uint8_t* dMode = (uint8_t*)(pic.dIntraCost + (size_t)lowCuW * lowCuH);
    rc = x265amd_lowres_intra_costs(laStream, planes[0], lowStride, lowCuW, lowCuH, lambda, pic.dIntraCost, dMode);
    if (rc != X265AMD_OK) return rc;
    /* the picture's sums for the weight analysis are measured in front of the one wait of this function */
    static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");          /* debugging aid: letters s / l / p switch the sums, the lookahead's analysis, the slice's analysis off */
    const bool sums = (p.bEnableWeightedPred || p.bEnableWeightedBiPred) && !(dbgWp && strchr(dbgWp, 's'));
    if (sums && !aqOn)
    {
        /* LookaheadTLD::calcAdaptiveQuantFrame with AQ off (slicetype.cpp:507-513): acEnergyCu over every 16x16 block for Lowres::wp_sum / wp_ssd, then :678-700 */
        const int bw = (W + 15) / 16, bh = (H + 15) / 16;
        if (!wpEnergy && (xa_scratch_alloc(&wpEnergy, (size_t)bw * bh * 4) != hipSuccess || xa_scratch_alloc(&wpSums, 6 * 8) != hipSuccess || xa_mapped_alloc(&wpSumsHost, 6 * 8, true) != hipSuccess ||
                          xa_mapped_alloc(&wpMvs, (size_t)lowCuW * lowCuH * 4, false) != hipSuccess))
            return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
        const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
        rc = x265amd_aq_energy(laStream, srcP, stride, cstride, W, H, 16, (uint32_t*)wpEnergy, (uint64_t*)wpSums);
        if (rc == X265AMD_OK && hipMemcpyAsync(wpSumsHost, wpSums, 6 * 8, hipMemcpyDeviceToHost, laStream) != hipSuccess) rc = xa_fail(X265AMD_EHIP, "encoder_encode: picture sums");
        if (rc != X265AMD_OK) return rc;
    }
    if (aqOn && (rc = adaptiveQuant(pic)) != X265AMD_OK) return rc;
    std::vector<int32_t> ic((size_t)lowCuW * lowCuH);
    if (hipMemcpyAsync(ic.data(), pic.dIntraCost, ic.size() * 4, hipMemcpyDeviceToHost, laStream) != hipSuccess || hipStreamSynchronize(laStream) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: lowres intra costs");
    if (aqOn)
    {
        /* the rest of calcAdaptiveQuantFrame (slicetype.cpp:513-640) on the block energies that have arrived with the intra costs */
        const int bw = (W + 15) / 16, bh = (H + 15) / 16, nb = bw * bh;
        pic.qpAqOffset.assign((size_t)nb, 0.0); pic.qpCuTreeOffset.assign((size_t)nb, 0.0); pic.invQscale.assign((size_t)nb, 256);
        /* (strength 0 -- cuTree without adaptive quantisation, --tune psnr: the arrays stay zero / 256, slicetype.cpp:483-505; the energies were only needed for the picture's sums) */
        if (p.aqStrength != 0)
            rc = x265amd_aq_offsets((const uint32_t*)aqEnergyHost, nb, lowCuW * lowCuH, p.aqMode, p.aqStrength, 1.0, 16, pic.qpAqOffset.data(), pic.qpCuTreeOffset.data(), pic.invQscale.data());
        if (rc != X265AMD_OK) return xa_fail(rc, "encoder_encode: adaptive quantisation");
        pic.intraCostHost = ic;
        if (p.cuTree) pic.propagateCost.assign((size_t)lowCuW * lowCuH, 0);
    }
    int64_t est = 0;
    const bool all = lowCuW <= 2 || lowCuH <= 2;
    for (int y = 0; y < lowCuH; y++)
        for (int x = 0; x < lowCuW; x++)
            if (all || (x > 0 && x < lowCuW - 1 && y > 0 && y < lowCuH - 1)) est += ic[(size_t)y * lowCuW + x];
    pic.costEst[0] = est;
    if (sums)
    {
        uint64_t wp[6];
        memcpy(wp, aqOn ? (const void*)((const char*)aqEnergyHost + (size_t)((W + 15) / 16) * ((H + 15) / 16) * 4) : wpSumsHost, sizeof(wp));
        const int maxCol = ((W + 8) >> 4) << 4, maxRow = ((H + 8) >> 4) << 4;
        const int width[3] = { maxCol, maxCol >> 1, maxCol >> 1 }, height[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
        {
            const uint64_t sum = wp[i], ssd = wp[3 + i];
            pic.wpSum[i] = sum;
            pic.wpSsd[i] = ssd - (sum * sum + (uint64_t)((width[i] * height[i]) / 2)) / (uint64_t)(width[i] * height[i]);
        }
    }
    return X265AMD_OK;
}

/* LookaheadTLD::calcAdaptiveQuantFrame's block loop (slicetype.cpp:560-600): acEnergyCu of every 16x16 block (luma + both chroma blocks) and the picture's sums for the weight
 * analysis, enqueued on the lookahead's stream with their read-back; the caller waits for the stream once (lowresInit) and turns the energies into offsets */
int x265amd_encoder::adaptiveQuant(Pic& pic)
{
    const int bw = (W + 15) / 16, bh = (H + 15) / 16;
    const size_t nb = (size_t)bw * bh;
    if (!aqEnergy && (xa_scratch_alloc(&aqEnergy, nb * 4) != hipSuccess || xa_scratch_alloc(&aqSums, 6 * 8) != hipSuccess || xa_mapped_alloc(&aqEnergyHost, nb * 4 + 6 * 8, true) != hipSuccess))
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if ((p.bEnableWeightedPred || p.bEnableWeightedBiPred) && !wpMvs && xa_mapped_alloc(&wpMvs, (size_t)lowCuW * lowCuH * 4, false) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    int rc = x265amd_aq_energy(laStream, srcP, stride, cstride, W, H, 16, (uint32_t*)aqEnergy, (uint64_t*)aqSums);
    if (rc != X265AMD_OK) return rc;
    if (hipMemcpyAsync(aqEnergyHost, aqEnergy, nb * 4, hipMemcpyDeviceToHost, laStream) != hipSuccess ||
        hipMemcpyAsync((char*)aqEnergyHost + nb * 4, aqSums, 6 * 8, hipMemcpyDeviceToHost, laStream) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: block energies");
    return X265AMD_OK;
}

namespace {
/* bs_size_ue / bs_size_se (common/bitstream.h:94-136) */
inline int bitSizeOf(unsigned v) { int n = 1; while (v > 1) { v >>= 1; n += 2; } return n; }
inline int bsSizeUe(unsigned val) { return bitSizeOf(val + 1); }
inline int bsSizeSe(int val) { int tmp = 1 - val * 2; if (tmp < 0) tmp = val * 2; return tmp < 256 ? bitSizeOf((unsigned)tmp) : bitSizeOf((unsigned)(tmp >> 8)) + 16; }
/* weight_pp_c's arguments for a WeightParam as weightCostLuma / weightCost pass them */
inline x265amd_weight_cand weightCand(int scale, int denom, int offset)
{
    const int correction = 14 - X265AMD_DEPTH;
    x265amd_weight_cand c;
    c.present = 1; c.w0 = scale; c.round = (denom ? 1 << (denom - 1) : 0) << correction; c.shift = denom + correction; c.offset = offset << (X265AMD_DEPTH - 8);
    return c;
}
}

/* LookaheadTLD::weightsAnalyse (slicetype.cpp:879-978) before a list-0 search of `fenc` against `ref`: the early exit when the two do not differ in mean or variance; else the
 * unweighted cost against one candidate (scale from the variances, offset from the means), a smaller denominator if the scale is even, and the 0.998 test.  weighted: the
 * reference's four planes are weighted for the search (scale / 2^denom, offset) */
/* (in two halves, so that the measurements of every search of a batch go out as one launch: the guess -- false: the early exit, no weight --, then the decision from
 * the two costs) */
bool x265amd_encoder::lookaheadWeightGuess(Pic& fenc, Pic& ref, LaWeight& g)
{
    static const float epsilon = 1.f / 128.f;
    float guessScale, fencMean, refMean;
    if (fenc.wpSsd[0] && ref.wpSsd[0]) guessScale = sqrtf((float)fenc.wpSsd[0] / ref.wpSsd[0]);
    else guessScale = 1.0f;
    fencMean = (float)fenc.wpSum[0] / (lowH * lowW) / (1 << (X265AMD_DEPTH - 8));
    refMean = (float)ref.wpSum[0] / (lowH * lowW) / (1 << (X265AMD_DEPTH - 8));
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon) return false;
    {
        /* WeightParam::setFromWeightAndOffset((int)(guessScale * 128 + 0.5f), 0, 7, true) (slice.h:304-316) */
        int w = (int)(guessScale * 128 + 0.5f), d = 7;
        while (d > 0 && w > 127) { d--; w >>= 1; }
        w = std::min(w, 127);
        g.mindenom = d; g.minscale = w;
    }
    g.curScale = g.minscale;
    g.curOffset = (int)(fencMean - refMean * g.curScale / (1 << g.mindenom) + 0.5f);
    if (g.curOffset < -128 || g.curOffset > 127)
    {
        g.curOffset = std::max(-128, std::min(127, g.curOffset));
        g.curScale = (int)((1 << g.mindenom) * (fencMean - g.curOffset) / refMean + 0.5f);
        g.curScale = std::max(0, std::min(127, g.curScale));
    }
    return true;
}
void x265amd_encoder::lookaheadWeightDecide(const LaWeight& g, const uint32_t costs[2], bool& weighted, int& scale, int& denom, int& offset)
{
    weighted = false;
    int minoff = 0, minscale = g.minscale, mindenom = g.mindenom;
    unsigned int minscore = costs[0], origscore = costs[0];
    int found = 0;
    if (!minscore) return;
    const unsigned int sc = costs[1];
    if (sc < minscore) { minscore = sc; minscale = g.curScale; minoff = g.curOffset; found = 1; }
    if (mindenom > 0 && !(minscale & 1))
    {
        const int idx = minscale ? __builtin_ctz((unsigned)minscale) : 32;
        const int shift = std::min(idx, mindenom);
        mindenom -= shift; minscale >>= shift;
    }
    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) return;
    weighted = true; scale = minscale; denom = mindenom; offset = minoff;
}
/* weightAnalyse (weightPrediction.cpp:222-540) for a P picture (list 0) or, with weighted bi-prediction, a B picture (both lists): the first reference of each list.  The chroma
 * denominator that fits both chroma scale guesses; per plane: the early exit, else the reference motion compensated with the lookahead's vectors of that distance (mcLuma on the
 * lowres planes, mcChroma on the SOURCE chroma planes) against every candidate scale (+-4 around the guess) and offset (+-2 around the mean's), each with the slice header's cost,
 * a smaller luma denominator if the scale is even, the 0.998 test.  Without a luma weight chroma is not looked at.  Leaves slice.m_weightPredTable in pic.wp and pic.weighted
 * (some reference carries a weight). */
int x265amd_encoder::sliceWeights(Pic& pic)
{
    pic.weighted = false;
    memset(pic.wp, 0, sizeof(pic.wp));
    const int numDirs = isBType(pic.type) ? 2 : 1;
    const float epsilon = 1.f / 128.f;
    const int w16 = ((W + 15) >> 4) << 4, h16 = ((H + 15) >> 4) << 4;
    int numpixels[3];
    numpixels[0] = w16 * h16; numpixels[1] = numpixels[2] = numpixels[0] >> 2;
    auto setW = [](x265amd_weight& w, bool present, int scale, int denom, int off) { w.present = present; w.w = (int16_t)scale; w.denom = (uint8_t)denom; w.o = (int16_t)off; };
    int chromaDenom = 7, lumaDenom = 7;
    const int lambda = X265AMD_DEPTH > 8 ? 16 : 1;          /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP] */
    for (int list = 0; list < numDirs; list++)
    {
        x265amd_weight* weights = pic.wp[list][0];
        Pic& ref = *pic.lists[list][0];
        const int diffPoc = abs(pic.poc - ref.poc);
        float guessScale[3], fencMean[3], refMean[3];
        for (int plane = 0; plane < 3; plane++)
        {
            setW(weights[plane], false, 1, 0, 0);
            const uint64_t fencVar = pic.wpSsd[plane] + !ref.wpSsd[plane], refVar = ref.wpSsd[plane] + !ref.wpSsd[plane];
            guessScale[plane] = sqrt((float)fencVar / refVar);
            fencMean[plane] = (float)pic.wpSum[plane] / (numpixels[plane]) / (1 << (X265AMD_DEPTH - 8));
            refMean[plane] = (float)ref.wpSum[plane] / (numpixels[plane]) / (1 << (X265AMD_DEPTH - 8));
        }
        while (!list && chromaDenom > 0)
        {
            const float thresh = 127.f / (1 << chromaDenom);
            if (guessScale[1] < thresh && guessScale[2] < thresh) break;
            chromaDenom--;
        }
        setW(weights[1], false, 1 << chromaDenom, chromaDenom, 0);
        setW(weights[2], false, 1 << chromaDenom, chromaDenom, 0);
        void* dMvs = nullptr;           /* the field of the luma analysis serves the chroma planes too */
        for (int plane = 0; plane < 3; plane++)
        {
            const int denom = plane ? chromaDenom : lumaDenom;
            if (plane && !weights[0].present) break;
            if (fabsf(refMean[plane] - fencMean[plane]) < 0.5f && fabsf(1.f - guessScale[plane]) < epsilon) { setW(weights[plane], false, 1 << denom, denom, 0); continue; }
            if (plane)
            {
                const int scale = std::max(0, std::min(255, (int)(guessScale[plane] * (1 << denom) + 0.5f)));
                if (scale > 127) continue;
                weights[plane].w = (int16_t)scale;
            }
            else
            {
                /* WeightParam::setFromWeightAndOffset(w, 0, denom, bNormalize = !list) (slice.h:304-316) */
                int w = (int)(guessScale[plane] * (1 << denom) + 0.5f), d = denom;
                while (!list && d > 0 && w > 127) { d--; w >>= 1; }
                w = std::min(w, 127);
                weights[plane].o = 0; weights[plane].denom = (uint8_t)d; weights[plane].w = (int16_t)w;
            }
            int mindenom = weights[plane].denom, minscale = weights[plane].w, minoff = 0;
            if (!plane && diffPoc <= p.bframes + 1)
            {
                const std::vector<int16_t>& f = list ? pic.lowMvs1[diffPoc < 18 ? diffPoc : 0] : pic.lowMvs[diffPoc < 18 ? diffPoc : 0];
                if (diffPoc < 18 && !f.empty())
                {
                    /* (a record the host writes in place: no copy from pageable memory) */
                    if (!wpMvs || f.size() * 2 > (size_t)lowCuW * lowCuH * 4) return xa_fail(X265AMD_EHIP, "encoder_encode: lowres vectors");
                    dMvs = wpMvs;
                    memcpy(dMvs, f.data(), f.size() * 2);
                }
            }
            /* the candidates in the order the reference tries them */
            struct Cand { int scale, off, startOffset, iter; };
            std::vector<Cand> order;
            std::vector<x265amd_weight_cand> cands(1);
            memset(&cands[0], 0, sizeof(cands[0]));
            const int startScale = std::max(0, std::min(127, minscale - 4)), endScale = std::max(0, std::min(127, minscale + 4));
            for (int scale = startScale; scale <= endScale; scale++)
            {
                const int deltaWeight = scale - (1 << mindenom);
                if (deltaWeight > 127 || deltaWeight <= -128) continue;
                int curScale = scale;
                int curOffset = (int)(fencMean[plane] - refMean[plane] * curScale / (1 << mindenom) + 0.5f);
                if (curOffset < -128 || curOffset > 127)
                {
                    curOffset = std::max(-128, std::min(127, curOffset));
                    curScale = (int)((1 << mindenom) * (fencMean[plane] - curOffset) / refMean[plane] + 0.5f);
                    curScale = std::max(0, std::min(127, curScale));
                }
                const int startOffset = std::max(-128, std::min(127, curOffset - 2)), endOffset = std::max(-128, std::min(127, curOffset + 2));
                for (int off = startOffset; off <= endOffset; off++) { order.push_back({ curScale, off, startOffset, scale }); cands.push_back(weightCand(curScale, mindenom, off)); }
            }
            std::vector<uint32_t> costs(cands.size(), 0);
            int rc;
            if (!plane)
            {
                const pixel* refPlanes[4];
                for (int t = 0; t < 4; t++) refPlanes[t] = ref.dLowres + (size_t)t * lowPlaneElems + lowOrg;
                rc = x265amd_lowres_weight_costs(laStream, pic.dLowres + lowOrg, refPlanes, (const int16_t*)dMvs, pic.dIntraCost, lowStride, lowW, lowH, cands.data(), (int)cands.size(), costs.data());
            }
            else
            {
                if (!pic.dSrc || !ref.dSrc) return xa_fail(X265AMD_EHIP, "encoder_encode: weight analysis without the reference's source picture");
                const int cw = ((W >> 4) << 4) >> 1, chh = ((H >> 4) << 4) >> 1;
                rc = x265amd_chroma_weight_costs(laStream, pic.dSrc + org[plane], ref.dSrc + org[plane], (const int16_t*)dMvs, cstride, cw, chh, lowCuW, lowCuH, cands.data(), (int)cands.size(), costs.data());
            }
            if (rc != X265AMD_OK) return rc;
            const uint32_t origscore = costs[0];
            if (!origscore) { setW(weights[plane], false, 1 << denom, denom, 0); continue; }
            uint32_t minscore = origscore;
            bool bFound = false;
            for (size_t k = 0; k < order.size(); k++)
            {
                const Cand& c = order[k];
                /* sliceHeaderCost(&wsp, lambda, !!plane): four times the lambda for chroma (analysed at full resolution), the denominator counted twice for luma */
                const int lam = plane ? lambda * 4 : lambda;
                const int hdr = lam * (10 + bsSizeUe((unsigned)mindenom) * (plane ? 1 : 2) + 2 * (bsSizeSe(c.scale) + bsSizeSe(c.off)));
                const uint32_t sc = costs[k + 1] + (uint32_t)hdr;
                if (sc < minscore) { minscore = sc; minscale = c.scale; minoff = c.off; bFound = true; }
                /* "Don't check any more offsets if the previous one had a lower cost than the current one": the rest of this scale's offsets are skipped */
                if (minoff == c.startOffset && c.off != c.startOffset)
                    while (k + 1 < order.size() && order[k + 1].iter == c.iter) k++;
            }
            if (!(plane || list) && mindenom > 0 && !(minscale & 1))
            {
                const int idx = minscale ? __builtin_ctz((unsigned)minscale) : 32;
                const int shift = std::min(idx, mindenom);
                mindenom -= shift; minscale >>= shift;
            }
            if (!bFound || (minscale == (1 << mindenom) && minoff == 0) || (float)minscore / origscore > 0.998f) setW(weights[plane], false, 1 << denom, denom, 0);
            else setW(weights[plane], true, minscale, mindenom, minoff);
        }
        if (weights[0].present && weights[1].present != weights[2].present)
        {
            /* "make sure both chroma channels match" */
            if (weights[1].present) weights[2] = weights[1]; else weights[1] = weights[2];
        }
        lumaDenom = weights[0].denom; chromaDenom = weights[1].denom;
        for (size_t r = 1; r < pic.lists[list].size(); r++)
        {
            setW(pic.wp[list][r][0], false, 1 << lumaDenom, lumaDenom, 0);
            setW(pic.wp[list][r][1], false, 1 << chromaDenom, chromaDenom, 0);
            setW(pic.wp[list][r][2], false, 1 << chromaDenom, chromaDenom, 0);
        }
        for (int plane = 0; plane < 3; plane++) pic.weighted |= weights[plane].present != 0;
    }
    pic.lumaDenom = pic.wp[0][0][0].denom; pic.chromaDenom = pic.wp[0][0][1].denom;         /* what pred_weight_table() codes once: the first reference's (entropy.cpp:1376-1387) */
    const bool wpLog = getenv("X265AMD_WP_LOG") != nullptr;         /* (read per picture: a test switches it on for one encode) */
    if (wpLog && pic.weighted)
    {
        /* the reference's --log-level full line */
        char buf[512]; int n = snprintf(buf, sizeof(buf), "poc: %d weights:", pic.poc);
        for (int list = 0; list < numDirs; list++)
        {
            const x265amd_weight* w = pic.wp[list][0];
            if (!(w[0].present || w[1].present || w[2].present)) continue;
            n += snprintf(buf + n, sizeof(buf) - n, " [L%d:R0 ", list);
            if (w[0].present) n += snprintf(buf + n, sizeof(buf) - n, "Y{%d/%d%+d}", w[0].w, 1 << w[0].denom, w[0].o);
            if (w[1].present) n += snprintf(buf + n, sizeof(buf) - n, "U{%d/%d%+d}", w[1].w, 1 << w[1].denom, w[1].o);
            if (w[2].present) n += snprintf(buf + n, sizeof(buf) - n, "V{%d/%d%+d}", w[2].w, 1 << w[2].denom, w[2].o);
            n += snprintf(buf + n, sizeof(buf) - n, "]");
        }
        fprintf(stderr, "x265amd: %s\n", buf);
    }
    return X265AMD_OK;
}

/* CostEstimateGroup::singleCost(p0, p1, b = p1) -> estimateFrameCost (slicetype.cpp:3882-4075) for a P candidate `dist` pictures behind its reference: the block
 * loop is x265amd_lowres_frame_cost (motion search of list 0 included: every (picture, distance) pair is estimated once); costEst / intraMbs are the sums over the
 * blocks that are not on the picture's edge (estimateCUCost's tail, :4220-4248) */
int x265amd_encoder::frameCostP(Pic& b, Pic& ref, int dist)
{
    int64_t score;
    return frameCostAt(b, ref, nullptr, dist, 0, score);
}

/* CostEstimateGroup::estimateFrameCost (slicetype.cpp:3975-4075) for candidate `fenc` against `ref0` d0 pictures before it and, for a B estimate, `ref1` d1 pictures behind
 * it: the searches a field still lacks run inside the block loop (bDoSearch), fields that exist are read again; the sum over the blocks that are not on the picture's
 * edge, scaled by 100 / (130 + bFrameBias) for a B estimate; intra blocks are counted for P estimates only */
int x265amd_encoder::frameCostAt(Pic& fenc, Pic& ref0, Pic* ref1, int d0, int d1, int64_t& score)
{
    if (d0 < 1 || d0 > 17 || d1 < 0 || d1 > 17 || (d1 > 0) != (ref1 != nullptr)) return xa_fail(X265AMD_EINVAL, "encoder: lookahead distance");
    if (fenc.cost2[d0][d1] >= 0) { score = fenc.cost2[d0][d1]; return X265AMD_OK; }
    /* the reference makes this estimate now, with the searches its fields still lack: what was searched ahead becomes the picture's */
    if (fenc.lowMvs[d0].empty() && !fenc.specMvs[d0].empty()) { fenc.lowMvs[d0].swap(fenc.specMvs[d0]); fenc.lowMvc[d0].swap(fenc.specMvc[d0]); }
    if (d1 > 0 && fenc.lowMvs1[d1].empty() && !fenc.specMvs1[d1].empty()) { fenc.lowMvs1[d1].swap(fenc.specMvs1[d1]); fenc.lowMvc1[d1].swap(fenc.specMvc1[d1]); }
    if (fenc.specCost2[d0][d1] >= 0 && !fenc.lowMvs[d0].empty() && (d1 == 0 || !fenc.lowMvs1[d1].empty()))
    {
        score = fenc.cost2[d0][d1] = fenc.specCost2[d0][d1];
        {
            /* ... and its block costs (cuTree) */
            const int key = d0 * 32 + d1;
            auto it = fenc.dSpecLc.find(key);
            if (it != fenc.dSpecLc.end()) { fenc.dropLc(fenc.dLc, key); fenc.dLc[key] = it->second; fenc.dSpecLc.erase(it); fenc.lcHost.erase(key); }
        }
        if (d1 == 0) { fenc.costEst[d0] = score; fenc.intraMbs[d0] = fenc.specIntraMbs[d0]; }
        return X265AMD_OK;
    }
    std::vector<CostJob> one(1);
    one[0].fenc = &fenc; one[0].ref0 = &ref0; one[0].ref1 = ref1; one[0].d0 = d0; one[0].d1 = d1;
    const int rc = frameCostMany(one);
    score = fenc.cost2[d0][d1];
    return rc;
}

void x265amd_encoder::laFieldPut(const void* key, void* mv, void* mc)
{
    auto it = laFields.find(key);
    if (it != laFields.end()) { laBufPut(it->second.mv); laBufPut(it->second.mc); it->second = DevField{ mv, mc, ++laFieldClock }; }
    else laFields.emplace(key, DevField{ mv, mc, ++laFieldClock });
}
/* (called when nothing of the lookahead's is in flight: behind frameCostMany's wait) */
void x265amd_encoder::laFieldsTrim()
{
    {
        /* fields of pictures that have gone since the last look */
        std::vector<const void*> dead;
        { std::lock_guard<std::mutex> lk(laPool->mu); dead.swap(laPool->deadFields); }
        for (const void* k : dead)
        {
            auto it = laFields.find(k);
            if (it != laFields.end()) { laBufPut(it->second.mv); laBufPut(it->second.mc); laFields.erase(it); }
        }
    }
    if (laFields.size() <= LA_FIELDS_MAX) return;
    std::vector<uint64_t> ages;
    for (auto& f : laFields) ages.push_back(f.second.used);
    std::nth_element(ages.begin(), ages.begin() + ages.size() / 4, ages.end());
    const uint64_t cut = ages[ages.size() / 4];
    for (auto it = laFields.begin(); it != laFields.end();)
        if (it->second.used < cut) { laBufPut(it->second.mv); laBufPut(it->second.mc); it = laFields.erase(it); } else ++it;
}
void x265amd_encoder::laFieldsFree()
{
    for (auto& f : laFields) { laBufPut(f.second.mv); laBufPut(f.second.mc); }
    laFields.clear();
}

/* Independent estimates side by side: every job on one of a handful of streams (the block loop of an estimate is a few dozen wavefronts chained row to row -- latency,
 * not throughput: a dozen of them overlap on the device), one wait for all, then the host sums.  Jobs of one call must not share a motion field they search or a cost
 * they fill (the callers' batches are by (picture, distance) pairs). */
int x265amd_encoder::frameCostMany(std::vector<CostJob>& jobs)
{
    if (jobs.empty()) return X265AMD_OK;
    laFieldsTrim();             /* (nothing of the lookahead's is in flight here either) */
    const size_t ncu = (size_t)lowCuW * lowCuH;
    int rc = X265AMD_OK;
    const auto tb0 = std::chrono::steady_clock::now();
    struct Tm { x265amd_encoder* e; std::chrono::steady_clock::time_point t0; size_t n; ~Tm() { const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); if (n > 1) { e->laBatchMs += ms; e->laBatches++; } else { e->laSingleMs += ms; e->laSingles++; } } } tm_{ this, tb0, jobs.size() };
    std::vector<x265amd_lowres_cost_job> kj(jobs.size());
    size_t issued = 0;
    /* a field that exists is read where its device copy lies (laFields); one that has none (evicted) is uploaded once and entered */
    auto shared = [&](const std::vector<int16_t>& mv, const std::vector<int32_t>& mc, void*& dMv, void*& dMc) -> bool {
        auto it = laFields.find(mv.data());
        if (it == laFields.end())
        {
            void* a = laBuf(); void* b = laBuf();
            if (!a || !b) { laBufPut(a); laBufPut(b); return false; }
            laFieldPut(mv.data(), a, b);
            it = laFields.find(mv.data());
            if (hipMemcpyAsync(a, mv.data(), ncu * 4, hipMemcpyHostToDevice, laStream) != hipSuccess || hipMemcpyAsync(b, mc.data(), ncu * 4, hipMemcpyHostToDevice, laStream) != hipSuccess)
            { laBufPut(a); laBufPut(b); laFields.erase(mv.data()); return false; }           /* (no entry for a field that did not arrive) */
        }
        it->second.used = ++laFieldClock;
        dMv = it->second.mv; dMc = it->second.mc;
        return true;
    };
    auto tph = std::chrono::steady_clock::now();
    auto phase = [&](int i) { const auto t = std::chrono::steady_clock::now(); laPhaseMs[i] += std::chrono::duration<double, std::milli>(t - tph).count(); tph = t; };
    /* first what every estimate searches, and for the list-0 searches the lookahead's weight guess: their measurements (two candidates each) go out as ONE launch */
    std::vector<LaWeight> lw(jobs.size());
    std::vector<int> wAt(jobs.size(), -1);
    std::vector<x265amd_weight_cost_job> wj;
    for (size_t k = 0; k < jobs.size(); k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        if (!j.spec && (!j.whole || laNumSlices <= 1))
        {
            /* made for good now: a field searched ahead of its time is the one this estimate would search (with cooperative slices only if this estimate is one of those
             * the reference makes in slices too: its batch searches whole pictures) */
            if (fenc.lowMvs[j.d0].empty() && !fenc.specMvs[j.d0].empty()) { fenc.lowMvs[j.d0].swap(fenc.specMvs[j.d0]); fenc.lowMvc[j.d0].swap(fenc.specMvc[j.d0]); }
            if (j.d1 > 0 && fenc.lowMvs1[j.d1].empty() && !fenc.specMvs1[j.d1].empty()) { fenc.lowMvs1[j.d1].swap(fenc.specMvs1[j.d1]); fenc.lowMvc1[j.d1].swap(fenc.specMvc1[j.d1]); }
        }
        /* (an estimate made ahead of its time reads and fills the fields made ahead of their time as well as the picture's own) */
        const bool have0 = !fenc.lowMvs[j.d0].empty() || (j.spec && !fenc.specMvs[j.d0].empty()), have1 = j.d1 > 0 && (!fenc.lowMvs1[j.d1].empty() || (j.spec && !fenc.specMvs1[j.d1].empty()));
        j.search0 = !have0; j.search1 = j.d1 > 0 && !have1;
        static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");
        if (p.bEnableWeightedPred && j.search0 && !(dbgWp && strchr(dbgWp, 'l')) && lookaheadWeightGuess(fenc, *j.ref0, lw[k]))
        {
            x265amd_weight_cost_job w;
            memset(&w, 0, sizeof(w));
            w.d_fenc = fenc.dLowres + lowOrg; w.d_intra_cost = fenc.dIntraCost;
            for (int t = 0; t < 4; t++) w.d_ref[t] = j.ref0->dLowres + (size_t)t * lowPlaneElems + lowOrg;
            w.cands[1] = weightCand(lw[k].curScale, lw[k].mindenom, lw[k].curOffset);
            wAt[k] = (int)wj.size();
            wj.push_back(w);
        }
        laJobs++; laSearches += (j.search0 ? 1 : 0) + (j.search1 ? 1 : 0);
    }
    std::vector<uint32_t> wCosts(2 * wj.size() + 2);
    laWeightJobs += wj.size();
    if (!wj.empty()) rc = x265amd_lowres_weight_costs_many(laStream, wj.data(), (int)wj.size(), lowStride, lowW, lowH, wCosts.data());
    for (size_t k = 0; k < jobs.size() && rc == X265AMD_OK; k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        bool weighted = false; int wScale = 0, wDenom = 0, wOffset = 0;
        if (wAt[k] >= 0) lookaheadWeightDecide(lw[k], &wCosts[2 * wAt[k]], weighted, wScale, wDenom, wOffset);
        /* the estimate's own buffers: its costs, and the fields it searches (the fields it reads are the call's shared copies) */
        void** bufs[6] = { &j.dLc, &j.dBc, &j.dMvs, &j.dMvc, &j.dMvs1, &j.dMvc1 };
        const size_t sizes[6] = { ncu * 2, ncu * 4, ncu * 4, ncu * 4, ncu * 4, ncu * 4 };
        const bool want[6] = { true, true, j.search0, j.search0, j.search1, j.search1 };
        (void)sizes;
        for (int b = 0; b < 6; b++) if (want[b] && !(*bufs[b] = laBuf())) rc = xa_fail(X265AMD_EHIP, "encoder: device allocation");
        issued = k + 1;
        if (rc != X265AMD_OK) break;
        x265amd_lowres_cost_job& q = kj[k];
        memset(&q, 0, sizeof(q));
        q.d_fenc = fenc.dLowres + lowOrg;
        for (int t = 0; t < 4; t++) { q.d_ref0[t] = j.ref0->dLowres + (size_t)t * lowPlaneElems + lowOrg; q.d_ref1[t] = j.ref1 ? j.ref1->dLowres + (size_t)t * lowPlaneElems + lowOrg : nullptr; }
        q.d_intra_cost = fenc.dIntraCost;
        q.d_lowres_costs = (uint16_t*)j.dLc; q.d_bcost = (int32_t*)j.dBc; q.do_search0 = j.search0; q.do_search1 = j.search1;
        if (!j.whole && laNumSlices > 1) { q.rows_per_slice = laRowsPerSlice; q.num_slices = laNumSlices; }
        if (weighted)
        {
            /* the four planes weighted, margins included, for this estimate's list-0 search (slicetype.cpp:962-977) */
            const int correction = 14 - X265AMD_DEPTH;
            if (xa_scratch_alloc(&j.dW, lowPlaneElems * 4 * sizeof(pixel)) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: device allocation"); break; }
            rc = x265amd_weight_buffer(laStream, j.ref0->dLowres, (pixel*)j.dW, lowPlaneElems * 4, wScale, (wDenom ? 1 << (wDenom - 1) : 0) << correction, wDenom + correction, wOffset << (X265AMD_DEPTH - 8));
            if (rc != X265AMD_OK) break;
            for (int t = 0; t < 4; t++) q.d_ref0w[t] = (pixel*)j.dW + (size_t)t * lowPlaneElems + lowOrg;
        }
        bool ok = true;
        const bool own0 = !fenc.lowMvs[j.d0].empty(), own1 = j.d1 > 0 && !fenc.lowMvs1[j.d1].empty();
        void* f0 = j.dMvs; void* c0 = j.dMvc; void* f1 = j.dMvs1; void* c1 = j.dMvc1;
        if (!j.search0) ok = shared((own0 ? fenc.lowMvs : fenc.specMvs)[j.d0], (own0 ? fenc.lowMvc : fenc.specMvc)[j.d0], f0, c0);
        if (ok && j.d1 > 0 && !j.search1) ok = shared((own1 ? fenc.lowMvs1 : fenc.specMvs1)[j.d1], (own1 ? fenc.lowMvc1 : fenc.specMvc1)[j.d1], f1, c1);
        q.d_mvs0 = (int16_t*)f0; q.d_mv_costs0 = (int32_t*)c0; q.d_mvs1 = (int16_t*)f1; q.d_mv_costs1 = (int32_t*)c1;
        if (!ok) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost set-up");
    }
    phase(0);
    /* one launch for all of them (blockIdx.y = the estimate): the device runs as many block rows side by side as it holds */
    if (rc == X265AMD_OK) rc = x265amd_lowres_frame_cost_batch(laStream, me, kj.data(), (int)jobs.size(), lowStride, lowCuW, lowCuH);
    /* the sums over the blocks, on the device too: two numbers per estimate come back instead of its two cost arrays */
    std::vector<int64_t> sums(2 * jobs.size());
    if (rc == X265AMD_OK) rc = x265amd_lowres_cost_sums(laStream, kj.data(), (int)issued, lowCuW, lowCuH, sums.data());
    phase(1);
    for (size_t k = 0; k < issued && rc == X265AMD_OK; k++)
    {
        CostJob& j = jobs[k];
        Pic& fenc = *j.fenc;
        bool ok = true;
        if (ok && j.search0)
        {
            std::vector<int16_t>& mv = (j.spec ? fenc.specMvs : fenc.lowMvs)[j.d0]; std::vector<int32_t>& mc = (j.spec ? fenc.specMvc : fenc.lowMvc)[j.d0];
            mv.resize(ncu * 2); mc.resize(ncu);
            ok = hipMemcpyAsync(mv.data(), j.dMvs, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipMemcpyAsync(mc.data(), j.dMvc, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess;
            if (ok) { laFieldPut(mv.data(), j.dMvs, j.dMvc); j.dMvs = j.dMvc = nullptr; }            /* the search's buffers ARE the field's device copy from now on */
        }
        if (ok && j.search1)
        {
            std::vector<int16_t>& mv = (j.spec ? fenc.specMvs1 : fenc.lowMvs1)[j.d1]; std::vector<int32_t>& mc = (j.spec ? fenc.specMvc1 : fenc.lowMvc1)[j.d1];
            mv.resize(ncu * 2); mc.resize(ncu);
            ok = hipMemcpyAsync(mv.data(), j.dMvs1, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipMemcpyAsync(mc.data(), j.dMvc1, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess;
            if (ok) { laFieldPut(mv.data(), j.dMvs1, j.dMvc1); j.dMvs1 = j.dMvc1 = nullptr; }
        }
        if (!ok) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost");
    }
    phase(2);
    if (hipStreamSynchronize(laStream) != hipSuccess && rc == X265AMD_OK) rc = xa_fail(X265AMD_EHIP, "encoder: lowres frame cost");
    phase(3);
    struct Ph { decltype(phase)& f; ~Ph() { f(4); } } ph_{ phase };
    laFieldsTrim();
    for (size_t k = 0; k < issued; k++)
    {
        CostJob& j = jobs[k];
        /* what the rate control reads of an estimate's blocks stays with the picture: with cuTree Lowres::lowresCosts[d0][d1] (cost capped at 14 bits + the lists used), without
         * it -- Lowres::costEstAq of a P estimate, the rate factor's complexity measure -- the blocks' uncapped costs */
        void* keepLc = rc != X265AMD_OK ? nullptr : (p.cuTree ? j.dLc : (rateCtl && j.d1 == 0 ? j.dBc : nullptr));
        void* bufs[6] = { j.dMvs, j.dMvc, keepLc == j.dLc ? nullptr : j.dLc, keepLc == j.dBc ? nullptr : j.dBc, j.dMvs1, j.dMvc1 };
        for (void* b : bufs) laBufPut(b);
        xa_scratch_free(j.dW);
        j.dMvs = j.dMvc = j.dLc = j.dBc = j.dMvs1 = j.dMvc1 = j.dW = nullptr;
        if (keepLc)
        {
            const int key = j.d0 * 32 + j.d1;
            std::map<int, void*>& m = j.spec ? j.fenc->dSpecLc : j.fenc->dLc;
            j.fenc->dropLc(m, key);
            m[key] = keepLc;
            if (!j.spec) j.fenc->lcHost.erase(key);
        }
        if (rc != X265AMD_OK)
        {
            if (j.search0) { (j.spec ? j.fenc->specMvs : j.fenc->lowMvs)[j.d0].clear(); (j.spec ? j.fenc->specMvc : j.fenc->lowMvc)[j.d0].clear(); }
            if (j.search1) { (j.spec ? j.fenc->specMvs1 : j.fenc->lowMvs1)[j.d1].clear(); (j.spec ? j.fenc->specMvc1 : j.fenc->lowMvc1)[j.d1].clear(); }
            continue;
        }
        int64_t est = sums[2 * k]; const int imb = (int)sums[2 * k + 1];
        if (j.d1 > 0) est = est * 100 / (130 + 0);          /* param.bFrameBias: the default */
        if (j.spec) { j.fenc->specCost2[j.d0][j.d1] = est; if (j.d1 == 0) j.fenc->specIntraMbs[j.d0] = imb; continue; }
        /* a field made for good replaces whatever was made ahead of its time for the same pair, and the estimates that were built on that */
        if (j.search0) { j.fenc->specMvs[j.d0].clear(); j.fenc->specMvc[j.d0].clear(); for (int t = 0; t < 18; t++) { j.fenc->specCost2[j.d0][t] = -1; j.fenc->dropLc(j.fenc->dSpecLc, j.d0 * 32 + t); } }
        if (j.search1) { j.fenc->specMvs1[j.d1].clear(); j.fenc->specMvc1[j.d1].clear(); for (int t = 0; t < 18; t++) { j.fenc->specCost2[t][j.d1] = -1; j.fenc->dropLc(j.fenc->dSpecLc, t * 32 + j.d1); } }
        j.fenc->cost2[j.d0][j.d1] = est;
        if (j.d1 == 0) { j.fenc->costEst[j.d0] = est; j.fenc->intraMbs[j.d0] = imb; }
    }
    return rc;
}
int x265amd_encoder::frameCost(std::vector<Pic*>& frames, int p0, int p1, int b, int64_t& score)
{
    return frameCostAt(*frames[b], *frames[p0], p1 > b ? frames[p1] : nullptr, b - p0, p1 - b, score);
}

/* The B-frame trellis (X265_B_ADAPT_TRELLIS; what Lookahead::slicetypePath / slicetypePathCost compute, slicetype.cpp:3218-3313).  A plan for the first n pictures of
 * the window is the list of its mini-GOPs' B runs (a run of k: k B pictures, then their P picture); planCost prices one -- per mini-GOP the P picture against the
 * mini-GOP's anchor, then its B pictures (with the pyramid: the middle one between anchor and P picture, the ones in front of it between anchor and middle, the ones behind
 * between middle and P picture) -- and gives up once the sum passes `limit`.  Which estimates are asked for, and in what order, is part of the result (an estimate that
 * is asked for exists afterwards: frameCostAt), so the order of the additions and of the limit checks is the reference's. */
int64_t x265amd_encoder::planCost(std::vector<Pic*>& frames, const std::vector<uint8_t>& runs, int64_t limit, int& rc)
{
    int64_t total = 0;
    int anchor = 0;
    for (size_t g = 0; g < runs.size() && rc == X265AMD_OK; g++)
    {
        const int pPic = anchor + runs[g] + 1;
        int64_t c = 0;
        rc = frameCost(frames, anchor, pPic, pPic, c);
        total += c;
        if (total > limit) break;
        if (p.bBPyramid && runs[g] > 1)
        {
            const int middle = anchor + (pPic - anchor) / 2;
            if (rc == X265AMD_OK) { rc = frameCost(frames, anchor, pPic, middle, c); total += c; }
            for (int b = anchor + 1; b < middle && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, anchor, middle, b, c); total += c; }
            for (int b = middle + 1; b < pPic && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, middle, pPic, b, c); total += c; }
        }
        else
            for (int b = anchor + 1; b < pPic && total < limit && rc == X265AMD_OK; b++) { rc = frameCost(frames, anchor, pPic, b, c); total += c; }
        anchor = pPic;
    }
    return total;
}
/* the cheapest plan for the first `length` pictures: the cheapest plan of a shorter prefix with one more mini-GOP behind it, the last run growing from 0; the cheapest so
 * far is the limit of the next one's pricing; the first of equals stays */
void x265amd_encoder::extendPlans(std::vector<Pic*>& frames, int length, std::vector<std::vector<uint8_t> >& plans, int& rc)
{
    const int longest = std::min(p.bframes, length - 1);
    int64_t cheapest = 1LL << 62;
    std::vector<uint8_t> winner;
    for (int run = 0; run <= longest && rc == X265AMD_OK; run++)
    {
        std::vector<uint8_t> plan = plans[length - (run + 1)];
        plan.push_back((uint8_t)run);
        const int64_t cost = planCost(frames, plan, cheapest, rc);
        if (cost < cheapest) { cheapest = cost; winner.swap(plan); }
    }
    plans[length] = winner;
}

/* Lookahead::scenecutInternal (slicetype.cpp:3016-3047): float / double arithmetic as written there */
bool x265amd_encoder::scenecutInternal(std::vector<Pic*>& frames, int p0, int p1, bool real, int& rc)
{
    Pic* frame = frames[p1];
    if (rc == X265AMD_OK) rc = frameCostP(*frame, *frames[p0], p1 - p0);
    if (rc != X265AMD_OK) return false;
    const int64_t icost = frame->costEst[0], pcost = frame->costEst[p1 - p0];
    const int gopSize = (int)(((int64_t)frame->poc - lastKeyframe) % p.keyframeMax);
    const float threshMax = (float)(p.scenecutThreshold / 100.0);
    float threshMin = (float)(threshMax * 0.25);
    double bias = 5.0 / 100;            /* param.scenecutBias: the default, scaled in Encoder::configure (encoder.cpp:3948) */
    if (real)
    {
        if (keyframeMin == p.keyframeMax) threshMin = threshMax;
        if (gopSize <= keyframeMin / 4) bias = threshMin / 4;
        else if (gopSize <= keyframeMin) bias = threshMin * gopSize / keyframeMin;
        else bias = threshMin + (threshMax - threshMin) * (gopSize - keyframeMin) / (p.keyframeMax - keyframeMin);
    }
    return pcost >= (1.0 - bias) * icost;
}

/* Lookahead::scenecut (slicetype.cpp:2921-3014) */
bool x265amd_encoder::scenecut(std::vector<Pic*>& frames, int p0, int p1, bool real, int numFrames, int& rc)
{
    if (real && p.bframes)
    {
        const int origmaxp1 = p0 + 1 + p.bframes, maxp1 = std::min(origmaxp1, numFrames);
        bool fluctuate = false, noScenecuts = false;
        int64_t avgSatdCost = 0;
        if (frames[p0]->costEst[p1 - p0] > -1) avgSatdCost = frames[p0]->costEst[p1 - p0];
        int cnt = 1;
        for (int cp1 = p1; cp1 <= maxp1; cp1++)
        {
            if (!scenecutInternal(frames, p0, cp1, false, rc))
            {
                for (int i = cp1; i > p0; i--) { frames[i]->bScenecut = false; noScenecuts = false; }
            }
            else if (scenecutInternal(frames, cp1 - 1, cp1, false, rc)) { frames[cp1]->bScenecut = true; noScenecuts = true; }
            if (rc != X265AMD_OK) return false;
            avgSatdCost += frames[cp1]->costEst[cp1 - p0];
            cnt++;
        }
        if (noScenecuts)
        {
            fluctuate = false;
            avgSatdCost /= cnt;
            for (int i = p1; i <= maxp1; i++)
            {
                const int64_t curCost = frames[i]->costEst[i - p0], prevCost = frames[i - 1]->costEst[i - 1 - p0];
                if (fabs((double)(curCost - avgSatdCost)) > 0.1 * avgSatdCost || fabs((double)(curCost - prevCost)) > 0.1 * prevCost)
                {
                    fluctuate = true;
                    if (!isSceneTransition && frames[i]->bScenecut)
                    {
                        isSceneTransition = true;
                        for (int j = i + 1; j <= maxp1; j++) frames[j]->bScenecut = false;
                        break;
                    }
                }
                frames[i]->bScenecut = false;
            }
        }
        if (!fluctuate && !noScenecuts) isSceneTransition = false;
    }
    if (!frames[p1]->bScenecut) return false;
    return scenecutInternal(frames, p0, p1, real, rc);
}

/* Lookahead::slicetypeAnalyse(frames, bKeyframe) (slicetype.cpp:2603-2919) without VBV / zones / gop-lookahead: frames[0] = the last non-B picture, frames[1..] = the
 * undecided pictures of the window.  bKeyframe: the pass behind a keyframe's mini-GOP that cuTree adds (slicetype.cpp:2469-2483): the same analysis with the keyframe as
 * frames[0], cuTree down to the keyframe itself, and every type taken back afterwards */
int x265amd_encoder::slicetypeAnalyse(std::vector<Pic*>& frames, bool bKeyframe)
{
    const int maxSearch = std::min(p.lookaheadDepth, 250);
    int framecnt = 0;
    for (; framecnt < maxSearch; framecnt++)
        if (framecnt + 1 >= (int)frames.size() || frames[framecnt + 1]->type != TYPE_AUTO) break;
    if (!framecnt) return p.cuTree ? runCuTree(frames, 0, bKeyframe) : X265AMD_OK;
    frames.resize((size_t)framecnt + 1);
    const int keyFrameLimit = (int)std::min<int64_t>((int64_t)p.keyframeMax + lastKeyframe - frames[0]->poc - 1, INT_MAX / 2), keyintLimit = keyFrameLimit;
    const int origNumFrames = std::min(framecnt, keyintLimit);
    int numFrames = origNumFrames;
    if (p.bOpenGOP && numFrames < framecnt) numFrames++;           /* open GOPs: the window takes in the keyframe (slicetype.cpp:2660-2661) */
    else if (numFrames == 0) { frames[1]->type = TYPE_I; return X265AMD_OK; }
    int rc = X265AMD_OK;
    if (p.bFrameAdaptive == 2 && p.bframes)
    {
        /* m_bBatchMotionSearch (slicetype.cpp:2668-2694; it stays on with a pool of four workers or more): every picture of the window is searched against the pictures
         * 1 .. bframes + 1 before it and, where the window allows, the same distance behind it -- whether or not the trellis below will ask for that pair.  The fields
         * stay with the pictures: the encoder's searches take candidates from them (Search::getLowresMV) */
        std::vector<CostJob> jobs;
        for (int b = 2; b < numFrames; b++)
            for (int i = 1; i <= p.bframes + 1; i++)
            {
                const int p0 = b - i;
                if (p0 < 0 || !frames[b]->lowMvs[i].empty()) continue;
                int p1 = b + i;
                if (p1 >= numFrames || !frames[b]->lowMvs1[i].empty()) p1 = b;
                if (frames[b]->cost2[i][p1 - b] >= 0) continue;
                CostJob j;
                j.fenc = frames[b]; j.ref0 = frames[p0]; j.ref1 = p1 > b ? frames[p1] : nullptr; j.d0 = i; j.d1 = p1 - b; j.whole = true;         /* (batch mode: no cooperative slices, :4004) */
                jobs.push_back(j);
            }
        /* (the first picture of the window is not in the reference's batch: its estimate against the last non-B picture is what the scene-cut check and every path
         * of the trellis start with) */
        if (frames[1]->lowMvs[1].empty()) { CostJob j; j.fenc = frames[1]; j.ref0 = frames[0]; j.d0 = 1; jobs.push_back(j); }
        /* ... nor is the last one, the P picture every path ends with: searched now, side by side with the batch, but AHEAD OF ITS TIME (CostJob::spec) -- the field and the
         * estimate wait in the picture's spec* members until the trellis asks for them, as everything below does */
        auto hasL0 = [](const Pic* f, int d) { return !f->lowMvs[d].empty() || !f->specMvs[d].empty(); };
        auto hasL1 = [](const Pic* f, int d) { return !f->lowMvs1[d].empty() || !f->specMvs1[d].empty(); };
        for (int i = 1; i <= p.bframes + 1 && i <= numFrames && numFrames > 1; i++)
            if (!hasL0(frames[numFrames], i)) { CostJob j; j.fenc = frames[numFrames]; j.ref0 = frames[numFrames - i]; j.d0 = i; j.spec = true; jobs.push_back(j); }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
        /* What the batch leaves to the trellis -- the fields towards pictures behind that it pairs with no distance before (the first picture's; every picture's towards
         * the window's last) -- searched side by side as well instead of one estimate at a time when a path asks.  Whether the reference ever makes one of them depends on
         * the paths it prices and where it gives them up, and with a B pyramid the encoder's pictures reference pictures the trellis did not price them against: so they
         * are made ahead of their time, and only what a path asks for becomes the picture's (frameCostAt). */
        jobs.clear();
        for (int b = 1; b < numFrames; b++)
            for (int jj = 1; jj <= p.bframes; jj++)
            {
                const int p1 = b + jj;
                if (p1 > numFrames) break;
                if (hasL1(frames[b], jj) || !hasL0(frames[b], 1)) continue;
                CostJob j;
                j.fenc = frames[b]; j.ref0 = frames[b - 1]; j.ref1 = frames[p1]; j.d0 = 1; j.d1 = jj; j.spec = true;
                jobs.push_back(j);
            }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
        /* ... and every cost the trellis can ask for of these pictures, side by side.  Those of m_bBatchFrameCosts (:2696-2734: pictures 2 .. numFrames - 1 against fields
         * that exist, the picture behind inside the window; the reference fills them with a pool of more than twelve workers) are the pictures' at once, the rest wait */
        jobs.clear();
        for (int b = 1; b < numFrames; b++)
            for (int i = 1; i <= p.bframes + 1; i++)
            {
                if (b < i || !hasL0(frames[b], i)) continue;
                for (int jj = 0; jj <= p.bframes; jj++)
                {
                    const int p1 = b + jj;
                    if (p1 > numFrames) break;
                    if ((jj && !hasL1(frames[b], jj)) || frames[b]->cost2[i][jj] >= 0 || frames[b]->specCost2[i][jj] >= 0) continue;
                    CostJob j;
                    j.fenc = frames[b]; j.ref0 = frames[b - i]; j.ref1 = jj ? frames[p1] : nullptr; j.d0 = i; j.d1 = jj;
                    j.spec = !(b >= 2 && p1 < numFrames && !frames[b]->lowMvs[i].empty() && (!jj || !frames[b]->lowMvs1[jj].empty()));
                    jobs.push_back(j);
                }
            }
        /* the last picture of the window as a P picture at every distance (the trellis' path ends) */
        for (int i = 1; i <= p.bframes + 1 && i <= numFrames; i++)
            if (hasL0(frames[numFrames], i) && frames[numFrames]->cost2[i][0] < 0 && frames[numFrames]->specCost2[i][0] < 0)
            { CostJob j; j.fenc = frames[numFrames]; j.ref0 = frames[numFrames - i]; j.d0 = i; j.spec = true; jobs.push_back(j); }
        rc = frameCostMany(jobs);
        if (rc != X265AMD_OK) return rc;
    }
    const bool isScenecut = scenecut(frames, 0, 1, true, origNumFrames, rc);       /* (run whatever the threshold: its estimates and marks stay) */
    if (rc != X265AMD_OK) return rc;
    if (p.scenecutThreshold > 0 && isScenecut) { frames[1]->type = TYPE_I; return X265AMD_OK; }
    int resetStart;
    if (p.bframes)
    {
        int numBFrames = std::min(numFrames - 1, p.bframes);
        if (p.bFrameAdaptive == 2)
        {
            /* X265_B_ADAPT_TRELLIS (slicetype.cpp:2776-2795): the cheapest path of P / B decisions through the window */
            numBFrames = 0;
            if (numFrames > 1)
            {
                std::vector<std::vector<uint8_t> > plans((size_t)numFrames + 1);      /* plans[n]: the cheapest plan for the first n pictures; plans[0] is empty, plans[1] one P picture */
                plans[1].push_back(0);
                for (int j = 2; j <= numFrames && rc == X265AMD_OK; j++) extendPlans(frames, j, plans, rc);
                if (rc != X265AMD_OK) return rc;
                const std::vector<uint8_t>& plan = plans[numFrames];
                numBFrames = plan.empty() ? 0 : plan[0];
                int at = 1;
                for (size_t g = 0; g < plan.size(); g++)
                {
                    for (int k = 0; k < plan[g] && at < numFrames; k++) frames[at++]->type = TYPE_B;
                    if (at < numFrames) frames[at++]->type = TYPE_P;
                }
            }
        }
        else if (p.bFrameAdaptive == 1)
        {
            /* X265_B_ADAPT_FAST (slicetype.cpp:2796-2848): pictures in pairs -- two P pictures when half the second one's blocks are intra, a P picture when P P is cheaper than B P,
             * else B pictures for as long as the P picture behind them stays cheap; every estimate made when it is asked for (no batch: slicetype.cpp:1024) */
            const int cuCount = lowBlocks;
            auto cost = [&](int p0, int p1, int b, bool intraPenalty, int64_t& out) -> int {
                int64_t sc = 0;
                const int r = frameCost(frames, p0, p1, b, sc);
                if (r != X265AMD_OK) return r;
                if (intraPenalty) sc += sc * frames[b]->intraMbs[b - p0] / (cuCount * 8);          /* estimateFrameCost's "arbitrary penalty for I-blocks after B-frames" (:4069-4071) */
                out = sc;
                return X265AMD_OK;
            };
            for (int i = 0; i <= numFrames - 2 && rc == X265AMD_OK; )
            {
                int64_t cost1p0 = 0, cost2p0 = 0, cost1b1 = 0, cost2p1 = 0;
                if ((rc = cost(i + 0, i + 2, i + 2, true, cost2p1)) != X265AMD_OK) break;
                if (frames[i + 2]->intraMbs[2] > cuCount / 2) { frames[i + 1]->type = TYPE_P; frames[i + 2]->type = TYPE_P; i += 2; continue; }
                if ((rc = cost(i + 0, i + 2, i + 1, false, cost1b1)) != X265AMD_OK || (rc = cost(i + 0, i + 1, i + 1, false, cost1p0)) != X265AMD_OK ||
                    (rc = cost(i + 1, i + 2, i + 2, false, cost2p0)) != X265AMD_OK) break;
                if (cost1p0 + cost2p0 < cost1b1 + cost2p1) { frames[i + 1]->type = TYPE_P; i += 1; continue; }
                frames[i + 1]->type = TYPE_B;
                int j;
                for (j = i + 2; j <= std::min(i + p.bframes, numFrames - 1); j++)
                {
                    const int64_t pthresh = std::max(300 - (50 - 0) * (j - i - 1), 300 / 10);          /* INTER_THRESH, P_SENS_BIAS with bFrameBias 0 */
                    int64_t pcost = 0;
                    if ((rc = cost(i + 0, j + 1, j + 1, true, pcost)) != X265AMD_OK) break;
                    if (pcost > pthresh * cuCount || frames[j + 1]->intraMbs[j - i + 1] > cuCount / 3) break;
                    frames[j]->type = TYPE_B;
                }
                if (rc != X265AMD_OK) break;
                frames[j]->type = TYPE_P;
                i = j;
            }
            if (rc != X265AMD_OK) return rc;
            frames[numFrames]->type = TYPE_P;
            numBFrames = 0;
            while (numBFrames < numFrames && frames[numBFrames + 1]->type == TYPE_B) numBFrames++;
        }
        else
            for (int j = 1; j < numFrames; j++) frames[j]->type = (j % (numBFrames + 1)) ? TYPE_B : TYPE_P;
        frames[numFrames]->type = TYPE_P;
        int numAnalyzed = numFrames;
        /* Check scenecut on the first minigop. */
        for (int j = 1; j < numBFrames + 1; j++)
        {
            const bool cut = scenecut(frames, j, j + 1, false, origNumFrames, rc);
            if (rc != X265AMD_OK) return rc;
            if (cut) { frames[j]->type = TYPE_P; numAnalyzed = j; break; }
        }
        resetStart = bKeyframe ? 1 : std::min(numBFrames + 2, numAnalyzed + 1);
    }
    else
    {
        for (int j = 1; j <= numFrames; j++) frames[j]->type = TYPE_P;
        resetStart = bKeyframe ? 1 : 2;
    }
    /* cuTree on the window as it is typed now (slicetype.cpp:2893-2894) */
    if (p.cuTree && (rc = runCuTree(frames, std::min(numFrames, p.keyframeMax), bKeyframe)) != X265AMD_OK) return rc;
    for (int64_t j = (int64_t)keyintLimit + 1; j <= numFrames; j += p.keyframeMax) { frames[j]->type = TYPE_I; resetStart = std::min(resetStart, (int)j + 1); }
    const int maxp1 = std::min(p.bframes + 1, origNumFrames);
    /* Restore frame types for all frames that haven't actually been decided yet. */
    for (int j = resetStart; j <= numFrames; j++)
    {
        frames[j]->type = TYPE_AUTO;
        if (j <= maxp1 && frames[j]->bScenecut && isSceneTransition) isSceneTransition = false;
    }
    return X265AMD_OK;
}

/* CostEstimateGroup::singleCost(p0, p1, b) for Lookahead::cuTree (x265amd_cutree's callback): the estimate is made if it does not exist (with the searches its fields
 * lack), its block costs come back from the device once, the fields are the picture's */
namespace { struct TreeCtx { x265amd_encoder* e; std::vector<Pic*>* frames; }; }
int x265amd_encoder::cuTreeEstimate(void* ctx, int p0, int p1, int b, const uint16_t** lc, const int16_t** mvs0, const int16_t** mvs1)
{
    TreeCtx& t = *(TreeCtx*)ctx;
    x265amd_encoder& e = *t.e;
    std::vector<Pic*>& frames = *t.frames;
    if (p0 < 0 || b < p0 || p1 < b || p1 >= (int)frames.size() || b - p0 > 17 || p1 - b > 17 || b == p0) return xa_fail(X265AMD_EINVAL, "encoder: cuTree estimate");
    int64_t score = 0;
    const int rc = e.frameCost(frames, p0, p1, b, score);
    if (rc != X265AMD_OK) return rc;
    Pic& f = *frames[b];
    const int d0 = b - p0, d1 = p1 - b, key = d0 * 32 + d1;
    auto h = f.lcHost.find(key);
    if (h == f.lcHost.end())
    {
        auto d = f.dLc.find(key);
        if (d == f.dLc.end()) return xa_fail(X265AMD_EHIP, "encoder: cuTree: the block costs of an estimate are gone");
        std::vector<uint16_t> v((size_t)e.lowCuW * e.lowCuH);
        if (hipMemcpyAsync(v.data(), d->second, v.size() * 2, hipMemcpyDeviceToHost, e.laStream) != hipSuccess || hipStreamSynchronize(e.laStream) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: cuTree: block costs");
        h = f.lcHost.emplace(key, std::move(v)).first;
    }
    *lc = h->second.data();
    *mvs0 = f.lowMvs[d0].empty() ? nullptr : f.lowMvs[d0].data();
    *mvs1 = d1 > 0 && !f.lowMvs1[d1].empty() ? f.lowMvs1[d1].data() : nullptr;
    return 0;
}
/* Lookahead::cuTree(frames, numframes, bIntra) (slicetype.cpp:3399-3500): host/fm_ratecontrol.cpp on the pictures' host arrays */
int x265amd_encoder::runCuTree(std::vector<Pic*>& frames, int numframes, bool bIntra)
{
    if (numframes >= (int)frames.size()) numframes = (int)frames.size() - 1;
    std::vector<x265amd_cutree_frame> recs(frames.size());
    std::vector<x265amd_cutree_frame*> ptrs(frames.size());
    for (size_t k = 0; k < frames.size(); k++)
    {
        Pic& f = *frames[k];
        if (f.intraCostHost.empty() || f.qpAqOffset.empty()) return xa_fail(X265AMD_EINVAL, "encoder: cuTree: a picture without block offsets");
        x265amd_cutree_frame& r = recs[k];
        memset(&r, 0, sizeof(r));
        r.slice_type = f.type; r.intra_cost = f.intraCostHost.data(); r.inv_qscale = f.invQscale.data(); r.qp_aq_offset = f.qpAqOffset.data();
        r.qp_cutree_offset = f.qpCuTreeOffset.data(); r.propagate_cost = f.propagateCost.data(); r.weighted_cost_delta = nullptr;       /* (Lowres::weightedCostDelta is an integer quotient below one: always 0, slicetype.cpp:964) */
        ptrs[k] = &r;
    }
    TreeCtx ctx{ this, &frames };
    const int rc = x265amd_cutree(&treeParams, ptrs.data(), numframes, bIntra, cuTreeEstimate, &ctx);
    return rc == X265AMD_OK ? X265AMD_OK : xa_fail(rc, "encoder: cuTree");
}

/* The typed mini-GOP input[0 .. b] goes to `ready` in coding order (slicetype.cpp:2372-2376, :2443-2470): with a B pyramid and two B pictures or more the middle one
 * becomes a reference (Lookahead::placeBref: index (0 + b) / 2); the non-B picture first, then the referenced B picture, then the other B pictures in display order */
void x265amd_encoder::pushMiniGop(int b)
{
    if (p.bBPyramid && b > 1) input[b / 2]->type = TYPE_BREF;
    if (b > 0) input[b - 1]->bLastMiniGopBFrame = true;
    ready.push_back(input[b]);
    for (int i = 0; i < b; i++) if (input[i]->type == TYPE_BREF) ready.push_back(input[i]);
    for (int i = 0; i < b; i++) if (input[i]->type != TYPE_BREF) ready.push_back(input[i]);
    input.erase(input.begin(), input.begin() + b + 1);
    first = false;
}

/* Lookahead::slicetypeDecide (slicetype.cpp:1802-2400) as far as the built subset goes: runs when the input queue holds lookaheadDepth pictures (Lookahead::findJob,
 * m_fullQueueSize; one picture is enough once the caller flushes), types the next mini-GOP and moves it to `ready` in coding order.  Returns 0, or an error code. */
int x265amd_encoder::decideLookahead(bool flush, int maxGops)
{
    const int fullQueue = flush ? 1 : std::max(1, p.lookaheadDepth);
    while ((int)input.size() >= fullQueue && maxGops-- > 0)
    {
        const int maxSearch = std::max(1, std::min(p.lookaheadDepth, 250));
        std::vector<Pic*> frames;
        frames.push_back(lastNonB.get());
        for (int j = 0; j < maxSearch && j < (int)input.size(); j++) frames.push_back(input[j].get());
        const int windowCount = (int)frames.size() - 1;         /* the reference's maxSearch: the pictures this decision looks at */
        if (lastNonB)
        {
            const int rc = slicetypeAnalyse(frames);
            if (rc != X265AMD_OK) return rc;
        }
        const int nlist = std::min((int)input.size(), p.bframes + 2);
        int b = 0;
        for (;; b++)
        {
            Pic& frm = *input[b];
            if ((int64_t)frm.poc - lastKeyframe >= p.keyframeMax && (frm.type == TYPE_AUTO || frm.type == TYPE_I)) frm.type = p.bOpenGOP && haveKeyframe ? TYPE_I : TYPE_IDR;
            if (frm.type == TYPE_I && (int64_t)frm.poc - lastKeyframe >= keyframeMin)
            {
                /* closed GOPs: a keyframe is an IDR picture; open GOPs: it stays an I picture (CRA) and the B pictures in front of it stay (slicetype.cpp:1985-1994) */
                if (p.bOpenGOP) { lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true; }
                else frm.type = TYPE_IDR;
            }
            if (frm.type == TYPE_IDR)
            {
                lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true;
                if (b > 0) { input[b - 1]->type = TYPE_P; b--; }
            }
            Pic& cur = *input[b];          /* (after the step back the tests below see the keyframe in the reference: they do nothing for it; the loop ends at the P picture) */
            if (&cur == &frm)
            {
                if (b == p.bframes || b + 1 >= nlist) { if (frm.type == TYPE_AUTO || frm.type == TYPE_B) frm.type = TYPE_P; }
                if (frm.type == TYPE_AUTO) frm.type = TYPE_B;
                else if (frm.type != TYPE_B) break;
            }
            else break;
        }
        if (p.rateControlMode == X265AMD_RC_CRF)
        {
            /* "calculate the frame costs ahead of time for estimateFrameCost while we still have lowres" (slicetype.cpp:2396-2440): the estimate of every picture of the
             * mini-GOP against the pictures it will reference -- made now if the decision above did not need it, with the searches its fields lack (the encoder's searches
             * take candidates from those fields, and cuTree and the rate control read the estimates) */
            if (p.bBPyramid && b > 1) input[b / 2]->type = TYPE_BREF;          /* placeBref comes first (:2386-2389) */
            std::vector<Pic*> est;
            est.push_back(lastNonB.get());
            for (int i = 0; i <= b; i++) est.push_back(input[i].get());
            int64_t score = 0;
            const int p1n = b + 1;
            const bool isI = est[p1n]->type == TYPE_I || est[p1n]->type == TYPE_IDR;
            if (!isI && est[0])
            {
                const int rc = frameCost(est, 0, p1n, p1n, score);
                if (rc != X265AMD_OK) return rc;
            }
            if (b && est[0])
            {
                int p0 = 0;
                bool isp0available = est[p1n]->type != TYPE_IDR;
                for (int bb = 1; bb <= b; bb++)
                {
                    if (!isp0available) p0 = bb;
                    int p1;
                    if (est[bb]->type == TYPE_B) for (p1 = bb; est[p1]->type == TYPE_B; p1++) { }
                    else p1 = b + 1;
                    if (p0 != bb)
                    {
                        const int rc = frameCost(est, p0, p1, bb, score);
                        if (rc != X265AMD_OK) return rc;
                    }
                    if (est[bb]->type == TYPE_BREF) { p0 = bb; isp0available = true; }
                }
            }
        }
        lastNonB = input[b];
        pushMiniGop(b);
        if (p.cuTree && (lastNonB->type == TYPE_I || lastNonB->type == TYPE_IDR))
        {
            /* the keyframe's own pass (slicetype.cpp:2469-2483): the pictures of the window that are left, behind the keyframe */
            std::vector<Pic*> kf;
            kf.push_back(lastNonB.get());
            const int left = windowCount - (b + 1);
            for (int j = 0; j < left && j < (int)input.size(); j++) kf.push_back(input[j].get());
            const int rc = slicetypeAnalyse(kf, true);
            if (rc != X265AMD_OK) return rc;
        }
    }
    return X265AMD_OK;
}

/* Lookahead::slicetypeDecide with bFrameAdaptive 0 and no scenecut (slicetype.cpp:1929-2040): the next mini-GOP, moved to `ready` in coding order */
void x265amd_encoder::decideMiniGop(bool flush)
{
    while (!input.empty())
    {
        if (!flush && (int)input.size() < p.bframes + 1 && !(first && !input.empty())) return;
        int b = 0;
        for (;; b++)
        {
            Pic& frm = *input[b];
            if ((int64_t)frm.poc - lastKeyframe >= p.keyframeMax) frm.type = p.bOpenGOP && haveKeyframe ? TYPE_I : TYPE_IDR;
            if (frm.type == TYPE_I)
            {
                /* open GOP: the keyframe is an I picture (CRA) that ends the mini-GOP; the B pictures in front of it stay and reference across it */
                lastKeyframe = frm.poc; frm.bKeyframe = true;
                break;
            }
            if (frm.type == TYPE_IDR)
            {
                /* closed GOP: the frame before a keyframe becomes P and ends the mini-GOP; the keyframe opens the next one */
                if (b > 0) { input[b - 1]->type = TYPE_P; b--; break; }
                lastKeyframe = frm.poc; frm.bKeyframe = true; haveKeyframe = true;
                break;
            }
            if (b == p.bframes || b + 1 >= (int)input.size()) { frm.type = TYPE_P; break; }
            frm.type = TYPE_B;
        }
        pushMiniGop(b);
        if (!flush) return;
    }
}

