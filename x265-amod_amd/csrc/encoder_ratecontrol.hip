/* The encoder object's DPB and rate control for the next picture in coding order (encoder_impl.h): DPB::prepareEncode, RateControl::rateControlStart (constant QP or the
 * constant rate factor: host/fm_ratecontrol.cpp), Lookahead::getEstimatedPictureCost, the QPs of the picture's quantisation groups (reference: source/encoder/dpb.cpp,
 * ratecontrol.cpp:1334-1643, slicetype.cpp:1327-1439, analysis.cpp:3634-3714). */
#include "encoder_impl.h"

/* DPB::prepareEncode for the next picture in coding order (main thread): NAL type, reference picture set, reference lists, slice QP */
int x265amd_encoder::prepare(const PicP& picp)
{
    Pic& pic = *picp;
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    /* DPB::getNalUnitType (dpb.cpp:486-506): IDR_N_LP 20; a keyframe of an open GOP CRA 21; pictures in front of the last CRA picture in output order RASL 9 / 8,
     * in front of the last IDR picture RADL 7 / 6; the rest TRAIL 1 / 0 (the second number: B pictures, which nobody references, prepareEncode dpb.cpp:156-172) */
    int nal;
    if (pic.type == TYPE_IDR) nal = 20;
    else if (pic.bKeyframe && p.bOpenGOP) nal = 21;
    else if (pocCRA && pic.poc < pocCRA) nal = 9;
    else if (lastIDR && pic.poc < lastIDR) nal = 7;
    else nal = 1;
    if (pic.type == TYPE_B && nal < 16) nal--;
    pic.nalType = nal;
    pic.rpsUsed = !(nal >= 16 && nal <= 23);
    if (nal == 20) lastIDR = pic.poc;
    pic.lastIDR = lastIDR;
    pic.hasReferences = pic.type != TYPE_B;
    /* recycleUnreferenced: pictures nobody references leave the list */
    picList.erase(std::remove_if(picList.begin(), picList.end(), [](const PicP& q) { return !q->hasReferences; }), picList.end());
    /* decodingRefreshMarking (dpb.cpp:357-399): an IDR picture empties the buffer; after a CRA picture the first picture behind it in output order does, keeping the CRA picture */
    if (nal == 20) { for (auto& q : picList) q->hasReferences = false; }
    else
    {
        if (refreshPending && pic.poc > pocCRA)
        {
            for (auto& q : picList) if (q->poc != pocCRA) q->hasReferences = false;
            refreshPending = false;
        }
        if (nal == 21) { refreshPending = true; pocCRA = pic.poc; }
    }
    std::vector<PicP> rps;                                                                              /* computeRPS */
    for (auto& q : picList)
    {
        if ((int)rps.size() >= maxDecPicBuffering - 1) break;
        if (q->poc != pic.poc && q->hasReferences && (lastIDR >= pic.poc || lastIDR <= q->poc)) rps.push_back(q);
    }
    for (auto& q : picList)                                                                             /* applyReferencePictureSet */
        if (q->hasReferences && std::find(rps.begin(), rps.end(), q) == rps.end()) q->hasReferences = false;
    pic.neg.clear(); pic.pos.clear(); pic.lists[0].clear(); pic.lists[1].clear();
    for (const PicP& q : rps) (q->poc < pic.poc ? pic.neg : pic.pos).push_back(q);
    std::sort(pic.neg.begin(), pic.neg.end(), [](const PicP& a, const PicP& b) { return a->poc > b->poc; });           /* RPS::sortDeltaPOC */
    std::sort(pic.pos.begin(), pic.pos.end(), [](const PicP& a, const PicP& b) { return a->poc < b->poc; });
    if (stype == 2) statPictures[0]++;
    if (stype != 2)
    {
        const int n0 = std::min(std::max(1, (int)pic.neg.size()), p.maxNumReferences), n1 = stype == 0 ? std::min(p.bBPyramid ? 2 : 1, (int)pic.pos.size()) : 0;       /* dpb.cpp:269-273 */
        std::vector<PicP> l0(pic.neg), l1(pic.pos);
        l0.insert(l0.end(), pic.pos.begin(), pic.pos.end()); l1.insert(l1.end(), pic.neg.begin(), pic.neg.end());
        if ((int)l0.size() < n0 || (int)l1.size() < n1 || (stype == 0 && !n1)) return xa_fail(X265AMD_EINVAL, "encoder_encode: reference lists");
        pic.lists[0].assign(l0.begin(), l0.begin() + n0); pic.lists[1].assign(l1.begin(), l1.begin() + n1);
        {
            /* x265amd_encoder_stats: the distinct reference pictures this picture reads (SURVEY section 8d's R) */
            std::vector<const Pic*> seen;
            for (int l = 0; l < 2; l++) for (const PicP& q : pic.lists[l]) if (std::find(seen.begin(), seen.end(), q.get()) == seen.end()) seen.push_back(q.get());
            statPictures[stype == 1 ? 1 : 2]++; statReferences += seen.size();
        }
    }
    static const char* const dbgWp = getenv("X265AMD_WP_DEBUG");
    memset(pic.wp, 0, sizeof(pic.wp)); pic.weighted = false;
    if (((p.bEnableWeightedPred && stype == 1) || (p.bEnableWeightedBiPred && stype == 0)) && !(dbgWp && strchr(dbgWp, 'p')))
    {
        /* FrameEncoder::compressFrame (frameencoder.cpp:553-582): weightAnalyse for P slices with --weightp, for B slices with --weightb */
        const int rcw = sliceWeights(pic);
        if (rcw != X265AMD_OK) return rcw;
    }
    if (rateCtl)
    {
        /* RateControl::rateControlStart, constant rate factor (coding order is the reference's m_startEndOrder; nothing it reads depends on how a picture was coded) */
        x265amd_rc_frame f;
        memset(&f, 0, sizeof(f));
        f.slice_type = stype; f.is_referenced = pic.type != TYPE_B; f.poc = pic.poc; f.scenecut = pic.bScenecut; f.last_minigop_b = pic.bLastMiniGopBFrame;
        if (stype != 2) f.ref0_scenecut = pic.lists[0][0]->bScenecut;
        f.satd_cost = estimatedPictureCost(pic);
        if (stype == 0)
            for (int l = 0; l < 2; l++)
            {
                const Pic& q = *pic.lists[l][0];
                f.ref_slice_type[l] = isBType(q.type) ? 0 : q.type == TYPE_P ? 1 : 2; f.ref_poc[l] = q.poc; f.ref_is_referenced[l] = q.type != TYPE_B; f.ref_avg_qp_rc[l] = q.avgQpRc;
            }
        const int qp = x265amd_rc_start(rateCtl, &f, &pic.avgQpRc);
        if (qp < 0) return xa_fail(X265AMD_EINVAL, "encoder_encode: rate control");
        pic.sliceQp = std::min(qp, 51);             /* FrameEncoder::compressFrame clips the slice QP to the range the syntax carries (frameencoder.cpp:612) */
        if (useDqp) cuQpTable(pic);
        static const bool rcLog = getenv("X265AMD_RC_LOG") != nullptr;
        if (rcLog) fprintf(stderr, "x265amd rc: poc %d type %d qp %d avgQpRc %.9f satd %lld scenecut %d\n", pic.poc, pic.type, pic.sliceQp, pic.avgQpRc, (long long)f.satd_cost, (int)pic.bScenecut);
    }
    else
    pic.sliceQp = pic.type == TYPE_BREF ? (qpConstant[0] + qpConstant[1]) / 2 : qpConstant[stype];                    /* rateControlStart, CQP (ratecontrol.cpp:1594-1597: a referenced B picture lies between B and P) */
    picList.insert(picList.begin(), picp);              /* PicList::pushFront */
    if (frameParallel)
    {
        /* what pictures coded beside this one read of it exists before any task starts: the maps (rows become valid as they are coded) and the POC lists */
        const size_t nUnits = (size_t)w4 * h4;
        pic.units.assign(nUnits, x265amd_cu_unit()); pic.motion.assign(nUnits, x265amd_mv_unit());
        memset(pic.units.data(), 0, sizeof(x265amd_cu_unit) * nUnits); memset(pic.motion.data(), 0, sizeof(x265amd_mv_unit) * nUnits);
        pic.registerMotion();
        memset(pic.refPoc, 0, sizeof(pic.refPoc));
        for (int l = 0; l < 2; l++)
            for (size_t r = 0; r < pic.lists[l].size(); r++) pic.refPoc[l][r] = pic.lists[l][r]->poc;
        if (p.limitTU >= 3) pic.tuRecs.assign((size_t)nctu * 21, -1);
        pic.finalX.resize(ctuH);
        for (int r = 0; r < ctuH; r++) pic.finalX[r] = xa_counter_alloc();
        pic.analysedCols.assign(ctuH, 0);
    }
    return 0;
}

/* Lookahead::getEstimatedPictureCost (slicetype.cpp:1327-1439) as far as the constant rate factor reads it (Lowres::satdCost; with cuTree RateControl only asks whether it is
 * zero, without it the value is the rate factor's complexity measure): I and P pictures -- with cuTree the estimate's block costs rescaled by the cuTree offsets
 * (frameCostRecalculate), without it the estimate weighted by the adaptive quantisation's factors (costEstAq), the plain estimate without either; B pictures, whose QP does not
 * read it, get the plain estimate */
int64_t x265amd_encoder::estimatedPictureCost(Pic& pic)
{
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    const size_t ncu = (size_t)lowCuW * lowCuH;
    const bool all = lowCuW <= 2 || lowCuH <= 2;
    auto inner = [&](size_t i) { const int x = (int)(i % lowCuW), y = (int)(i / lowCuW); return all || (x > 0 && x < lowCuW - 1 && y > 0 && y < lowCuH - 1); };
    if (stype == 2)
    {
        if (pic.intraCostHost.size() != ncu) return std::max<int64_t>(pic.costEst[0], 1);
        if (p.cuTree)
        {
            std::vector<uint16_t> lc(ncu);
            for (size_t i = 0; i < ncu; i++) lc[i] = (uint16_t)std::min(pic.intraCostHost[i], (1 << 14) - 1);        /* lowresIntraEstimate: lowresCosts[0][0] (slicetype.cpp:806) */
            return x265amd_frame_cost_recalculate(&treeParams, lc.data(), pic.qpCuTreeOffset.data());
        }
        /* Lowres::costEstAq[0][0] (slicetype.cpp:809-823) */
        int64_t sum = 0;
        for (size_t i = 0; i < ncu; i++) if (inner(i)) sum += pic.invQscale.size() == ncu ? (pic.intraCostHost[i] * pic.invQscale[i] + 128) >> 8 : pic.intraCostHost[i];
        return sum;
    }
    const int d0 = pic.poc - pic.lists[0][0]->poc;
    if (stype == 1 && d0 > 0 && d0 < 18)
    {
        const int key = d0 * 32;
        auto d = pic.dLc.find(key);
        if (p.cuTree)
        {
            auto h = pic.lcHost.find(key);
            if (h == pic.lcHost.end() && d != pic.dLc.end())
            {
                std::vector<uint16_t> v(ncu);
                if (hipMemcpyAsync(v.data(), d->second, ncu * 2, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipStreamSynchronize(laStream) == hipSuccess)
                    h = pic.lcHost.emplace(key, std::move(v)).first;
            }
            if (h != pic.lcHost.end()) return x265amd_frame_cost_recalculate(&treeParams, h->second.data(), pic.qpCuTreeOffset.data());
        }
        else if (d != pic.dLc.end() && pic.invQscale.size() == ncu)
        {
            /* Lowres::costEstAq[d0][0] (estimateCUCost, slicetype.cpp:4218-4233): the blocks' costs weighted by the adaptive quantisation's factors */
            std::vector<int32_t> bc(ncu);
            if (hipMemcpyAsync(bc.data(), d->second, ncu * 4, hipMemcpyDeviceToHost, laStream) == hipSuccess && hipStreamSynchronize(laStream) == hipSuccess)
            {
                int64_t sum = 0;
                for (size_t i = 0; i < ncu; i++) if (inner(i)) sum += (bc[i] * pic.invQscale[i] + 128) >> 8;
                return sum;
            }
        }
    }
    if (d0 > 0 && d0 < 18 && pic.costEst[d0] > 0) return pic.costEst[d0];
    return 1;
}

/* Analysis::calculateQpforCuSize (analysis.cpp:3634-3714) for every quantisation group of the picture, ahead of the analysis: per CTU the QP of the 64x64 CU, then (qgSize 32)
 * of its four 32x32 CUs in z order -- values up to 69 (what setLambdaFromQP is given; it clips the QP that is coded to 51).  A referenced picture with cuTree takes the cuTree
 * offsets, every other picture the adaptive quantisation's. */
void x265amd_encoder::cuQpTable(Pic& pic)
{
    const int per = maxCuDqpDepth >= 1 ? 5 : 1;
    pic.cuQp.assign((size_t)nctu * per, (int8_t)pic.sliceQp);
    const double* offs = (p.cuTree && pic.type != TYPE_B) ? pic.qpCuTreeOffset.data() : pic.qpAqOffset.data();
    for (int a = 0; a < nctu; a++)
    {
        const int x = (a % ctuW) * 64, y = (a / ctuW) * 64;
        pic.cuQp[(size_t)a * per] = (int8_t)x265amd_cu_qp(pic.avgQpRc, offs, W, H, x, y, 64, p.qpMin, p.qpMax);
        for (int q = 0; q < 4 && per == 5; q++)
        {
            const int cx = x + (q & 1) * 32, cy = y + (q >> 1) * 32;
            if (cx < W && cy < H) pic.cuQp[(size_t)a * per + 1 + q] = (int8_t)x265amd_cu_qp(pic.avgQpRc, offs, W, H, cx, cy, 32, p.qpMin, p.qpMax);
        }
    }
}

