/* FrameEncoder::compressFrame for one picture of the encoder object (encoder_impl.h): the picture's context (reference lists, weighted copies, slice and analysis parameters),
 * the row gates of pictures coded in parallel, the in-loop filters by rows and by columns, the slice NAL unit (reference: source/encoder/frameencoder.cpp:470-1130, :880-960,
 * framefilter.cpp:559-664, reference.cpp:51-185). */
#include "encoder_impl.h"

/* what the analysis and the slice header of one picture need, derived from the picture's lists (DPB::prepareEncode has run) */
/* MotionReference with weights (reference.cpp:51-185): the weighted copy of a reference picture that the motion searches of ONE slice read (luma; chroma too when the
 * sub-sample refinement measures chroma, subme > 2), made CTU row by CTU row as the reference picture's rows become final (applyWeight).  The copy is the pointwise
 * weight_pp_c of the padded plane: the margins of the reconstruction repeat its edge samples, so weighting them is what extending the weighted rows gives. */
struct WPlane
{
    Pic* src = nullptr; x265amd_weight w[3]; pixel* buf = nullptr; bool chroma[3] = { false, false, false };
    std::mutex mu; std::vector<uint8_t> rowDone; hipStream_t st = nullptr;
    ~WPlane() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } xa_scratch_free(buf); }
};
struct FrameCtx
{
    int stype = 0;
    std::vector<uint64_t> planes;           /* reference pictures (distinct), then the weighted copies of this slice (wplanes), then the reconstruction, then the source: 3 addresses each */
    int numRefs = 0;
    std::vector<std::unique_ptr<WPlane>> wplanes;
    int32_t mePic[2][16];
    x265amd_mvpred_info info;
    x265amd_inter_search_params sp;
    x265amd_slice_info si;
    x265amd_analysis_params ap;
    const Pic* colPic = nullptr;
    bool failed = false;
};

static void frameContext(const x265amd_encoder& e, Pic& pic, FrameCtx& c)
{
    const x265amd_param& p = e.p;
    const int stype = isBType(pic.type) ? 0 : pic.type == TYPE_P ? 1 : 2;
    c.stype = stype;
    const std::vector<PicP>* lists = pic.lists;
    std::vector<Pic*> index;
    int32_t refPic[2][16];
    memset(refPic, 0, sizeof(refPic));
    if (!e.frameParallel) memset(pic.refPoc, 0, sizeof(pic.refPoc));        /* coded in parallel: set by prepare(), other pictures' tasks may be reading it */
    for (int l = 0; l < 2; l++)
        for (size_t r = 0; r < lists[l].size(); r++)
        {
            Pic* q = lists[l][r].get();
            size_t k = std::find(index.begin(), index.end(), q) - index.begin();
            if (k == index.size()) { index.push_back(q); for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(q->finalPlanes(), cc)); }
            refPic[l][r] = (int32_t)k;
            if (!e.frameParallel) pic.refPoc[l][r] = q->poc;
        }
    c.numRefs = (int)index.size();
    memcpy(c.mePic, refPic, sizeof(c.mePic));
    if (pic.weighted)
        for (int l = 0; l < 2; l++)
            for (size_t r = 0; r < lists[l].size(); r++)
            {
                if (!pic.wp[l][r][0].present) continue;         /* MotionReference::init is given weights only when the luma weight is there (frameencoder.cpp:573-577) */
                std::unique_ptr<WPlane> wpl(new WPlane);
                wpl->src = lists[l][r].get();
                memcpy(wpl->w, pic.wp[l][r], sizeof(wpl->w));
                wpl->rowDone.assign(e.ctuH, 0);
                if (xa_scratch_alloc((void**)&wpl->buf, e.picElems * sizeof(pixel)) != hipSuccess || hipStreamCreateWithFlags(&wpl->st, hipStreamNonBlocking) != hipSuccess) { c.failed = true; return; }
                wpl->chroma[0] = true;
                for (int cc = 1; cc < 3; cc++) wpl->chroma[cc] = p.subpelRefine > 2 && pic.wp[l][r][cc].present;     /* numInterpPlanes (reference.cpp:56) */
                c.mePic[l][r] = (int32_t)(c.planes.size() / 3);
                for (int cc = 0; cc < 3; cc++) c.planes.push_back(wpl->chroma[cc] ? e.planeAddr(wpl->buf, cc) : e.planeAddr(wpl->src->finalPlanes(), cc));
                c.wplanes.push_back(std::move(wpl));
            }
    for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(pic.dRec, cc));
    for (int cc = 0; cc < 3; cc++) c.planes.push_back(e.planeAddr(pic.dSrc, cc));

    x265amd_mvpred_info& info = c.info;
    memset(&info, 0, sizeof(info));
    info.pic_width = e.W; info.pic_height = e.H; info.is_inter_b = stype == 0; info.max_num_merge_cand = p.maxNumMergeCand;
    info.num_ref_idx[0] = (int32_t)lists[0].size(); info.num_ref_idx[1] = (int32_t)lists[1].size();
    info.temporal_mvp = p.bEnableTemporalMvp != 0; info.col_from_l0 = stype != 0; info.check_ldc = stype != 0; info.poc = pic.poc;
    memcpy(info.ref_poc, pic.refPoc, sizeof(info.ref_poc));
    c.colPic = stype == 2 ? nullptr : (stype == 1 ? lists[0][0].get() : lists[1][0].get());
    if (c.colPic) { info.col_poc = c.colPic->poc; memcpy(info.col_ref_poc, c.colPic->refPoc, sizeof(info.col_ref_poc)); }

    x265amd_inter_search_params& sp = c.sp;
    memset(&sp, 0, sizeof(sp));
    sp.search_method = p.searchMethod; sp.subpel_refine = p.subpelRefine; sp.search_range = p.searchRange; sp.qp = pic.sliceQp; sp.chroma_mc = 1;
    sp.frame_parallel = e.frameParallel;
    memcpy(sp.ref_pic, refPic, sizeof(sp.ref_pic));
    memcpy(sp.me_pic, c.mePic, sizeof(sp.me_pic));
    if (pic.weighted) { sp.weighted = stype == 1 ? 1 : 2; memcpy(sp.wp, pic.wp, sizeof(sp.wp)); }
    sp.lowres_blocks_in_row = e.lowCuW;
    for (int l = 0; l < 2; l++)
        for (size_t r = 0; r < lists[l].size(); r++)
        {
            const int diffPoc = abs(pic.poc - lists[l][r]->poc);
            const std::vector<int16_t>& f = l ? pic.lowMvs1[diffPoc < 18 ? diffPoc : 0] : pic.lowMvs[diffPoc < 18 ? diffPoc : 0];
            if (diffPoc <= p.bframes + 1 && diffPoc < 18 && !f.empty()) sp.lowres_mvs[l][r] = (uint64_t)(uintptr_t)f.data();
        }

    x265amd_slice_info& si = c.si;
    memset(&si, 0, sizeof(si));
    si.pic_width = e.W; si.pic_height = e.H; si.slice_type = stype; si.slice_qp = pic.sliceQp;
    si.num_ref_idx[0] = info.num_ref_idx[0]; si.num_ref_idx[1] = info.num_ref_idx[1];
    si.max_num_merge_cand = p.maxNumMergeCand; si.sign_hide = p.bEnableSignHiding != 0; si.wpp = p.bEnableWavefront != 0;
    si.use_dqp = e.useDqp; si.max_cu_dqp_depth = e.maxCuDqpDepth;
    si.max_cu_depth = 3; si.max_amp_depth = p.bEnableAMP ? 3 : 0; si.tu_log2_min = 2; si.tu_log2_max = 5;
    si.tu_max_depth_inter = p.tuQTMaxInterDepth; si.tu_max_depth_intra = p.tuQTMaxIntraDepth;

    x265amd_analysis_params& ap = c.ap;
    memset(&ap, 0, sizeof(ap));
    ap.psy_rd = p.psyRd; ap.rd_level = p.rdLevel; ap.early_skip = p.bEnableEarlySkip != 0; ap.rskip = p.recursionSkipMode; ap.limit_refs = p.limitReferences;
    ap.b_intra = p.bIntraInBFrames != 0; ap.rect = p.bEnableRectInter != 0; ap.amp = p.bEnableAMP != 0; ap.limit_modes = p.limitModes != 0;
    ap.strong_intra_smoothing = p.bEnableStrongIntraSmoothing != 0; ap.use_sao = p.bEnableSAO != 0;
    ap.fast_intra = p.bEnableFastIntra != 0; ap.limit_tu = p.limitTU;
    ap.rdoq_level = p.rdoqLevel; ap.psy_rdoq_scale = p.rdoqLevel ? p.psyRdoqFix8 : 0;      /* encoder.cpp:3667: no psy-rdoq without RDOQ */
}

/* slice header (Entropy::codeSliceHeader inputs as DPB / Encoder set them) + the sub-streams -> the picture's NAL unit */
static int sliceNal(const x265amd_encoder& e, Pic& pic, const FrameCtx& c, const int32_t* saoFlags, const std::vector<uint8_t>& data, const std::vector<uint32_t>& sizes, int nsub)
{
    const x265amd_param& p = e.p;
    x265amd_slice_header h;
    memset(&h, 0, sizeof(h));
    h.nal_unit_type = pic.nalType; h.temporal_id_plus1 = 1; h.first_in_access_unit = 1;
    h.slice_type = c.stype; h.poc = pic.poc; h.last_idr_poc = pic.lastIDR; h.log2_max_poc_lsb = 8; h.rps_idx = -1; h.num_rps_in_sps = 0;
    h.num_negative = (int32_t)pic.neg.size(); h.num_positive = (int32_t)pic.pos.size();
    {
        int j = 0;
        for (const PicP& q : pic.neg) { h.delta_poc[j] = q->poc - pic.poc; h.used[j++] = pic.rpsUsed; }
        for (const PicP& q : pic.pos) { h.delta_poc[j] = q->poc - pic.poc; h.used[j++] = pic.rpsUsed; }
    }
    h.temporal_mvp_enabled = p.bEnableTemporalMvp != 0;
    h.use_sao = p.bEnableSAO != 0; h.sao_luma = saoFlags[0]; h.sao_chroma = saoFlags[1];
    h.num_ref_idx[0] = c.info.num_ref_idx[0]; h.num_ref_idx[1] = c.info.num_ref_idx[1]; h.num_ref_idx_default[0] = h.num_ref_idx_default[1] = 1;
    h.col_from_l0 = c.stype != 0; h.col_ref_idx = 0; h.max_num_merge_cand = p.maxNumMergeCand;
    h.slice_qp = pic.sliceQp; h.pps_init_qp = 26; h.deblocking_disabled = !p.bEnableLoopFilter;
    h.slfase_flag = (0x5f4e4a53u >> (pic.poc % 31)) & 1;                                              /* SLFASE_CONSTANT (dpb.cpp:294) */
    h.wpp = p.bEnableWavefront != 0;
    h.weighted_pred = p.bEnableWeightedPred != 0; h.weighted_bipred = p.bEnableWeightedBiPred != 0; h.luma_log2_weight_denom = pic.lumaDenom; h.chroma_log2_weight_denom = pic.chromaDenom;
    memcpy(h.wp, pic.wp, sizeof(h.wp));
    size_t dataBytes = 0;
    for (int s = 0; s < nsub; s++) dataBytes += sizes[s];
    pic.nalBytes.assign(dataBytes * 3 / 2 + 4096, 0);
    const size_t n = x265amd_write_slice_nal(&h, data.data(), sizes.data(), nsub, pic.nalBytes.data(), pic.nalBytes.size());
    if (!n || n > pic.nalBytes.size()) return xa_fail(X265AMD_EINVAL, "encoder: slice NAL");
    pic.nalBytes.resize(n);
    return 0;
}

/* FrameEncoder::compressFrame for one picture (its own thread and HIP stream).  The analysis starts when every reference picture is final; the
 * in-loop filters, SAO (its decision carries state from picture to picture, SAO::m_depthSaoRate) and the shared filter scratch run in coding
 * order, i.e. after the previous picture's task. */
int x265amd_encoder::runFrame(const PicP& picp, std::shared_future<int> prev)
{
    Pic& pic = *picp;
    for (int l = 0; l < 2; l++)
        for (const PicP& q : pic.lists[l]) if (q->done.valid() && q->done.get() != 0) return X265AMD_EHIP;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const std::vector<PicP>* lists = pic.lists;
    FrameCtx fc;
    frameContext(*this, pic, fc);
    if (fc.failed) return xa_fail(X265AMD_EHIP, "encoder: weighted reference planes");
    for (auto& wpl : fc.wplanes)            /* one picture at a time: the reference pictures are complete, so are their weighted copies */
        if (weightRows(*wpl, 0, ctuH - 1) != X265AMD_OK) return X265AMD_EHIP;
    const int stype = fc.stype;
    std::vector<uint64_t>& planes = fc.planes;
    x265amd_mvpred_info& info = fc.info;
    x265amd_inter_search_params& sp = fc.sp;
    x265amd_slice_info& si = fc.si;
    x265amd_analysis_params& ap = fc.ap;
    const Pic* colPic = fc.colPic;
    (void)stype;

    const size_t nUnits = (size_t)w4 * h4;
    pic.units.assign(nUnits, x265amd_cu_unit()); pic.motion.assign(nUnits, x265amd_mv_unit());
    memset(pic.units.data(), 0, sizeof(x265amd_cu_unit) * nUnits); memset(pic.motion.data(), 0, sizeof(x265amd_mv_unit) * nUnits);
    std::vector<x265amd_mv_unit> noCol;
    if (!colPic) { noCol.resize(nUnits); memset(noCol.data(), 0, sizeof(x265amd_mv_unit) * nUnits); }
    std::vector<uint8_t> refDepth(2 * nUnits, 0);
    std::vector<int8_t> refQp0(2 * (size_t)nctu, 0);
    for (int l = 0; l < 2; l++)
        if (!lists[l].empty())
        {
            const Pic* q = lists[l][0].get();
            for (size_t i = 0; i < nUnits; i++) refDepth[l * nUnits + i] = q->units[i].depth;
            for (int i = 0; i < nctu; i++) refQp0[(size_t)l * nctu + i] = useDqp ? q->units[(size_t)(i / ctuW) * 16 * w4 + (size_t)(i % ctuW) * 16].qp : (int8_t)q->sliceQp;      /* CUData::m_qp[0] of the co-located CTU */
        }
    std::vector<x265amd_cu_stat> stat((size_t)nctu + 1);
    memset(stat.data(), 0, sizeof(x265amd_cu_stat) * stat.size());
    std::vector<int16_t> coeff((size_t)nctu * RD_TILE_ELEMS, 0);
    std::vector<uint8_t> data((size_t)W * H * 3 + (1u << 16));
    std::vector<uint32_t> sizes((size_t)ctuH + 1, 0);
    int nsub = 0;
    const bool sao = p.bEnableSAO != 0;
    if (useDqp && pic.cuQp.empty()) return xa_fail(X265AMD_EINVAL, "encoder: the picture has no CU QPs");
    XaTuRecs tuRecs = { nullptr, { nullptr, nullptr } };
    if (p.limitTU >= 3)
    {
        pic.tuRecs.assign((size_t)nctu * 21, -1);
        tuRecs.cur = pic.tuRecs.data();
        for (int l = 0; l < 2; l++) if (!pic.lists[l].empty() && pic.lists[l][0]->tuRecs.size() == pic.tuRecs.size()) tuRecs.ref[l] = pic.lists[l][0]->tuRecs.data();
    }
    int rc = xa_analyse_frame(me, st, &info, &sp, &si, &ap, pic.units.data(), pic.motion.data(), colPic ? colPic->motion.data() : noCol.data(),
                              refDepth.data(), refQp0.data(), planes.data(), (int)(planes.size() / 3), stride, cstride, stat.data(), coeff.data(), nullptr,
                              sao ? nullptr : data.data(), data.size(), sizes.data(), &nsub, nullptr, useDqp ? pic.cuQp.data() : nullptr, p.limitTU >= 3 ? &tuRecs : nullptr);
    if (rc != X265AMD_OK) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: analysis");

    /* ---- from here on in coding order ---- */
    if (prev.valid() && prev.get() != 0) return X265AMD_EHIP;
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    if (p.bEnableLoopFilter)
    {
        std::vector<x265amd_deblock_unit> dbu(nUnits);
        rc = x265amd_deblock_units(&si, &info, pic.units.data(), pic.motion.data(), dbu.data());
        if (rc != X265AMD_OK) return rc;
        if (hipMemcpyAsync(dDbUnits, dbu.data(), sizeof(x265amd_deblock_unit) * nUnits, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: deblock upload");
        rc = x265amd_deblock_picture(st, recY, recU, recV, stride, cstride, W, H, dDbUnits, p.deblockingFilterBetaOffset, p.deblockingFilterTCOffset, 0, 0, 0, 3);
        if (rc != X265AMD_OK) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: deblock");
    }
    int32_t saoFlags[2] = { 0, 0 };
    if (sao)
    {
        const size_t nstat = (size_t)nctu * 3 * 5 * 32;
        const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
        const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
        if (hipMemsetAsync(dSaoCount, 0, nstat * 4, st) != hipSuccess || hipMemsetAsync(dSaoOrg, 0, nstat * 4, st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: sao memset");
        rc = x265amd_sao_stats(st, recP, srcP, stride, cstride, W, H, dSaoCount, dSaoOrg);
        if (rc != X265AMD_OK) return rc;
        std::vector<int32_t> cnt(nstat), orgs(nstat);
        if (hipMemcpyAsync(cnt.data(), dSaoCount, nstat * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(orgs.data(), dSaoOrg, nstat * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: sao download");
        std::vector<x265amd_sao_ctu> sparams((size_t)nctu);
        memset(sparams.data(), 0, sizeof(x265amd_sao_ctu) * nctu);
        rc = x265amd_sao_rdo(&si, pic.type != TYPE_B ? 1 : 0, 1, p.qpMin, p.qpMax,       /* IS_REFERENCED: fixed by the type (hasReferences changes as later pictures are prepared) */
                             pic.units.data(), cnt.data(), orgs.data(), depthSaoRate, sparams.data(), saoFlags);
        if (rc != X265AMD_OK) return rc;
        if (hipMemcpyAsync(dSaoParams, sparams.data(), sizeof(x265amd_sao_ctu) * nctu, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(dSaoTmp, pic.dRec, picElems * sizeof(pixel), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: sao upload");
        const uint64_t dstP[3] = { planeAddr(dSaoTmp, 0), planeAddr(dSaoTmp, 1), planeAddr(dSaoTmp, 2) };
        rc = x265amd_sao_apply(st, recP, dstP, stride, cstride, W, H, dSaoParams);
        if (rc != X265AMD_OK) return rc;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: sao");
        std::swap(pic.dRec, dSaoTmp);
        recY = pic.dRec + org[0]; recU = pic.dRec + org[1]; recV = pic.dRec + org[2];
        rc = x265amd_encode_slice_data(&si, pic.units.data(), coeff.data(), sparams.data(), saoFlags, data.data(), data.size(), sizes.data(), &nsub);
        if (rc != X265AMD_OK) return rc;
    }
    /* the reconstruction becomes a reference: extend its borders */
    rc = x265amd_extend_pic_border(st, recY, stride, W, H, marginX, marginY);
    if (rc == X265AMD_OK) rc = x265amd_extend_pic_border(st, recU, cstride, W / 2, H / 2, marginX / 2, marginY / 2);
    if (rc == X265AMD_OK) rc = x265amd_extend_pic_border(st, recV, cstride, W / 2, H / 2, marginX / 2, marginY / 2);
    if (rc != X265AMD_OK) return rc;
    if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: border extension");

    rc = sliceNal(*this, pic, fc, saoFlags, data, sizes, nsub);
    if (rc) return rc;
    /* the source is no longer needed (unless the weight analysis of later pictures reads its chroma planes: weightAnalyse works on source pictures); the reconstruction
     * stays while the picture is referenced.  The reference lists are only needed by pictures that are still to come through their own lists */
    if (!keepSources()) { xa_scratch_free(pic.dSrc); pic.dSrc = nullptr; }
    return 0;
}

/* ---- pictures coded in parallel (param.frameNumThreads > 1) ----
 * FrameEncoder::compressFrame as the reference runs it with several frame encoders (frameencoder.cpp:880-960, :1930-1960; framefilter.cpp:559-664): a CTU row
 * of this picture starts when every reference picture has finished the rows down to refLagRows below it, and the in-loop filters follow the analysis row by
 * row so that the rows of this picture become available to the pictures that reference it while its lower rows are still being analysed. */
struct RowGate { x265amd_encoder* e; Pic* pic; std::vector<Pic*> refs; std::vector<uint8_t>* refDepth; size_t nUnits; FrameCtx* fc = nullptr; std::vector<int8_t>* refQp0 = nullptr; };

/* What a CTU may read of a reference picture, and when.  The reference waits for whole rows: row + refLagRows rows of every reference picture before a row
 * starts (frameencoder.cpp:893-908).  What the row's commands can actually read is less -- vectors end searchRange samples below the block (search.cpp:92,
 * :2763; merge / AMVP candidates beyond are left out), plus sub-sample steps and interpolation taps: the rows row - 2 .. row + 2 at most -- and pictures here
 * are published by COLUMNS (Pic::finalX, filterRowsCols): a CTU starts when those rows of every reference picture are final two CTUs to its right, which covers
 * ordinary vectors, the co-located CTUs' motion and the co-located depths; every command that reads reference samples first asks gateRefReady with its exact
 * reach (xa_ref_guard_*), so a long vector waits for exactly what it needs.  A picture therefore follows its reference pictures a few CTUs behind instead of
 * rows behind, and the slow last CTU row of a picture (cut CTUs when the height is no multiple of 64) no longer holds up every picture behind it.  Results do
 * not depend on any of this: a sample is only ever read when it is final.  (Deadlock freedom with few device queues: a row takes its queue when it starts, and
 * it starts only when finalX of the rows it will follow is above zero, i.e. when those rows hold their queues and run.) */
/* the per-CTU gate: the rows row - 2 .. row + 1 of every reference picture final up to 56 samples beyond the CTU to the right (vectors up to that length, the
 * co-located CTUs' motion and depths), row + 2 begun (a vector reaching into its first lines is rare: gateRefWait then waits for it, and the row holds its queue) */
static inline int gateNeed(const x265amd_encoder& e, int col) { return std::min(e.W, 64 * col + 120); }
static int gateCtuReady(void* ctx, int row, int col)        /* polled (the start condition of a row task): 1 yes, 0 not yet, -1 a reference picture failed */
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    const int need = gateNeed(e, col), r0 = std::max(0, row - 2), r1 = std::min(e.ctuH - 1, row + 1);
    for (Pic* q : g.refs)
    {
        if (q->failed.load(std::memory_order_acquire)) return -1;
        if (row + 2 < e.ctuH && q->published(row + 2) < 1) return 0;
        for (int r = r1; r >= r0; r--) if (q->published(r) < need) return 0;
    }
    return 1;
}
static int gateRowReady(void* ctx, int row) { return gateCtuReady(ctx, row, 0); }
static int gateCtuWait(void* ctx, int row, int col)         /* blocking: the task parks on the counters */
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    const int need = gateNeed(e, col), r0 = std::max(0, row - 2), r1 = std::min(e.ctuH - 1, row + 1);
    static const bool pubLog = getenv("X265AMD_PUB_LOG") != nullptr;
    for (Pic* q : g.refs)
        for (int r = r1; r >= r0; r--)
        {
            if (q->published(r) < need)
            {
                const double t0 = pubLog ? Pic::pubClockMs() : 0;
                xa_wait_counter(q->finalX[r], (uint64_t)need);
                if (pubLog) fprintf(stderr, "x265amd gate: poc %d row %d col %d waited from %.2f to %.2f for poc %d row %d x %d\n", g.pic->poc, row, col, t0, Pic::pubClockMs(), q->poc, r, need);
            }
            if (q->failed.load(std::memory_order_acquire)) return -1;
        }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
static int gateRefWait(void* ctx, int picIdx, int yMin, int yMax, int xMax)
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    if (picIdx >= (int)g.refs.size() && g.fc && picIdx - (int)g.refs.size() < (int)g.fc->wplanes.size())
    {
        /* a weighted copy (a motion search of a slice with weights): whole CTU rows of the reference picture, then the copy's rows (MotionReference::applyWeight at the
         * row's start, frameencoder.cpp:900-908) */
        WPlane& wpl = *g.fc->wplanes[picIdx - (int)g.refs.size()];
        Pic* q = wpl.src;
        const int r0 = std::min(std::max(yMin, 0), e.H - 1) >> 6, r1 = std::min(std::max(yMax, 0), e.H - 1) >> 6;
        const int wc = xa_task_wait_class(3);
        for (int r = r1; r >= r0; r--)
        {
            if (q->published(r) < e.W) xa_wait_counter(q->finalX[r], (uint64_t)e.W);
            if (q->failed.load(std::memory_order_acquire)) { xa_task_wait_class(wc); return -1; }
        }
        xa_task_wait_class(wc);
        std::atomic_thread_fence(std::memory_order_acquire);
        return g.e->weightRows(wpl, r0, r1) == X265AMD_OK ? 0 : -1;
    }
    if (picIdx < 0 || picIdx >= (int)g.refs.size()) return 0;          /* the picture itself / the source: not a reference */
    Pic* q = g.refs[picIdx];
    const int need = xMax >= e.W - 1 ? e.W : std::max(0, xMax + 1);
    const int r0 = std::min(std::max(yMin, 0), e.H - 1) >> 6, r1 = std::min(std::max(yMax, 0), e.H - 1) >> 6;
    const int wc = xa_task_wait_class(3);
    for (int r = r1; r >= r0; r--)
    {
        if (q->published(r) < need) xa_wait_counter(q->finalX[r], (uint64_t)need);
        if (q->failed.load(std::memory_order_acquire)) { xa_task_wait_class(wc); return -1; }
    }
    xa_task_wait_class(wc);
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}
/* what gateCtuWait(row, col) has waited for (XaRowHooks::ctu_reach) */
static void gateCtuReach(void* ctx, int row, int col, int* r0, int* r1, int* need)
{
    const x265amd_encoder& e = *((RowGate*)ctx)->e;
    *need = gateNeed(e, col); *r0 = std::max(0, row - 2); *r1 = std::min(e.ctuH - 1, row + 1);
}
static void gateBeforeRow(void*, int) {}
static void gateBeforeCtu(void* ctx, int row, int col)
{
    RowGate& g = *(RowGate*)ctx;
    const x265amd_encoder& e = *g.e;
    /* the co-located CTU's depths (topSkipMinDepth reads refFrameList[l][0] at this CTU's address): that CTU of the reference picture is coded now */
    for (int l = 0; l < 2; l++)
        if (!g.pic->lists[l].empty())
        {
            const Pic* q = g.pic->lists[l][0].get();
            const int y0 = row * 16, y1 = std::min(e.h4, y0 + 16), x0 = col * 16, x1 = std::min(e.w4, x0 + 16);
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) (*g.refDepth)[l * g.nUnits + (size_t)y * e.w4 + x] = q->units[(size_t)y * e.w4 + x].depth;
            /* ... and its first unit's QP (topSkipMinDepth's previousQP: CUData::m_qp[0] of that CTU as it was coded) */
            if (e.useDqp && g.refQp0) (*g.refQp0)[(size_t)l * e.nctu + (size_t)row * e.ctuW + col] = q->units[(size_t)y0 * e.w4 + x0].qp;
        }
}
static void gateAfterCtu(void* ctx, int row, int col)
{
    RowGate& g = *(RowGate*)ctx;
    { std::lock_guard<std::mutex> lk(g.pic->mu); g.pic->analysedCols[row] = col + 1; }
    g.pic->cv.notify_all();
}
static void gateAfterRow(void* ctx, int row)
{
    RowGate& g = *(RowGate*)ctx;
    { std::lock_guard<std::mutex> lk(g.pic->mu); g.pic->analysedRows = row + 1; }
    g.pic->cv.notify_all();
}

/* CTU rows r0 .. r1 of a weighted copy (MotionReference::applyWeight, reference.cpp:118-185): weight_pp_c over the rows' padded lines -- with the first row the top margin, with
 * the last the bottom margin -- of the planes that carry a weight.  The rows of the reference picture are final (the caller has waited for them). */
int x265amd_encoder::weightRows(WPlane& wpl, int r0, int r1)
{
    std::lock_guard<std::mutex> lk(wpl.mu);
    bool any = false;
    xa_thread_device();
    for (int r = r0; r <= r1; r++)
    {
        if (r < 0 || r >= ctuH || wpl.rowDone[r]) continue;
        for (int cc = 0; cc < 3; cc++)
        {
            if (!wpl.chroma[cc]) continue;
            const int sh = cc ? 1 : 0, h = H >> sh, my = marginY >> sh, mx = marginX >> sh, rows = 64 >> sh;
            const intptr_t st = cc ? cstride : stride;
            const int y0 = r == 0 ? -my : rows * r, y1 = r == ctuH - 1 ? h + my : rows * (r + 1);
            const intptr_t at = (intptr_t)org[cc] + (intptr_t)y0 * st - mx;
            const x265amd_weight& w = wpl.w[cc];
            const int correction = 14 - X265AMD_DEPTH;
            const int rc = x265amd_weight_buffer(wpl.st, wpl.src->finalPlanes() + at, wpl.buf + at, (size_t)(y1 - y0) * st, w.w, (w.denom ? 1 << (w.denom - 1) : 0) << correction,
                                                 w.denom + correction, w.o * (1 << (X265AMD_DEPTH - 8)));
            if (rc != X265AMD_OK) return rc;
        }
        any = true;
    }
    if (any && hipStreamSynchronize(wpl.st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: weighted reference rows");
    for (int r = std::max(r0, 0); r <= r1 && r < ctuH; r++) wpl.rowDone[r] = 1;
    return X265AMD_OK;
}

/* The filter thread of a picture: FrameFilter::processRow / processPostRow for each CTU row as the analysis delivers it.  Row r is deblocked when row r + 1
 * is analysed (its vertical edges, then its horizontal edges, which reach three samples up into row r - 1); its SAO statistics follow (they leave out the samples
 * the rows below still change) and its parameters are decided; row r - 1 can then be offset (its last lines and the line below them are final), its borders
 * extended and the row published.  The last row publishes itself. */
int x265amd_encoder::filterRows(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags)
{
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const bool sao = p.bEnableSAO != 0, dbl = p.bEnableLoopFilter != 0;
    const size_t nUnits = (size_t)w4 * h4, nstat = (size_t)nctu * 3 * 5 * 32, rowStat = (size_t)ctuW * 3 * 5 * 32;
    struct Scratch { void* p = nullptr; ~Scratch() { xa_scratch_free(p); } } dDb, dCnt, dOrg, dPar;
    std::vector<x265amd_deblock_unit> dbu;
    std::vector<int32_t> cnt, orgs;
    if (dbl) { if (xa_scratch_alloc(&dDb.p, sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: device allocation"); dbu.resize(nUnits); }
    if (sao)
    {
        if (xa_scratch_alloc(&dCnt.p, nstat * 4) != hipSuccess || xa_scratch_alloc(&dOrg.p, nstat * 4) != hipSuccess || xa_scratch_alloc(&dPar.p, sizeof(x265amd_sao_ctu) * nctu) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: device allocation");
        cnt.resize(nstat); orgs.resize(nstat);
        saoFlags[0] = saoFlags[1] = 1;          /* SAO::startSlice: never switched off when pictures are coded in parallel (sao.cpp:264) */
    }
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    pixel* fin = sao ? pic.dFin : pic.dRec;
    const uint64_t finP[3] = { planeAddr(fin, 0), planeAddr(fin, 1), planeAddr(fin, 2) };
    double unusedRate[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int rc = X265AMD_OK;
    /* offsets, borders, publication of CTU row k */
    auto finish = [&](int k) -> int {
        const int y0 = k * 64, y1 = std::min(H, y0 + 64);
        if (sao)
        {
            int r = x265amd_sao_apply_rows(st, recP, finP, stride, cstride, W, H, (const x265amd_sao_ctu*)dPar.p, k, k + 1);
            if (r != X265AMD_OK) return r;
        }
        int r = x265amd_extend_border_rows(st, fin + org[0], stride, W, H, marginX, marginY, y0, y1);
        if (r == X265AMD_OK) r = x265amd_extend_border_rows(st, fin + org[1], cstride, W / 2, H / 2, marginX / 2, marginY / 2, y0 / 2, y1 / 2);
        if (r == X265AMD_OK) r = x265amd_extend_border_rows(st, fin + org[2], cstride, W / 2, H / 2, marginX / 2, marginY / 2, y0 / 2, y1 / 2);
        if (r != X265AMD_OK) return r;
        if (hipStreamSynchronize(st) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: row filters");
        pic.publish(k, W);
        return X265AMD_OK;
    };
    static const bool timing = getenv("X265AMD_TIMING") != nullptr;
    double tPh[6] = { 0, 0, 0, 0, 0, 0 };
    auto tLast = std::chrono::steady_clock::now();
    auto stamp = [&](int k) { if (!timing) return; const auto n = std::chrono::steady_clock::now(); tPh[k] += std::chrono::duration<double, std::milli>(n - tLast).count(); tLast = n; };
    struct Report { const bool& on; double* t; int poc; ~Report() { if (on) fprintf(stderr, "x265amd: filter rows of poc %d (ms): waiting %.1f, deblock units + upload %.1f, deblock %.1f, sao statistics %.1f, sao decision + upload %.1f, offsets + borders %.1f\n", poc, t[0], t[1], t[2], t[3], t[4], t[5]); } } report{ timing, tPh, pic.poc };
    for (int r = 0; r < ctuH && rc == X265AMD_OK; r++)
    {
        stamp(5);
        {
            std::unique_lock<std::mutex> lk(pic.mu);
            /* intra prediction of row r + 1 reads the unfiltered last line of row r: FrameEncoder::m_filterRowDelay (frameencoder.cpp:124-126, :1936-1950) */
            const int needRows = (dbl || sao) ? std::min(ctuH, r + 2) : r + 1;
            pic.cv.wait(lk, [&] { return pic.analysedRows >= needRows || pic.failed; });
            if (pic.failed) return X265AMD_EHIP;
        }
        stamp(0);
        const int y4b = r * 16, y4e = std::min(h4, y4b + 16);
        if (dbl)
        {
            rc = x265amd_deblock_units_rows(&si, &info, pic.units.data(), pic.motion.data(), dbu.data(), y4b, y4e);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync((x265amd_deblock_unit*)dDb.p + (size_t)y4b * w4, dbu.data() + (size_t)y4b * w4, sizeof(x265amd_deblock_unit) * (size_t)(y4e - y4b) * w4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: deblock upload"); break; }
            stamp(1);
            rc = x265amd_deblock_rows(st, recY, recU, recV, stride, cstride, W, H, (const x265amd_deblock_unit*)dDb.p, p.deblockingFilterBetaOffset, p.deblockingFilterTCOffset, 0, 0, 0, 3, y4b, y4e);
            if (rc != X265AMD_OK) break;
        }
        if (sao)
        {
            if (hipMemsetAsync((int32_t*)dCnt.p + r * rowStat, 0, rowStat * 4, st) != hipSuccess || hipMemsetAsync((int32_t*)dOrg.p + r * rowStat, 0, rowStat * 4, st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao memset"); break; }
            rc = x265amd_sao_stats_rows(st, recP, srcP, stride, cstride, W, H, (int32_t*)dCnt.p, (int32_t*)dOrg.p, r, r + 1);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync(cnt.data() + r * rowStat, (int32_t*)dCnt.p + r * rowStat, rowStat * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(orgs.data() + r * rowStat, (int32_t*)dOrg.p + r * rowStat, rowStat * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao download"); break; }
            stamp(3);
            int32_t flags[2] = { 1, 1 };
            rc = x265amd_sao_rdo_rows(&si, pic.type != TYPE_B ? 1 : 0, 2, p.qpMin, p.qpMax, pic.units.data(), cnt.data(), orgs.data(), unusedRate, sparams.data(), flags, r, r + 1);
            if (rc != X265AMD_OK) break;
            if (hipMemcpyAsync((x265amd_sao_ctu*)dPar.p + (size_t)r * ctuW, sparams.data() + (size_t)r * ctuW, sizeof(x265amd_sao_ctu) * ctuW, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
            { rc = xa_fail(X265AMD_EHIP, "encoder: sao upload"); break; }
            stamp(4);
        }
        if (!dbl && !sao) { rc = finish(r); continue; }         /* nothing below changes this row */
        if (r > 0) rc = finish(r - 1);
        if (rc == X265AMD_OK && r == ctuH - 1) rc = finish(r);
    }
    return rc;
}

/* Waiting for a stream without burning a core: hipStreamSynchronize polls flat out, and the filter threads of twenty pictures in flight did that beside the worker
 * threads -- past the CPU quota of the box (16 cores), where the kernel then freezes every thread of the process for the rest of its 100 ms period (cgroup cpu.stat:
 * nr_throttled; a dozen milliseconds each time, in the middle of the encode).  An event, a short poll for the common case (the work is a few kernels), then naps. */
static hipError_t streamWaitPolite(hipStream_t st, hipEvent_t ev)
{
    static const bool off = getenv("X265AMD_FILTER_SPIN") && atoi(getenv("X265AMD_FILTER_SPIN")) != 0;
    static const int spinUs = getenv("X265AMD_FILTER_SPIN_US") ? atoi(getenv("X265AMD_FILTER_SPIN_US")) : 30;
    if (off || !ev) return hipStreamSynchronize(st);
    hipError_t e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;)
    {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spinUs)) { for (int k = 0; k < 16; k++) __builtin_ia32_pause(); continue; }
        struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr);
        /* a filter stream that stands for seconds: say so once (the kernels behind the event are a few microseconds each) */
        static std::atomic<int> said{ 0 };
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(3) && said.fetch_add(1) < 4)
            fprintf(stderr, "x265amd: a filter stream has not reached its event for 3 s (stream %p, hipStreamQuery says %s)\n", (void*)st, hipGetErrorName(hipStreamQuery(st)));
    }
}

/* The filter thread of a picture, by columns.  A UNIT is a CTU row r and a range of its CTU columns [c0, c1): the deblocking of the unit's edges (vertical edges
 * right of c0's left boundary up to and including c1's left boundary, then the horizontal edges of the columns, the top one reaching three samples into row
 * r - 1), the SAO statistics and decisions of its CTUs; behind it row r - 1 (and the last row itself) is offset, its borders extended and its columns published
 * up to eight samples short of the unit's right end (the offsets of those need the next unit's horizontal edges).  A unit is ready when
 *   - row r is analysed through CTU c1 (the vertical edge at its right boundary reads both sides' coding data),
 *   - row r + 1 is analysed through CTU c1 (its intra prediction has then read everything it needs of row r's last line UNFILTERED:
 *     FrameEncoder::m_filterRowDelay, frameencoder.cpp:124-126, :1936-1950),
 *   - the units of row r - 1 cover the columns (their vertical edges precede this unit's top horizontal edge, their decisions are the merge-up candidates).
 * The same samples as FrameFilter's row order produce (framefilter.cpp:559-664): vertical edges lie eight samples apart and touch three on either side, so
 * their order is free; a horizontal edge reads its own columns behind the vertical edges on both sides; a CTU's statistics leave out what the CTUs to its right
 * and below still change (sao.cpp:760-776); an offset sample is written when it and its neighbours are final.  Units are taken as large as the analysis
 * allows (a row the filter falls behind on is caught up in one unit), at least `minChunk` CTUs. */
int x265amd_encoder::filterRowsCols(Pic& pic, const x265amd_slice_info& si, const x265amd_mvpred_info& info, std::vector<x265amd_sao_ctu>& sparams, int32_t* saoFlags)
{
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    hipEvent_t ev = nullptr;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    struct EventGuard { hipEvent_t e; ~EventGuard() { if (e) (void)hipEventDestroy(e); } } evGuard{ ev };
    const bool sao = p.bEnableSAO != 0, dbl = p.bEnableLoopFilter != 0;
    const size_t nUnits = (size_t)w4 * h4, ctuStat = (size_t)3 * 5 * 32, nstat = (size_t)nctu * ctuStat;
    /* device: deblocking records, SAO parameters.  Pinned host memory the device reads / writes in place (no staging copies, no synchronisation to free a
     * staging buffer): the records as the host derives them (copied to the device in stream order), the statistics as the kernel stores them, the parameters as
     * decided (copied in stream order). */
    struct Scratch { void* p = nullptr; ~Scratch() { xa_scratch_free(p); } } dDb, dPar;
    XaMapped hDb, hPar; XaMappedOut hCnt, hOrg;
    if (dbl && (xa_scratch_alloc(&dDb.p, sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess || hDb.alloc(sizeof(x265amd_deblock_unit) * nUnits) != hipSuccess))
        return xa_fail(X265AMD_EHIP, "encoder: device allocation");
    if (sao)
    {
        if (xa_scratch_alloc(&dPar.p, sizeof(x265amd_sao_ctu) * nctu) != hipSuccess || hPar.alloc(sizeof(x265amd_sao_ctu) * nctu) != hipSuccess ||
            hCnt.alloc(nstat * 4) != hipSuccess || hOrg.alloc(nstat * 4) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder: device allocation");
        saoFlags[0] = saoFlags[1] = 1;          /* SAO::startSlice: never switched off when pictures are coded in parallel (sao.cpp:264) */
    }
    x265amd_deblock_unit* dbu = (x265amd_deblock_unit*)hDb.p;
    int32_t* cnt = (int32_t*)hCnt.p; int32_t* orgs = (int32_t*)hOrg.p;
    pixel* recY = pic.dRec + org[0]; pixel* recU = pic.dRec + org[1]; pixel* recV = pic.dRec + org[2];
    const uint64_t recP[3] = { planeAddr(pic.dRec, 0), planeAddr(pic.dRec, 1), planeAddr(pic.dRec, 2) };
    const uint64_t srcP[3] = { planeAddr(pic.dSrc, 0), planeAddr(pic.dSrc, 1), planeAddr(pic.dSrc, 2) };
    pixel* fin = sao ? pic.dFin : pic.dRec;
    const uint64_t finP[3] = { planeAddr(fin, 0), planeAddr(fin, 1), planeAddr(fin, 2) };
    static const bool timing = getenv("X265AMD_TIMING") != nullptr;
    double tWait = 0, tWork = 0; int numUnits = 0, numSweeps = 0;
    auto tLast = std::chrono::steady_clock::now();
    auto lap = [&](double& acc) { if (!timing) return; const auto n = std::chrono::steady_clock::now(); acc += std::chrono::duration<double, std::milli>(n - tLast).count(); tLast = n; };
    struct Report { const bool& on; double& w; double& k; int& n; int& sw; int poc; ~Report() { if (on) fprintf(stderr, "x265amd: filter units of poc %d: %d units in %d sweeps, %.1f ms waiting for the analysis, %.1f ms filtering\n", poc, n, sw, w, k); } } report{ timing, tWait, tWork, numUnits, numSweeps, pic.poc };
    /* a picture nobody references is waited for by nobody: whole rows */
    /* the offsets' parameters are read by the kernel where the host wrote them (mapped memory: device memory behind the BAR unless X265AMD_PUSH_RECORDS=0 put the pools
     * into host memory, where a read per sample would cross PCIe: then they are copied as before) */
    static const bool parCopy = (getenv("X265AMD_SAO_PARAMS_COPY") && atoi(getenv("X265AMD_SAO_PARAMS_COPY")) != 0) || (getenv("X265AMD_PUSH_RECORDS") && atoi(getenv("X265AMD_PUSH_RECORDS")) == 0);
    static const int minChunkEnv = getenv("X265AMD_FILTER_CHUNK") ? atoi(getenv("X265AMD_FILTER_CHUNK")) : 0;
    const int minChunk = pic.type == TYPE_B ? ctuW : (minChunkEnv > 0 ? minChunkEnv : 2);
    /* the last rows are where a chain of pictures waits for each other (they finish last, and cut CTUs make the last row the slowest): every CTU of them at once */
    auto minChunkOf = [&](int r) { return (pic.type != TYPE_B && r >= ctuH - 3) ? 1 : minChunk; };
    std::vector<int> doneTop((size_t)ctuH, 0), doneFull((size_t)ctuH, 0), pubX((size_t)ctuH, 0), a((size_t)ctuH, 0);
    std::vector<uint8_t> carry((size_t)ctuH * (X265AMD_CTX_STRIDE + 8), 0);
    struct Unit { int r, c0, c1; };
    std::vector<Unit> todoTop, todoFull;
    int rc = X265AMD_OK;
    /* offsets and borders of the sample columns [pubX[k], newX) of CTU row k (enqueued; published behind the sweep's synchronisation) */
    auto finishCols = [&](int k, int newX) -> int {
        const int x0 = pubX[k];
        if (newX <= x0) return X265AMD_OK;
        const int y0 = k * 64, y1 = std::min(H, y0 + 64);
        if (sao)
        {
            int r = x265amd_sao_apply_rows_cols(st, recP, finP, stride, cstride, W, H, parCopy ? (const x265amd_sao_ctu*)dPar.p : (const x265amd_sao_ctu*)hPar.p, k, k + 1, x0, newX);
            if (r != X265AMD_OK) return r;
        }
        return xa_extend_border_band_420(st, fin + org[0], fin + org[1], fin + org[2], stride, cstride, W, H, marginX, marginY, y0, y1, x0, newX, x0 == 0, newX == W);
    };
    /* A CTU row's unit in two steps (round 4).  TOP: the vertical edges of the row's first eight lines and its top horizontal edge -- which completes the deblocking of
     * the row ABOVE -- as soon as the row itself is analysed (nothing of this touches the row's last line, which the row below still reads unfiltered); the row above can
     * then be offset, extended and published: one CTU row earlier than when everything waited for the row below.  FULL: the other vertical edges, the inner horizontal
     * edges, the statistics and the decisions, when the row below is analysed (FrameEncoder::m_filterRowDelay).  Vertical edges are decided per four lines and touch only
     * their own lines, the top horizontal edge touches lines 0-2: the samples are those of the reference's order (X265AMD_FILTER_EARLY_TOP=0: both steps together). */
    static const bool earlyTop = !(getenv("X265AMD_FILTER_EARLY_TOP") && atoi(getenv("X265AMD_FILTER_EARLY_TOP")) == 0);
    auto colsOf = [&](const std::vector<int>& an, int r) -> int { return an[r] == ctuW ? ctuW : an[r] - 1; };
    auto limTop = [&](const std::vector<int>& an, int r) -> int {
        int lim = colsOf(an, r);
        if (r > 0) lim = std::min(lim, doneFull[r - 1]);
        if (!earlyTop && r + 1 < ctuH) lim = std::min(lim, colsOf(an, r + 1));
        return lim;
    };
    auto limFull = [&](const std::vector<int>& an, int r) -> int {
        int lim = doneTop[r];
        if (r + 1 < ctuH) lim = std::min(lim, colsOf(an, r + 1));
        return lim;
    };
    auto chunkOk = [&](int r, int c0, int c1) { return c1 > c0 && (c1 == ctuW || c1 - c0 >= minChunkOf(r)); };
    for (;;)
    {
        {
            std::unique_lock<std::mutex> lk(pic.mu);
            /* something to do? (a snapshot of the analysis: the rows only advance) */
            auto ready = [&]() -> bool {
                if (pic.failed) return true;
                bool allDone = true;
                for (int r = 0; r < ctuH; r++)
                {
                    if (doneFull[r] == ctuW) continue;
                    allDone = false;
                    if (chunkOk(r, doneTop[r], limTop(pic.analysedCols, r))) return true;
                    /* (a FULL step may become possible through the TOP step of the same sweep: the TOP test above covers that case) */
                    if (chunkOk(r, doneFull[r], limFull(pic.analysedCols, r))) return true;
                }
                return allDone;
            };
            pic.cv.wait(lk, ready);
            if (pic.failed) return X265AMD_EHIP;
            a = pic.analysedCols;
        }
        lap(tWait);
        /* ---- one sweep: every step that is ready, top row first (the stream orders them: FULL of row r - 1, TOP of row r, FULL of row r).  First the edges and the
         * statistics of all of them, one synchronisation, then the decisions on the host, then offsets + borders, a second synchronisation, then the publication:
         * two waits per sweep however many rows are in flight. ---- */
        todoTop.clear(); todoFull.clear();
        bool all = true;
        static const bool dbCopy = getenv("X265AMD_DEBLOCK_UNITS_COPY") && atoi(getenv("X265AMD_DEBLOCK_UNITS_COPY")) != 0;
        for (int r = 0; r < ctuH && rc == X265AMD_OK; r++)
        {
            if (doneFull[r] == ctuW) continue;
            all = false;
            const int y4b = r * 16, y4e = std::min(h4, y4b + 16), y4t = std::min(y4e, y4b + 2);
            {
                const int c0 = doneTop[r], c1 = limTop(a, r);
                if (chunkOk(r, c0, c1))
                {
                    const int x4b = c0 * 16, x4e = std::min(w4, c1 * 16 + 1);       /* + the unit column right of the boundary edge */
                    if (dbl)
                    {
                        /* the edge records of the whole row height (both steps read them) where the kernels read them: mapped memory, no copy */
                        rc = x265amd_deblock_units_rect(&si, &info, pic.units.data(), pic.motion.data(), dbu, y4b, y4e, x4b, x4e);
                        if (rc != X265AMD_OK) break;
                        _mm_sfence();       /* the records went through the write-combining BAR mapping: out of this core's buffers before the launch that reads them */
                        if (dbCopy && hipMemcpy2DAsync((x265amd_deblock_unit*)dDb.p + (size_t)y4b * w4 + x4b, sizeof(x265amd_deblock_unit) * w4, dbu + (size_t)y4b * w4 + x4b, sizeof(x265amd_deblock_unit) * w4,
                                             sizeof(x265amd_deblock_unit) * (size_t)(x4e - x4b), (size_t)(y4e - y4b), hipMemcpyHostToDevice, st) != hipSuccess)
                        { rc = xa_fail(X265AMD_EHIP, "encoder: deblock upload"); break; }
                        rc = x265amd_deblock_rows_cols(st, recY, recU, recV, stride, cstride, W, H, dbCopy ? (const x265amd_deblock_unit*)dDb.p : dbu, p.deblockingFilterBetaOffset, p.deblockingFilterTCOffset, 0, 0, 0, 3, y4b, y4t, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    todoTop.push_back(Unit{ r, c0, c1 });
                    doneTop[r] = c1;
                }
            }
            {
                const int c0 = doneFull[r], c1 = limFull(a, r);
                if (chunkOk(r, c0, c1))
                {
                    if (dbl && y4e > y4t)
                    {
                        rc = x265amd_deblock_rows_cols(st, recY, recU, recV, stride, cstride, W, H, dbCopy ? (const x265amd_deblock_unit*)dDb.p : dbu, p.deblockingFilterBetaOffset, p.deblockingFilterTCOffset, 0, 0, 0, 3, y4t, y4e, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    if (sao)
                    {
                        /* every workgroup stores all 160 sums and counts of its (CTU, plane): nothing to clear; the host reads them where the kernel leaves them */
                        rc = x265amd_sao_stats_rows_cols(st, recP, srcP, stride, cstride, W, H, cnt, orgs, r, r + 1, c0, c1);
                        if (rc != X265AMD_OK) break;
                    }
                    todoFull.push_back(Unit{ r, c0, c1 });
                    doneFull[r] = c1;
                }
            }
        }
        if (rc != X265AMD_OK) break;
        if (todoTop.empty() && todoFull.empty()) { if (all) break; continue; }
        numUnits += (int)(todoTop.size() + todoFull.size()); numSweeps++;
        if (sao && !todoFull.empty())
        {
            if (streamWaitPolite(st, ev) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: sao statistics"); break; }
            for (const Unit& u : todoFull)
            {
                int32_t flags[2] = { 1, 1 };
                rc = x265amd_sao_rdo_cols(&si, pic.type != TYPE_B ? 1 : 0, 2, p.qpMin, p.qpMax, pic.units.data(), cnt, orgs, sparams.data(), flags, u.r, u.c0, u.c1,
                                          carry.data() + (size_t)u.r * (X265AMD_CTX_STRIDE + 8));
                if (rc != X265AMD_OK) break;
                const size_t off = (size_t)u.r * ctuW + u.c0, n = (size_t)(u.c1 - u.c0);
                memcpy((x265amd_sao_ctu*)hPar.p + off, sparams.data() + off, sizeof(x265amd_sao_ctu) * n);
                _mm_sfence();               /* as for the deblocking records above */
                if (parCopy && hipMemcpyAsync((x265amd_sao_ctu*)dPar.p + off, (const x265amd_sao_ctu*)hPar.p + off, sizeof(x265amd_sao_ctu) * n, hipMemcpyHostToDevice, st) != hipSuccess)
                { rc = xa_fail(X265AMD_EHIP, "encoder: sao upload"); break; }
            }
            if (rc != X265AMD_OK) break;
        }
        /* final now: the row above a TOP step (its parameters were decided by its own FULL step, in this sweep at the latest), and the last row behind its FULL step --
         * up to eight samples short of the step's right end */
        for (const Unit& u : todoTop)
        {
            if (u.r == 0) continue;
            rc = finishCols(u.r - 1, u.c1 == ctuW ? W : 64 * u.c1 - 8);
            if (rc != X265AMD_OK) break;
        }
        if (rc != X265AMD_OK) break;
        for (const Unit& u : todoFull)
        {
            if (u.r != ctuH - 1) continue;
            rc = finishCols(u.r, u.c1 == ctuW ? W : 64 * u.c1 - 8);
            if (rc != X265AMD_OK) break;
        }
        if (rc != X265AMD_OK) break;
        if (streamWaitPolite(st, ev) != hipSuccess) { rc = xa_fail(X265AMD_EHIP, "encoder: row filters"); break; }
        for (const Unit& u : todoTop)
        {
            const int newX = u.c1 == ctuW ? W : 64 * u.c1 - 8;
            if (u.r > 0 && newX > pubX[u.r - 1]) { pubX[u.r - 1] = newX; pic.publish(u.r - 1, newX); }
        }
        for (const Unit& u : todoFull)
        {
            const int newX = u.c1 == ctuW ? W : 64 * u.c1 - 8;
            if (u.r == ctuH - 1 && newX > pubX[u.r]) { pubX[u.r] = newX; pic.publish(u.r, newX); }
        }
        lap(tWork);
    }
    return rc;
}

static uint64_t thread_cpu_ns()
{
    struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
int x265amd_encoder::runFrameParallel(const PicP& picp)
{
    Pic& pic = *picp;
    /* whatever happens, the pictures waiting for rows of this one are released */
    struct Release { Pic& pic; int* rc; ~Release() { if (*rc) pic.fail(); } };
    int rc = X265AMD_EHIP;
    Release release{ pic, &rc };
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder: stream");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{ st };
    const std::vector<PicP>* lists = pic.lists;
    FrameCtx fc;
    frameContext(*this, pic, fc);
    const Pic* colPic = fc.colPic;
    const size_t nUnits = (size_t)w4 * h4;
    std::vector<x265amd_mv_unit> noCol;
    if (!colPic) { noCol.resize(nUnits); memset(noCol.data(), 0, sizeof(x265amd_mv_unit) * nUnits); }
    std::vector<uint8_t> refDepth(2 * nUnits, 0);
    std::vector<int8_t> refQp0(2 * (size_t)nctu, 0);
    if (fc.failed) return xa_fail(X265AMD_EHIP, "encoder: weighted reference planes");
    RowGate gate{ this, &pic, {}, &refDepth, nUnits, &fc, &refQp0 };
    if (useDqp && pic.cuQp.empty()) return xa_fail(X265AMD_EINVAL, "encoder: the picture has no CU QPs");
    for (int l = 0; l < 2; l++)
    {
        for (const PicP& q : lists[l]) if (std::find(gate.refs.begin(), gate.refs.end(), q.get()) == gate.refs.end()) gate.refs.push_back(q.get());
        if (!lists[l].empty()) for (int i = 0; i < nctu; i++) refQp0[(size_t)l * nctu + i] = (int8_t)lists[l][0]->sliceQp;
    }
    std::vector<x265amd_cu_stat> stat((size_t)nctu + 1);
    memset(stat.data(), 0, sizeof(x265amd_cu_stat) * stat.size());
    std::vector<int16_t> coeff((size_t)nctu * RD_TILE_ELEMS, 0);
    std::vector<uint8_t> data((size_t)W * H * 3 + (1u << 16));
    std::vector<uint32_t> sizes((size_t)ctuH + 1, 0);
    int nsub = 0;
    const bool sao = p.bEnableSAO != 0;
    std::vector<x265amd_sao_ctu> sparams((size_t)nctu);
    memset(sparams.data(), 0, sizeof(x265amd_sao_ctu) * nctu);
    int32_t saoFlags[2] = { 0, 0 };
    int filterRc = X265AMD_OK;
    /* by columns when there is something to filter and the rows run as a wavefront; the row-by-row form otherwise */
    static const bool colsOff = getenv("X265AMD_FILTER_COLS") && atoi(getenv("X265AMD_FILTER_COLS")) == 0;
    const bool byCols = !colsOff && p.bEnableWavefront && (p.bEnableLoopFilter || p.bEnableSAO) && ctuH > 1 && ctuW > 1;
    std::thread filters([&, byCols] { xa_thread_device(); filterRc = byCols ? filterRowsCols(pic, fc.si, fc.info, sparams, saoFlags) : filterRows(pic, fc.si, fc.info, sparams, saoFlags); if (filterRc) pic.fail();
                                      cpuFilterNs += thread_cpu_ns(); });
    /* the rows' priority among the row tasks of all pictures in flight: the picture's place in coding order -- an I picture some places earlier (X265AMD_I_BOOST): its
     * chain of 8x8 CUs is the longest thing in flight, nothing it needs comes from another picture, and the pictures behind the scene cut wait for it */
    static const uint64_t iBoost = getenv("X265AMD_I_BOOST") ? (uint64_t)atoi(getenv("X265AMD_I_BOOST")) : 0;
    const bool isI = pic.type == TYPE_IDR || pic.type == TYPE_I;
    const uint64_t rowOrder = isI ? (pic.codingOrder + 1 > iBoost ? pic.codingOrder + 1 - iBoost : 1) : pic.codingOrder + 1;
    const XaRowHooks hooks{ &gate, gateRowReady, gateBeforeRow, gateAfterRow, gateCtuWait, gateBeforeCtu, gateAfterCtu, gateRefWait, rowOrder, gateCtuReach };
    XaTuRecs tuRecs = { nullptr, { nullptr, nullptr } };
    if (p.limitTU >= 3)
    {
        /* (the records were sized when the picture was prepared: pictures coded beside this one read them CTU by CTU behind the gate) */
        tuRecs.cur = pic.tuRecs.data();
        for (int l = 0; l < 2; l++) if (!lists[l].empty() && lists[l][0]->tuRecs.size() == pic.tuRecs.size()) tuRecs.ref[l] = lists[l][0]->tuRecs.data();
    }
    int arc = xa_analyse_frame(me, st, &fc.info, &fc.sp, &fc.si, &fc.ap, pic.units.data(), pic.motion.data(), colPic ? colPic->motion.data() : noCol.data(),
                               refDepth.data(), refQp0.data(), fc.planes.data(), (int)(fc.planes.size() / 3), stride, cstride, stat.data(), coeff.data(), nullptr,
                               sao ? nullptr : data.data(), data.size(), sizes.data(), &nsub, &hooks, useDqp ? pic.cuQp.data() : nullptr, p.limitTU >= 3 ? &tuRecs : nullptr);
    if (arc != X265AMD_OK) pic.fail();
    filters.join();
    if (arc != X265AMD_OK) return rc = arc;
    if (filterRc != X265AMD_OK) return rc = filterRc;
    if (sao)
    {
        arc = x265amd_encode_slice_data(&fc.si, pic.units.data(), coeff.data(), sparams.data(), saoFlags, data.data(), data.size(), sizes.data(), &nsub);
        if (arc != X265AMD_OK) return rc = arc;
    }
    arc = sliceNal(*this, pic, fc, saoFlags, data, sizes, nsub);
    if (arc) return rc = arc;
    if (!keepSources()) { xa_scratch_free(pic.dSrc); pic.dSrc = nullptr; }
    return rc = 0;
}

