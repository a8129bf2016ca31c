/* The encoder object behind include/x265amd_encoder.h: the frame-level host loop of the reference (SURVEY section 8b outer boundary and 8f rank 1),
 * restated for a device-resident encode.  What it follows in the reference:
 *   - mini-GOP formation with bFrameAdaptive 0:   Lookahead::slicetypeDecide        source/encoder/slicetype.cpp:1929-2040
 *   - constant QP per slice type:                  RateControl::init / rateControlStart  source/encoder/ratecontrol.cpp:321-346, :1592-1597
 *   - decoded picture buffer, RPS, NAL type:       DPB::prepareEncode / computeRPS / applyReferencePictureSet / decodingRefreshMarking /
 *                                                  getNalUnitType                  source/encoder/dpb.cpp:134-330, :336-470, :487-510
 *   - reference lists:                             Slice::setRefPicList            source/common/slice.cpp:32-140
 *   - DPB sizes, level:                            Encoder::initVPS/initSPS, determineLevel   source/encoder/encoder.cpp:3340-3470, level.cpp:44-230, :290-300
 *   - per frame: FrameEncoder::compressFrame (analysis rows, deblocking, SAO, border extension, slice NAL)   source/encoder/frameencoder.cpp:470-1130
 * Pictures (source and reconstruction) are padded planes in device memory with the reference's PicYuv margins (maxCUSize + 32 / + 16). */
#include "encoder_impl.h"

/* ---- configuration ---- */
extern "C" void x265amd_param_default(x265amd_param* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->fpsNum = 25; p->fpsDenom = 1;
    p->bframes = 0; p->keyframeMax = 250; p->maxNumReferences = 3; p->qp = 30; p->ipFactor = 1.4f; p->pbFactor = 1.3f;      /* (float literals, as common/param.cpp:276-277 has them) */
    p->rateControlMode = X265AMD_RC_CQP; p->rfConstant = 28; p->aqStrength = 1.0; p->qCompress = 0.6; p->aqMode = 0; p->cuTree = 0; p->qgSize = 32; p->qpMin = 0; p->qpMax = 69; p->vuiVideoFormat = 5; p->vuiColorPrimaries = 2; p->vuiTransfer = 2; p->vuiMatrix = 2; p->bEmitCLL = 1;
    p->rdLevel = 3; p->limitReferences = 3; p->bEnableEarlySkip = 1; p->recursionSkipMode = 1; p->bIntraInBFrames = 1; p->psyRd = 2.0;
    p->searchMethod = X265AMD_ME_HEX; p->subpelRefine = 2; p->searchRange = 57; p->maxNumMergeCand = 3;
    p->bEnableSignHiding = 1; p->bEnableStrongIntraSmoothing = 1; p->bEnableTemporalMvp = 1; p->tuQTMaxInterDepth = 1; p->tuQTMaxIntraDepth = 1;
    p->bEnableLoopFilter = 1; p->bEnableSAO = 1; p->bEnableWavefront = 1; p->aspectRatioIdc = 0;
}

void x265amd_encoder::fillStreamParams(x265amd_stream_params& s) const
{
    memset(&s, 0, sizeof(s));
    /* determineLevel (level.cpp:76-230): Main / Main 10, main tier, the lowest level that holds the picture size, rate and DPB */
    static const struct { uint32_t maxLumaSamples, maxLumaSamplesPerSecond; int idc; } levels[] = {
        { 36864, 552960, 30 }, { 122880, 3686400, 60 }, { 245760, 7372800, 63 }, { 552960, 16588800, 90 }, { 983040, 33177600, 93 },
        { 2228224, 66846720, 120 }, { 2228224, 133693440, 123 }, { 8912896, 267386880, 150 }, { 8912896, 534773760, 153 },
        { 8912896, 1069547520, 156 }, { 35651584, 1069547520, 180 }, { 35651584, 2139095040u, 183 }, { 35651584, 4278190080u, 186 } };
    s.profile_idc = X265AMD_DEPTH <= 8 ? 1 : 2;
    s.profile_compatibility_flags = X265AMD_DEPTH <= 8 ? (1u << 1) | (1u << 2) : (1u << 2);
    s.progressive_source = 1; s.frame_only_constraint = 1;
    s.bit_depth_constraint = X265AMD_DEPTH; s.chroma_format_constraint = 1; s.lower_bit_rate_constraint = 1;
    s.intra_constraint = p.keyframeMax <= 1;
    if (s.intra_constraint) { s.profile_idc = 4; s.profile_compatibility_flags = 1u << 4; }         /* Main Intra / Main 10 Intra: the range extensions' profile with its constraint flags (level.cpp:84-107) */
    const uint32_t lumaSamples = (uint32_t)(W * H);
    const uint32_t samplesPerSec = (uint32_t)(lumaSamples * ((double)p.fpsNum / p.fpsDenom));
    s.level_idc = 255;
    for (const auto& l : levels)
    {
        if (lumaSamples > l.maxLumaSamples || samplesPerSec > l.maxLumaSamplesPerSecond) continue;
        if (W > sqrt(l.maxLumaSamples * 8.0f) || H > sqrt(l.maxLumaSamples * 8.0f)) continue;
        uint32_t maxDpbSize = 6;
        if (lumaSamples <= (l.maxLumaSamples >> 2)) maxDpbSize = 16;
        else if (lumaSamples <= (l.maxLumaSamples >> 1)) maxDpbSize = 12;
        else if (lumaSamples <= ((3 * l.maxLumaSamples) >> 2)) maxDpbSize = 8;
        if ((uint32_t)maxDecPicBuffering > maxDpbSize) continue;
        s.level_idc = l.idc;
        break;
    }
    s.max_temporal_sub_layers = 1;
    s.max_dec_pic_buffering[0] = maxDecPicBuffering; s.num_reorder_pics[0] = numReorderPics; s.max_latency_increase[0] = p.bframes;
    s.chroma_format_idc = 1; s.pic_width = W; s.pic_height = H; s.bit_depth = X265AMD_DEPTH; s.log2_max_poc_lsb = 8;
    s.conformance_window = W != srcW || H != srcH; s.conf_win_offsets[1] = W - srcW; s.conf_win_offsets[3] = H - srcH;
    s.log2_min_cu_size = 3; s.log2_diff_max_min_cu_size = 3; s.tu_log2_min = 2; s.tu_log2_max = 5;
    s.tu_max_depth_inter = p.tuQTMaxInterDepth; s.tu_max_depth_intra = p.tuQTMaxIntraDepth;
    s.amp = p.bEnableAMP != 0; s.sao = p.bEnableSAO != 0; s.temporal_mvp = p.bEnableTemporalMvp != 0; s.strong_intra_smoothing = p.bEnableStrongIntraSmoothing != 0;
    s.aspect_ratio_idc = p.aspectRatioIdc; s.sar_width = p.vuiSarWidth; s.sar_height = p.vuiSarHeight;
    s.overscan_info_present = p.vuiOverscanInfoPresent != 0; s.overscan_appropriate = p.vuiOverscanAppropriate != 0;
    s.video_signal_type_present = p.vuiVideoSignalTypePresent != 0; s.video_format = p.vuiVideoFormat; s.video_full_range = p.vuiFullRange != 0;
    s.colour_description_present = p.vuiColorDescriptionPresent != 0; s.colour_primaries = p.vuiColorPrimaries; s.transfer_characteristics = p.vuiTransfer; s.matrix_coefficients = p.vuiMatrix;
    s.chroma_loc_info_present = p.vuiChromaLocPresent != 0; s.chroma_sample_loc_top = p.vuiChromaLocTop; s.chroma_sample_loc_bottom = p.vuiChromaLocBottom;
    s.default_display_window = p.vuiDisplayWindow != 0;
    s.def_disp_win_offsets[0] = p.vuiDispWinLeft; s.def_disp_win_offsets[1] = p.vuiDispWinRight; s.def_disp_win_offsets[2] = p.vuiDispWinTop; s.def_disp_win_offsets[3] = p.vuiDispWinBottom;
    s.emit_timing_info = 1; s.num_units_in_tick = p.fpsDenom; s.time_scale = p.fpsNum;
    s.weighted_pred = p.bEnableWeightedPred != 0; s.weighted_bipred = p.bEnableWeightedBiPred != 0;
    s.sign_hide = p.bEnableSignHiding != 0; s.num_ref_idx_default[0] = s.num_ref_idx_default[1] = 1; s.init_qp_minus26 = 0;
    s.use_dqp = useDqp; s.max_cu_dqp_depth = maxCuDqpDepth;           /* Encoder::initPPS (encoder.cpp:3424-3445) */
    s.wpp = p.bEnableWavefront != 0; s.loop_filter_across_slices = 1;
    s.deblocking_filter_control_present = !p.bEnableLoopFilter || p.deblockingFilterBetaOffset || p.deblockingFilterTCOffset; s.pic_disable_deblocking = !p.bEnableLoopFilter;          /* Encoder::initPPS (encoder.cpp:3458-3461) */
    s.beta_offset_div2 = p.deblockingFilterBetaOffset; s.tc_offset_div2 = p.deblockingFilterTCOffset;
}

extern "C" x265amd_encoder* x265amd_encoder_open(const x265amd_param* p)
{
    if (!p) { xa_fail(X265AMD_EINVAL, "encoder_open: null param"); return nullptr; }
    /* Encoder::configure's rules for the rate control's switches (encoder.cpp:3721-3754): constant QP switches adaptive quantisation and cuTree off; cuTree without AQ gets
     * aq-mode 1 at strength 0 (cuTree needs the offset arrays; delta QP is on); strength 0 without cuTree is no AQ at all */
    x265amd_param norm = *p;
    /* --keyint -1 (encoder.cpp:3627-3635): "only one I frame at the start of the stream": an infinite GOP distance and no adaptive I frame placement */
    if (norm.keyframeMax < 0) { norm.keyframeMax = INT_MAX; norm.scenecutThreshold = 0; }
    if (norm.keyframeMax <= 1 && norm.keyframeMax >= 0)
    {
        /* all-intra encodes (encoder.cpp:3636-3658): no lookahead, no B pictures, no cuTree, no weights, one reference, the parameter sets with every picture */
        norm.keyframeMax = 1; norm.keyframeMin = 1; norm.bFrameAdaptive = 0; norm.bframes = 0; norm.bOpenGOP = 0; norm.bRepeatHeaders = 1; norm.lookaheadDepth = 0;
        norm.scenecutThreshold = 0; norm.cuTree = 0; norm.bEnableWeightedPred = 0; norm.bEnableWeightedBiPred = 0; norm.maxNumReferences = 1;
    }
    /* the HDR10 SEI units come with the parameter sets at every keyframe ("Turning on repeat-headers for HDR compatibility", encoder.cpp:4347-4353) */
    if (norm.bEmitHDR10SEI || norm.hasMasteringDisplay || norm.maxCLL || norm.maxFALL) { norm.bEmitHDR10SEI = 1; norm.bRepeatHeaders = 1; }
    if (norm.limitTU && norm.tuQTMaxInterDepth < 2) norm.limitTU = 0;          /* "limit-tu disabled, requires tu-inter-depth > 1" (encoder.cpp:4103-4107) */
    if (norm.rateControlMode != X265AMD_RC_CRF) { norm.aqMode = 0; norm.cuTree = 0; }
    if (norm.lookaheadDepth == 0) norm.cuTree = 0;
    if (!norm.aqMode && norm.cuTree) { norm.aqMode = 1; norm.aqStrength = 0.0; }
    if (norm.aqStrength == 0 && !norm.cuTree) norm.aqMode = 0;
    /* Encoder::create (encoder.cpp:249-254): "Do not allow WPP if only one row or fewer than 3 columns, it is pointless and unstable" */
    if ((((norm.sourceHeight + 7) & ~7) + 63) / 64 == 1 || (((norm.sourceWidth + 7) & ~7) + 63) / 64 < 3) norm.bEnableWavefront = 0;
    p = &norm;
    if (p->sourceWidth < 16 || p->sourceHeight < 16 || (p->sourceWidth & 1) || (p->sourceHeight & 1) || p->sourceWidth > 8192 || p->sourceHeight > 4320)
    { xa_fail(X265AMD_EINVAL, "encoder_open: picture size must be even (4:2:0) and within 16..8192 x 16..4320"); return nullptr; }
    /* (found at the end of round 6 and not understood yet: the first P picture of a 64x64 clip is coded as one 64x64 CU where the reference splits it -- every other small
     * shape tried, 128x128, 256x64, 128x256, 136x72, is identical.  Refused rather than coded differently) */
    if (p->sourceWidth <= 64 && p->sourceHeight <= 64)
    { xa_fail(X265AMD_EINVAL, "encoder_open: sourceWidth / sourceHeight: a picture of a single CTU is not built"); return nullptr; }
    {
        /* every field outside the built subset is named (the reference logs "x265 [error]: <what>" per field, encoder/api.cpp:96-239 -> x265_check_params) */
        static thread_local char why[160];
        const char* bad = nullptr;
#define XA_REQUIRE(cond, text) do { if (!bad && !(cond)) bad = text; } while (0)
        XA_REQUIRE(p->fpsNum && p->fpsDenom, "fpsNum / fpsDenom must be non-zero");
        XA_REQUIRE(p->bframes >= 0 && p->bframes <= 16, "bframes outside 0..16");
        XA_REQUIRE(p->keyframeMax >= 1, "keyframeMax below 1");
        XA_REQUIRE(p->maxNumReferences >= 1 && p->maxNumReferences <= 8, "maxNumReferences outside 1..8");
        XA_REQUIRE(p->rateControlMode == 0 || p->rateControlMode == X265AMD_RC_CQP || p->rateControlMode == X265AMD_RC_CRF, "rc.rateControlMode: constant QP (1) and constant rate factor (2) are built, ABR (0 with a bitrate) is not");
        XA_REQUIRE(p->rateControlMode == X265AMD_RC_CRF || (p->qp >= 0 && p->qp <= 51), "qp outside 0..51");
        XA_REQUIRE(p->rateControlMode != X265AMD_RC_CRF || (p->rfConstant >= 0 && p->rfConstant <= 51), "rfConstant outside 0..51");
        XA_REQUIRE(p->aqMode >= 0 && p->aqMode <= 3, "aqMode outside 0..3 (the edge-based modes are not built)");
        XA_REQUIRE(p->qpMin >= 0 && p->qpMin <= p->qpMax && p->qpMax <= 69, "qpMin / qpMax outside 0..69 (or crossed)");
        XA_REQUIRE(p->aspectRatioIdc >= 0 && (p->aspectRatioIdc <= 16 || p->aspectRatioIdc == 255), "aspectRatioIdc outside 0..16 / 255");
        XA_REQUIRE(p->deblockingFilterTCOffset >= -6 && p->deblockingFilterTCOffset <= 6 && p->deblockingFilterBetaOffset >= -6 && p->deblockingFilterBetaOffset <= 6, "deblocking offsets outside -6..6");
        XA_REQUIRE(p->limitTU == 0 || (p->limitTU >= 2 && p->limitTU <= 4), "limitTU: 0, 2, 3 and 4 are built (1, the breadth-first form, is not)");
        XA_REQUIRE(p->limitTU < 3 || p->shardCount <= 1, "limitTU 3 / 4 with pictures coded on several GPUs is not built (the transform depth records do not travel)");
        XA_REQUIRE(p->decodedPictureHashSEI >= 0 && p->decodedPictureHashSEI <= 3 && p->maxCLL >= 0 && p->maxCLL <= 65535 && p->maxFALL >= 0 && p->maxFALL <= 65535, "decodedPictureHashSEI outside 0..3 or a light level outside 16 bits");
        XA_REQUIRE(p->vuiVideoFormat >= 0 && p->vuiVideoFormat <= 5 && p->vuiColorPrimaries >= 0 && p->vuiColorPrimaries <= 255 && p->vuiTransfer >= 0 && p->vuiTransfer <= 255 &&
                   p->vuiMatrix >= 0 && p->vuiMatrix <= 255 && p->vuiChromaLocTop >= 0 && p->vuiChromaLocTop <= 5 && p->vuiChromaLocBottom >= 0 && p->vuiChromaLocBottom <= 5, "vui: a value outside its range");
        XA_REQUIRE(!p->aqMode || p->aqStrength >= 0, "aqStrength negative");
        XA_REQUIRE(!p->aqMode || p->qgSize == 32 || p->qgSize == 64, "qgSize: 64 and 32 are built");
        XA_REQUIRE(!p->cuTree || p->aqMode, "cuTree needs adaptive quantisation (Encoder::configure switches it on with cuTree; say aqMode)");
        XA_REQUIRE(!p->cuTree || p->rateControlMode == X265AMD_RC_CRF, "cuTree needs rate control (the reference switches it off under constant QP, encoder.cpp:3721-3728)");
        XA_REQUIRE(!p->cuTree || p->lookaheadDepth > 0, "cuTree needs the lookahead (lookaheadDepth > 0)");
        XA_REQUIRE(!(p->aqMode && p->rateControlMode != X265AMD_RC_CRF), "adaptive quantisation needs rate control (the reference switches it off under constant QP, encoder.cpp:3721-3728)");
        XA_REQUIRE(p->rdLevel >= 2 && p->rdLevel <= 6, "rdLevel outside 2..6 (rd 0-1 are not built)");
        XA_REQUIRE(p->maxNumMergeCand >= 1 && p->maxNumMergeCand <= 5, "maxNumMergeCand outside 1..5");
        XA_REQUIRE(p->tuQTMaxInterDepth >= 1 && p->tuQTMaxInterDepth <= 4, "tuQTMaxInterDepth outside 1..4");
        XA_REQUIRE(p->tuQTMaxIntraDepth >= 1 && p->tuQTMaxIntraDepth <= 4, "tuQTMaxIntraDepth outside 1..4");
        XA_REQUIRE(p->searchMethod == X265AMD_ME_DIA || p->searchMethod == X265AMD_ME_HEX || p->searchMethod == X265AMD_ME_STAR, "searchMethod: only dia, hex and star are built (no umh / sea / full)");
        XA_REQUIRE(p->subpelRefine >= 0 && p->subpelRefine <= 7, "subpelRefine outside 0..7");
        XA_REQUIRE(p->rdoqLevel >= 0 && p->rdoqLevel <= 2, "rdoqLevel outside 0..2");
        XA_REQUIRE(p->psyRdoqFix8 >= 0, "psyRdoqFix8 negative");
        XA_REQUIRE(p->recursionSkipMode >= 0 && p->recursionSkipMode <= 1, "recursionSkipMode: only 0 and 1 are built (no edge-based rskip)");
        XA_REQUIRE(p->limitReferences >= 0 && p->limitReferences <= 3, "limitReferences outside 0..3");
        XA_REQUIRE(!p->bEnableAMP || p->bEnableRectInter, "bEnableAMP needs bEnableRectInter");
#undef XA_REQUIRE
        if (bad) { snprintf(why, sizeof(why), "encoder_open: %s", bad); xa_fail(X265AMD_EINVAL, why); return nullptr; }
    }
    xa_bind_device();           /* the threads of this library work on the opening thread's GPU */
    std::unique_ptr<x265amd_encoder> e(new x265amd_encoder);
    e->p = *p;
    /* a size that is no multiple of the smallest CU is coded padded to one, the pad replicating the last column / row, and the SPS's conformance window takes it off again
     * (Encoder::configure, encoder.cpp:4081-4090, :4300-4308; PicYuv::copyFromPicture) */
    e->srcW = p->sourceWidth; e->srcH = p->sourceHeight;
    e->W = (p->sourceWidth + 7) & ~7; e->H = (p->sourceHeight + 7) & ~7; e->w4 = e->W / 4; e->h4 = e->H / 4;
    e->ctuW = (e->W + 63) / 64; e->ctuH = (e->H + 63) / 64; e->nctu = e->ctuW * e->ctuH;
    e->stride = e->W + 2 * e->marginX; e->cstride = e->W / 2 + e->marginX;
    const size_t ysz = (size_t)(e->H + 2 * e->marginY) * e->stride, csz = (size_t)(e->H / 2 + e->marginY) * e->cstride;
    e->org[0] = (size_t)e->marginY * e->stride + e->marginX;
    e->org[1] = ysz + (size_t)(e->marginY / 2) * e->cstride + e->marginX / 2;
    e->org[2] = ysz + csz + (size_t)(e->marginY / 2) * e->cstride + e->marginX / 2;
    e->picElems = ysz + 2 * csz;
    /* RateControl (ratecontrol.cpp:321-346): constant QPs of the three slice types */
    const double ipOffset = 6.0 * log2(p->ipFactor), pbOffset = 6.0 * log2(p->pbFactor);
    auto clipQp = [](int q) { return q < 0 ? 0 : q > 69 ? 69 : q; };
    e->qpConstant[1] = p->qp;
    e->qpConstant[2] = clipQp((int)(p->qp - ipOffset + 0.5));
    e->qpConstant[0] = clipQp((int)(p->qp + pbOffset + 0.5));
    if (e->qpConstant[0] > 51 || e->qpConstant[2] > 51) { xa_fail(X265AMD_EINVAL, "encoder_open: slice QP above 51"); return nullptr; }
    /* level.cpp:290-296 */
    e->numReorderPics = (p->bBPyramid && p->bframes > 1) ? 2 : (p->bframes ? 1 : 0);           /* enforceLevel (level.cpp:295-296) */
    e->maxDecPicBuffering = std::min(16, std::max(e->numReorderPics + 2, p->maxNumReferences) + 1);
    if (p->firstFrame < 0) { xa_fail(X265AMD_EINVAL, "encoder_open: firstFrame"); return nullptr; }
    e->frameCount = p->firstFrame; e->lastKeyframe = p->firstFrame - p->keyframeMax; e->lastIDR = p->firstFrame;
    if (p->scenecutThreshold < 0 || p->scenecutThreshold > 100 || p->lookaheadDepth < 0 || p->lookaheadDepth > 250 || p->keyframeMin < 0 || p->keyframeMin > p->keyframeMax)
    { xa_fail(X265AMD_EINVAL, "encoder_open: scenecutThreshold outside 0..100, lookaheadDepth outside 0..250 or keyframeMin outside 0..keyframeMax"); return nullptr; }
    if (p->shardCount < 0 || p->shardCount > 64 || (p->shardCount > 1 && (p->shardRank < 0 || p->shardRank >= p->shardCount || p->frameNumThreads <= 1)))
    { xa_fail(X265AMD_EINVAL, "encoder_open: shardRank / shardCount (frame-per-GPU needs 0 <= rank < count and frameNumThreads > 1: rows are published by pictures coded in parallel)"); return nullptr; }
    if (p->bFrameAdaptive < 0 || p->bFrameAdaptive > 2) { xa_fail(X265AMD_EINVAL, "encoder_open: bFrameAdaptive: 0 (fixed mini-GOPs), 1 (fast) or 2 (trellis)"); return nullptr; }
    e->lookahead = p->scenecutThreshold > 0 || (p->bFrameAdaptive && p->bframes) || p->cuTree || p->aqMode;
    if ((p->bEnableWeightedPred || p->bEnableWeightedBiPred) && !e->lookahead) { xa_fail(X265AMD_EINVAL, "encoder_open: bEnableWeightedPred needs the lookahead (scenecutThreshold > 0 or bFrameAdaptive 2 with B frames)"); return nullptr; }
    {
        /* Encoder::configure (encoder.cpp:3658-3663) */
        int kmin = p->keyframeMin;
        if (!kmin) { const double fps = (double)p->fpsNum / p->fpsDenom; kmin = std::min((int)fps, p->keyframeMax / 10); }
        e->keyframeMin = std::max(1, kmin);
        /* Lowres::create (lowres.cpp:52-110): half size rounded up to whole 8x8 blocks, the picture's margins, stride a multiple of 32 */
        e->lowCuW = (e->W / 2 + 7) >> 3; e->lowCuH = (e->H / 2 + 7) >> 3;
        if (p->lookaheadSlices > 1 && e->H >= 720)
        {
            e->laRowsPerSlice = std::min(std::max(e->lowCuH / p->lookaheadSlices, 10), e->lowCuH);
            e->laNumSlices = e->lowCuH / e->laRowsPerSlice;
        }
        /* (found at the end of round 6 and not understood yet: without B pictures, with a lookahead that runs in slices -- 720 rows or more -- cuTree's offsets of a few blocks
         * at the picture's right edge differ from the reference's (76 of 8160 at 1080p), and the streams with them; --lookahead-slices 0 is identical, so are --tune
         * zerolatency (no lookahead) and every size below 720 rows.  Refused rather than coded differently) */
        if (e->laNumSlices > 1 && p->bframes == 0 && p->lookaheadDepth > 0)
        { xa_fail(X265AMD_EINVAL, "encoder_open: bframes 0 with lookaheadSlices above 1 in a picture of 720 rows or more is not built (lookaheadSlices 0 is)"); return nullptr; }
        e->lowW = e->lowCuW * 8; e->lowH = e->lowCuH * 8;
        e->lowBlocks = (e->lowCuW > 2 && e->lowCuH > 2) ? (e->lowCuW - 2) * (e->lowCuH - 2) : e->lowCuW * e->lowCuH;
        e->lowStride = e->W / 2 + 2 * e->marginX;
        e->lowStride += (32 - (e->lowStride & 31)) & 31;
        e->lowPlaneElems = (size_t)(e->lowH + 2 * e->marginY) * e->lowStride;
        e->lowOrg = (size_t)e->marginY * e->lowStride + e->marginX;
        if (e->lookahead)
        {
            /* The lookahead's kernels run for tens of milliseconds (a batch of cost estimates: hundreds of rows chained through progress words), and streams of one priority
             * share hardware queues: a picture's in-loop filter launch -- one workgroup, microseconds -- queued behind such a batch on the same hardware queue waited for it to END
             * (k_deblock_unit: 47 ms at worst in round 4's trace), and every picture that references that row waited with it.  Streams of another priority get hardware queues
             * of their own (device_queue.hip: the resident kernel's is the highest): the lookahead takes the lowest.  X265AMD_LA_PRIORITY=0: the ordinary one, as before. */
            static const bool laLow = !(getenv("X265AMD_LA_PRIORITY") && atoi(getenv("X265AMD_LA_PRIORITY")) == 0);
            int least = 0, greatest = 0;
            if (!laLow || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest || hipStreamCreateWithPriority(&e->laStream, hipStreamNonBlocking, least) != hipSuccess)
                if (hipStreamCreateWithFlags(&e->laStream, hipStreamNonBlocking) != hipSuccess) { xa_fail(X265AMD_EHIP, "encoder_open: stream"); return nullptr; }
        }
    }
    {
        /* pictures whose references are complete are analysed concurrently (B frames of a mini-GOP, the next P): the reference's frame threads, but
         * a picture only starts when its references are final, so the output does not depend on the thread count */
        const char* ft = getenv("X265AMD_FRAME_THREADS");
        e->frameThreads = ft ? atoi(ft) : 3;
        if (e->frameThreads < 1) e->frameThreads = 1;
    }
    if (p->frameNumThreads < 0 || p->frameNumThreads > 16) { xa_fail(X265AMD_EINVAL, "encoder_open: frameNumThreads"); return nullptr; }
    e->frameParallel = p->frameNumThreads > 1;
    const int queues = xa_queues_hint(0);           /* the number of device job queues in force (224 unless X265AMD_QUEUES says otherwise) */
    if (e->frameParallel)
    {
        /* FrameEncoder::init (frameencoder.cpp:170-175): rows of a reference picture that must be final before a row of this picture starts */
        static const int hpelIters[8] = { 1, 1, 1, 2, 3, 1, 2, 3 };         /* MotionEstimate::hpelIterationCount: hpel_iters + qpel_iters / 2 (motion.cpp:48-58, :155) */
        int range = p->searchRange;
        range += p->searchMethod < 2;
        range += 8 / 2;
        range += 2 + (hpelIters[p->subpelRefine] + 1) / 2;
        e->refLagRows = 1 + ((range + 63) / 64);
        if (!getenv("X265AMD_FRAME_THREADS"))
        {
            /* every CTU row in flight holds a device job queue; pictures in flight never wait for one */
            const int rowsInFlight = p->bEnableWavefront ? std::max(1, std::min(e->ctuH, (e->ctuW + 1) / 2)) : 1;
            e->frameThreads = std::max(2, std::min(16, (queues > 0 ? queues : 16) / rowsInFlight));
        }
    }
    e->me = x265amd_me_open();
    if (!e->me) return nullptr;
    e->laPool.reset(new LaPool);
    e->laPool->one = (((size_t)e->lowCuW * e->lowCuH * 4) + 255) & ~(size_t)255;
    /* adaptive quantisation, cuTree, the rate factor (Encoder::configure / initPPS, RateControl::RateControl, Lookahead::Lookahead) */
    e->aqOn = p->aqMode != 0 && p->rateControlMode == X265AMD_RC_CRF;
    e->useDqp = e->aqOn;
    e->maxCuDqpDepth = e->aqOn ? (p->qgSize == 64 ? 0 : 1) : 0;
    memset(&e->treeParams, 0, sizeof(e->treeParams));
    e->treeParams.width8 = e->lowCuW; e->treeParams.height8 = e->lowCuH; e->treeParams.fps_num = p->fpsNum; e->treeParams.fps_denom = p->fpsDenom;
    e->treeParams.b_pyramid = p->bBPyramid != 0; e->treeParams.weighted_bipred = p->bEnableWeightedBiPred != 0; e->treeParams.lookahead_depth = p->lookaheadDepth;
    e->treeParams.strength = 5.0 * (1.0 - p->qCompress);
    if (p->rateControlMode == X265AMD_RC_CRF)
    {
        x265amd_rc_params rp;
        memset(&rp, 0, sizeof(rp));
        rp.width = e->W; rp.height = e->H; rp.fps_num = p->fpsNum; rp.fps_denom = p->fpsDenom; rp.bframes = p->bframes; rp.keyframe_max = p->keyframeMax; rp.cu_tree = p->cuTree != 0;
        rp.qp_min = p->qpMin; rp.qp_max = p->qpMax; rp.rf_constant = p->rfConstant; rp.q_compress = p->qCompress; rp.ip_factor = p->ipFactor; rp.pb_factor = p->pbFactor;
        e->rateCtl = x265amd_rc_open(&rp);
        if (!e->rateCtl) { xa_fail(X265AMD_EINVAL, "encoder_open: rate control parameters"); return nullptr; }
    }
    const size_t nstat = (size_t)e->nctu * 3 * 5 * 32;
    if (hipMalloc((void**)&e->dSaoCount, nstat * 4) != hipSuccess || hipMalloc((void**)&e->dSaoOrg, nstat * 4) != hipSuccess ||
        hipMalloc((void**)&e->dSaoParams, sizeof(x265amd_sao_ctu) * e->nctu) != hipSuccess ||
        hipMalloc((void**)&e->dDbUnits, sizeof(x265amd_deblock_unit) * e->w4 * e->h4) != hipSuccess ||
        xa_scratch_alloc((void**)&e->dSaoTmp, e->picElems * sizeof(pixel)) != hipSuccess)
    { xa_fail(X265AMD_EHIP, "encoder_open: device allocation"); return nullptr; }
    x265amd_stream_params sp;
    e->fillStreamParams(sp);
    e->headerBytes.resize(512);
    const size_t n = x265amd_write_stream_headers(&sp, e->headerBytes.data(), e->headerBytes.size());
    if (!n || n > e->headerBytes.size()) { xa_fail(X265AMD_EINVAL, "encoder_open: stream headers"); return nullptr; }
    e->headerBytes.resize(n);
    if (p->bEmitHDR10SEI || p->hasMasteringDisplay || p->maxCLL || p->maxFALL)
    {
        /* Encoder::getStreamHeaders (encoder.cpp:3264-3282): content light level (payload type 144), then the mastering display colour volume (137) */
        uint8_t sei[64], payload[24];
        if (p->bEmitCLL)
        {
            payload[0] = (uint8_t)(p->maxCLL >> 8); payload[1] = (uint8_t)p->maxCLL; payload[2] = (uint8_t)(p->maxFALL >> 8); payload[3] = (uint8_t)p->maxFALL;
            const size_t m = x265amd_write_sei(0, 144, payload, 4, sei, sizeof(sei));
            if (!m) { xa_fail(X265AMD_EINVAL, "encoder_open: content light level SEI"); return nullptr; }
            e->headerBytes.insert(e->headerBytes.end(), sei, sei + m);
        }
        if (p->hasMasteringDisplay)
        {
            for (int i = 0; i < 8; i++) { payload[2 * i] = (uint8_t)(p->masteringDisplay[i] >> 8); payload[2 * i + 1] = (uint8_t)p->masteringDisplay[i]; }
            for (int i = 0; i < 2; i++)
                for (int k = 0; k < 4; k++) payload[16 + 4 * i + k] = (uint8_t)(p->masteringDisplay[8 + i] >> (24 - 8 * k));
            const size_t m = x265amd_write_sei(0, 137, payload, 24, sei, sizeof(sei));
            if (!m) { xa_fail(X265AMD_EINVAL, "encoder_open: mastering display SEI"); return nullptr; }
            e->headerBytes.insert(e->headerBytes.end(), sei, sei + m);
        }
    }
    if (p->bEmitInfoSEI)
    {
        /* Encoder::getStreamHeaders' fourth unit (encoder.cpp:3260-3280): who coded this and with what -- the reference's own text names ITS build, this one names this library */
        char text[1024];
        snprintf(text, sizeof(text), "x265amd (x265 build 209 interface) - %s - H.265/HEVC codec on AMD Instinct MI355X - options: %dx%d fps=%u/%u bitdepth=%d %s=%g aq-mode=%d aq-strength=%.2f "
                 "cutree=%d qcomp=%.2f qg-size=%d bframes=%d b-adapt=%d b-pyramid=%d open-gop=%d keyint=%d min-keyint=%d scenecut=%d rc-lookahead=%d lookahead-slices=%d ref=%d limit-refs=%d "
                 "rd=%d rdoq-level=%d psy-rd=%.2f me=%d subme=%d merange=%d max-merge=%d rect=%d amp=%d limit-modes=%d early-skip=%d rskip=%d weightp=%d weightb=%d sao=%d deblock=%d wpp=%d "
                 "tu-intra-depth=%d tu-inter-depth=%d signhide=%d strong-intra-smoothing=%d temporal-mvp=%d b-intra=%d fast-intra=%d",
                 x265amd_version(), p->sourceWidth, p->sourceHeight, p->fpsNum, p->fpsDenom, X265AMD_DEPTH, p->rateControlMode == X265AMD_RC_CRF ? "crf" : "qp",
                 p->rateControlMode == X265AMD_RC_CRF ? p->rfConstant : (double)p->qp, e->aqOn ? p->aqMode : 0, p->aqStrength, p->cuTree != 0, p->qCompress, p->qgSize, p->bframes, p->bFrameAdaptive,
                 p->bBPyramid != 0, p->bOpenGOP != 0, p->keyframeMax, e->keyframeMin, p->scenecutThreshold, p->lookaheadDepth, p->lookaheadSlices, p->maxNumReferences, p->limitReferences,
                 p->rdLevel, p->rdoqLevel, p->psyRd, p->searchMethod, p->subpelRefine, p->searchRange, p->maxNumMergeCand, p->bEnableRectInter != 0, p->bEnableAMP != 0, p->limitModes != 0,
                 p->bEnableEarlySkip != 0, p->recursionSkipMode, p->bEnableWeightedPred != 0, p->bEnableWeightedBiPred != 0, p->bEnableSAO != 0, p->bEnableLoopFilter != 0, p->bEnableWavefront != 0,
                 p->tuQTMaxIntraDepth, p->tuQTMaxInterDepth, p->bEnableSignHiding != 0, p->bEnableStrongIntraSmoothing != 0, p->bEnableTemporalMvp != 0, p->bIntraInBFrames != 0, p->bEnableFastIntra != 0);
        uint8_t sei[1400];
        const size_t m = x265amd_write_info_sei(text, sei, sizeof(sei));
        if (!m) { xa_fail(X265AMD_EINVAL, "encoder_open: info SEI"); return nullptr; }
        e->headerBytes.insert(e->headerBytes.end(), sei, sei + m);
    }
    return e.release();
}

/* ---- frame-per-GPU: rows of a picture between the objects of a set (include/x265amd_encoder.h) ---- */
static PicP picByCoding(x265amd_encoder* e, uint64_t k)
{
    std::lock_guard<std::mutex> lk(e->byCodingMu);
    auto it = e->byCoding.find(k);
    return it == e->byCoding.end() ? PicP() : it->second;
}
static void rowRanges(const x265amd_encoder& e, int row, uint64_t off[3], uint64_t bytes[3])
{
    for (int k = 0; k < 3; k++)
    {
        const int sh = k ? 1 : 0, my = e.marginY >> sh, mx = e.marginX >> sh, h = e.H >> sh, rowH = 64 >> sh;
        const intptr_t st = k ? e.cstride : e.stride;
        const int y0 = row == 0 ? -my : row * rowH, y1 = row == e.ctuH - 1 ? h + my : std::min(h, (row + 1) * rowH);
        const int64_t first = (int64_t)e.org[k] + (int64_t)y0 * st - mx;
        off[k] = (uint64_t)first * sizeof(pixel);
        bytes[k] = (uint64_t)((int64_t)(y1 - y0) * st) * sizeof(pixel);
    }
}
extern "C" int x265amd_encoder_ctu_rows(const x265amd_encoder* e) { return e ? e->ctuH : -1; }
extern "C" int x265amd_encoder_stats(const x265amd_encoder* e, uint64_t* out, int n)
{
    if (!e || !out || n < 4) return xa_fail(X265AMD_EINVAL, "encoder_stats: arguments"), -1;
    out[0] = e->statPictures[0]; out[1] = e->statPictures[1]; out[2] = e->statPictures[2]; out[3] = e->statReferences;
    if (n >= 13)
        for (int t = 0; t < 3; t++) { out[4 + t] = e->statEmitted[t]; out[7 + t] = e->statBits[t]; memcpy(&out[10 + t], &e->statQpSum[t], 8); }
    return 0;
}
extern "C" int x265amd_encoder_row_geometry(const x265amd_encoder* e, int row, x265amd_row_export* out)
{
    if (!e || !out || row < 0 || row >= e->ctuH) return xa_fail(X265AMD_EINVAL, "encoder_row_geometry: bad arguments"), -1;
    memset(out, 0, sizeof(*out));
    out->ctu_row = row;
    rowRanges(*e, row, out->plane_offset, out->plane_bytes);
    const size_t u0 = (size_t)row * 16 * e->w4, u1 = (size_t)std::min(e->h4, (row + 1) * 16) * e->w4;
    out->units_bytes = (u1 - u0) * sizeof(x265amd_cu_unit); out->map_offset_units = u0 * sizeof(x265amd_cu_unit);
    out->motion_bytes = (u1 - u0) * sizeof(x265amd_mv_unit); out->map_offset_motion = u0 * sizeof(x265amd_mv_unit);
    return 0;
}
extern "C" int x265amd_encoder_is_referenced(x265amd_encoder* e, uint64_t k)
{
    if (!e) return xa_fail(X265AMD_EINVAL, "encoder_is_referenced: null"), -1;
    PicP pic = picByCoding(e, k);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (k < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_is_referenced: that picture has been collected and released"), -1;
        return 2;
    }
    return pic->type != TYPE_B;         /* what DPB::prepareEncode fixes with the slice type: a plain B picture is never a reference */
}
extern "C" int x265amd_encoder_owns(const x265amd_encoder* e, uint64_t k) { return e ? (e->p.shardCount <= 1 || (int)(k % (uint64_t)e->p.shardCount) == e->p.shardRank) : 0; }
extern "C" int x265amd_encoder_export_row(x265amd_encoder* e, uint64_t codingIndex, int row, x265amd_row_export* out, int timeoutMs)
{
    if (!e || !out || row < 0 || row >= e->ctuH) return xa_fail(X265AMD_EINVAL, "encoder_export_row: bad arguments"), -1;
    if (!e->frameParallel) return xa_fail(X265AMD_EINVAL, "encoder_export_row: rows are published by objects that code pictures in parallel (frameNumThreads > 1)"), -1;
    PicP pic = picByCoding(e, codingIndex);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (codingIndex < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_export_row: that picture has been collected and released: the pump is more than eight pictures late"), -1;
        return 1;
    }
    if (!pic->owned) return xa_fail(X265AMD_EINVAL, "encoder_export_row: this object does not code that picture"), -1;
    const auto t0 = std::chrono::steady_clock::now();
    while (pic->published(row) < e->W)
    {
        if (pic->failed.load()) return xa_fail(X265AMD_EHIP, "encoder_export_row: the picture failed"), -1;
        if (timeoutMs <= 0) return 2;           /* a look, not a wait: the row is not final yet (the one-thread pump of frame_rows.py asks like this) */
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeoutMs) return xa_fail(X265AMD_EHIP, "encoder_export_row: time-out"), -1;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    /* Pic::fail() sets every counter to the end: a picture that failed before the loop was entered looks published */
    if (pic->failed.load()) return xa_fail(X265AMD_EHIP, "encoder_export_row: the picture failed"), -1;
    memset(out, 0, sizeof(*out));
    out->coding_index = codingIndex; out->ctu_row = row;
    rowRanges(*e, row, out->plane_offset, out->plane_bytes);
    for (int k = 0; k < 3; k++) out->src[k] = (const uint8_t*)pic->finalPlanes() + out->plane_offset[k];
    const size_t u0 = (size_t)row * 16 * e->w4, u1 = (size_t)std::min(e->h4, (row + 1) * 16) * e->w4;
    out->units = pic->units.data() + u0; out->units_bytes = (u1 - u0) * sizeof(x265amd_cu_unit); out->map_offset_units = u0 * sizeof(x265amd_cu_unit);
    out->motion = pic->motion.data() + u0; out->motion_bytes = (u1 - u0) * sizeof(x265amd_mv_unit); out->map_offset_motion = u0 * sizeof(x265amd_mv_unit);
    return 0;
}
extern "C" int x265amd_encoder_import_row(x265amd_encoder* e, const x265amd_row_export* in)
{
    if (!e || !in || in->ctu_row < 0 || in->ctu_row >= e->ctuH || !in->src[0] || !in->src[1] || !in->src[2] || !in->units || !in->motion) return xa_fail(X265AMD_EINVAL, "encoder_import_row: bad arguments"), -1;
    PicP pic = picByCoding(e, in->coding_index);
    if (!pic)
    {
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        if (in->coding_index < e->collectedCoding) return xa_fail(X265AMD_EINVAL, "encoder_import_row: that picture has been collected and released"), -1;
        return 1;
    }
    if (pic->owned) return xa_fail(X265AMD_EINVAL, "encoder_import_row: this object codes that picture itself"), -1;
    uint64_t off[3], bytes[3];
    rowRanges(*e, in->ctu_row, off, bytes);
    for (int k = 0; k < 3; k++)
        if (off[k] != in->plane_offset[k] || bytes[k] != in->plane_bytes[k]) return xa_fail(X265AMD_EINVAL, "encoder_import_row: the row comes from a picture of another geometry"), -1;
    if (in->map_offset_units + in->units_bytes > pic->units.size() * sizeof(x265amd_cu_unit) || in->map_offset_motion + in->motion_bytes > pic->motion.size() * sizeof(x265amd_mv_unit))
        return xa_fail(X265AMD_EINVAL, "encoder_import_row: map range"), -1;
    xa_thread_device();
    uint8_t* dst = (uint8_t*)(pic->dFin ? pic->dFin : pic->dRec);
    {
        /* a stream of the object's own, waited for: a device-to-device hipMemcpy may return before the bytes have landed, and the counter below lets readers in */
        std::lock_guard<std::mutex> lk(e->importMu);
        if (!e->importStream && hipStreamCreateWithFlags(&e->importStream, hipStreamNonBlocking) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_import_row: stream"), -1;
        for (int k = 0; k < 3; k++)
            if (hipMemcpyAsync(dst + off[k], in->src[k], bytes[k], hipMemcpyDefault, e->importStream) != hipSuccess) { pic->fail(); return xa_fail(X265AMD_EHIP, "encoder_import_row: copy"), -1; }
        if (hipStreamSynchronize(e->importStream) != hipSuccess) { pic->fail(); return xa_fail(X265AMD_EHIP, "encoder_import_row: copy"), -1; }
    }
    memcpy((uint8_t*)pic->units.data() + in->map_offset_units, in->units, in->units_bytes);
    memcpy((uint8_t*)pic->motion.data() + in->map_offset_motion, in->motion, in->motion_bytes);
    xa_devmap_push_rows(pic->motion.data(), pic->units.data(), e->w4, in->ctu_row * 16, std::min(e->h4, (in->ctu_row + 1) * 16));      /* the field's mirror in device memory */
    pic->publish(in->ctu_row, e->W);
    { std::lock_guard<std::mutex> lk(pic->mu); pic->importedRows++; }
    pic->cv.notify_all();
    return 0;
}

extern "C" void x265amd_encoder_close(x265amd_encoder* e) { delete e; }

/* splits a byte stream of NAL units behind 4-byte start codes into x265_nal records (payload includes the start code, as the reference's do) */
static void splitNals(std::vector<uint8_t>& bytes, std::vector<x265amd_nal>& nals)
{
    /* NAL units behind start codes of four bytes (the first unit of an access unit, parameter sets) or three (NALList::serialize, nal.cpp:85-160).  Emulation prevention keeps
     * 00 00 01 out of every payload. */
    nals.clear();
    std::vector<size_t> starts;
    for (size_t i = 0; i + 3 <= bytes.size(); i++)
        if (!bytes[i] && !bytes[i + 1] && bytes[i + 2] == 1) { starts.push_back(i > 0 && !bytes[i - 1] && (starts.empty() || starts.back() + 3 <= i - 1) ? i - 1 : i); i += 2; }
    for (size_t k = 0; k < starts.size(); k++)
    {
        const size_t start = starts[k], end = k + 1 < starts.size() ? starts[k + 1] : bytes.size();
        const size_t hdr = start + (bytes[start + 2] == 1 ? 3 : 4);
        x265amd_nal n;
        n.type = (bytes[hdr] >> 1) & 63; n.sizeBytes = (uint32_t)(end - start); n.payload = bytes.data() + start;
        nals.push_back(n);
    }
}

extern "C" int x265amd_encoder_headers(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal)
{
    if (!e || !ppNal || !piNal) return xa_fail(X265AMD_EINVAL, "encoder_headers: null argument");
    e->outBytes = e->headerBytes;
    splitNals(e->outBytes, e->nals);
    *ppNal = e->nals.data(); *piNal = (uint32_t)e->nals.size();
    return (int)e->outBytes.size();
}

/* ---- pictures ---- */
int x265amd_encoder::uploadPicture(const x265amd_picture* in, Pic& pic, bool onDevice)
{
    if (onDevice)
    {
        /* the planes are device memory: the picture area by a device-to-device copy, the margins by the kernel that extends the reconstructed planes (the same edge
         * replication: PicYuv::copyFromPicture pads, extendPicBorder); what lies outside the margins is zero as in the host form */
        if (xa_scratch_alloc((void**)&pic.dSrc, picElems * sizeof(pixel)) != hipSuccess || xa_scratch_alloc((void**)&pic.dRec, picElems * sizeof(pixel)) != hipSuccess)
            return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
        if (hipMemsetAsync(pic.dSrc, 0, picElems * sizeof(pixel), nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
        for (int k = 0; k < 3; k++)
        {
            const int w = k ? W / 2 : W, h = k ? H / 2 : H, mx = k ? marginX / 2 : marginX, my = k ? marginY / 2 : marginY;
            const intptr_t st = k ? cstride : stride;
            const int sw = k ? srcW / 2 : srcW, sh = k ? srcH / 2 : srcH;
            if (!in->planes[k] || in->stride[k] < (int)(sw * sizeof(pixel))) return xa_fail(X265AMD_EINVAL, "encoder_encode: input plane");
            if (hipMemcpy2DAsync(pic.dSrc + org[k], (size_t)st * sizeof(pixel), in->planes[k], (size_t)in->stride[k], (size_t)sw * sizeof(pixel), (size_t)sh, hipMemcpyDeviceToDevice, nullptr) != hipSuccess)
                return xa_fail(X265AMD_EHIP, "encoder_encode: the input picture is not device memory of this device");
            /* the pad up to the coded size first (the same kernel with the pad as its margin: what it writes left of and above the picture the second call overwrites) */
            if ((sw != w || sh != h) && x265amd_extend_pic_border(nullptr, (x265amd_pixel*)(pic.dSrc + org[k]), st, sw, sh, w - sw, h - sh) != X265AMD_OK) return -1;
            if (x265amd_extend_pic_border(nullptr, (x265amd_pixel*)(pic.dSrc + org[k]), st, w, h, mx, my) != X265AMD_OK) return -1;
        }
        if (hipMemsetAsync(pic.dRec, 0, picElems * sizeof(pixel), nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
        if (frameParallel && p.bEnableSAO)
        {
            if (xa_scratch_alloc((void**)&pic.dFin, picElems * sizeof(pixel)) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
            if (hipMemsetAsync(pic.dFin, 0, picElems * sizeof(pixel), nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
        }
        /* the frame tasks run on their own non-blocking streams: make sure the picture is in place before one can start */
        if (hipStreamSynchronize(nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: the input picture's copy");
        return 0;
    }
    /* the picture area of each plane with its margins filled by edge replication (PicYuv::copyFromPicture pads, extendPicBorder): put together in a pinned buffer of the
     * encoder's own (the copy below goes straight from it; what lies outside the margins was zeroed once and is never written) */
    if (!uploadBuf)
    {
        if (hipHostMalloc((void**)&uploadBuf, picElems * sizeof(pixel), hipHostMallocDefault) != hipSuccess) { uploadBuf = nullptr; return xa_fail(X265AMD_EHIP, "encoder_encode: pinned input buffer"); }
        memset(uploadBuf, 0, picElems * sizeof(pixel));
    }
    for (int k = 0; k < 3; k++)
    {
        const int w = k ? W / 2 : W, h = k ? H / 2 : H, mx = k ? marginX / 2 : marginX, my = k ? marginY / 2 : marginY;
        const intptr_t st = k ? cstride : stride;
        const int sw = k ? srcW / 2 : srcW, sh = k ? srcH / 2 : srcH;            /* what the caller hands over; w x h is what is coded */
        if (!in->planes[k] || in->stride[k] < (int)(sw * sizeof(pixel))) return xa_fail(X265AMD_EINVAL, "encoder_encode: input plane");
        pixel* base = uploadBuf + org[k];
        for (int y = 0; y < h; y++)
        {
            const pixel* src = (const pixel*)((const uint8_t*)in->planes[k] + (size_t)(y < sh ? y : sh - 1) * in->stride[k]);
            pixel* row = base + (intptr_t)y * st;
            memcpy(row, src, sizeof(pixel) * sw);
            for (int x = sw; x < w; x++) row[x] = row[sw - 1];
            for (int x = 1; x <= mx; x++) { row[-x] = row[0]; row[w - 1 + x] = row[w - 1]; }
        }
        for (int y = 1; y <= my; y++)
        {
            memcpy(base + (intptr_t)(-y) * st - mx, base - mx, sizeof(pixel) * (w + 2 * mx));
            memcpy(base + (intptr_t)(h - 1 + y) * st - mx, base + (intptr_t)(h - 1) * st - mx, sizeof(pixel) * (w + 2 * mx));
        }
    }
    if (xa_scratch_alloc((void**)&pic.dSrc, picElems * sizeof(pixel)) != hipSuccess || xa_scratch_alloc((void**)&pic.dRec, picElems * sizeof(pixel)) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
    if (hipMemcpy(pic.dSrc, uploadBuf, picElems * sizeof(pixel), hipMemcpyHostToDevice) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: upload");
    /* the frame tasks run on their own non-blocking streams: make sure the picture is in place before one can start */
    if (hipMemset(pic.dRec, 0, picElems * sizeof(pixel)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    if (frameParallel && p.bEnableSAO)
    {
        if (xa_scratch_alloc((void**)&pic.dFin, picElems * sizeof(pixel)) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: device allocation");
        if (hipMemset(pic.dFin, 0, picElems * sizeof(pixel)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return xa_fail(X265AMD_EHIP, "encoder_encode: memset");
    }
    return 0;
}

static int encoder_encode_impl(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal, const x265amd_picture* picIn, x265amd_picture* picOut, bool inputOnDevice);
extern "C" int x265amd_encoder_encode(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal, const x265amd_picture* picIn, x265amd_picture* picOut)
{
    return encoder_encode_impl(e, ppNal, piNal, picIn, picOut, false);
}
extern "C" int x265amd_encoder_encode_device(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal, const x265amd_picture* picIn, x265amd_picture* picOut)
{
    return encoder_encode_impl(e, ppNal, piNal, picIn, picOut, true);
}
static int encoder_encode_impl(x265amd_encoder* e, x265amd_nal** ppNal, uint32_t* piNal, const x265amd_picture* picIn, x265amd_picture* picOut, bool inputOnDevice)
{
    if (!e) return xa_fail(X265AMD_EINVAL, "encoder_encode: null encoder");
    if (ppNal) *ppNal = nullptr;
    if (piNal) *piNal = 0;
    if (picIn)
    {
        PicP pic(new Pic);
        pic->pool = e->laPool;
        if (e->firstInMs < 0) e->firstInMs = Pic::pubClockMs();
        pic->poc = e->frameCount++;
        const auto tu0 = std::chrono::steady_clock::now();
        int rc = e->uploadPicture(picIn, *pic, inputOnDevice);
        if (rc) return -1;
        const auto tl0 = std::chrono::steady_clock::now();
        e->uploadMs += std::chrono::duration<double, std::milli>(tl0 - tu0).count();
        if (e->lookahead && (rc = e->lowresInit(*pic)) != X265AMD_OK) return -1;
        e->laInitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count();
        e->input.push_back(pic);
    }
    const bool flushing = picIn == nullptr;
    /* The slice-type decisions.  With pictures coming in, whatever a full queue allows.  When the caller flushes, ONE mini-GOP at a time (below): each decision of
     * the lookahead takes as long as a P picture, and the pictures of the first mini-GOP have no reason to wait for the decisions about the last -- a clip shorter than
     * the lookahead is decided entirely while it is flushed, and its first P picture used to start when the last decision was made (X265AMD_FLUSH_DECIDE_ALL=1: that
     * form).  The decisions themselves do not depend on when they are made. */
    static const bool decideAll = getenv("X265AMD_FLUSH_DECIDE_ALL") && atoi(getenv("X265AMD_FLUSH_DECIDE_ALL")) != 0;
    auto decide = [e, flushing]() -> int {
        const auto tl1 = std::chrono::steady_clock::now();
        int rc = X265AMD_OK;
        if (e->lookahead) rc = e->decideLookahead(flushing, flushing && !decideAll ? 1 : 1 << 30);
        else e->decideMiniGop(flushing);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl1).count();
        e->laDecideMs += ms;
        static const bool timingD = getenv("X265AMD_TIMING") != nullptr;
        if (timingD && ms > 0.5) fprintf(stderr, "x265amd: decision: %.1f ms, %d pictures typed, %d still in the lookahead%s\n", ms, (int)e->ready.size(), (int)e->input.size(), flushing ? " (flushing)" : "");
        return rc;
    };
    /* every typed picture is prepared in coding order here (DPB::prepareEncode is bookkeeping: it does not wait for any picture to be coded); the frame itself is a task */
    auto admit = [e]() -> int {
        while (!e->ready.empty())
        {
            PicP pic = e->ready.front();
            e->ready.pop_front();
            if (e->prepare(pic)) return -1;
            pic->codingOrder = e->codingCount++;
            pic->owned = e->p.shardCount <= 1 || (int)(pic->codingOrder % (uint64_t)e->p.shardCount) == e->p.shardRank;
            e->inflight.push_back(pic);
            { std::lock_guard<std::mutex> lk(e->byCodingMu); e->byCoding[pic->codingOrder] = pic; }      /* stays until the picture has been collected (below): however many pictures are in flight */
        }
        return 0;
    };
    if (decide() != X265AMD_OK || admit()) return -1;
    const bool timing = getenv("X265AMD_TIMING") != nullptr;
    auto start = [e, timing](const PicP& pic) {
        std::shared_future<int> prev = e->lastTask;
        pic->started = true;
        if (timing) fprintf(stderr, "x265amd: poc %d handed to a frame task at %.1f ms (%d running)\n", pic->poc, Pic::pubClockMs(), e->running);
        pic->done = std::async(std::launch::async, [e, pic, prev, timing]() {
            xa_thread_device();
            const auto t0 = std::chrono::steady_clock::now();
            if (!pic->owned)
            {
                /* another object codes this picture: its rows arrive through x265amd_encoder_import_row -- unless nobody will ever read them (a plain B picture is
                 * no reference: the row pump does not send it) */
                if (pic->type == TYPE_B) { if (!e->keepSources()) { xa_scratch_free(pic->dSrc); pic->dSrc = nullptr; } return (int)X265AMD_OK; }
                std::unique_lock<std::mutex> lk(pic->mu);
                static const int importWaitS = getenv("X265AMD_IMPORT_WAIT_S") ? atoi(getenv("X265AMD_IMPORT_WAIT_S")) : 300;       /* (debugging a stalled pump: a short wait shows where it stands) */
                const bool ok = pic->cv.wait_for(lk, std::chrono::seconds(importWaitS), [&] { return pic->importedRows >= e->ctuH || pic->failed.load(); });
                if (!ok || pic->failed.load()) { lk.unlock(); pic->fail(); return xa_fail(X265AMD_EHIP, "encoder: a picture coded elsewhere did not arrive"); }
                lk.unlock();
                /* (weightAnalyse of later pictures reads the chroma planes of its references' SOURCE pictures on every object of the set: sliceWeights) */
                if (!e->keepSources()) { xa_scratch_free(pic->dSrc); pic->dSrc = nullptr; }
                return (int)X265AMD_OK;
            }
            const int rc = e->frameParallel ? e->runFrameParallel(pic) : e->runFrame(pic, prev);
            { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); e->cpuPictureNs += (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
            if (timing)
                fprintf(stderr, "x265amd: poc %d type %d qp %d: %.2f ms\n", pic->poc, pic->type, pic->sliceQp,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            return rc;
        }).share();
        e->lastTask = pic->done;
        e->running++;
    };
    /* Tasks start in coding order while fewer than frameThreads + 1 run.  Coded in parallel, a picture without references does not wait for its turn: nothing it needs
     * comes from another picture, and an I picture takes as long as a dozen of the others -- started when the lookahead hands it over, it is coded beside the pictures
     * in front of it instead of holding up the ones behind it (X265AMD_EARLY_I=0: in turn).  Output stays in coding order. */
    static const bool earlyI = !(getenv("X265AMD_EARLY_I") && atoi(getenv("X265AMD_EARLY_I")) == 0);
    static const int earlyIMax = getenv("X265AMD_EARLY_I_MAX") ? atoi(getenv("X265AMD_EARLY_I_MAX")) : 1 << 20;
    static const bool earlyP = !(getenv("X265AMD_EARLY_P") && atoi(getenv("X265AMD_EARLY_P")) == 0);
    static const bool earlyPAlways = getenv("X265AMD_EARLY_P") && atoi(getenv("X265AMD_EARLY_P")) == 2;
    static const bool earlyBref = getenv("X265AMD_EARLY_BREF") && atoi(getenv("X265AMD_EARLY_BREF")) != 0;      /* a referenced B picture is a link of the same chain */
    static const int earlyPMax = getenv("X265AMD_EARLY_P_MAX") ? atoi(getenv("X265AMD_EARLY_P_MAX")) : 6;
    /* (measured, profiles/r05_early_b_sweep.txt: 2160p clips 15-22 % shorter with 12, 8-bit, Main 10 and --preset slow alike; the 1080p clips unchanged or -- the
     * sixty-frame clip with its two scene cuts -- 15 % longer: there an I picture behind a scene cut shares the device with a dozen pictures more.  So: by size) */
    const int earlyBMax = getenv("X265AMD_EARLY_B_MAX") ? atoi(getenv("X265AMD_EARLY_B_MAX")) : (e->ctuH > 24 ? 12 : 0);
    auto launch = [&]() {
    const bool headLong = !e->inflight.empty() && (e->inflight.front()->type == TYPE_IDR || e->inflight.front()->type == TYPE_I) && e->inflight.front()->started &&
                          e->inflight.front()->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready;
    for (auto& q : e->inflight)
    {
        if (q->started) continue;
        /* the first picture in coding order always runs: it is the one collected next, whatever started ahead of its turn */
        if (e->running <= e->frameThreads || q == e->inflight.front()) { start(q); if (e->frameThreads <= 1) q->done.wait(); continue; }
        if (!(e->frameParallel && earlyI)) break;
        if (q->type == TYPE_IDR || q->type == TYPE_I)
        {
            /* (X265AMD_EARLY_I_MAX: no more of them at once than this -- an experiment of round 5's end: while P pictures started ahead of their turn all the time, a limit
             * of one helped a long 2160p clip; with the P pictures held to their turn outside an I picture's time it does not, and there is none.  profiles/r05_sched_sweep.txt) */
            int runningI = 0;
            for (auto& o : e->inflight)
                if (o->started && (o->type == TYPE_IDR || o->type == TYPE_I) && o->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready) runningI++;
            if (runningI < earlyIMax) start(q);
        }
        /* The P pictures are the chain every other picture hangs on (each follows its reference by a few CTU rows, the B pictures between two of them follow both): a P
         * picture held back until the B pictures in front of it have been collected starts with nothing to trail and takes its full latency, so it starts when the
         * lookahead hands it over, too (every picture it references is in front of it in coding order and therefore started; X265AMD_EARLY_P=0: in turn). */
        /* (round 5's end: like the B pictures below, only while a running I picture holds the head of the coding order -- X265AMD_EARLY_P=2: always, as round 4 had it.
         * With P pictures ahead of their turn ALL the time a long clip stood at 52 frames/s where it reaches 100 without: the pictures far ahead held the places
         * and the queues that the pictures collected next were waiting for; profiles/r05_sched_sweep.txt) */
        else if (earlyP && (headLong || earlyPAlways) && (q->type == TYPE_P || (earlyBref && q->type == TYPE_BREF)) && e->running <= e->frameThreads + earlyPMax) start(q);
        /* The B pictures, too (round 5), while an I picture that still runs holds the head of the coding order: `running` counts every picture that trails it and is not
         * collected yet (collection is in coding order), and the B pictures of the mini-GOPs whose P pictures ran already waited for the I picture's END although their
         * references were rows ahead of them -- at 2160p a third of a twenty-frame clip, at --preset slow more.  How many pictures run side by side changes nothing in
         * what they code (the vertical reach of the vectors follows from the parameter frameNumThreads, not from this count): up to X265AMD_EARLY_B_MAX more than the
         * parameter (0: in turn).  Only then, and only for large pictures (see earlyBMax above). */
        else if (earlyBMax > 0 && headLong && e->running <= e->frameThreads + earlyBMax) start(q);
    }
    };
    static const bool holdUntilFlush = getenv("X265AMD_HOLD_UNTIL_FLUSH") != nullptr;      /* an experiment: no picture starts before the caller flushes (what the clip costs when every decision is made beforehand) */
    if (holdUntilFlush)
    {
        if (!flushing) return 0;
        while (!e->input.empty()) { const size_t before = e->input.size(); if (decide() != X265AMD_OK || admit()) return -1; if (e->input.size() >= before) break; }
        static bool said = false;
        if (!said) { said = true; fprintf(stderr, "x265amd: every decision made %.1f ms after the encoder's first picture came in; the pictures start now\n", Pic::pubClockMs() - e->firstInMs); }
    }
    launch();
    /* flushing: the next mini-GOP is decided while the picture the caller will get next is still being coded */
    while (flushing && !e->input.empty())
    {
        if (!e->inflight.empty() && e->inflight.front()->started && e->inflight.front()->done.wait_for(std::chrono::seconds(0)) == std::future_status::ready) break;
        const size_t before = e->input.size();
        if (decide() != X265AMD_OK || admit()) return -1;
        launch();
        if (e->input.size() >= before) break;
    }
    if (e->inflight.empty()) return 0;
    PicP front = e->inflight.front();
    if (!front->started) { xa_fail(X265AMD_EINVAL, "encoder_encode: the first picture in coding order has no task"); return -1; }
    int waiting = 0;
    for (auto& q : e->inflight) waiting += !q->started;
    /* the caller is held when enough pictures run and enough wait behind them (the lookahead may run ahead of the frame tasks by a window of its own) */
    const bool mustWait = !picIn || (e->running > e->frameThreads && waiting > 2 * e->p.lookaheadDepth + 8);
    if (!mustWait && front->done.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return 0;
    const int rc = front->done.get();
    e->inflight.pop_front();
    e->running--;
    {
        /* a collected picture is complete -- every row exported or imported -- so the row pump has no more business with it; a few stay for a pump that asks late */
        std::lock_guard<std::mutex> lk(e->byCodingMu);
        e->collectedCoding = front->codingOrder + 1;
        while (!e->byCoding.empty() && e->byCoding.begin()->first + 8 < e->collectedCoding) e->byCoding.erase(e->byCoding.begin());
    }
    if (rc) { xa_fail(rc, "encoder_encode: a frame task failed"); return -1; }
    if (const char* dumpPath = getenv("X265AMD_RC_DUMP"))
    {
        /* debugging aid: the picture's record in the layout of oracle/ref_rc_dump.cpp (the reference's decisions for the same picture), appended to the file named */
        if (FILE* f = fopen(dumpPath, "ab"))
        {
            const Pic& q = *front;
            const int blocks16 = ((e->W + 15) / 16) * ((e->H + 15) / 16), lowresBlocks = e->lowCuW * e->lowCuH;
            int32_t hdr[44];
            memset(hdr, 0, sizeof(hdr));
            hdr[0] = 0x52434450; hdr[1] = q.poc; hdr[2] = q.type; hdr[3] = q.type != TYPE_B; hdr[4] = q.sliceQp; hdr[5] = q.bScenecut;
            for (int l = 0; l < 2; l++) { hdr[6 + l] = (int32_t)q.lists[l].size(); for (size_t r = 0; r < q.lists[l].size() && r < 16; r++) hdr[8 + 16 * l + r] = q.lists[l][r]->poc; }
            hdr[40] = e->w4; hdr[41] = e->h4; hdr[42] = blocks16; hdr[43] = lowresBlocks;
            fwrite(hdr, sizeof(hdr), 1, f);
            const int64_t satd = 0; fwrite(&satd, 8, 1, f);
            const double qq[2] = { q.avgQpRc, 0 }; fwrite(qq, 8, 2, f);
            std::vector<double> zd((size_t)blocks16, 0.0); std::vector<int32_t> zi((size_t)std::max(blocks16, lowresBlocks), 0); std::vector<uint16_t> zs((size_t)lowresBlocks, 0);
            fwrite(q.qpAqOffset.size() == (size_t)blocks16 ? q.qpAqOffset.data() : zd.data(), 8, blocks16, f);
            fwrite(q.qpCuTreeOffset.size() == (size_t)blocks16 ? q.qpCuTreeOffset.data() : zd.data(), 8, blocks16, f);
            fwrite(q.invQscale.size() == (size_t)blocks16 ? q.invQscale.data() : zi.data(), 4, blocks16, f);
            fwrite(q.intraCostHost.size() == (size_t)lowresBlocks ? q.intraCostHost.data() : zi.data(), 4, lowresBlocks, f);
            fwrite(q.propagateCost.size() == (size_t)lowresBlocks ? q.propagateCost.data() : zs.data(), 2, lowresBlocks, f);
            const size_t n = (size_t)e->w4 * e->h4;
            std::vector<uint8_t> b(n);
            for (size_t i = 0; i < n; i++) b[i] = (uint8_t)q.units[i].qp; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].depth; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].pred_mode; fwrite(b.data(), 1, n, f);
            for (size_t i = 0; i < n; i++) b[i] = q.units[i].cbf[0]; fwrite(b.data(), 1, n, f);
            fclose(f);
        }
    }
    e->outBytes.swap(front->nalBytes);
    /* --repeat-headers (and every all-intra encode): the parameter sets in front of a keyframe's slice units (FrameEncoder::compressFrame, frameencoder.cpp:465-480) */
    const bool withAud = e->p.bEnableAccessUnitDelimiters && (front->poc || e->p.bRepeatHeaders) && !e->outBytes.empty();
    const bool withHeaders = e->p.bRepeatHeaders && front->bKeyframe && !e->outBytes.empty();
    if (withAud || withHeaders)
    {
        /* ... and the slice units are no longer the first of their access unit: start codes of three bytes (nal.cpp:110-118).  --aud: the delimiter opens the access unit
         * (frameencoder.cpp:497-506), the parameter sets follow it */
        std::vector<uint8_t> au;
        if (withAud)
        {
            uint8_t aud[16];
            const size_t m = x265amd_write_aud(isBType(front->type) ? 0 : front->type == TYPE_P ? 1 : 2, aud, sizeof(aud));
            au.insert(au.end(), aud, aud + m);
        }
        if (withHeaders) au.insert(au.end(), e->headerBytes.begin(), e->headerBytes.end());
        const std::vector<uint8_t>& b = e->outBytes;
        for (size_t i = 0; i < b.size(); i++)
        {
            if (i + 4 <= b.size() && !b[i] && !b[i + 1] && !b[i + 2] && b[i + 3] == 1) continue;         /* the zero_byte in front of a start code goes */
            au.push_back(b[i]);
        }
        e->outBytes.swap(au);
    }
    bool haveStaging = false;
    if (e->p.decodedPictureHashSEI && !e->outBytes.empty())
    {
        /* --hash: the digest of the finished picture in a suffix SEI unit behind its slice units (FrameEncoder::writeTrailingSEIMessages, frameencoder.cpp:418-460) */
        e->staging.resize(e->picElems);
        if (hipMemcpy(e->staging.data(), front->finalPlanes(), e->picElems * sizeof(pixel), hipMemcpyDeviceToHost) != hipSuccess) { xa_fail(X265AMD_EHIP, "encoder: recon download"); return -1; }
        haveStaging = true;
        const void* planes[3] = { e->staging.data() + e->org[0], e->staging.data() + e->org[1], e->staging.data() + e->org[2] };
        const intptr_t strides[3] = { (intptr_t)(e->stride * sizeof(pixel)), (intptr_t)(e->cstride * sizeof(pixel)), (intptr_t)(e->cstride * sizeof(pixel)) };
        uint8_t payload[64], sei[96];
        const size_t n = x265amd_picture_hash(e->p.decodedPictureHashSEI, planes, strides, e->W, e->H, X265AMD_DEPTH, 64, payload, sizeof(payload));
        const size_t m = n ? x265amd_write_sei(1, 132, payload, n, sei, sizeof(sei)) : 0;
        if (!m) { xa_fail(X265AMD_EINVAL, "encoder: picture hash SEI"); return -1; }
        e->outBytes.insert(e->outBytes.end(), sei, sei + m);
    }
    splitNals(e->outBytes, e->nals);
    if (!e->outBytes.empty())
    {
        /* what x265_encoder_get_stats totals per slice type (Encoder::finishFrameStats, encoder.cpp:2960-3060): the access unit's bits and the picture's average QP -- the mean of
         * its CUs' QPs weighted by their size (FrameEncoder::collectCTUStatistics; taken over the picture's own 4x4 units here, the reference also counts the absent units of
         * cut CTUs) */
        const int t = front->type == TYPE_B || front->type == TYPE_BREF ? 2 : (front->type == TYPE_P ? 1 : 0);
        const size_t n = (size_t)e->w4 * e->h4;
        double q = 0;
        if (e->useDqp && front->units.size() >= n) { int64_t sum = 0; for (size_t i = 0; i < n; i++) sum += front->units[i].qp; q = (double)sum / (double)n; }
        else q = front->sliceQp;
        e->statEmitted[t]++; e->statBits[t] += (uint64_t)e->outBytes.size() * 8; e->statQpSum[t] += q;
    }
    if (picOut)
    {
        e->staging.resize(e->picElems);
        if (!haveStaging && hipMemcpy(e->staging.data(), front->finalPlanes(), e->picElems * sizeof(pixel), hipMemcpyDeviceToHost) != hipSuccess) { xa_fail(X265AMD_EHIP, "encoder: recon download"); return -1; }
        for (int k = 0; k < 3; k++)
        {
            if (!picOut->planes[k]) continue;
            const int w = k ? e->srcW / 2 : e->srcW, hh = k ? e->srcH / 2 : e->srcH;           /* the picture inside the conformance window: what the caller's planes hold */
            const intptr_t st = k ? e->cstride : e->stride;
            for (int y = 0; y < hh; y++)
                memcpy((uint8_t*)picOut->planes[k] + (size_t)y * picOut->stride[k], e->staging.data() + e->org[k] + (intptr_t)y * st, sizeof(pixel) * w);
        }
        picOut->poc = front->poc; picOut->sliceType = front->type; picOut->qp = front->sliceQp;
    }
    /* a finished picture that nobody references any more releases its lists (and with them the pictures only it kept alive) */
    front->lists[0].clear(); front->lists[1].clear(); front->neg.clear(); front->pos.clear();
    if (ppNal) *ppNal = e->nals.data();
    if (piNal) *piNal = (uint32_t)e->nals.size();
    return 1;
}
