/* The reference's public ABI over the encoder object: `x265_api_get_209` / `x265_api_query` returning a table with the layout of `struct x265_api`
 * (reference: source/x265.h:2561-2614; the library's own table `libapi`, source/encoder/api.cpp:1034-1085; the two entry points api.cpp:1107-1279), so that a
 * libx265 client can dlopen libx265amd_main.so / libx265amd_main10.so in place of libx265_main.so / libx265_main10.so.  Host C++.
 *
 * The table's encoder entries take the reference's own structures: x265_encoder_open reads a real `x265_param` (x265.h:1034-2275), x265_encoder_encode real
 * `x265_picture`s (x265.h:397-490).  Their members are read at the byte offsets of x265_abi_layout.h, which oracle/gen_abi_layout.cpp GENERATES from the
 * reference's header (numbers only) and tests/layout_check.cpp pins against it again; the reference's header itself is not part of this build.
 *
 * Built entries: param_alloc / param_free, picture_alloc / picture_free / picture_init, encoder_open, encoder_parameters, encoder_headers, encoder_encode,
 * encoder_get_stats (zeroes), encoder_log (no-op), encoder_close, cleanup.  Entries of features outside the built subset FAIL CLEANLY: param_default fills the
 * reference's defaults for the members encoder_open reads and zeroes the rest, param_default_preset accepts NULL / "medium" only (the other presets' option
 * tables are not restated), param_parse / zone_param_parse / scenecut_aware_qp_param_parse report a bad name, encoder_reconfig* / intra_refresh / ctu_info /
 * get_slicetype_poc_and_scenecut / get_ref_frame_list / set_analysis_data return -1, csvlog_open returns NULL.  encoder_open rejects every parameter outside the
 * subset BY NAME (x265amd_last_error).
 */
#include "../../include/x265amd.h"
#include "../../include/x265amd_encoder.h"
#include "x265_abi_layout.h"
#include "x265_api_table.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>
#include <time.h>
#include <vector>

int xa_fail(int code, const char* msg);

namespace {

template<typename T> inline T rd(const void* base, size_t off) { T v; memcpy(&v, (const char*)base + off, sizeof(T)); return v; }
template<typename T> inline void wr(void* base, size_t off, T v) { memcpy((char*)base + off, &v, sizeof(T)); }
#define PI(p, f) rd<int32_t>(p, X265ABI_PARAM_##f)
#define PU(p, f) rd<uint32_t>(p, X265ABI_PARAM_##f)
#define PD(p, f) rd<double>(p, X265ABI_PARAM_##f)

struct AbiEncoder
{
    x265amd_encoder* enc = nullptr;
    std::vector<uint8_t> param;         /* the caller's x265_param as it was at encoder_open (encoder_parameters hands it back) */
    int width = 0, height = 0;
    /* time stamps (Encoder::encode, encoder.cpp:1695-1697, :2286-2295; Lookahead::slicetypeDecide hands the display-order stamps of a mini-GOP to its pictures in
     * coding order, slicetype.cpp:2134-2469 -- so the i-th coded picture's m_reorderedPts is the i-th input picture's pts) */
    std::vector<int64_t> ptsIn;         /* by input (display) order */
    int64_t prevReordered[2] = { 0, 0 };
    int bframeDelay = 0;                /* Encoder::m_bframeDelay: 2 with the B pyramid, 1 with B pictures, else 0 (encoder.cpp:3945) */
    uint64_t coded = 0;                 /* pictures handed out so far */
    double fps = 25.0;
    struct timespec opened = { 0, 0 };   /* x265_stats.elapsedEncodeTime counts from encoder_open */
    std::vector<uint8_t> recon;         /* the reconstruction handed out by the last encoder_encode: valid until the next call on this encoder (the reference hands out its own picture) */
};

/* ---- x265_param ---- */
void* abi_param_alloc(void) { return calloc(1, X265ABI_SIZEOF_PARAM); }
void abi_param_free(void* p) { free(p); }

/* x265_param_default (source/common/param.cpp:100-420) for the members x265_encoder_open reads here; everything else zero */
void abi_param_default(void* p)
{
    if (!p) return;
    memset(p, 0, X265ABI_SIZEOF_PARAM);
    wr<int32_t>(p, X265ABI_PARAM_cpuid, 0); wr<int32_t>(p, X265ABI_PARAM_bEnableWavefront, 1); wr<int32_t>(p, X265ABI_PARAM_frameNumThreads, 0);
    wr<int32_t>(p, X265ABI_PARAM_logLevel, 2); wr<int32_t>(p, X265ABI_PARAM_internalBitDepth, X265AMD_DEPTH); wr<int32_t>(p, X265ABI_PARAM_internalCsp, 1);
    wr<int32_t>(p, X265ABI_PARAM_levelIdc, 0); wr<int32_t>(p, X265ABI_PARAM_bHighTier, 1); wr<int32_t>(p, X265ABI_PARAM_bAnnexB, 1); wr<int32_t>(p, X265ABI_PARAM_bEmitInfoSEI, 1);
    wr<int32_t>(p, X265ABI_PARAM_maxCUSize, 64); wr<int32_t>(p, X265ABI_PARAM_minCUSize, 8); wr<int32_t>(p, X265ABI_PARAM_maxTUSize, 32);
    wr<int32_t>(p, X265ABI_PARAM_tuQTMaxInterDepth, 1); wr<int32_t>(p, X265ABI_PARAM_tuQTMaxIntraDepth, 1);
    wr<int32_t>(p, X265ABI_PARAM_maxNumReferences, 3); wr<int32_t>(p, X265ABI_PARAM_limitReferences, 3);
    wr<int32_t>(p, X265ABI_PARAM_bOpenGOP, 1); wr<int32_t>(p, X265ABI_PARAM_keyframeMin, 0); wr<int32_t>(p, X265ABI_PARAM_keyframeMax, 250);
    wr<int32_t>(p, X265ABI_PARAM_bframes, 4); wr<int32_t>(p, X265ABI_PARAM_bFrameAdaptive, 2); wr<int32_t>(p, X265ABI_PARAM_bBPyramid, 1);
    wr<int32_t>(p, X265ABI_PARAM_lookaheadDepth, 20); wr<int32_t>(p, X265ABI_PARAM_lookaheadSlices, 8); wr<int32_t>(p, X265ABI_PARAM_scenecutThreshold, 40);
    wr<int32_t>(p, X265ABI_PARAM_searchMethod, 1); wr<int32_t>(p, X265ABI_PARAM_subpelRefine, 2); wr<int32_t>(p, X265ABI_PARAM_searchRange, 57);
    wr<int32_t>(p, X265ABI_PARAM_maxNumMergeCand, 3); wr<int32_t>(p, X265ABI_PARAM_bEnableWeightedPred, 1); wr<int32_t>(p, X265ABI_PARAM_bEnableEarlySkip, 1);
    wr<int32_t>(p, X265ABI_PARAM_recursionSkipMode, 1); wr<int32_t>(p, X265ABI_PARAM_bEnableSignHiding, 1); wr<int32_t>(p, X265ABI_PARAM_bEnableStrongIntraSmoothing, 1);
    wr<int32_t>(p, X265ABI_PARAM_bEnableTemporalMvp, 1); wr<int32_t>(p, X265ABI_PARAM_bEnableLoopFilter, 1); wr<int32_t>(p, X265ABI_PARAM_bEnableSAO, 1);
    wr<int32_t>(p, X265ABI_PARAM_rdLevel, 3); wr<int32_t>(p, X265ABI_PARAM_bIntraInBFrames, 1); wr<double>(p, X265ABI_PARAM_psyRd, 2.0); wr<double>(p, X265ABI_PARAM_psyRdoq, 0.0);
    wr<int32_t>(p, X265ABI_PARAM_bEmitCLL, 1);
    wr<int32_t>(p, X265ABI_PARAM_vui_videoFormat, 5); wr<int32_t>(p, X265ABI_PARAM_vui_colorPrimaries, 2); wr<int32_t>(p, X265ABI_PARAM_vui_transferCharacteristics, 2);
    wr<int32_t>(p, X265ABI_PARAM_vui_matrixCoeffs, 2);         /* unspecified (param.cpp:358-366) */
    wr<int32_t>(p, X265ABI_PARAM_rc_rateControlMode, 2 /* X265_RC_CRF */); wr<int32_t>(p, X265ABI_PARAM_rc_qp, 32); wr<double>(p, X265ABI_PARAM_rc_ipFactor, 1.4f);
    wr<double>(p, X265ABI_PARAM_rc_pbFactor, 1.3f); wr<int32_t>(p, X265ABI_PARAM_rc_aqMode, 2); wr<int32_t>(p, X265ABI_PARAM_rc_cuTree, 1);
    wr<double>(p, X265ABI_PARAM_rc_rfConstant, 28); wr<double>(p, X265ABI_PARAM_rc_aqStrength, 1.0); wr<double>(p, X265ABI_PARAM_rc_qCompress, 0.6);
    wr<int32_t>(p, X265ABI_PARAM_rc_qgSize, 32); wr<int32_t>(p, X265ABI_PARAM_rc_qpStep, 4);
    wr<int32_t>(p, X265ABI_PARAM_rc_qpMin, 0); wr<int32_t>(p, X265ABI_PARAM_rc_qpMax, 69); wr<int32_t>(p, X265ABI_PARAM_maxSlices, 1);
}
/* x265_param_default_preset (source/common/param.cpp:405-600): the ten presets' option tables and the tunes that only touch members read here (psnr, ssim).  Members outside
 * the built subset are written as the reference writes them -- maxCUSize 32 of ultrafast / superfast, limitTU of slower, transform skip of placebo: x265_encoder_open names
 * what it cannot code. */
int abi_param_default_preset(void* p, const char* preset, const char* tune)
{
    if (!p) return -1;
    abi_param_default(p);
#define SI(f, v) wr<int32_t>(p, X265ABI_PARAM_##f, v)
#define SD(f, v) wr<double>(p, X265ABI_PARAM_##f, v)
    if (preset)
    {
        static const char* const names[] = { "ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo" };
        char* end;
        const long i = strtol(preset, &end, 10);
        if (*end == 0 && i >= 0 && i < 10) preset = names[i];
        if (!strcmp(preset, "ultrafast"))
        {
            SI(maxNumMergeCand, 2); SI(bIntraInBFrames, 0); SI(lookaheadDepth, 5); SI(scenecutThreshold, 0); SI(maxCUSize, 32); SI(minCUSize, 16); SI(bframes, 3); SI(bFrameAdaptive, 0);
            SI(subpelRefine, 0); SI(searchMethod, 0); SI(bEnableSAO, 0); SI(bEnableSignHiding, 0); SI(bEnableWeightedPred, 0); SI(rdLevel, 2); SI(maxNumReferences, 1); SI(limitReferences, 0);
            SD(rc_aqStrength, 0.0); SI(rc_aqMode, 0); SI(rc_hevcAq, 0); SI(rc_qgSize, 32); SI(bEnableFastIntra, 1);
        }
        else if (!strcmp(preset, "superfast"))
        {
            SI(maxNumMergeCand, 2); SI(bIntraInBFrames, 0); SI(lookaheadDepth, 10); SI(maxCUSize, 32); SI(bframes, 3); SI(bFrameAdaptive, 0); SI(subpelRefine, 1); SI(bEnableWeightedPred, 0);
            SI(rdLevel, 2); SI(maxNumReferences, 1); SI(limitReferences, 0); SD(rc_aqStrength, 0.0); SI(rc_aqMode, 0); SI(rc_hevcAq, 0); SI(rc_qgSize, 32); SI(bEnableSAO, 0); SI(bEnableFastIntra, 1);
        }
        else if (!strcmp(preset, "veryfast"))
        { SI(maxNumMergeCand, 2); SI(bIntraInBFrames, 0); SI(lookaheadDepth, 15); SI(bFrameAdaptive, 0); SI(subpelRefine, 1); SI(rdLevel, 2); SI(maxNumReferences, 2); SI(rc_qgSize, 32); SI(bEnableFastIntra, 1); }
        else if (!strcmp(preset, "faster"))
        { SI(maxNumMergeCand, 2); SI(bIntraInBFrames, 0); SI(lookaheadDepth, 15); SI(bFrameAdaptive, 0); SI(rdLevel, 2); SI(maxNumReferences, 2); SI(bEnableFastIntra, 1); }
        else if (!strcmp(preset, "fast"))
        { SI(maxNumMergeCand, 2); SI(bEnableEarlySkip, 0); SI(bIntraInBFrames, 0); SI(lookaheadDepth, 15); SI(bFrameAdaptive, 0); SI(rdLevel, 2); SI(maxNumReferences, 3); SI(bEnableFastIntra, 1); }
        else if (!strcmp(preset, "medium")) { }
        else if (!strcmp(preset, "slow"))
        {
            SI(bEnableEarlySkip, 0); SI(bIntraInBFrames, 0); SI(bEnableRectInter, 1); SI(lookaheadDepth, 25); SI(rdLevel, 4); SI(rdoqLevel, 2); SD(psyRdoq, 1.0); SI(subpelRefine, 3);
            SI(searchMethod, 3); SI(maxNumReferences, 4); SI(limitModes, 1); SI(lookaheadSlices, 4);
        }
        else if (!strcmp(preset, "slower"))
        {
            SI(bEnableEarlySkip, 0); SI(bEnableWeightedBiPred, 1); SI(bEnableAMP, 1); SI(bEnableRectInter, 1); SI(lookaheadDepth, 40); SI(bframes, 8); SI(tuQTMaxInterDepth, 3); SI(tuQTMaxIntraDepth, 3);
            SI(rdLevel, 6); SI(rdoqLevel, 2); SD(psyRdoq, 1.0); SI(subpelRefine, 4); SI(maxNumMergeCand, 4); SI(searchMethod, 3); SI(maxNumReferences, 5); SI(limitModes, 1); SI(limitReferences, 1);
            SI(lookaheadSlices, 0); SI(limitTU, 4);
        }
        else if (!strcmp(preset, "veryslow"))
        {
            SI(bEnableEarlySkip, 0); SI(bEnableWeightedBiPred, 1); SI(bEnableAMP, 1); SI(bEnableRectInter, 1); SI(lookaheadDepth, 40); SI(bframes, 8); SI(tuQTMaxInterDepth, 3); SI(tuQTMaxIntraDepth, 3);
            SI(rdLevel, 6); SI(rdoqLevel, 2); SD(psyRdoq, 1.0); SI(subpelRefine, 4); SI(maxNumMergeCand, 5); SI(searchMethod, 3); SI(maxNumReferences, 5); SI(limitReferences, 0); SI(limitModes, 0);
            SI(lookaheadSlices, 0); SI(limitTU, 0);
        }
        else if (!strcmp(preset, "placebo"))
        {
            SI(bEnableEarlySkip, 0); SI(bEnableWeightedBiPred, 1); SI(bEnableAMP, 1); SI(bEnableRectInter, 1); SI(lookaheadDepth, 60); SI(searchRange, 92); SI(bframes, 8); SI(tuQTMaxInterDepth, 4);
            SI(tuQTMaxIntraDepth, 4); SI(rdLevel, 6); SI(rdoqLevel, 2); SD(psyRdoq, 1.0); SI(subpelRefine, 5); SI(maxNumMergeCand, 5); SI(searchMethod, 3); SI(bEnableTransformSkip, 1);
            SI(recursionSkipMode, 0); SI(maxNumReferences, 5); SI(limitReferences, 0); SI(lookaheadSlices, 0);
        }
        else return -1;
    }
    if (tune && *tune)
    {
        if (!strcmp(tune, "psnr")) { SD(rc_aqStrength, 0.0); SD(psyRd, 0.0); SD(psyRdoq, 0.0); }
        else if (!strcmp(tune, "ssim")) { SI(rc_aqMode, 2); SD(psyRd, 0.0); SD(psyRdoq, 0.0); }
        else if (!strcmp(tune, "fastdecode") || !strcmp(tune, "fast-decode")) { SI(bEnableLoopFilter, 0); SI(bEnableSAO, 0); SI(bEnableWeightedPred, 0); SI(bEnableWeightedBiPred, 0); SI(bIntraInBFrames, 0); }
        else if (!strcmp(tune, "zerolatency") || !strcmp(tune, "zero-latency"))
        { SI(bFrameAdaptive, 0); SI(bframes, 0); SI(lookaheadDepth, 0); SI(scenecutThreshold, 0); SI(bHistBasedSceneCut, 0); SI(rc_cuTree, 0); SI(frameNumThreads, 1); }
        else if (!strcmp(tune, "grain"))
        {
            /* (encoder_open refuses rc.bEnableGrain: its rate control is not built; the table still says what the reference's says) */
            SD(rc_ipFactor, 1.1); SD(rc_pbFactor, 1.0); SI(rc_cuTree, 0); SI(rc_aqMode, 0); SI(rc_hevcAq, 0); SI(rc_qpStep, 1); SI(rc_bEnableGrain, 1); SI(recursionSkipMode, 0);
            SD(psyRd, 4.0); SD(psyRdoq, 10.0); SI(bEnableSAO, 0); SI(rc_bEnableConstVbv, 1);
        }
        else if (!strcmp(tune, "animation"))
        {
            const int bf = PI(p, bframes);
            SI(bframes, (bf + 2) >= PI(p, lookaheadDepth) ? bf : bf + 2);
            SD(psyRd, 0.4); SD(rc_aqStrength, 0.4); SI(deblockingFilterBetaOffset, 1); SI(deblockingFilterTCOffset, 1);
        }
        else if (!strcmp(tune, "vmaf")) { }
        else return -1;
    }
    return 0;
}

/* x265_param_parse (source/common/param.cpp:845-1420) for the options whose members x265_encoder_open reads here: the long names of the reference's command line, "no-<name>"
 * for the switches, NULL or "" = true.  Returns 0, X265_PARAM_BAD_NAME (-1) for a name that is not restated, X265_PARAM_BAD_VALUE (-2). */
int abi_param_parse(void* p, const char* name, const char* value)
{
    if (!p || !name) return -1;
    char nm[64];
    size_t n = 0;
    for (; name[n] && n + 1 < sizeof(nm); n++) nm[n] = name[n] == '_' ? '-' : name[n];
    nm[n] = 0;
    const char* key = nm;
    if (key[0] == '-' && key[1] == '-') key += 2;
    bool neg = false;
    if (!strncmp(key, "no-", 3)) { neg = true; key += 3; }
    auto truth = [&](bool& bad) -> int {
        if (!value || !*value || !strcmp(value, "1") || !strcmp(value, "true") || !strcmp(value, "yes")) return neg ? 0 : 1;
        if (!strcmp(value, "0") || !strcmp(value, "false") || !strcmp(value, "no")) return neg ? 1 : 0;
        bad = true; return 0;
    };
    auto num = [&](bool& bad) -> int { if (!value || !*value) { bad = true; return 0; } char* e; const long v = strtol(value, &e, 0); if (*e) bad = true; return (int)v; };
    auto real = [&](bool& bad) -> double { if (!value || !*value) { bad = true; return 0; } char* e; const double v = strtod(value, &e); if (*e) bad = true; return v; };
    bool bad = false;
    static const struct { const char* name; size_t off; } switches[] = {
        { "wpp", X265ABI_PARAM_bEnableWavefront }, { "open-gop", X265ABI_PARAM_bOpenGOP }, { "b-pyramid", X265ABI_PARAM_bBPyramid }, { "rect", X265ABI_PARAM_bEnableRectInter },
        { "amp", X265ABI_PARAM_bEnableAMP }, { "signhide", X265ABI_PARAM_bEnableSignHiding }, { "strong-intra-smoothing", X265ABI_PARAM_bEnableStrongIntraSmoothing },
        { "temporal-mvp", X265ABI_PARAM_bEnableTemporalMvp }, { "weightp", X265ABI_PARAM_bEnableWeightedPred }, { "weightb", X265ABI_PARAM_bEnableWeightedBiPred },
        { "deblock", X265ABI_PARAM_bEnableLoopFilter }, { "sao", X265ABI_PARAM_bEnableSAO }, { "early-skip", X265ABI_PARAM_bEnableEarlySkip }, { "fast-intra", X265ABI_PARAM_bEnableFastIntra },
        { "b-intra", X265ABI_PARAM_bIntraInBFrames }, { "limit-modes", X265ABI_PARAM_limitModes }, { "cutree", X265ABI_PARAM_rc_cuTree }, { "info", X265ABI_PARAM_bEmitInfoSEI },
        { "annexb", X265ABI_PARAM_bAnnexB }, { "repeat-headers", X265ABI_PARAM_bRepeatHeaders }, { "aud", X265ABI_PARAM_bEnableAccessUnitDelimiters }, { "hdr10", X265ABI_PARAM_bEmitHDR10SEI },
        { "hdr", X265ABI_PARAM_bEmitHDR10SEI }, { "cll", X265ABI_PARAM_bEmitCLL }, { "tskip", X265ABI_PARAM_bEnableTransformSkip }, { "lossless", X265ABI_PARAM_bLossless } };
    if (!strcmp(key, "deblock") && !neg && value)
    {
        /* --deblock tc:beta | tc,beta | tc | a truth value (param.cpp:1083-1097) */
        int tc = 0, beta = 0;
        if (sscanf(value, "%d:%d", &tc, &beta) == 2 || sscanf(value, "%d,%d", &tc, &beta) == 2)
        { wr<int32_t>(p, X265ABI_PARAM_deblockingFilterTCOffset, tc); wr<int32_t>(p, X265ABI_PARAM_deblockingFilterBetaOffset, beta); wr<int32_t>(p, X265ABI_PARAM_bEnableLoopFilter, 1); return 0; }
        if (sscanf(value, "%d", &tc) == 1)
        { wr<int32_t>(p, X265ABI_PARAM_deblockingFilterTCOffset, tc); wr<int32_t>(p, X265ABI_PARAM_deblockingFilterBetaOffset, tc); wr<int32_t>(p, X265ABI_PARAM_bEnableLoopFilter, 1); return 0; }
    }
    for (const auto& sw : switches)
        if (!strcmp(key, sw.name)) { const int v = truth(bad); if (bad) return -2; wr<int32_t>(p, sw.off, v); return 0; }
    if (neg && !strcmp(key, "scenecut")) { wr<int32_t>(p, X265ABI_PARAM_scenecutThreshold, 0); return 0; }          /* --no-scenecut (param.cpp: atobool of "false") */
    if (neg) return -1;
    static const struct { const char* name; size_t off; } ints[] = {
        { "frame-threads", X265ABI_PARAM_frameNumThreads }, { "bframes", X265ABI_PARAM_bframes }, { "b-adapt", X265ABI_PARAM_bFrameAdaptive }, { "rc-lookahead", X265ABI_PARAM_lookaheadDepth },
        { "lookahead-slices", X265ABI_PARAM_lookaheadSlices }, { "scenecut", X265ABI_PARAM_scenecutThreshold }, { "keyint", X265ABI_PARAM_keyframeMax }, { "min-keyint", X265ABI_PARAM_keyframeMin },
        { "ref", X265ABI_PARAM_maxNumReferences }, { "limit-refs", X265ABI_PARAM_limitReferences }, { "rd", X265ABI_PARAM_rdLevel }, { "rdoq-level", X265ABI_PARAM_rdoqLevel },
        { "subme", X265ABI_PARAM_subpelRefine }, { "merange", X265ABI_PARAM_searchRange }, { "max-merge", X265ABI_PARAM_maxNumMergeCand }, { "tu-intra-depth", X265ABI_PARAM_tuQTMaxIntraDepth },
        { "tu-inter-depth", X265ABI_PARAM_tuQTMaxInterDepth }, { "limit-tu", X265ABI_PARAM_limitTU }, { "rskip", X265ABI_PARAM_recursionSkipMode }, { "aq-mode", X265ABI_PARAM_rc_aqMode },
        { "qg-size", X265ABI_PARAM_rc_qgSize }, { "ctu", X265ABI_PARAM_maxCUSize }, { "min-cu-size", X265ABI_PARAM_minCUSize }, { "max-tu-size", X265ABI_PARAM_maxTUSize },
        { "slices", X265ABI_PARAM_maxSlices }, { "level-idc", X265ABI_PARAM_levelIdc }, { "log-level", X265ABI_PARAM_logLevel }, { "qpmin", X265ABI_PARAM_rc_qpMin }, { "qpmax", X265ABI_PARAM_rc_qpMax },
        { "qpstep", X265ABI_PARAM_rc_qpStep }, { "vbv-bufsize", X265ABI_PARAM_rc_vbvBufferSize }, { "vbv-maxrate", X265ABI_PARAM_rc_vbvMaxBitrate } };
    for (const auto& it : ints)
        if (!strcmp(key, it.name)) { const int v = num(bad); wr<int32_t>(p, it.off, v); return bad ? -2 : 0; }          /* (the reference stores what atoi made of a bad value, too) */
    static const struct { const char* name; size_t off; } reals[] = {
        { "psy-rd", X265ABI_PARAM_psyRd }, { "psy-rdoq", X265ABI_PARAM_psyRdoq }, { "aq-strength", X265ABI_PARAM_rc_aqStrength }, { "qcomp", X265ABI_PARAM_rc_qCompress },
        { "ipratio", X265ABI_PARAM_rc_ipFactor }, { "pbratio", X265ABI_PARAM_rc_pbFactor } };
    for (const auto& it : reals)
        if (!strcmp(key, it.name)) { const double v = real(bad); wr<double>(p, it.off, v); return bad ? -2 : 0; }
    if (!strcmp(key, "crf")) { const double v = real(bad); if (bad) return -2; wr<double>(p, X265ABI_PARAM_rc_rfConstant, v); wr<int32_t>(p, X265ABI_PARAM_rc_rateControlMode, 2); return 0; }
    if (!strcmp(key, "qp")) { const int v = num(bad); if (bad) return -2; wr<int32_t>(p, X265ABI_PARAM_rc_qp, v); wr<int32_t>(p, X265ABI_PARAM_rc_rateControlMode, 1); return 0; }
    if (!strcmp(key, "bitrate")) { const int v = num(bad); if (bad) return -2; wr<int32_t>(p, X265ABI_PARAM_rc_bitrate, v); wr<int32_t>(p, X265ABI_PARAM_rc_rateControlMode, 0); return 0; }
    if (!strcmp(key, "me"))
    {
        static const char* const me[] = { "dia", "hex", "umh", "star", "sea", "full" };
        if (!value) return -2;
        for (int i = 0; i < 6; i++) if (!strcmp(value, me[i])) { wr<int32_t>(p, X265ABI_PARAM_searchMethod, i); return 0; }
        const int v = num(bad); if (bad || v < 0 || v > 5) return -2;
        wr<int32_t>(p, X265ABI_PARAM_searchMethod, v); return 0;
    }
    if (!strcmp(key, "fps"))
    {
        if (!value) return -2;
        unsigned a = 0, b = 0;
        if (sscanf(value, "%u/%u", &a, &b) == 2 && a && b) { wr<uint32_t>(p, X265ABI_PARAM_fpsNum, a); wr<uint32_t>(p, X265ABI_PARAM_fpsDenom, b); return 0; }
        /* anything that is not a fraction goes in as thousandths, whole numbers too ("60" is 60000 / 1000: param.cpp:936-941 -- the VUI's timing info carries the pair as it is) */
        const float fps = (float)atof(value);
        if (fps > 0 && fps <= INT_MAX / 1000) { wr<uint32_t>(p, X265ABI_PARAM_fpsNum, (uint32_t)(int)(fps * 1000 + .5)); wr<uint32_t>(p, X265ABI_PARAM_fpsDenom, 1000); }
        else { wr<uint32_t>(p, X265ABI_PARAM_fpsNum, (uint32_t)atoi(value)); wr<uint32_t>(p, X265ABI_PARAM_fpsDenom, 1); }
        return 0;
    }
    if (!strcmp(key, "input-res"))
    {
        int w = 0, h = 0;
        if (!value || sscanf(value, "%dx%d", &w, &h) != 2) return -2;
        wr<int32_t>(p, X265ABI_PARAM_sourceWidth, w); wr<int32_t>(p, X265ABI_PARAM_sourceHeight, h); return 0;
    }
    if (!strcmp(key, "pools") || !strcmp(key, "numa-pools"))
    {
        /* the string is kept for the life of the param, as strdup in the reference (param.cpp:949); x265_param_free does not own it either */
        wr<const char*>(p, X265ABI_PARAM_numaPools, value ? strdup(value) : nullptr); return 0;
    }
    /* the video usability information's options (param.cpp:1173-1240): names from the header's tables, or numbers */
    auto byName = [&](const char* const* names, bool& err) -> int {
        if (!value) { err = true; return 0; }
        for (int i = 0; names[i]; i++) if (!strcmp(value, names[i])) return i;
        return num(err);
    };
    static const char* const sarNames[] = { "unknown", "1:1", "12:11", "10:11", "16:11", "40:33", "24:11", "20:11", "32:11", "80:33", "18:11", "15:11", "64:33", "160:99", "4:3", "3:2", "2:1", 0 };
    static const char* const formatNames[] = { "component", "pal", "ntsc", "secam", "mac", "unknown", 0 };
    static const char* const rangeNames[] = { "limited", "full", 0 };
    static const char* const primNames[] = { "reserved", "bt709", "unknown", "reserved", "bt470m", "bt470bg", "smpte170m", "smpte240m", "film", "bt2020", "smpte428", "smpte431", "smpte432", 0 };
    static const char* const transferNames[] = { "reserved", "bt709", "unknown", "reserved", "bt470m", "bt470bg", "smpte170m", "smpte240m", "linear", "log100", "log316", "iec61966-2-4", "bt1361e",
                                                 "iec61966-2-1", "bt2020-10", "bt2020-12", "smpte2084", "smpte428", "arib-std-b67", 0 };
    static const char* const matrixNames[] = { "gbr", "bt709", "unknown", "", "fcc", "bt470bg", "smpte170m", "smpte240m", "ycgco", "bt2020nc", "bt2020c", "smpte2085", "chroma-derived-nc", "chroma-derived-c", "ictcp", 0 };
    if (!strcmp(key, "sar"))
    {
        int v = byName(sarNames, bad);
        if (bad)
        {
            int w = 0, h = 0;
            v = 255;            /* X265_EXTENDED_SAR */
            bad = !value || sscanf(value, "%d:%d", &w, &h) != 2;
            if (!bad) { wr<int32_t>(p, X265ABI_PARAM_vui_sarWidth, w); wr<int32_t>(p, X265ABI_PARAM_vui_sarHeight, h); }
        }
        wr<int32_t>(p, X265ABI_PARAM_vui_aspectRatioIdc, v); return bad ? -2 : 0;
    }
    if (!strcmp(key, "overscan"))
    {
        if (value && !strcmp(value, "show")) wr<int32_t>(p, X265ABI_PARAM_vui_bEnableOverscanInfoPresentFlag, 1);
        else if (value && !strcmp(value, "crop")) { wr<int32_t>(p, X265ABI_PARAM_vui_bEnableOverscanInfoPresentFlag, 1); wr<int32_t>(p, X265ABI_PARAM_vui_bEnableOverscanAppropriateFlag, 1); }
        else if (value && !strcmp(value, "unknown")) wr<int32_t>(p, X265ABI_PARAM_vui_bEnableOverscanInfoPresentFlag, 0);
        else return -2;
        return 0;
    }
    if (!strcmp(key, "videoformat")) { wr<int32_t>(p, X265ABI_PARAM_vui_bEnableVideoSignalTypePresentFlag, 1); wr<int32_t>(p, X265ABI_PARAM_vui_videoFormat, byName(formatNames, bad)); return bad ? -2 : 0; }
    if (!strcmp(key, "range")) { wr<int32_t>(p, X265ABI_PARAM_vui_bEnableVideoSignalTypePresentFlag, 1); wr<int32_t>(p, X265ABI_PARAM_vui_bEnableVideoFullRangeFlag, byName(rangeNames, bad)); return bad ? -2 : 0; }
    if (!strcmp(key, "colorprim") || !strcmp(key, "transfer") || !strcmp(key, "colormatrix"))
    {
        wr<int32_t>(p, X265ABI_PARAM_vui_bEnableVideoSignalTypePresentFlag, 1); wr<int32_t>(p, X265ABI_PARAM_vui_bEnableColorDescriptionPresentFlag, 1);
        if (key[0] == 'c' && key[5] == 'p') wr<int32_t>(p, X265ABI_PARAM_vui_colorPrimaries, byName(primNames, bad));
        else if (key[0] == 't') wr<int32_t>(p, X265ABI_PARAM_vui_transferCharacteristics, byName(transferNames, bad));
        else wr<int32_t>(p, X265ABI_PARAM_vui_matrixCoeffs, byName(matrixNames, bad));
        return bad ? -2 : 0;
    }
    if (!strcmp(key, "hash")) { if (!value) return -2; wr<int32_t>(p, X265ABI_PARAM_decodedPictureHashSEI, atoi(value)); return 0; }
    if (!strcmp(key, "master-display")) { wr<const char*>(p, X265ABI_PARAM_masteringDisplayColorVolume, value ? strdup(value) : nullptr); return value ? 0 : -2; }
    if (!strcmp(key, "max-cll"))
    {
        unsigned short a = 0, b = 0;
        const int got = value ? sscanf(value, "%hu,%hu", &a, &b) : 0;
        if (got >= 1) wr<uint16_t>(p, X265ABI_PARAM_maxCLL, a);
        if (got >= 2) wr<uint16_t>(p, X265ABI_PARAM_maxFALL, b);
        return got == 2 ? 0 : -2;
    }
    if (!strcmp(key, "chromaloc"))
    {
        if (!value) return -2;
        wr<int32_t>(p, X265ABI_PARAM_vui_bEnableChromaLocInfoPresentFlag, 1);
        wr<int32_t>(p, X265ABI_PARAM_vui_chromaSampleLocTypeTopField, atoi(value)); wr<int32_t>(p, X265ABI_PARAM_vui_chromaSampleLocTypeBottomField, atoi(value));
        return 0;
    }
    if (!strcmp(key, "display-window") || !strcmp(key, "crop-rect"))
    {
        int l = 0, t = 0, r = 0, b = 0;
        wr<int32_t>(p, X265ABI_PARAM_vui_bEnableDefaultDisplayWindowFlag, 1);
        /* (sscanf straight into the members in the reference: what it got before it failed stays) */
        const int got = value ? sscanf(value, "%d,%d,%d,%d", &l, &t, &r, &b) : 0;
        if (got >= 1) wr<int32_t>(p, X265ABI_PARAM_vui_defDispWinLeftOffset, l);
        if (got >= 2) wr<int32_t>(p, X265ABI_PARAM_vui_defDispWinTopOffset, t);
        if (got >= 3) wr<int32_t>(p, X265ABI_PARAM_vui_defDispWinRightOffset, r);
        if (got >= 4) wr<int32_t>(p, X265ABI_PARAM_vui_defDispWinBottomOffset, b);
        return got == 4 ? 0 : -2;
    }
    return -1;
}
int abi_param_apply_profile(void* p, const char* profile)
{
    if (!p || !profile) return 0;
    if (X265AMD_DEPTH == 8 && (!strcmp(profile, "main") || !strcmp(profile, "main10"))) return 0;
    if (X265AMD_DEPTH == 10 && !strcmp(profile, "main10")) return 0;
    return -1;
}

const char* const g_versionStr = "x265amd 0.3 (parity target x265 3.6+1-aa7f602f7)";

/* ---- x265_picture ---- */
void abi_picture_init(void* param, void* pic)
{
    if (!pic) return;
    memset(pic, 0, X265ABI_SIZEOF_PICTURE);
    wr<int32_t>(pic, X265ABI_PIC_bitDepth, param ? PI(param, internalBitDepth) : X265AMD_DEPTH);
    wr<int32_t>(pic, X265ABI_PIC_colorSpace, param ? PI(param, internalCsp) : 1);
    wr<int32_t>(pic, X265ABI_PIC_forceqp, 0);
}
void* abi_picture_alloc(void) { return calloc(1, X265ABI_SIZEOF_PICTURE); }
void abi_picture_free(void* p) { free(p); }

/* ---- encoder ---- */
void* abi_encoder_open(void* p)
{
    if (!p) { xa_fail(X265AMD_EINVAL, "x265_encoder_open: null param"); return nullptr; }
    static thread_local char why[200];
    const char* bad = nullptr;
#define REQUIRE(cond, text) do { if (!bad && !(cond)) bad = text; } while (0)
    REQUIRE(PI(p, internalBitDepth) == X265AMD_DEPTH, "internalBitDepth differs from this library's (libx265amd_main: 8, libx265amd_main10: 10)");
    REQUIRE(PI(p, internalCsp) == 1, "internalCsp: only X265_CSP_I420 is built");
    REQUIRE(PI(p, rc_rateControlMode) == 1 || PI(p, rc_rateControlMode) == 2, "rc.rateControlMode: X265_RC_CQP (--qp) and X265_RC_CRF (--crf) are built, ABR (--bitrate) is not");
    REQUIRE(!PI(p, rc_hevcAq) && PI(p, rc_vbvMaxBitrate) == 0, "rc: hevc-aq and VBV are not built");
    REQUIRE(!PI(p, rc_bEnableGrain), "rc.bEnableGrain (--tune grain): the grain rate control is not built");
    REQUIRE(PI(p, rc_vbvBufferSize) == 0 && !PI(p, rc_bStatRead) && !PI(p, rc_bStatWrite), "rc: VBV and multi-pass statistics are not built");
    REQUIRE(PI(p, bFrameAdaptive) >= 0 && PI(p, bFrameAdaptive) <= 2, "bFrameAdaptive (--b-adapt): 0, 1 or 2");
    REQUIRE(!PI(p, bHistBasedSceneCut), "bHistBasedSceneCut: histogram scene-cut detection is not built");
    REQUIRE(PI(p, maxCUSize) == 64 && PI(p, minCUSize) == 8 && PI(p, maxTUSize) == 32, "maxCUSize / minCUSize / maxTUSize: only 64 / 8 / 32 are built");
    REQUIRE(!PI(p, interlaceMode) && !PI(p, bField), "interlaced coding is not built");
    REQUIRE(!PI(p, bLossless) && !PI(p, bCULossless), "lossless coding is not built");
    REQUIRE(!PI(p, bEnableTransformSkip), "bEnableTransformSkip is not built");
    REQUIRE(PI(p, maxSlices) <= 1, "maxSlices above 1 is not built");
    REQUIRE(!PI(p, bIntraRefresh) && !PI(p, bEnableHME) && !PI(p, bEnableConstrainedIntra), "intra refresh / HME / constrained intra are not built");
    REQUIRE(!PI(p, noiseReductionIntra) && !PI(p, noiseReductionInter) && !rd<const char*>(p, X265ABI_PARAM_scalingLists), "noise reduction / scaling lists are not built");
    REQUIRE(!PI(p, cbQpOffset) && !PI(p, crQpOffset), "chroma QP offsets (cbQpOffset / crQpOffset) must be 0");
    REQUIRE(!PI(p, bSaoNonDeblocked) && !PI(p, selectiveSAO), "sao-non-deblock / selective-sao are not built");
    REQUIRE(!PI(p, bEmitHRDSEI), "the HRD SEI (bEmitHRDSEI) is not built");
    REQUIRE(!PI(p, bEnableTemporalSubLayers) && !PI(p, uhdBluray) && !PI(p, bEnableSvtHevc), "temporal layers / uhd-bd / svt are not built");
    REQUIRE(!PI(p, analysisReuseMode) && !PI(p, bDynamicRefine) && !PI(p, rdPenalty) && !PI(p, bEnableRdRefine) && !PI(p, dynamicRd) && !PI(p, bSsimRd), "analysis reuse / rd-refine / dynamic-rd / ssim-rd are not built");
    REQUIRE(!PI(p, bDistributeModeAnalysis) && !PI(p, bDistributeMotionEstimation), "pmode / pme are not built");
    REQUIRE(!PI(p, bAQMotion) && !PI(p, gopLookahead) && !PI(p, radl) && !PI(p, bEnableSceneCutAwareQp) && !PI(p, bEnableFades), "aq-motion / gop-lookahead / radl / scenecut-aware-qp / fades are not built");
    REQUIRE(PI(p, levelIdc) == 0, "levelIdc: the level is derived (determineLevel), not forced");
    REQUIRE(PI(p, searchMethod) == 0 || PI(p, searchMethod) == 1 || PI(p, searchMethod) == 3, "searchMethod: only dia, hex and star are built");
#undef REQUIRE
    if (bad) { snprintf(why, sizeof(why), "x265_encoder_open: %s", bad); xa_fail(X265AMD_EINVAL, why); return nullptr; }
    x265amd_param q;
    x265amd_param_default(&q);
    q.sourceWidth = PI(p, sourceWidth); q.sourceHeight = PI(p, sourceHeight); q.fpsNum = PU(p, fpsNum); q.fpsDenom = PU(p, fpsDenom);
    q.bframes = PI(p, bframes); q.keyframeMax = PI(p, keyframeMax); q.maxNumReferences = PI(p, maxNumReferences);
    q.scenecutThreshold = PI(p, scenecutThreshold); q.lookaheadDepth = PI(p, lookaheadDepth); q.keyframeMin = PI(p, keyframeMin); q.bFrameAdaptive = PI(p, bFrameAdaptive); q.bOpenGOP = PI(p, bOpenGOP) != 0; q.bBPyramid = PI(p, bBPyramid) != 0; q.lookaheadSlices = PI(p, lookaheadSlices); q.bEnableWeightedPred = PI(p, bEnableWeightedPred) != 0; q.bEnableWeightedBiPred = PI(p, bEnableWeightedBiPred) != 0;
    q.qp = PI(p, rc_qp); q.ipFactor = PD(p, rc_ipFactor); q.pbFactor = PD(p, rc_pbFactor);
    q.rateControlMode = PI(p, rc_rateControlMode); q.rfConstant = PD(p, rc_rfConstant); q.qCompress = PD(p, rc_qCompress); q.qgSize = PI(p, rc_qgSize);
    q.aqMode = PI(p, rc_aqMode); q.aqStrength = PD(p, rc_aqStrength); q.cuTree = PI(p, rc_cuTree) != 0;
    q.bEmitInfoSEI = PI(p, bEmitInfoSEI) != 0; q.bRepeatHeaders = PI(p, bRepeatHeaders) != 0; q.qpMin = PI(p, rc_qpMin); q.qpMax = PI(p, rc_qpMax);
    /* (Encoder::configure's rules for these switches, encoder.cpp:3721-3754, are x265amd_encoder_open's) */
    q.rdLevel = PI(p, rdLevel); q.bEnableRectInter = PI(p, bEnableRectInter); q.bEnableAMP = PI(p, bEnableAMP); q.limitModes = PI(p, limitModes); q.limitReferences = PI(p, limitReferences);
    q.bEnableEarlySkip = PI(p, bEnableEarlySkip); q.recursionSkipMode = PI(p, recursionSkipMode); q.bIntraInBFrames = PI(p, bIntraInBFrames); q.psyRd = PD(p, psyRd);
    q.searchMethod = PI(p, searchMethod); q.subpelRefine = PI(p, subpelRefine); q.searchRange = PI(p, searchRange); q.maxNumMergeCand = PI(p, maxNumMergeCand);
    q.bEnableSignHiding = PI(p, bEnableSignHiding); q.bEnableStrongIntraSmoothing = PI(p, bEnableStrongIntraSmoothing); q.bEnableTemporalMvp = PI(p, bEnableTemporalMvp);
    q.tuQTMaxInterDepth = PI(p, tuQTMaxInterDepth); q.tuQTMaxIntraDepth = PI(p, tuQTMaxIntraDepth);
    q.bEnableLoopFilter = PI(p, bEnableLoopFilter); q.bEnableSAO = PI(p, bEnableSAO); q.bEnableWavefront = PI(p, bEnableWavefront);
    {
        /* without a thread pool the reference switches WPP off (Encoder::create, encoder.cpp: "no thread pool ... WPP disabled"): --pools none */
        const char* pools = rd<const char*>(p, X265ABI_PARAM_numaPools);
        if (pools && (!strcmp(pools, "none") || !strcmp(pools, "NONE") || !strcmp(pools, "0"))) q.bEnableWavefront = 0;
    }
    q.aspectRatioIdc = PI(p, vui_aspectRatioIdc); q.rdoqLevel = PI(p, rdoqLevel);
    q.limitTU = PI(p, limitTU);
    q.deblockingFilterTCOffset = PI(p, deblockingFilterTCOffset); q.deblockingFilterBetaOffset = PI(p, deblockingFilterBetaOffset);
    q.bEnableAccessUnitDelimiters = PI(p, bEnableAccessUnitDelimiters) != 0; q.decodedPictureHashSEI = PI(p, decodedPictureHashSEI);
    q.maxCLL = rd<uint16_t>(p, X265ABI_PARAM_maxCLL); q.maxFALL = rd<uint16_t>(p, X265ABI_PARAM_maxFALL); q.bEmitCLL = PI(p, bEmitCLL) != 0;
    {
        /* x265_check_params (param.cpp:1875-1876): any of the values switches the SEI units on; SEIMasteringDisplayColorVolume::parse (sei.h:205-213) */
        const char* md = rd<const char*>(p, X265ABI_PARAM_masteringDisplayColorVolume);
        q.bEmitHDR10SEI = PI(p, bEmitHDR10SEI) != 0 || md || q.maxCLL || q.maxFALL;
        unsigned short v[8]; unsigned l[2];
        if (md)
        {
            if (sscanf(md, "G(%hu,%hu)B(%hu,%hu)R(%hu,%hu)WP(%hu,%hu)L(%u,%u)", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &l[0], &l[1]) == 10)
            {
                q.hasMasteringDisplay = 1;
                for (int i = 0; i < 8; i++) q.masteringDisplay[i] = v[i];
                q.masteringDisplay[8] = l[0]; q.masteringDisplay[9] = l[1];
            }
            else fprintf(stderr, "x265amd [warning]: unable to parse mastering display color volume info\n");
        }
    }
    q.vuiSarWidth = PI(p, vui_sarWidth); q.vuiSarHeight = PI(p, vui_sarHeight);
    q.vuiOverscanInfoPresent = PI(p, vui_bEnableOverscanInfoPresentFlag); q.vuiOverscanAppropriate = PI(p, vui_bEnableOverscanAppropriateFlag);
    q.vuiVideoSignalTypePresent = PI(p, vui_bEnableVideoSignalTypePresentFlag); q.vuiVideoFormat = PI(p, vui_videoFormat); q.vuiFullRange = PI(p, vui_bEnableVideoFullRangeFlag);
    q.vuiColorDescriptionPresent = PI(p, vui_bEnableColorDescriptionPresentFlag); q.vuiColorPrimaries = PI(p, vui_colorPrimaries); q.vuiTransfer = PI(p, vui_transferCharacteristics);
    q.vuiMatrix = PI(p, vui_matrixCoeffs);
    q.vuiChromaLocPresent = PI(p, vui_bEnableChromaLocInfoPresentFlag); q.vuiChromaLocTop = PI(p, vui_chromaSampleLocTypeTopField); q.vuiChromaLocBottom = PI(p, vui_chromaSampleLocTypeBottomField);
    q.vuiDisplayWindow = PI(p, vui_bEnableDefaultDisplayWindowFlag); q.vuiDispWinLeft = PI(p, vui_defDispWinLeftOffset); q.vuiDispWinRight = PI(p, vui_defDispWinRightOffset);
    q.vuiDispWinTop = PI(p, vui_defDispWinTopOffset); q.vuiDispWinBottom = PI(p, vui_defDispWinBottomOffset);
    q.psyRdoqFix8 = q.rdoqLevel ? (int32_t)(PD(p, psyRdoq) * 256.0) : 0;         /* Quant::init: m_psyRdoqScale = (int32_t)(psyScale * 256.0) (quant.cpp:188) */
    q.bEnableFastIntra = PI(p, bEnableFastIntra);
    /* frame threads: 0 = by core count, which is more than one on any machine with four cores or more (threadpool.cpp:661-677); the stream of the
     * frame-parallel rules does not depend on the number */
    /* Encoder::create (encoder.cpp:199-254): no wavefronts in a picture of one CTU row or fewer than three CTU columns ("pointless and unstable") -- and without wavefronts the
     * frame threads chosen by core count are as many as half the rows, which is ONE for a picture of one or two rows (threadpool.cpp:664-666) */
    const int ctuRows = (((q.sourceHeight + 7) & ~7) + 63) / 64, ctuCols = (((q.sourceWidth + 7) & ~7) + 63) / 64;
    if (ctuRows == 1 || ctuCols < 3) q.bEnableWavefront = 0;
    const int ft = PI(p, frameNumThreads);
    const int ftAuto = q.bEnableWavefront ? 3 : ((ctuRows + 1) / 2 < 16 ? (ctuRows + 1) / 2 : 16);
    q.frameNumThreads = ft == 1 ? 1 : (ft > 16 ? 16 : (ft <= 0 ? ftAuto : ft));
    x265amd_encoder* e = x265amd_encoder_open(&q);
    if (!e) return nullptr;
    AbiEncoder* a = new AbiEncoder;
    a->enc = e; a->width = q.sourceWidth; a->height = q.sourceHeight;
    a->bframeDelay = q.bframes ? (q.bBPyramid ? 2 : 1) : 0;
    a->param.assign((const uint8_t*)p, (const uint8_t*)p + X265ABI_SIZEOF_PARAM);         /* api.cpp:96-116: the encoder keeps a copy */
    if ((q.keyframeMax >= 0 && q.keyframeMax <= 1) || q.bEmitHDR10SEI) wr<int32_t>(a->param.data(), X265ABI_PARAM_bRepeatHeaders, 1);          /* ... as Encoder::configure left it: x265_encoder_parameters tells the caller (the reference's program asks before it writes the headers) */
    a->fps = (double)q.fpsNum / (double)q.fpsDenom;
    clock_gettime(CLOCK_MONOTONIC, &a->opened);
    return a;
}
void abi_encoder_parameters(void* enc, void* out) { if (enc && out) memcpy(out, ((AbiEncoder*)enc)->param.data(), X265ABI_SIZEOF_PARAM); }
int abi_encoder_headers(void* enc, x265amd_nal** ppNal, uint32_t* piNal)
{
    if (!enc) return -1;
    return x265amd_encoder_headers(((AbiEncoder*)enc)->enc, ppNal, piNal);         /* x265_nal and x265amd_nal have one layout (x265.h:94-99) */
}
int abi_encoder_encode(void* enc, x265amd_nal** ppNal, uint32_t* piNal, void* picIn, void* picOut)
{
    if (!enc) return -1;
    AbiEncoder& a = *(AbiEncoder*)enc;
    x265amd_picture in, out;
    memset(&in, 0, sizeof(in)); memset(&out, 0, sizeof(out));
    if (picIn)
    {
        if (rd<int32_t>(picIn, X265ABI_PIC_bitDepth) != X265AMD_DEPTH) { xa_fail(X265AMD_EINVAL, "x265_encoder_encode: picture bit depth differs from the library's (no conversion)"); return -1; }
        if (rd<int32_t>(picIn, X265ABI_PIC_colorSpace) != 1) { xa_fail(X265AMD_EINVAL, "x265_encoder_encode: only X265_CSP_I420 pictures"); return -1; }
        const int st = rd<int32_t>(picIn, X265ABI_PIC_sliceType);
        if (st != 0) { xa_fail(X265AMD_EINVAL, "x265_encoder_encode: forced slice types are not built (sliceType must be X265_TYPE_AUTO)"); return -1; }
        for (int k = 0; k < 3; k++)
        {
            in.planes[k] = rd<void*>(picIn, X265ABI_PIC_planes + 8 * k);
            in.stride[k] = rd<int32_t>(picIn, X265ABI_PIC_stride + 4 * k);
        }
        a.ptsIn.push_back(rd<int64_t>(picIn, X265ABI_PIC_pts));
    }
    /* the reconstruction is returned through planes the encoder owns in the reference (pic_out->planes point into its reconstructed picture); here the
     * caller's pic_out receives pointers to a buffer of this encoder that stays valid until its next call */
    std::vector<uint8_t>& recon = a.recon;
    if (picOut)
    {
        const size_t isz = X265AMD_DEPTH > 8 ? 2 : 1, ysz = (size_t)a.width * a.height * isz, csz = ysz / 4;
        recon.resize(ysz + 2 * csz);
        out.planes[0] = recon.data(); out.planes[1] = recon.data() + ysz; out.planes[2] = recon.data() + ysz + csz;
        out.stride[0] = (int32_t)(a.width * isz); out.stride[1] = out.stride[2] = (int32_t)(a.width / 2 * isz);
    }
    const int ret = x265amd_encoder_encode(a.enc, ppNal, piNal, picIn ? &in : nullptr, picOut ? &out : nullptr);
    if (ret > 0 && picOut)
    {
        for (int k = 0; k < 3; k++) { wr<void*>(picOut, X265ABI_PIC_planes + 8 * k, out.planes[k]); wr<int32_t>(picOut, X265ABI_PIC_stride + 4 * k, out.stride[k]); }
        wr<int32_t>(picOut, X265ABI_PIC_bitDepth, X265AMD_DEPTH); wr<int32_t>(picOut, X265ABI_PIC_colorSpace, 1);
        wr<int32_t>(picOut, X265ABI_PIC_poc, out.poc); wr<int32_t>(picOut, X265ABI_PIC_sliceType, out.sliceType);
        wr<int32_t>(picOut, X265ABI_PIC_width, a.width); wr<int32_t>(picOut, X265ABI_PIC_height, a.height);
    }
    if (ret > 0)
    {
        /* the picture's POC is its place in display order from the first picture on (Frame::m_poc; the slice header's count restarts at an IDR picture, this one does not) */
        const uint64_t display = (uint64_t)(out.poc < 0 ? 0 : out.poc);
        const int64_t pts = display < a.ptsIn.size() ? a.ptsIn[display] : 0;
        const int64_t reordered = a.coded < a.ptsIn.size() ? a.ptsIn[a.coded] : pts;
        int64_t dts = reordered;
        if (a.bframeDelay)
        {
            const int64_t delayTime = (size_t)a.bframeDelay < a.ptsIn.size() ? a.ptsIn[a.bframeDelay] - a.ptsIn[0] : 0;
            dts = a.coded > (uint64_t)a.bframeDelay ? a.prevReordered[(a.coded - a.bframeDelay) % a.bframeDelay] : reordered - delayTime;
            a.prevReordered[a.coded % a.bframeDelay] = reordered;
        }
        a.coded++;
        if (picOut) { wr<int64_t>(picOut, X265ABI_PIC_pts, pts); wr<int64_t>(picOut, X265ABI_PIC_dts, dts); }
    }
    return ret;
}
/* x265_encoder_get_stats (Encoder::fetchStats, encoder.cpp:2870-2960): what this encoder counts -- pictures, bits and average QP by slice type, the times and the bit rate.
 * PSNR / SSIM (off unless asked for, and then not built), the light levels and the weighted-frame count stay 0. */
void abi_encoder_get_stats(void* enc, void* stats, uint32_t bytes)
{
    if (!stats) return;
    uint8_t st[X265ABI_SIZEOF_STATS];
    memset(st, 0, sizeof(st));
    if (enc)
    {
        AbiEncoder& a = *(AbiEncoder*)enc;
        uint64_t w[13] = { 0 };
        if (x265amd_encoder_stats(a.enc, w, 13) == 0)
        {
            uint64_t pics = 0, bits = 0;
            static const size_t at[3] = { X265ABI_STATS_statsI, X265ABI_STATS_statsP, X265ABI_STATS_statsB };
            for (int t = 0; t < 3; t++)
            {
                double qsum; memcpy(&qsum, &w[10 + t], 8);
                const uint64_t n = w[4 + t];
                pics += n; bits += w[7 + t];
                wr<uint32_t>(st, at[t] + X265ABI_SLICESTATS_numPics, (uint32_t)n);
                if (n)
                {
                    wr<double>(st, at[t] + X265ABI_SLICESTATS_avgQp, qsum / (double)n);
                    /* EncStats: bitrate of the type's pictures at the frame rate, in kbps (encoder.cpp:2885: m_accBits * scale / m_numPics, scale = fps / 1000) */
                    wr<double>(st, at[t] + X265ABI_SLICESTATS_bitrate, (double)w[7 + t] * a.fps / 1000.0 / (double)n);
                }
            }
            struct timespec now; clock_gettime(CLOCK_MONOTONIC, &now);
            const double elapsed = (double)(now.tv_sec - a.opened.tv_sec) + 1e-9 * (double)(now.tv_nsec - a.opened.tv_nsec);
            const double video = (double)pics / a.fps;
            wr<double>(st, X265ABI_STATS_elapsedEncodeTime, elapsed); wr<double>(st, X265ABI_STATS_elapsedVideoTime, video);
            wr<double>(st, X265ABI_STATS_bitrate, video > 0 ? 0.001 * (double)bits / video : 0.0);
            wr<uint64_t>(st, X265ABI_STATS_accBits, bits); wr<uint32_t>(st, X265ABI_STATS_encodedPictureCount, (uint32_t)pics);
        }
    }
    memcpy(stats, st, bytes < (uint32_t)X265ABI_SIZEOF_STATS ? bytes : (uint32_t)X265ABI_SIZEOF_STATS);
}
void abi_encoder_log(void*, int, char**) {}
void abi_encoder_close(void* enc)
{
    if (!enc) return;
    AbiEncoder* a = (AbiEncoder*)enc;
    x265amd_encoder_close(a->enc);
    delete a;
}
void abi_cleanup(void) { x265amd_release_scratch(); }
/* not built: fail cleanly */
int abi_fail_encoder_param(void*, void*) { return -1; }
int abi_fail_encoder(void*) { return -1; }
int abi_fail_ctu_info(void*, int, void**) { return -1; }
int abi_fail_slicetype(void*, int*, int*, int*) { return -1; }
int abi_fail_ref_list(void*, void**, void**, int, int, int*, int*) { return -1; }
/* x265_csvlog_open / x265_csvlog_encode (api.cpp:1281-1403, :1511-1636) at the default log level (csvLogLevel 0: one summary line per encode; the per-frame levels 1 and 2
 * are not built: NULL): a new file gets the header line, an existing one is appended to */
static const char* const kSummaryHeader =
    "Command, Date/Time, Elapsed Time, FPS, Bitrate, "
    "Y PSNR, U PSNR, V PSNR, Global PSNR, SSIM, SSIM (dB), "
    "I count, I ave-QP, I kbps, I-PSNR Y, I-PSNR U, I-PSNR V, I-SSIM (dB), "
    "P count, P ave-QP, P kbps, P-PSNR Y, P-PSNR U, P-PSNR V, P-SSIM (dB), "
    "B count, B ave-QP, B kbps, B-PSNR Y, B-PSNR U, B-PSNR V, B-SSIM (dB), ";
void* abi_csvlog_open(const void* p)
{
    if (!p) return nullptr;
    const char* fn = rd<const char*>(p, X265ABI_PARAM_csvfn);
    if (!fn || PI(p, csvLogLevel) != 0) { xa_fail(X265AMD_EINVAL, "x265_csvlog_open: no file name, or a per-frame log level (only the summary level 0 is built)"); return nullptr; }
    if (FILE* f = fopen(fn, "r")) { fclose(f); return fopen(fn, "ab"); }
    FILE* f = fopen(fn, "wb");
    if (!f) return nullptr;
    fputs(kSummaryHeader, f);
    if (rd<uint16_t>(p, X265ABI_PARAM_maxCLL) || rd<uint16_t>(p, X265ABI_PARAM_maxFALL)) fputs("MaxCLL, MaxFALL,", f);
    fputs(" Version\n", f);
    return f;
}
void abi_csvlog_frame(const void*, const void*) {}
void abi_csvlog_encode(const void* p, const void* stats, int, int, int argc, char** argv)
{
    FILE* f = p ? rd<FILE*>(p, X265ABI_PARAM_csvfpt) : nullptr;
    if (!f || !stats || PI(p, csvLogLevel) != 0) return;
    fputc('"', f);
    for (int i = 1; i < argc; i++) { fputc(' ', f); fputs(argv[i], f); }         /* (without a command line the reference prints the option string; that string is not built) */
    fputc('"', f);
    time_t now; time(&now);
    char buffer[200];
    strftime(buffer, 128, "%c", localtime(&now));
    fprintf(f, ", %s, ", buffer);
    const double elapsed = rd<double>(stats, X265ABI_STATS_elapsedEncodeTime);
    const uint32_t pics = rd<uint32_t>(stats, X265ABI_STATS_encodedPictureCount);
    fprintf(f, "%.2f, %.2f, %.2f,", elapsed, elapsed > 0 ? pics / elapsed : 0.0, rd<double>(stats, X265ABI_STATS_bitrate));
    fprintf(f, " -, -, -, -,");          /* PSNR: not measured */
    fprintf(f, " -, -,");                /* SSIM */
    static const size_t at[3] = { X265ABI_STATS_statsI, X265ABI_STATS_statsP, X265ABI_STATS_statsB };
    for (int t = 0; t < 3; t++)
    {
        const uint32_t n = rd<uint32_t>(stats, at[t] + X265ABI_SLICESTATS_numPics);
        if (n) fprintf(f, " %-6u, %2.2lf, %-8.2lf, -, -, -, -,", n, rd<double>(stats, at[t] + X265ABI_SLICESTATS_avgQp), rd<double>(stats, at[t] + X265ABI_SLICESTATS_bitrate));
        else fprintf(f, " -, -, -, -, -, -, -,");
    }
    if (rd<uint16_t>(p, X265ABI_PARAM_maxCLL) || rd<uint16_t>(p, X265ABI_PARAM_maxFALL)) fprintf(f, " %-6u, %-6u,", 0u, 0u);
    fprintf(f, " %s\n", g_versionStr);
}
void abi_dither_image(void*, int, int, int16_t*, int) {}
int abi_fail_analysis(void*, void*, int, uint32_t) { return -1; }
int abi_fail_parse3(void*, const char*, const char*) { return -1; }

/* struct x265_api (x265.h:2561-2614) member for member: 3 + 4 ints, bit depth, two strings, 20 function pointers, sizeof_frame_stats, 9 function pointers,
 * zone_param_parse (ENABLE_LIBVMAF is off in the reference build this library replaces) */
typedef X265ApiTable AbiTable;          /* x265_api_table.h */
static_assert(sizeof(AbiTable) == X265ABI_SIZEOF_API, "struct x265_api layout");
static_assert(offsetof(AbiTable, bit_depth) == X265ABI_API_bit_depth && offsetof(AbiTable, version_str) == X265ABI_API_version_str, "struct x265_api layout");
static_assert(offsetof(AbiTable, fn) == X265ABI_API_param_alloc && offsetof(AbiTable, fn) + 10 * sizeof(void*) == X265ABI_API_encoder_open, "struct x265_api layout");
static_assert(offsetof(AbiTable, fn) + 15 * sizeof(void*) == X265ABI_API_encoder_encode && offsetof(AbiTable, fn) + 18 * sizeof(void*) == X265ABI_API_encoder_close, "struct x265_api layout");
static_assert(offsetof(AbiTable, fn) + 19 * sizeof(void*) == X265ABI_API_cleanup && offsetof(AbiTable, sizeof_frame_stats) == X265ABI_API_sizeof_frame_stats, "struct x265_api layout");
static_assert(offsetof(AbiTable, fn2) == X265ABI_API_encoder_intra_refresh && offsetof(AbiTable, zone_param_parse) == X265ABI_API_zone_param_parse, "struct x265_api layout");

const AbiTable g_api = {
    X265ABI_MAJOR_VERSION, X265ABI_BUILD, X265ABI_SIZEOF_PARAM, X265ABI_SIZEOF_PICTURE, X265ABI_SIZEOF_ANALYSIS_DATA, X265ABI_SIZEOF_ZONE, X265ABI_SIZEOF_STATS,
    X265AMD_DEPTH,
    g_versionStr, "[Linux][hipcc gfx950][64 bit] MI355X",
    { (void*)abi_param_alloc, (void*)abi_param_free, (void*)abi_param_default, (void*)abi_param_parse, (void*)abi_fail_parse3, (void*)abi_param_apply_profile,
      (void*)abi_param_default_preset, (void*)abi_picture_alloc, (void*)abi_picture_free, (void*)abi_picture_init, (void*)abi_encoder_open, (void*)abi_encoder_parameters,
      (void*)abi_fail_encoder_param, (void*)abi_fail_encoder_param, (void*)abi_encoder_headers, (void*)abi_encoder_encode, (void*)abi_encoder_get_stats, (void*)abi_encoder_log,
      (void*)abi_encoder_close, (void*)abi_cleanup },
    X265ABI_SIZEOF_FRAME_STATS,
    { (void*)abi_fail_encoder, (void*)abi_fail_ctu_info, (void*)abi_fail_slicetype, (void*)abi_fail_ref_list, (void*)abi_csvlog_open, (void*)abi_csvlog_frame,
      (void*)abi_csvlog_encode, (void*)abi_dither_image, (void*)abi_fail_analysis },
    (void*)abi_fail_parse3
};

}

/* x265_api_get_209 (x265.h:2619-2635, api.cpp:1107-1182): the table for `bitDepth` (0: this library's).  Another depth is looked for in the sibling library the
 * way the reference looks for libx265_main10.so: dlopen by name next to this one -- here the caller is simply told there is none (NULL), as the reference does
 * when the sibling is missing. */
extern "C" const void* x265_api_get_209(int bitDepth)
{
    if (bitDepth && bitDepth != X265AMD_DEPTH) return nullptr;
    return &g_api;
}
/* x265_api_query (api.cpp:1184-1279): refuses builds older than 51 and any build that is not this one */
extern "C" const void* x265_api_query(int bitDepth, int apiVersion, int* err)
{
    if (apiVersion < 51) { if (err) *err = 1 /* X265_API_QUERY_ERR_VER_REFUSED */; return nullptr; }
    if (bitDepth && bitDepth != X265AMD_DEPTH) { if (err) *err = 2 /* X265_API_QUERY_ERR_LIB_NOT_FOUND */; return nullptr; }
    if (err) *err = 0;
    return &g_api;
}
