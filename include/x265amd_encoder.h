/* x265amd_encoder.h -- the OUTER drop-in boundary: the encoder object behind the reference's public API (reference: source/x265.h,
 * `x265_encoder_open` :2412, `x265_encoder_headers` :2427, `x265_encoder_encode` :2437, `x265_encoder_close` :2471, struct x265_api
 * :2561-2614; implementation source/encoder/api.cpp:96-600, encoder.cpp `Encoder::encode`, dpb.cpp `DPB::prepareEncode`,
 * slicetype.cpp `Lookahead::slicetypeDecide`, ratecontrol.cpp CQP branch).
 *
 * One process per GPU; every picture of the encode lives in device memory from the moment it is handed in.  The host loop below the
 * API is C++ (csrc/encoder_api.hip): mini-GOP formation, decoded picture buffer and reference picture sets, reference lists, slice QPs,
 * per-frame calls into the frame pipeline (x265amd_analyse_frame, deblocking, SAO, border extension, slice NAL) and the stream headers.
 *
 * Built subset (everything else is rejected by x265amd_encoder_open with NULL + x265amd_last_error): 4:2:0, bit depth of the library,
 * constant QP or constant rate factor (rc.rateControlMode = X265_RC_CQP / X265_RC_CRF; no ABR, VBV or second pass), adaptive quantisation (aq-mode 0-3) and cuTree,
 * mini-GOPs fixed or chosen by the lookahead (bFrameAdaptive 0 / 1 / 2), scene-cut detection, open or
 * closed GOPs, the B pyramid, the lookahead in slices, weighted prediction, CTU 64 / min CU 8, one slice, even picture sizes (a size that is no multiple of 8 is coded padded to one, the SPS's conformance window takes the pad off: encoder.cpp:4081-4090).  Within that subset the byte stream
 * is the reference encoder's (tests/test_encoder_api.py compares whole streams with the reference command line program's). */
#ifndef X265AMD_ENCODER_H
#define X265AMD_ENCODER_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* the fields of x265_param (x265.h:1034-2275) the built subset reads, under the reference's names */
typedef struct x265amd_param
{
    int32_t sourceWidth, sourceHeight;      /* luma samples; even */
    uint32_t fpsNum, fpsDenom;
    int32_t bframes;                        /* consecutive B frames of a mini-GOP (0..16); bFrameAdaptive is 0 */
    int32_t keyframeMax;                    /* keyframe interval (--keyint): IDR pictures with closed GOPs, CRA pictures with bOpenGOP */
    int32_t maxNumReferences;               /* --ref */
    int32_t qp;                             /* rc.qp (CQP) */
    double ipFactor, pbFactor;              /* rc.ipFactor / rc.pbFactor: QP offsets of I and B slices */
    int32_t rdLevel;                        /* 2..6 */
    int32_t bEnableRectInter, bEnableAMP, limitModes, limitReferences;
    int32_t bEnableEarlySkip, recursionSkipMode, bIntraInBFrames;
    double psyRd;
    int32_t searchMethod, subpelRefine, searchRange;       /* X265AMD_ME_* (dia / hex / star), subme, merange */
    int32_t maxNumMergeCand;
    int32_t bEnableSignHiding, bEnableStrongIntraSmoothing, bEnableTemporalMvp;
    int32_t tuQTMaxInterDepth, tuQTMaxIntraDepth;
    int32_t bEnableLoopFilter, bEnableSAO, bEnableWavefront;
    int32_t aspectRatioIdc;                 /* vui.aspectRatioIdc (1 = square samples; 0 = not signalled) */
    int32_t rdoqLevel;                      /* 0..2 */
    int32_t psyRdoqFix8;                    /* (int)(psyRdoq * 256), i.e. Quant::m_psyRdoqScale; needs rdoqLevel > 0 */
    int32_t bEnableFastIntra;
    int32_t firstFrame;                     /* display-order number of the first picture handed in (0 for a whole encode).  A closed GOP depends on nothing
                                             * before its IDR picture, so a stream can be cut at IDR pictures and the pieces encoded by different encoder objects
                                             * (different GPUs): each piece starts with firstFrame = its first picture and the slice NAL units concatenate to the
                                             * single-encoder stream (headers from one of them) */
    int32_t frameNumThreads;                /* 0 / 1: a picture is coded when its reference pictures are complete (the reference's --frame-threads 1).  > 1: the
                                             * reference's frame-parallel rules, which are what its default gives on any machine with four cores or more
                                             * (threadpool.cpp:661-677): a CTU row starts when the reference pictures have finished the rows down to refLagRows below it
                                             * (frameencoder.cpp:170-175, :893-908), motion vectors stop at searchRange below the block (search.cpp:92, :2763) and
                                             * candidates beyond are left out (search.cpp:1934, :2009; analysis.cpp:2803, :2933), SAO is never switched off from
                                             * picture to picture (sao.cpp:264).  The stream is the same for every value > 1; the number of pictures in flight is the
                                             * library's own choice (X265AMD_FRAME_THREADS) */
    int32_t scenecutThreshold;              /* param.scenecutThreshold (--scenecut; the preset's 40): 0 = no scene-cut detection.  > 0: the lookahead's scene-cut decision
                                             * (Lookahead::slicetypeAnalyse / scenecut / scenecutInternal, slicetype.cpp:2603-3047, with bFrameAdaptive 0) on the
                                             * lowres cost estimates (x265amd_lowres_*): a scene change becomes an I picture (IDR when keyframeMin pictures have passed
                                             * since the last keyframe), the picture before it P.  scenecutBias is the reference's default (5) */
    int32_t lookaheadDepth;                 /* param.lookaheadDepth (--rc-lookahead): pictures the slice-type decision looks at (only read when scenecutThreshold > 0) */
    int32_t keyframeMin;                    /* param.keyframeMin (--min-keyint); 0 = the reference's default min(fps, keyframeMax / 10) (encoder.cpp:3658-3663) */
    int32_t shardRank, shardCount;          /* frame-per-GPU: with shardCount > 1 this object codes the pictures whose place in CODING order
                                             * k satisfies k % shardCount == shardRank (the reference's frame k -> frame encoder k mod G, encoder.cpp:1872) and
                                             * takes the others' finished CTU rows from the objects that code them (x265amd_encoder_export_row / _import_row).
                                             * Every object of the set is fed every picture (slice types, DPB and reference lists are decided identically by all);
                                             * the stream is the owners' NAL units in coding order.  0 / 0 or count 1: one object codes everything */
    int32_t bFrameAdaptive;                 /* param.bFrameAdaptive (--b-adapt): 0 fixed mini-GOPs of `bframes` B pictures; 2 the trellis (Lookahead::slicetypeAnalyse with
                                             * slicetypePath / slicetypePathCost, slicetype.cpp:2776-2795, :3218-3313) on P and B cost estimates of the lowres pictures,
                                             * every pair of the window searched in advance as the reference's batch does with four pool workers or more; 1 the fast
                                             * decision (X265_B_ADAPT_FAST, slicetype.cpp:2796-2848: pictures in pairs, estimates made when they are asked for) */
    int32_t bOpenGOP;                       /* param.bOpenGOP (--open-gop, the reference's default; 0 = --no-open-gop): keyframes after the first are I pictures with
                                             * NAL type CRA instead of IDR (Lookahead::slicetypeDecide, slicetype.cpp:1956-1993; DPB::getNalUnitType, dpb.cpp:486-506):
                                             * the POC count runs on, the B pictures in front of a keyframe stay B (leading pictures, RASL_N) and reference across it,
                                             * the pictures before the keyframe leave the DPB with the first picture after it in output order (decodingRefreshMarking,
                                             * dpb.cpp:357-399); the lookahead's window reaches one picture beyond the keyframe interval (slicetype.cpp:2660-2661) */
    int32_t bBPyramid;                      /* param.bBPyramid (--b-pyramid, the reference's default; 0 = --no-b-pyramid): in a mini-GOP of two B pictures or more the middle
                                             * one is a reference picture (Lookahead::placeBref, slicetype.cpp:1755-1762, :2372-2376): coded right behind the P picture,
                                             * NAL type TRAIL_R, slice QP between P and B (ratecontrol.cpp:1594-1595), referenced by the B pictures on either side of it
                                             * (two L1 references at most, dpb.cpp:273) and by later pictures while it stays in the DPB; two reorder pictures in the
                                             * VPS / SPS (level.cpp:295); the trellis prices the B pictures against it (slicetype.cpp:3291-3302) */
    int32_t lookaheadSlices;                /* param.lookaheadSlices (--lookahead-slices; the reference's default 8, 0 and 1 = off; off below 720 lines whatever it says):
                                             * the lookahead's estimates outside its batches run in cooperative slices of max(rows / slices, 10) block rows
                                             * (slicetype.cpp:1035-1059, :3957-3970, :4004-4036) -- a slice's bottom row takes no motion predictors from the row below, so
                                             * the fields (the encoder's search candidates) and costs differ from the whole-picture ones */
    int32_t bEnableWeightedPred;            /* param.bEnableWeightedPred (--weightp, the reference's default; needs the lookahead: scenecutThreshold > 0 or bFrameAdaptive 2):
                                             * every picture's sums and squared sums (LookaheadTLD::calcAdaptiveQuantFrame, slicetype.cpp:507-513, :678-700); the lookahead's weight
                                             * analysis before every list-0 search (LookaheadTLD::weightsAnalyse, slicetype.cpp:879-978); the slice's analysis for P pictures
                                             * (weightAnalyse, weightPrediction.cpp:222-540: luma on the lowres planes, chroma on the source planes, list 0's first reference,
                                             * motion compensated with the lookahead's vectors); pps.weighted_pred_flag and pred_weight_table().  A slice whose analysis picks
                                             * weights is coded with them (round 5): the motion searches read weighted copies of the reference (MotionReference, reference.cpp:
                                             * 51-185), every prediction is weighted (Predict::motionCompensation, predict.cpp:85-232) */
    int32_t bEnableWeightedBiPred;          /* param.bEnableWeightedBiPred (--weightb; on in the presets slower and veryslow): the same for B pictures, both lists
                                             * (pps.weighted_bipred_flag; addWeightBi, predict.cpp:411-518) */
    /* ---- rate control as the presets come (round 6) ---- */
    int32_t rateControlMode;                /* param.rc.rateControlMode with the reference's numbers (x265.h X265_RC_*): X265AMD_RC_CQP 1 = constant QP (`qp` above; also what 0 means);
                                             * X265AMD_RC_CRF 2 = constant rate factor, the reference's default: RateControl::rateControlStart / rateEstimateQscale / getQScale
                                             * (ratecontrol.cpp:1334-1643, :1900-2375, :2931-2954) without VBV, zones or a second pass -- every picture's QP follows from the slice
                                             * types, the scene-cut marks and the QPs of the pictures before it (ABR 0 is not built) */
    double rfConstant;                      /* param.rc.rfConstant (--crf; 28) */
    double aqStrength;                      /* param.rc.aqStrength (--aq-strength; 1.0) */
    double qCompress;                       /* param.rc.qCompress (--qcomp; 0.6): cuTree's strength 5 (1 - qcomp) and, without cuTree, the exponent of the blurred complexity */
    int32_t aqMode;                         /* param.rc.aqMode (--aq-mode): 0 off, 1 variance, 2 auto-variance (the default), 3 auto-variance biased to dark scenes
                                             * (LookaheadTLD::calcAdaptiveQuantFrame, slicetype.cpp:452-713): a QP offset per 16x16 block, the QP of a CU = the picture's
                                             * QP + the mean offset of the blocks under its quantisation group (Analysis::calculateQpforCuSize, analysis.cpp:3634-3714),
                                             * cu_qp_delta in the stream (pps.cu_qp_delta_enabled_flag, diff_cu_qp_delta_depth from qgSize) */
    int32_t cuTree;                         /* param.rc.cuTree (--cutree, the default; needs the lookahead and aqMode != 0 as in the reference's presets): Lookahead::cuTree /
                                             * estimateCUPropagate / cuTreeFinish (slicetype.cpp:3399-3800) on the lookahead's block costs and motion fields: referenced
                                             * pictures take their block offsets from it */
    int32_t qgSize;                         /* param.rc.qgSize (--qg-size): 64 or 32 (the default); 16 and 8 are not built */
    int32_t bEmitInfoSEI;                   /* param.bEmitInfoSEI (--info, the reference's default): a user-data SEI NAL unit behind the parameter sets that names the encoder and its
                                             * options (Encoder::getStreamHeaders, encoder.cpp:3260-3280).  The reference's text carries ITS version and build strings, so this
                                             * unit is the one part of a stream that can never be byte-equal between two builds of anything: every parity test runs --no-info */
    int32_t qpMin, qpMax;                   /* param.rc.qpMin / qpMax (--qpmin / --qpmax; 0 / 69): the range of the rate control's QP and of every CU's (ratecontrol.cpp clipQscale,
                                             * analysis.cpp:3712) */
    int32_t bRepeatHeaders;                 /* param.bRepeatHeaders (--repeat-headers; Encoder::configure switches it on for all-intra encodes, --keyint 1): VPS / SPS / PPS in front
                                             * of every keyframe's slice units (Encoder::encode, encoder.cpp:2035-2045) */
    int32_t reserved2;
    /* param.vui (x265.h: the video usability information of the SPS, Encoder::initSPS encoder.cpp:3388-3423; aspectRatioIdc is further up): signalling only, nothing here changes a
     * coded sample.  --sar W:H (aspectRatioIdc 255), --overscan, --videoformat, --range, --colorprim, --transfer, --colormatrix, --chromaloc, --display-window */
    int32_t vuiSarWidth, vuiSarHeight;
    int32_t vuiOverscanInfoPresent, vuiOverscanAppropriate;
    int32_t vuiVideoSignalTypePresent, vuiVideoFormat, vuiFullRange;                 /* videoFormat 5 = unspecified */
    int32_t vuiColorDescriptionPresent, vuiColorPrimaries, vuiTransfer, vuiMatrix;   /* 2 = unspecified */
    int32_t vuiChromaLocPresent, vuiChromaLocTop, vuiChromaLocBottom;
    int32_t vuiDisplayWindow, vuiDispWinLeft, vuiDispWinRight, vuiDispWinTop, vuiDispWinBottom;
    int32_t reserved3;
    /* units around the slices (signalling only) */
    int32_t bEnableAccessUnitDelimiters;    /* param.bEnableAccessUnitDelimiters (--aud): an access unit delimiter in front of every picture but the first -- of every picture with
                                             * bRepeatHeaders (frameencoder.cpp:497-506) */
    int32_t bEmitHDR10SEI, bEmitCLL;        /* param.bEmitHDR10SEI (--hdr10; x265_check_params switches it on with any of the values below), param.bEmitCLL (--cll, the default):
                                             * the content light level and mastering display colour volume SEI units behind the parameter sets (encoder.cpp:3264-3282) */
    int32_t maxCLL, maxFALL;                /* --max-cll "cll,fall" */
    int32_t hasMasteringDisplay;            /* --master-display "G(x,y)B(x,y)R(x,y)WP(x,y)L(max,min)": the ten numbers below */
    uint32_t masteringDisplay[10];
    int32_t decodedPictureHashSEI;          /* param.decodedPictureHashSEI (--hash): 0 none, 1 MD5, 2 CRC, 3 checksum of each reconstructed picture in a suffix SEI unit */
    int32_t reserved4;
    int32_t deblockingFilterTCOffset, deblockingFilterBetaOffset;      /* param.deblockingFilter*Offset (--deblock tc:beta, each -6 .. 6): pps_tc_offset_div2 / pps_beta_offset_div2 */
    int32_t limitTU;                        /* param.limitTU (--limit-tu; --preset slower has 4): 0, 2 (depth first), 3 (neighbourhood), 4 (both); 1 (breadth first) is not built.
                                             * Needs tuQTMaxInterDepth > 1 (else it is switched off, encoder.cpp:4103-4107) */
    int32_t reserved5;
} x265amd_param;
enum { X265AMD_RC_CQP = 1, X265AMD_RC_CRF = 2 };

/* x265_param_default + --preset medium for the fields above, but CQP 30 without adaptive quantisation and cuTree, --bframes 0 (what the block-level tests build on);
 * the preset's own rate control is rateControlMode = X265AMD_RC_CRF, rfConstant 28, aqMode 2, aqStrength 1, cuTree 1, qCompress 0.6, qgSize 32 */
void x265amd_param_default(x265amd_param* p);

typedef struct x265amd_nal { uint32_t type; uint32_t sizeBytes; uint8_t* payload; } x265amd_nal;      /* x265_nal (x265.h:94-99) */
typedef struct x265amd_picture                                                                        /* the used part of x265_picture (x265.h:397-490) */
{
    void* planes[3];            /* host pointers: Y, U, V (8-bit samples for the 8-bit library, 16-bit little endian otherwise) */
    int32_t stride[3];          /* bytes */
    int32_t poc, sliceType;     /* filled on output: display order count, X265_TYPE_* (1 IDR, 3 P, 5 B) */
    int32_t qp;                 /* filled on output: the slice QP */
} x265amd_picture;

typedef struct x265amd_encoder x265amd_encoder;
x265amd_encoder* x265amd_encoder_open(const x265amd_param* p);
/* VPS, SPS, PPS; the array and payloads stay valid until the next call on this encoder.  Returns the total payload size or -1. */
int x265amd_encoder_headers(x265amd_encoder* enc, x265amd_nal** pp_nal, uint32_t* pi_nal);
/* pic_in == NULL flushes.  Returns 1 when a coded picture was emitted (its NAL units in *pp_nal, its reconstruction copied to
 * pic_out's planes when pic_out is given), 0 when none is ready yet (or the flush is complete), -1 on error. */
int x265amd_encoder_encode(x265amd_encoder* enc, x265amd_nal** pp_nal, uint32_t* pi_nal, const x265amd_picture* pic_in, x265amd_picture* pic_out);
/* The same call with the input picture ALREADY ON THE DEVICE: pic_in->planes are device addresses (hipMalloc memory of the encoder's device; strides in bytes as above),
 * read by a device-to-device copy and a margin kernel instead of being padded on the host and sent over the bus.  There is no counterpart in the reference's API (x265_picture
 * holds host pointers, x265.h:397-490): this is the form for callers whose frames are made or decoded on the GPU, and the one bench.py times ("inputs already resident in
 * HBM when the timed region starts").  Everything else -- pic_out, the NAL units, flushing with pic_in == NULL -- as x265amd_encoder_encode. */
int x265amd_encoder_encode_device(x265amd_encoder* enc, x265amd_nal** pp_nal, uint32_t* pi_nal, const x265amd_picture* pic_in, x265amd_picture* pic_out);
void x265amd_encoder_close(x265amd_encoder* enc);

/* ---- frame-per-GPU: a finished CTU row travels from the object that codes a picture to the objects that reference it ----
 * What a reference picture's consumers read, published at the reference's m_reconRowFlag point (framefilter.cpp:654-664; consumers wait at
 * frameencoder.cpp:893-908): the filtered samples of the CTU row in the three planes with their side margins (and the top / bottom margin with the first / last
 * row), and the row's unit and motion records (the co-located motion field, cudata.cpp:1858-1862; the depths topSkipMinDepth reads).  The flat picture buffers of all
 * objects of a set have one geometry, so a row is three byte ranges of the picture buffer plus two ranges of the host maps. */
typedef struct x265amd_row_export
{
    uint64_t coding_index; int32_t ctu_row, reserved;
    const void* src[3];                 /* DEVICE memory: where the row's bytes of plane 0 / 1 / 2 are read from -- the exporting object's picture, or wherever the
                                         * transport has put them on the importing side (a receive buffer) */
    uint64_t plane_offset[3], plane_bytes[3];       /* the row's byte range in each plane, relative to the start of the picture's flat buffer (Y | U | V, padded):
                                                     * where the importing object puts them */
    const void* units; uint64_t units_bytes;        /* HOST memory: the row's x265amd_cu_unit / x265amd_mv_unit records */
    const void* motion; uint64_t motion_bytes;
    uint64_t map_offset_units, map_offset_motion;   /* where those records sit in the importing object's maps (bytes) */
} x265amd_row_export;
/* The object that codes picture `coding_index`: blocks until CTU row `ctu_row` of it is final (at most timeout_ms), then describes it.  Returns 0; 1 when the picture is
 * not known yet (not handed over by the lookahead: ask again); 2 when timeout_ms <= 0 and the row is not final yet (a look instead of a wait: ask again);
 * -1 on error / time-out / a failed picture.  The described memory stays valid while the picture can be
 * referenced. */
int x265amd_encoder_export_row(x265amd_encoder* enc, uint64_t coding_index, int ctu_row, x265amd_row_export* out, int timeout_ms);
/* An object that does not code the picture: copies the row in (src[] may be another object's picture -- the same device or a peer of this process -- or the buffer
 * this rank received the row in) and opens the gates of the pictures that wait for it.  Same return values. */
int x265amd_encoder_import_row(x265amd_encoder* enc, const x265amd_row_export* row);
/* the ranges of CTU row `ctu_row` (offsets and byte counts; no addresses): what an importing rank sizes its receive buffers by.  Returns 0 or -1. */
int x265amd_encoder_row_geometry(const x265amd_encoder* enc, int ctu_row, x265amd_row_export* out);
/* the number of CTU rows of a picture, and whether this object codes picture `coding_index` */
int x265amd_encoder_ctu_rows(const x265amd_encoder* enc);
int x265amd_encoder_owns(const x265amd_encoder* enc, uint64_t coding_index);
/* counters of the pictures handed over by the lookahead so far (the counterpart of what x265_encoder_get_stats totals, x265.h:2475): out[0] I, out[1] P, out[2] B pictures,
 * out[3] the sum over them of the DISTINCT reference pictures each reads (DPB::prepareEncode's lists, dpb.cpp:101-290) -- SURVEY section 8d's R per picture, which the
 * bench's roofline figure is built from.  With n >= 13 also, of the pictures HANDED OUT so far by I / P / B: out[4..6] their number, out[7..9] the bits of their NAL units,
 * out[10..12] the sums of their average QPs (IEEE doubles, bit for bit in the words) -- x265_stats' statsI / statsP / statsB.  n: words available (>= 4).  Returns 0 or -1. */
int x265amd_encoder_stats(const x265amd_encoder* enc, uint64_t* out, int n);
/* whether pictures coded later may reference picture `coding_index` (DPB::prepareEncode: every picture but a plain B picture; reference: source/encoder/dpb.cpp:101-140):
 * 1 yes, 0 no -- its rows need not travel and an object that does not code it does not wait for them --, 2 not known yet (not handed over by the lookahead: ask again),
 * -1 on error.  The same answer on every object of a set. */
int x265amd_encoder_is_referenced(x265amd_encoder* enc, uint64_t coding_index);

#ifdef __cplusplus
}
#endif
#endif
