/* x265amd_ratecontrol.h -- the host arithmetic of the reference's rate control as `--preset medium` comes (rc.rateControlMode = X265_RC_CRF, rc.aqMode = 2, rc.cuTree = 1):
 *   - cuTree: Lookahead::cuTree / estimateCUPropagate / cuTreeFinish, frameCostRecalculate   (reference: source/encoder/slicetype.cpp:3399-3500, :3502-3608, :3750-3800, :3802-3880)
 *     with primitives.propagateCost = estimateCUPropagateCost                               (source/common/pixel.cpp:931-959)
 *   - constant rate factor: RateControl::RateControl / rateControlStart / rateEstimateQscale / getQScale / accumPQpUpdate   (source/encoder/ratecontrol.cpp:184-360, :1334-1643,
 *     :1900-2375, :2931-2954) without VBV, ABR, two passes, zones, grain, scene-cut aware QP
 *   - the QP of a CU from the picture's QP and the block offsets: Analysis::calculateQpforCuSize (source/encoder/analysis.cpp:3634-3714)
 * Host code only (x265-amod_amd/host/fm_ratecontrol.cpp), double precision in the reference's order and operand types, compiled with the reference's own floating-point
 * flags (-O2 -ffast-math) so that every value that is rounded to an integer downstream comes out bit for bit (tests/test_ratecontrol.py pins each function against the
 * reference's own classes and against records of whole encodes).  The lowres cost estimates and motion fields cuTree reads come from the device
 * (x265amd_lowres_frame_cost_batch); nothing here touches the GPU. */
#ifndef X265AMD_RATECONTROL_H
#define X265AMD_RATECONTROL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- cuTree ---- */
/* What cuTree reads and writes of a picture's Lowres (common/lowres.h): one record per picture of the lookahead window, frames[0] = the last non-B picture.
 * Arrays are per 8x8 block of the half-resolution picture (width8 x height8, raster), which with qgSize 16 / 32 / 64 is also the 16x16 grid of the full picture. */
typedef struct x265amd_cutree_frame
{
    int32_t slice_type;                 /* X265_TYPE_*: 5 (B) is a B picture, everything else is not */
    int32_t reserved;
    const int32_t* intra_cost;          /* Lowres::intraCost */
    const int32_t* inv_qscale;          /* Lowres::invQscaleFactor */
    const double* qp_aq_offset;         /* Lowres::qpAqOffset */
    double* qp_cutree_offset;           /* Lowres::qpCuTreeOffset: written by the finish step */
    uint16_t* propagate_cost;           /* Lowres::propagateCost */
    const double* weighted_cost_delta;  /* Lowres::weightedCostDelta[d - 1] for the list-0 estimate at distance d (0 where the lookahead found no weight); NULL: all zero */
} x265amd_cutree_frame;

typedef struct x265amd_cutree_params
{
    int32_t width8, height8;            /* Lookahead::m_8x8Width / m_8x8Height */
    uint32_t fps_num, fps_denom;
    int32_t b_pyramid, weighted_bipred; /* param.bBPyramid, param.bEnableWeightedBiPred */
    int32_t lookahead_depth;            /* param.lookaheadDepth (> 0: the lookahead-less extrapolation is not built) */
    int32_t reserved;
    double strength;                    /* Lookahead::m_cuTreeStrength = 5.0 * (1.0 - rc.qCompress) */
} x265amd_cutree_params;

/* CostEstimateGroup::singleCost(p0, p1, b) as cuTree asks for it: makes sure the estimate of frames[b] against frames[p0] (and frames[p1] when p1 > b) exists and hands
 * out Lowres::lowresCosts[b - p0][p1 - b] and the two motion fields Lowres::lowresMvs[0][b - p0] / [1][p1 - b] (x, y pairs; mvs1 NULL when p1 == b).
 * Returns 0, or an error code that x265amd_cutree passes on. */
typedef int (*x265amd_cutree_estimate_fn)(void* ctx, int p0, int p1, int b, const uint16_t** lowres_costs, const int16_t** mvs0, const int16_t** mvs1);

/* Lookahead::cuTree(frames, numframes, bIntra).  Returns 0 or the estimate callback's error. */
int x265amd_cutree(const x265amd_cutree_params* p, x265amd_cutree_frame* const* frames, int numframes, int b_intra, x265amd_cutree_estimate_fn estimate, void* ctx);

/* Lookahead::frameCostRecalculate for a P / I picture (a B picture's value is its costEstAq): the block costs rescaled by the cuTree offsets, summed over the blocks that are
 * not on the picture's edge */
int64_t x265amd_frame_cost_recalculate(const x265amd_cutree_params* p, const uint16_t* lowres_costs, const double* qp_cutree_offset);

/* ---- constant rate factor ---- */
typedef struct x265amd_rc_params
{
    int32_t width, height;
    uint32_t fps_num, fps_denom;
    int32_t bframes, keyframe_max, cu_tree, qp_min, qp_max, reserved;
    double rf_constant, q_compress, ip_factor, pb_factor;
} x265amd_rc_params;

/* what rateControlStart reads of the picture it is called for (in coding order) */
typedef struct x265amd_rc_frame
{
    int32_t slice_type;                 /* 0 B, 1 P, 2 I (Slice::m_sliceType) */
    int32_t is_referenced;              /* IS_REFERENCED(frame) */
    int32_t poc;
    int32_t scenecut;                   /* Lowres::bScenecut */
    int32_t ref0_scenecut;              /* refFrameList[0][0]->m_lowres.bScenecut (P / B) */
    int32_t last_minigop_b;             /* Lowres::bLastMiniGopBFrame */
    int64_t satd_cost;                  /* Lowres::satdCost (getEstimatedPictureCost) */
    /* B pictures: the first reference of each list */
    int32_t ref_slice_type[2], ref_poc[2], ref_is_referenced[2];
    double ref_avg_qp_rc[2];            /* FrameData::m_avgQpRc of those pictures (what this function returned for them in *avg_qp_rc) */
} x265amd_rc_frame;

typedef struct x265amd_rc x265amd_rc;
x265amd_rc* x265amd_rc_open(const x265amd_rc_params* p);
void x265amd_rc_close(x265amd_rc* rc);
/* RateControl::rateControlStart for the next picture in coding order: returns the slice QP (m_qp) and, in *avg_qp_rc, FrameData::m_avgQpRc -- the double that is the base of
 * every CU's QP (FrameEncoder: m_cuStat[].baseQp) and of later B pictures' QPs. */
int x265amd_rc_start(x265amd_rc* rc, const x265amd_rc_frame* f, double* avg_qp_rc);

/* Analysis::calculateQpforCuSize (analysis.cpp:3634-3714) for the CU at (x, y) of `size` samples: base_qp + the mean of the 16x16 block offsets under the CU, rounded, clipped
 * to [qp_min, qp_max].  offsets: Lowres::qpCuTreeOffset for a referenced picture with cuTree, else Lowres::qpAqOffset (maxBlocksInRow = (width + 15) / 16 per row). */
int x265amd_cu_qp(double base_qp, const double* offsets, int width, int height, int x, int y, int size, int qp_min, int qp_max);

#ifdef __cplusplus
}
#endif
#endif
