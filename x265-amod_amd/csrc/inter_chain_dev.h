/* The skip chain of a CTU of a P / B picture as ONE device command (round 4).
 *
 * What it restates: the part of Analysis::compressInterCU_rd0_4 (reference: source/encoder/analysis.cpp:1146-1848) that ends a CU early -- checkMerge2Nx2N_rd0_4
 * (:2750-2880) choosing the merge candidate by SA8D, encodeResAndCalcRdSkipCU / encodeResAndCalcRdInterCU (search.cpp:2770-2975) finding that no transform unit keeps
 * a level, early skip + recursion skip (:1302-1330) -- together with CUData::getInterMergeCandidates (common/cudata.cpp:1458-1712) on a device-resident motion map.
 *
 * Why: on the host every CU was two round trips (candidates' predictions + measurements; transform chains + measurement) with the candidate derivation, the job
 * records and the bookkeeping in between: 82 us per skipped CU, 3.4 ms for a cut CTU of the last row (1080 = 16 x 64 + 56: fourteen CUs each), and the last row of a
 * P picture is what every picture behind it waits for.  None of the decisions that END a CU as a skip depends on the entropy coder's state: the candidate is chosen by
 * SA8D + a fixed index cost, and the CU is skipped exactly when the chosen candidate's residual quantises to nothing (then the residual mode equals the skip mode and
 * `tempPred->rdCost < bestPred->rdCost` is false, see xa_merge_rd in inter_rd.hip).  So the device runs CU after CU in the CTU's coding order on its own -- candidates
 * from its motion map, predictions, SA8D, the choice, the transform + quantisation of the winner's residual, and when nothing is left the CU's motion into the map and
 * its prediction into the picture and into the enclosing CUs' reconstruction tiles -- until a CU does NOT end as a skip (or a vector reaches beyond what the reference
 * pictures have published).  The host then prices the skipped CUs from the compact results (bits, contexts, costs: x265amd_skip_rd_host, as before) and takes the CU
 * the chain stopped at through the ordinary path.
 *
 * Nodes: the CTU's quad-tree in pre-order (depth-first, z-order), at most 85; a node that may not be coded at its own depth (cut by the picture edge, or above
 * topSkipMinDepth's depth) is descended into, a checked node that ends as a skip jumps to `next` (the node behind its subtree).
 */
#ifndef X265AMD_INTER_CHAIN_DEV_H
#define X265AMD_INTER_CHAIN_DEV_H
#include <stdint.h>

/* one 4x4 unit of the device-resident motion map: x265amd_mv_unit + the CU depth, 16 bytes so that a unit is two 8-byte words */
struct alignas(16) XaMapUnit { uint8_t pred_mode, inter_dir; int8_t ref_idx[2]; int16_t mv[2][2]; uint8_t depth, pad[3]; };

struct XaChainNode { int16_t x, y; uint8_t log2, flags, next, parent; };       /* flags: 1 = merge check at this depth (inside the picture, depth >= topSkipMinDepth);
                                                                                   2 = inside the picture but not coded at this depth: the split flag (1) is counted behind its sub-CUs.
                                                                                   Bits 0-1 hold when the CTU's QP is not below the reference picture's (topSkipMinDepth's currentQP >=
                                                                                   previousQP, analysis.cpp:3432-3470), bits 2-3 the same two flags when it is */
enum { XA_CHAIN_END = 0, XA_CHAIN_NOTSKIP = 1, XA_CHAIN_GUARD = 2 };            /* why the chain stopped */
enum { XA_CHAIN_MAX_NODES = 85 };

struct XaChainCuOut             /* one skipped CU: pinned host memory */
{
    uint32_t node; uint8_t cand, dir; int8_t ref_idx[2];
    int16_t mv[2][2];
    x265amd_cu_measure meas;    /* the chosen candidate's prediction against the source */
};
/* the merge check of the CU the chain stopped at, when the device has made it (a CU of 8x8 .. 32x32 whose residual mode beats its skip mode): what
 * checkMerge2Nx2N_rd0_4 leaves in its two modes.  The host prices the skip mode itself (x265amd_skip_rd_host on `meas`); the residual mode's figures are here, its
 * reconstruction is in the depth's PRED_MERGE reconstruction tile, the skip mode's (the prediction) in the PRED_SKIP tile */
struct XaChainStop
{
    uint32_t valid, node;
    uint8_t cand, dir; int8_t ref_idx[2]; int16_t mv[2][2];
    uint8_t cbf[3], reserved0;
    uint32_t total_bits, mv_bits, coeff_bits, psy_energy, sa8d, sa8d_luma;
    uint64_t rd_cost, luma_dist, chroma_dist, frac;
    x265amd_cu_measure meas;
    uint8_t ctx[X265AMD_CTX_STRIDE];
    int16_t levels[1024 + 2 * 256];         /* Y (N x N), U, V (N/2 x N/2 each, from [1024] and [1280]) */
};
/* the stop record's head as one store sequence (xa_st_result): everything in front of XaChainStop::ctx */
struct alignas(8) XaChainStopHead
{
    uint32_t valid, node; uint8_t cand, dir; int8_t ref_idx[2]; int16_t mv[2][2]; uint8_t cbf[3], reserved0;
    uint32_t total_bits, mv_bits, coeff_bits, psy_energy, sa8d, sa8d_luma; uint64_t rd_cost, luma_dist, chroma_dist, frac; x265amd_cu_measure meas;
};
static_assert(sizeof(XaChainStopHead) == offsetof(XaChainStop, ctx), "the stop record's head");
struct XaChainOut { uint32_t count, stop_node, reason, reserved; uint64_t frac; uint64_t ticks[8]; uint8_t ctx[X265AMD_CTX_STRIDE]; XaChainStop stop; XaChainCuOut cu[XA_CHAIN_MAX_NODES]; };     /* frac / ctx: the coder's state where the chain stopped */

struct alignas(8) XaChainJob
{
    x265amd_mvpred_info info;
    int32_t ref_pic[2][16];                 /* plane-table index of reference r of list l */
    uint64_t planes;                        /* device array: num_pics x 3 addresses of sample (0,0); the last picture is the source, the one before it the reconstruction */
    uint64_t cur, col;                      /* XaMapUnit maps: this picture / the co-located picture */
    uint64_t tiles; uint64_t tile_bytes;    /* the Analyzer's tile arena */
    uint64_t out;                           /* XaChainOut */
    uint64_t nodes;                         /* XaChainNode[num_nodes] */
    int32_t stride, cstride, num_pics, w4;
    int32_t start, end, num_nodes;
    int32_t tiles_per_depth, cand_tile0, split_recon_tile;      /* tile index = depth * tiles_per_depth + ... */
    int32_t skip_recon_tile, merge_recon_tile, chain64_off;     /* chain64_off: a CU above the largest transform with a level somewhere goes to the host at once (the default; X265AMD_CHAIN_64=1: chain_merge_rd64) */
    int32_t frame_parallel, search_range, chroma_sa8d, slice_type, qp_luma, qp_chroma, tu_log2_max;
    int32_t guard_on, guard_r0, guard_r1, guard_need;           /* reference rows guard_r0 .. guard_r1 are final up to column guard_need (the picture width: all of them) */
    int32_t ctu_x, ctu_y, reserved1;
    uint64_t lambda;                        /* RDCost::m_lambda (FIX8) of the CTU's QP: calcRdSADCost, calcPsyRdCost */
    uint64_t lambda2;                       /* RDCost::m_lambda2 */
    uint32_t psy_rd;                        /* RDCost::m_psyRd (0: off) */
    int32_t rd_level, sign_hide, max_cu_depth, dbg;
    uint64_t scratch;                       /* levels / residual / reconstruction dumps of the winner's transform units (x265amd_inter_rd_scratch_bytes) */
    uint64_t frac;                          /* the entropy coder's state the first CU of the chain starts from (Entropy::m_fracBits, contexts) */
    uint8_t ctx[X265AMD_CTX_STRIDE];
    /* ---- delta QP (PPS cu_qp_delta_enabled; section 4.27 of DESIGN.md).  Quantisation group k of the CTU: 0 when the groups are CTUs, else the 32x32 quadrant ---- */
    int32_t use_dqp, max_dqp_depth;
    int32_t first_qp;                       /* the QP of the CTU's first unit in the picture's records as the chain starts (topSkipMinDepth's currentQP) */
    int32_t previous_qp;                    /* ... and what it is compared with: the first reference picture's CTU (previousQP) */
    int32_t prev_qp;                        /* CUData::getLastCodedQP in front of the CTU */
    int32_t dqp_reserved;
    struct QpSet { uint64_t lambda, lambda2; uint32_t psy_rd; int32_t qp_luma, qp_chroma, qp; } qps[5];       /* [0] the 64x64 CU, [1 + k] group k and everything inside it */
    int8_t last_src[4];                     /* getLastCodedQP in front of group k: the group whose unit it arrives at, -1: the one in front of the CTU (picture geometry alone) */
    int8_t last_val[4], left_val[4], above_val[4];      /* those units' QPs in the picture's records, for groups complete before the chain starts (the chain knows the ones it completes) */
    uint8_t anc_flags[4];                   /* per depth: the flags (bits 0-1) the host's recursion used for the start node's ancestors */
    uint8_t dqp_pad[4];
};

#ifdef XA_CHAIN_DEVICE

struct ChainCand { int16_t mv[2][2]; int8_t ref_idx[2]; uint8_t dir, valid; };

struct ChainLds
{
    XaChainJob job;
    XaChainNode nodes[XA_CHAIN_MAX_NODES + 3];
    x265amd_mc_job mc[5];
    CuMeasureJob mj;
    x265amd_cu_measure meas[5];
    ChainCand cand[5];
    int nc, stop, best, anyLevel, count, skipWins;
    unsigned int acc[5][3][16];         /* candidates' Hadamard sums per plane and 16x16 group */
    unsigned long long red[XA_SERVER_WAVES];
    x265amd_tu_job tu[12];              /* the winner's transform units: Y, U, V -- or, a CU one size above the largest transform, its four luma units, then U's four, then V's */
    x265amd_tu_result tr[12];
    uint64_t frac;                      /* the coder's state in front of the current CU */
    uint8_t ctx[X265AMD_CTX_STRIDE], ctxS[X265AMD_CTX_STRIDE], ctxB[X265AMD_CTX_STRIDE], ctxD[X265AMD_CTX_STRIDE], ctxQ[X265AMD_CTX_STRIDE];
    uint64_t fracS, fracD;
    uint32_t rdBits[3], rdPsy, rdCbf[3], rdPad; uint64_t rdCost, rdLuma, rdChroma;     /* the residual mode when it wins: bits (total, prediction info, skip flag), psy energy, coded block flags, cost, distortions */
    uint32_t step[256];
    unsigned long long ticks[8]; long long tprev;       /* stage clock (thread 0): candidates, predictions, measurements, units, RD, placing, CUs, - */
    uint8_t used[XA_CHAIN_MAX_NODES + 3];               /* the flags (bits 0-1) each node was visited with */
    int firstQp, dqp, dqpTwice;                         /* the CTU's first unit's QP as the recursion would find it; the current CU's cu_qp_delta and whether checkDQP prices it again */
    int qgVal[4]; uint32_t qgDone;                      /* groups this chain has completed (all skipped): every unit carries the group's predicted QP */
};
#define XA_CHAIN_HEADER 8192
#define XA_CHAIN_T(k) do { if (tid == 0) { const long long t_ = wall_clock64(); S.ticks[k] += (unsigned long long)(t_ - S.tprev); S.tprev = t_; } } while (0)

XA_DEV XaMapUnit chain_ld_unit(const XaMapUnit* p)
{
    union { XaMapUnit u; uint64_t w[2]; } v;
    v.w[0] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v.w[1] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return v.u;
}
XA_DEV void chain_st_unit(XaMapUnit* p, const XaMapUnit& u)
{
    union { XaMapUnit u; uint64_t w[2]; } v;
    v.u = u;
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), v.w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p) + 1, v.w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

XA_DEV unsigned chain_z(int x, int y)          /* z-order of the 4x4 unit holding sample (x, y) inside its CTU */
{
    unsigned r = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) r |= (((unsigned)(x >> 2) >> b) & 1u) << (2 * b) | (((unsigned)(y >> 2) >> b) & 1u) << (2 * b + 1);
    return r;
}
XA_DEV bool chain_inter(bool have, const XaMapUnit& u) { return have && (u.pred_mode == X265AMD_MODE_INTER || u.pred_mode == X265AMD_MODE_SKIP); }
XA_DEV bool chain_same_motion(const XaMapUnit& a, const XaMapUnit& b)      /* hasEqualMotion (cudata.cpp:1439-1456) */
{
    if (a.inter_dir != b.inter_dir) return false;
    for (int l = 0; l < 2; l++)
        if ((a.inter_dir & (1 << l)) && (a.mv[l][0] != b.mv[l][0] || a.mv[l][1] != b.mv[l][1] || a.ref_idx[l] != b.ref_idx[l])) return false;
    return true;
}
XA_DEV void chain_scale(int& mx, int& my, int curPOC, int curRefPOC, int colPOC, int colRefPOC)      /* scaleMvByPOCDist + scaleMv (cudata.cpp:2030-2045, :104-110) */
{
    const int d = colPOC - colRefPOC, b = curPOC - curRefPOC;
    if (d == b) return;
    const int tdb = b < -128 ? -128 : (b > 127 ? 127 : b), tdd = d < -128 ? -128 : (d > 127 ? 127 : d);
    const int x = (0x4000 + abs(tdd / 2)) / tdd;
    int s = (tdb * x + 32) >> 6;
    s = s < -4096 ? -4096 : (s > 4095 ? 4095 : s);
    int ax = (s * mx + 127 + (s * mx < 0)) >> 8, ay = (s * my + 127 + (s * my < 0)) >> 8;
    mx = ax < -32768 ? -32768 : (ax > 32767 ? 32767 : ax); my = ay < -32768 ? -32768 : (ay > 32767 ? 32767 : ay);
}
/* getColMVP (cudata.cpp:1968-2001) on the two units it reads: the one at the position, and the first of its 16x16 block (the compressed field) */
XA_DEV bool chain_col_mvp(const XaChainJob& J, int& mx, int& my, int list, const XaMapUnit& u, const XaMapUnit& c)
{
    const x265amd_mvpred_info& I = J.info;
    if (u.pred_mode == X265AMD_MODE_NONE || c.pred_mode == X265AMD_MODE_INTRA) return false;
    int cl = I.check_ldc ? list : I.col_from_l0;
    int ci = c.ref_idx[cl];
    if (ci < 0) { cl = !cl; ci = c.ref_idx[cl]; if (ci < 0) return false; }
    mx = c.mv[cl][0]; my = c.mv[cl][1];
    chain_scale(mx, my, I.poc, I.ref_poc[list][0], I.col_poc, I.col_ref_poc[cl][ci]);
    return true;
}

/* CUData::getInterMergeCandidates for a 2Nx2N CU (host form: host/mv_pred.cpp, x265amd_merge_candidates); every lane of the calling wavefront computes the same */
XA_DEV int chain_merge_candidates(const XaChainJob& J, int px, int py, int size, ChainCand* out)
{
    const x265amd_mvpred_info& I = J.info;
    const XaMapUnit* cur = reinterpret_cast<const XaMapUnit*>(J.cur);
    const int maxCand = I.max_num_merge_cand, isB = I.is_inter_b, w4 = J.w4, W = I.pic_width, H = I.pic_height;
    for (int i = 0; i < 5; i++) { out[i].mv[0][0] = out[i].mv[0][1] = out[i].mv[1][0] = out[i].mv[1][1] = 0; out[i].ref_idx[0] = out[i].ref_idx[1] = -1; out[i].dir = 0; out[i].valid = 0; }
    const int ltx = px, lty = py, rtx = px + size - 4, lby = py + size - 4;
    /* the five neighbours (cudata.cpp:605-760): inside the picture and coded before this CU */
    bool hA1 = ltx - 4 >= 0, hB1 = lty - 4 >= 0, hB2 = ltx - 4 >= 0 && lty - 4 >= 0, hB0, hA0;
    {
        /* above right of (rtx, lty) */
        hB0 = false;
        if (rtx + 4 < W && lty - 4 >= 0)
        {
            const int ux = rtx & 63, uy = lty & 63;
            if (ux < 60) hB0 = uy ? chain_z(ux, uy) > chain_z(ux + 4, uy - 4) : true;
            else hB0 = uy == 0;
        }
        /* below left of (ltx, lby) */
        hA0 = false;
        if (lby + 4 < H && ltx - 4 >= 0)
        {
            const int ux = ltx & 63, uy = lby & 63;
            if (uy < 60) hA0 = ux ? chain_z(ux, uy) > chain_z(ux - 4, uy + 4) : true;
        }
    }
    XaMapUnit a1, b1, b0, a0, b2;
    a1 = b1 = b0 = a0 = b2 = XaMapUnit{};
    if (hA1) a1 = chain_ld_unit(cur + (lby >> 2) * w4 + ((ltx - 4) >> 2));
    if (hB1) b1 = chain_ld_unit(cur + ((lty - 4) >> 2) * w4 + (rtx >> 2));
    if (hB0) b0 = chain_ld_unit(cur + ((lty - 4) >> 2) * w4 + ((rtx + 4) >> 2));
    if (hA0) a0 = chain_ld_unit(cur + ((lby + 4) >> 2) * w4 + ((ltx - 4) >> 2));
    if (hB2) b2 = chain_ld_unit(cur + ((lty - 4) >> 2) * w4 + ((ltx - 4) >> 2));
    /* the temporal candidate's units: the right-bottom position H (cudata.cpp:1628-1652) and the centre -- fetched with the others (every load is a memory round trip) */
    XaMapUnit hU = XaMapUnit{}, hC = XaMapUnit{}, cU = XaMapUnit{}, cC = XaMapUnit{};
    bool haveH = false;
    if (I.temporal_mvp)
    {
        const XaMapUnit* col = reinterpret_cast<const XaMapUnit*>(J.col);
        const int rbx = px + size - 4, rby = py + size - 4;
        if (rbx + 4 < W && rby + 4 < H && (rby & 63) < 60)
        {
            const int hx = rbx + 4, hy = rby + 4;
            haveH = true;
            hU = chain_ld_unit(col + (hy >> 2) * w4 + (hx >> 2)); hC = chain_ld_unit(col + ((hy & ~15) >> 2) * w4 + ((hx & ~15) >> 2));
        }
        const int cx = px + size / 2, cy = py + size / 2;
        cU = chain_ld_unit(col + (cy >> 2) * w4 + (cx >> 2)); cC = chain_ld_unit(col + ((cy & ~15) >> 2) * w4 + ((cx & ~15) >> 2));
    }
    int count = 0;
    auto take = [&](const XaMapUnit& u) {
        out[count].dir = u.inter_dir;
        out[count].mv[0][0] = u.mv[0][0]; out[count].mv[0][1] = u.mv[0][1]; out[count].ref_idx[0] = u.ref_idx[0];
        if (isB) { out[count].mv[1][0] = u.mv[1][0]; out[count].mv[1][1] = u.mv[1][1]; out[count].ref_idx[1] = u.ref_idx[1]; }
        return ++count == maxCand;
    };
    const bool availA1 = chain_inter(hA1, a1);
    if (availA1 && take(a1)) return maxCand;
    const bool availB1 = chain_inter(hB1, b1);
    if (availB1 && (!availA1 || !chain_same_motion(a1, b1)) && take(b1)) return maxCand;
    const bool availB0 = chain_inter(hB0, b0);
    if (availB0 && (!availB1 || !chain_same_motion(b1, b0)) && take(b0)) return maxCand;
    const bool availA0 = chain_inter(hA0, a0);
    if (availA0 && (!availA1 || !chain_same_motion(a1, a0)) && take(a0)) return maxCand;
    if (count < 4)
    {
        if (chain_inter(hB2, b2) && (!availA1 || !chain_same_motion(a1, b2)) && (!availB1 || !chain_same_motion(b1, b2)) && take(b2)) return maxCand;
    }
    if (I.temporal_mvp)
    {
        int dir = 0;
        for (int list = 0; list < (isB ? 2 : 1); list++)
        {
            int mx = 0, my = 0;
            bool ok = haveH && chain_col_mvp(J, mx, my, list, hU, hC);
            if (!ok) ok = chain_col_mvp(J, mx, my, list, cU, cC);
            if (ok) { dir |= 1 << list; out[count].mv[list][0] = (int16_t)mx; out[count].mv[list][1] = (int16_t)my; out[count].ref_idx[list] = 0; }
        }
        if (dir) { out[count].dir = (uint8_t)dir; if (++count == maxCand) return maxCand; }
    }
    if (isB)
    {
        const unsigned cutoff = (unsigned)(count * (count - 1));
        unsigned pl0 = 0xEDC984, pl1 = 0xB73621;
        for (unsigned k = 0; k < cutoff; k++, pl0 >>= 2, pl1 >>= 2)
        {
            const int i = pl0 & 3, j = pl1 & 3;
            if ((out[i].dir & 1) && (out[j].dir & 2))
            {
                const int r0 = out[i].ref_idx[0], r1 = out[j].ref_idx[1];
                if (!(I.ref_poc[0][r0] == I.ref_poc[1][r1] && out[i].mv[0][0] == out[j].mv[1][0] && out[i].mv[0][1] == out[j].mv[1][1]))
                {
                    out[count].mv[0][0] = out[i].mv[0][0]; out[count].mv[0][1] = out[i].mv[0][1]; out[count].ref_idx[0] = (int8_t)r0;
                    out[count].mv[1][0] = out[j].mv[1][0]; out[count].mv[1][1] = out[j].mv[1][1]; out[count].ref_idx[1] = (int8_t)r1;
                    out[count].dir = 3;
                    if (++count == maxCand) return maxCand;
                }
            }
        }
    }
    const int numRef = isB ? (I.num_ref_idx[0] < I.num_ref_idx[1] ? I.num_ref_idx[0] : I.num_ref_idx[1]) : I.num_ref_idx[0];
    int r = 0, refcnt = 0;
    while (count < maxCand)
    {
        out[count].dir = isB ? 3 : 1;
        out[count].mv[0][0] = out[count].mv[0][1] = 0; out[count].ref_idx[0] = (int8_t)r;
        if (isB) { out[count].mv[1][0] = out[count].mv[1][1] = 0; out[count].ref_idx[1] = (int8_t)r; }
        count++;
        if (refcnt == numRef - 1) r = 0;
        else { ++r; ++refcnt; }
    }
    return count;
}

/* what the prediction of a candidate reads of its reference pictures, against what they have published (ctu_analysis.hip: xa_ref_guard_mc; encoder_api.hip:
 * gateRefWait / gateCtuWait) */
XA_DEV bool chain_reach_ok(const XaChainJob& J, int x, int y, int size, const ChainCand& c)
{
    if (!J.guard_on) return true;
    const int W = J.info.pic_width, H = J.info.pic_height;
    for (int l = 0; l < 2; l++)
    {
        if (!(c.dir & (1 << l)) || c.ref_idx[l] < 0) continue;
        /* CUData::clipMv first: the prediction reads with the clipped vector */
        const int xmax = (W + 8 - x - 1) << 2, xmin = -((64 + 8 + x - 1) << 2), ymax = (H + 8 - y - 1) << 2, ymin = -((64 + 8 + y - 1) << 2);
        const int mvx = min(xmax, max(xmin, (int)c.mv[l][0])), mvy = min(ymax, max(ymin, (int)c.mv[l][1]));
        const int y0 = y + (mvy >> 2) - 4, y1 = y + size - 1 + (mvy >> 2) + 5, x1 = x + size - 1 + (mvx >> 2) + 5;
        const int need = x1 >= W - 1 ? W : max(0, x1 + 1);
        const int r0 = min(max(y0, 0), H - 1) >> 6, r1 = min(max(y1, 0), H - 1) >> 6;
        if (r0 < J.guard_r0 || r1 > J.guard_r1 || need > J.guard_need) return false;
    }
    return true;
}

/* an N x N block, rows in pieces of `unit` samples (8 luma / 4 chroma: block positions and strides are multiples of that) */
template<class V> XA_DEV void chain_copy_rows(pixel* dst, long ds, const pixel* src, int ss, int log2N, int tid, int nthr)
{
    constexpr int U = (int)(sizeof(V) / sizeof(pixel));
    const int N = 1 << log2N, perRow = N / U, total = perRow * N;
    for (int i = tid; i < total; i += nthr)
    {
        const int yy = i / perRow, xx = (i - yy * perRow) * U;
        *reinterpret_cast<V*>(dst + (long)yy * ds + xx) = *reinterpret_cast<const V*>(src + yy * ss + xx);
    }
}
XA_DEV void chain_copy_plane(pixel* dst, long ds, const pixel* src, int ss, int log2N, bool chroma, int tid, int nthr)
{
    if (sizeof(pixel) == 1) { if (chroma) chain_copy_rows<uint32_t>(dst, ds, src, ss, log2N, tid, nthr); else chain_copy_rows<uint64_t>(dst, ds, src, ss, log2N, tid, nthr); }
    else { if (chroma) chain_copy_rows<uint64_t>(dst, ds, src, ss, log2N, tid, nthr); else chain_copy_rows<uint4>(dst, ds, src, ss, log2N, tid, nthr); }
}

/* one 8x8 tile of a candidate's prediction: a lane per sample -- the sample predicted (Predict::motionCompensation), stored in the candidate's tile, and the
 * Hadamard sum of its difference to the source left in every lane (sa8d_8x8 before its rounding, pixel.cpp:342-360) */
template<int TAPS> XA_DEV int chain_pred_tile(const x265amd_mc_job& j, const McSetup& su, const McPlane& P, const pixel* fenc, int fs, int tx, int ty, int lane)
{
    const int x = 8 * tx + (lane & 7), y = 8 * ty + (lane >> 3);
    const int v = mc_one<TAPS>(P, j, su.mode, su.lsel, x, y);
    P.dst[(long)y * P.dstStride + x] = (pixel)v;
    return xa_wave_sum(abs(xa_lane_had8x8((int)fenc[y * fs + x] - v, lane)));
}

/* the residual of one transform unit: does it quantise to nothing?  (Quant::transformNxN without sign-bit hiding: a unit without levels has none to hide) */
template<class G> XA_DEV uint32_t chain_tu_levels(TuLds& s, const pixel* fenc, int fs, const pixel* pred, int ps, int log2N, int ttype, int sliceType, int qpScaled, const G& g)
{
    const int N = 1 << log2N;
    for (int i = g.idx; i < N * N; i += g.step())
    {
        const int y = i >> log2N, x = i & (N - 1);
        s.a[i] = (int16_t)((int)fenc[y * fs + x] - (int)pred[y * ps + x]);
    }
    g.sync();
    return grp_tu_forward(s, log2N, ttype, 0, 0, sliceType, qpScaled, 0, g);
}


#if XA_DEPTH < 10
typedef uint32_t chain_sse_t;
#else
typedef uint64_t chain_sse_t;
#endif
/* RDCost::calcRdCost / calcPsyRdCost (rdcost.h:99-153) */
XA_DEV uint64_t chain_cost(const XaChainJob& J, chain_sse_t dist, uint32_t bits, uint32_t energy)
{
    return J.psy_rd ? (uint64_t)dist + ((J.lambda * J.psy_rd * energy) >> 24) + (((uint64_t)bits * J.lambda2) >> 8) : (uint64_t)dist + (((uint64_t)bits * J.lambda2 + 128) >> 8);
}
enum { CC_SPLIT = 0, CC_SKIP = 3, CC_MERGE_FLAG = 6, CC_MERGE_IDX = 7, CC_PART_SIZE = 8, CC_PRED_MODE = 12, CC_QT_CBF = 28, CC_QT_ROOT_CBF = 38 };      /* host/cabac_coder.h */
/* Entropy::codeMergeIndex in counting mode (entropy.cpp:1572-1590) */
XA_DEV uint64_t chain_merge_index(uint8_t* ctx, uint32_t idx, uint32_t numCand)
{
    uint64_t f = 0;
    if (numCand > 1)
    {
        f += cb_bin(ctx + CC_MERGE_IDX, idx != 0);
        if (idx != 0) f += 32768ull * (idx - (idx == numCand - 1));
    }
    return f;
}
/* the split flag's context (CUData::getCtxSplitFlag): neighbours inside the picture are coded when a CU is reached */
XA_DEV int chain_split_ctx(const XaChainJob& J, int x, int y, int depth)
{
    const XaMapUnit* cur = reinterpret_cast<const XaMapUnit*>(J.cur);
    int c = 0;
    if (x > 0) { const XaMapUnit l = chain_ld_unit(cur + (y >> 2) * J.w4 + ((x - 4) >> 2)); c += l.pred_mode != X265AMD_MODE_NONE && l.depth > depth; }
    if (y > 0) { const XaMapUnit a = chain_ld_unit(cur + ((y - 4) >> 2) * J.w4 + (x >> 2)); c += a.pred_mode != X265AMD_MODE_NONE && a.depth > depth; }
    return c;
}
XA_DEV int chain_skip_ctx(const XaChainJob& J, int x, int y)
{
    const XaMapUnit* cur = reinterpret_cast<const XaMapUnit*>(J.cur);
    int c = 0;
    if (x > 0) { const XaMapUnit l = chain_ld_unit(cur + (y >> 2) * J.w4 + ((x - 4) >> 2)); c += l.pred_mode == X265AMD_MODE_SKIP; }
    if (y > 0) { const XaMapUnit a = chain_ld_unit(cur + ((y - 4) >> 2) * J.w4 + (x >> 2)); c += a.pred_mode == X265AMD_MODE_SKIP; }
    return c;
}
/* Analysis::addSplitFlagCost on the running state (analysis.cpp:3405-3426); rd level 2 counts a bit without touching the coder */
XA_DEV void chain_split_flag(const XaChainJob& J, uint8_t* ctx, uint64_t& frac, int x, int y, int depth, uint32_t flag)
{
    if (J.rd_level == 2) return;
    frac &= 32767;
    frac += cb_bin(ctx + CC_SPLIT + chain_split_ctx(J, x, y, depth), flag);
}

/* Entropy::codeDeltaQP's bins (entropy.cpp:1737-1756) in counting mode: the first bin on context 16, the rest of the unary prefix (to 5) on 17, EG0 suffix and sign in bypass */
XA_DEV uint64_t chain_dqp_bits(uint8_t* cx, int dqp)
{
    const uint32_t a = (uint32_t)(dqp < 0 ? -dqp : dqp);
    uint64_t f = cb_bin(&cx[16], a ? 1u : 0u);
    if (a)
    {
        uint32_t sym = a < 5 ? a : 5;
        const bool codeLast = 5 > sym;
        while (--sym) f += cb_bin(&cx[17], 1u);
        if (codeLast) f += cb_bin(&cx[17], 0u);
        if (a >= 5)
        {
            uint32_t symbol = a - 5, count = 0, n = 0;
            while (symbol >= (1u << count)) { n++; symbol -= 1u << count; count++; }
            f += (uint64_t)32768 * (n + 1 + count);
        }
        f += 32768;         /* the sign */
    }
    return f;
}
/* CUData::getRefQP (cudata.cpp:814-855) of quantisation group k from what the job says about the groups in front of it and what the chain has completed itself */
XA_DEV int chain_qg_qp(const ChainLds& S, int j, int hostVal) { return (S.qgDone >> j) & 1u ? S.qgVal[j] : hostVal; }
XA_DEV int chain_ref_qp(const ChainLds& S, int k)
{
    const XaChainJob& J = S.job;
    const int last = J.last_src[k] < 0 ? J.prev_qp : chain_qg_qp(S, J.last_src[k], J.last_val[k]);
    if (J.max_dqp_depth == 0 || k == 0) return last;
    const int l = (k & 1) ? chain_qg_qp(S, k - 1, J.left_val[k]) : last;
    const int a = (k & 2) ? chain_qg_qp(S, k - 2, J.above_val[k]) : last;
    return (l + a + 1) >> 1;
}

/* encodeResAndCalcRdSkipCU and encodeResAndCalcRdInterCU (search.cpp:2770-2975) of the chosen merge candidate of a CU with one transform unit per plane (8x8 ..
 * 32x32: the residual quad-tree is its root), from the units' results S.tr[] and levels, and the choice between them as checkMerge2Nx2N_rd0_4 makes it
 * (analysis.cpp:2852-2880: the residual mode only when strictly cheaper).  Host form: x265amd_skip_rd_host, inter_rd_walk_impl, x265amd_inter_rd_finish (inter_rd.hip).
 * One wavefront; leaves S.skipWins and, in S.ctxS / S.fracS, the skip mode's coder state. */
XA_DEV void chain_merge_rd(ChainLds& S, int x, int y, int log2, int best, int lane)
{
    const XaChainJob& J = S.job;
    const uint32_t numCand = (uint32_t)J.info.max_num_merge_cand;
    for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) { const uint8_t v = S.ctx[i]; S.ctxS[i] = v; S.ctxB[i] = v; S.ctxD[i] = v; }
    xa_wave_sync();
    const int skipCtx = chain_skip_ctx(J, x, y);
    const x265amd_cu_measure m0 = S.meas[best];
    const chain_sse_t predDist = (chain_sse_t)((chain_sse_t)m0.sse[0] + (chain_sse_t)m0.sse[1] + (chain_sse_t)m0.sse[2]);
    const uint32_t predPsy = J.psy_rd ? m0.psy : 0;
    /* ---- the skip mode ---- */
    uint64_t fS = S.frac & 32767;
    if (lane == 0)
    {
        fS += cb_bin(S.ctxS + CC_SKIP + skipCtx, 1);
        fS += chain_merge_index(S.ctxS, (uint32_t)best, numCand);
        S.fracS = fS;
    }
    xa_wave_sync();
    fS = S.fracS;
    const uint64_t skipCost = chain_cost(J, predDist, (uint32_t)(fS >> 15), predPsy);
    bool anyLevel = false;
    for (int p = 0; p < 3; p++) anyLevel |= S.tr[p].num_sig != 0;
    if (!anyLevel) { if (lane == 0) S.skipWins = 1; return; }
    /* ---- the residual mode: estimateResidualQT of the root (search.cpp:3276-3497) ---- */
    const int logs[3] = { log2, log2 - 1 < 2 ? 2 : log2 - 1, log2 - 1 < 2 ? 2 : log2 - 1 };
    uint32_t cbf[3], singleBits[3];
    chain_sse_t singleDist[3];
    uint32_t energyY = 0;
    uint64_t fB = S.frac & 32767;                /* resetBits() in front of the luma unit */
    for (int p = 0; p < 3; p++)
    {
        const x265amd_tu_result r = S.tr[p];
        cbf[p] = r.num_sig != 0;
        const uint32_t latest = (uint32_t)(fB >> 15);
        if (cbf[p]) fB += wave_coeff_bits(S.ctxB, S.ctxB, reinterpret_cast<const int16_t*>(S.tu[p].coeff), logs[p], p, 0, 0, J.sign_hide, S.step, lane);
        xa_wave_sync();
        singleBits[p] = (uint32_t)(fB >> 15) - (p ? latest : 0);
        const chain_sse_t zeroDist = (chain_sse_t)r.zero_dist;
        const uint32_t zeroEnergy = J.psy_rd ? r.zero_energy : 0;
        if (cbf[p])
        {
            const uint8_t st = S.ctxB[CC_QT_CBF + (p ? 2 : 1)];                 /* estimateCbfBits: ctxCbf[ttype][0] */
            const uint32_t nzCbfBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 1]) >> 15), nullBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 0]) >> 15);
            const chain_sse_t nzDist = (chain_sse_t)r.nz_dist;
            const uint32_t nzEnergy = J.psy_rd ? r.nz_energy : 0;
            const uint64_t singleCost = chain_cost(J, nzDist, nzCbfBits + singleBits[p], nzEnergy), nullCost = chain_cost(J, zeroDist, nullBits, zeroEnergy);
            if (nullCost < singleCost) { cbf[p] = 0; singleBits[p] = 0; singleDist[p] = zeroDist; if (!p) energyY = zeroEnergy; }
            else { singleDist[p] = nzDist; if (!p) energyY = nzEnergy; }
        }
        else { singleBits[p] = 0; singleDist[p] = zeroDist; if (!p) energyY = zeroEnergy; }
    }
    chain_sse_t fullDist = 0;
    fullDist += singleDist[0]; fullDist += singleDist[1]; fullDist += singleDist[2];
    const uint64_t fullCost = chain_cost(J, fullDist, singleBits[0] + singleBits[1] + singleBits[2], energyY);
    /* the cost of not signalling any residual (search.cpp:2869-2895) */
    const uint32_t cbf0Bits = (uint32_t)(((S.frac & 32767) + en_bits[S.ctx[CC_QT_ROOT_CBF] ^ 0]) >> 15);
    if (chain_cost(J, predDist, cbf0Bits, predPsy) < fullCost) cbf[0] = cbf[1] = cbf[2] = 0;
    if (!(cbf[0] | cbf[1] | cbf[2])) { if (lane == 0) S.skipWins = 1; return; }         /* the residual mode has become the skip mode: not cheaper */
    /* ---- the bits of the CU coded with its residual (search.cpp:2900-2930; Entropy::encodeTransform, entropy.cpp:930-1063) ---- */
    uint64_t fD = S.frac & 32767;
    if (lane == 0)
    {
        fD += cb_bin(S.ctxD + CC_SKIP + skipCtx, 0);
        S.rdBits[2] = (uint32_t)(fD >> 15);
        fD += cb_bin(S.ctxD + CC_PRED_MODE, 0);
        fD += cb_bin(S.ctxD + CC_PART_SIZE, 1);
        fD += cb_bin(S.ctxD + CC_MERGE_FLAG, 1);
        fD += chain_merge_index(S.ctxD, (uint32_t)best, numCand);
        S.rdBits[1] = (uint32_t)(fD >> 15) - S.rdBits[2];
        fD += cb_bin(S.ctxD + CC_QT_CBF + 2, cbf[1]);
        fD += cb_bin(S.ctxD + CC_QT_CBF + 2, cbf[2]);
        if (cbf[1] | cbf[2]) fD += cb_bin(S.ctxD + CC_QT_CBF + 1, cbf[0]);
        /* cu_qp_delta with the first coded block flag (encodeTransform with bCodeDQP, entropy.cpp:1207-1222) */
        if (J.use_dqp) fD += chain_dqp_bits(S.ctxD, S.dqp);
    }
    xa_wave_sync();
    fD = __shfl(fD, 0, 64);
    for (int p = 0; p < 3; p++)
    {
        if (cbf[p]) fD += wave_coeff_bits(S.ctxD, S.ctxD, reinterpret_cast<const int16_t*>(S.tu[p].coeff), logs[p], p, 0, 0, J.sign_hide, S.step, lane);
        xa_wave_sync();
    }
    const uint32_t bits0 = (uint32_t)(fD >> 15);
    uint32_t again = 0;
    if (J.use_dqp && S.dqpTwice)
    {
        /* Search::checkDQP inside encodeResAndCalcRdInterCU (search.cpp:3974-4003) for a CU at or above the groups' depth: resetBits(), codeDeltaQP, the bits added */
        if (lane == 0) S.fracD = (fD & 32767) + chain_dqp_bits(S.ctxD, S.dqp);
        xa_wave_sync();
        fD = S.fracD;
        again = (uint32_t)(fD >> 15);
    }
    chain_sse_t dist = 0;
    dist += cbf[0] ? (chain_sse_t)S.tr[0].nz_dist : (chain_sse_t)S.tr[0].zero_dist;
    chain_sse_t cd = cbf[1] ? (chain_sse_t)S.tr[1].nz_dist : (chain_sse_t)S.tr[1].zero_dist;
    cd += cbf[2] ? (chain_sse_t)S.tr[2].nz_dist : (chain_sse_t)S.tr[2].zero_dist;
    dist += cd;
    const uint32_t psy = J.psy_rd ? (cbf[0] ? S.tr[0].nz_energy : S.tr[0].zero_energy) : 0;
    const uint32_t totalBits = bits0 + again;
    const uint64_t mergeCost = chain_cost(J, dist, totalBits, psy);
    if (lane == 0)
    {
        S.skipWins = mergeCost < skipCost ? 0 : 1;
        S.fracD = fD; S.rdBits[0] = totalBits; S.rdPad = again; S.rdPsy = psy; S.rdCbf[0] = cbf[0]; S.rdCbf[1] = cbf[1]; S.rdCbf[2] = cbf[2];
        S.rdCost = mergeCost; S.rdLuma = (uint64_t)(dist - cd); S.rdChroma = (uint64_t)cd;
    }
}

/* The same decision for a CU one size above the largest transform (64x64 with 32x32 transforms): its residual is a tree of four nodes one level down, each a luma
 * unit and the two chroma units under it (S.tu / S.tr: Y 0..3, U 4..7, V 8..11 in z-order).  estimateResidualQT at the root cannot code the CU as one node, so it is
 * splitTU alone (search.cpp:3126-3176, :3178-3497): every node starts from the state the node before it left -- which is the CU's start state plus the coded block flags
 * coded so far, because each node's coefficient contexts are dropped again (load(rqtRoot), :3652) -- the nodes' bits are their coefficients' alone (no subdivision flag
 * to choose), and the flags of the whole tree are priced once from the CU's start state (codeInterSubdivCbfQT, :3859-3887).  Then the cost of signalling nothing and
 * the CU's real bit count (Entropy::encodeTransform, entropy.cpp:930-1063: chroma flags of the root, then per node its chroma flags when the root's are set, its luma
 * flag, cu_qp_delta with the first node that has anything, its coefficients).  Only S.skipWins is the caller's: a residual mode that wins goes back to the host. */
XA_DEV void chain_merge_rd64(ChainLds& S, int x, int y, int log2, int best, int lane, uint32_t sig)
{
    const XaChainJob& J = S.job;
    const uint32_t numCand = (uint32_t)J.info.max_num_merge_cand;
    const int L = log2 - 1, C = L - 1;
    for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) { const uint8_t v = S.ctx[i]; S.ctxS[i] = v; S.ctxB[i] = v; S.ctxD[i] = v; S.ctxQ[i] = v; }
    xa_wave_sync();
    const int skipCtx = chain_skip_ctx(J, x, y);
    /* sig: the units that have levels (bit k); the others' share of every sum is what the prediction leaves, which the CU's measurement has for the whole CU */
    const x265amd_cu_measure m0 = S.meas[best];
    const chain_sse_t predDist = (chain_sse_t)((chain_sse_t)m0.sse[0] + (chain_sse_t)m0.sse[1] + (chain_sse_t)m0.sse[2]);
    const uint32_t predPsy = J.psy_rd ? m0.psy : 0;
    /* ---- the skip mode ---- */
    uint64_t fS = S.frac & 32767;
    if (lane == 0)
    {
        fS += cb_bin(S.ctxS + CC_SKIP + skipCtx, 1);
        fS += chain_merge_index(S.ctxS, (uint32_t)best, numCand);
        S.fracS = fS;
    }
    xa_wave_sync();
    fS = S.fracS;
    const uint64_t skipCost = chain_cost(J, predDist, (uint32_t)(fS >> 15), predPsy);
    /* ---- the residual mode: the four nodes ---- */
    uint32_t cbf[12];
    uint32_t treeBits = 0, treeEnergy = predPsy;
    chain_sse_t treeDist = predDist;
    uint64_t fN = S.frac & 32767;               /* the state a node starts from: S.ctxQ and this fraction */
    for (int q = 0; q < 4; q++)
    {
        uint64_t fB = fN & 32767;               /* resetBits() in front of the luma unit */
        for (int p = 0; p < 3; p++)
        {
            const int k = p * 4 + q, lg = p ? C : L;
            if (!((sig >> k) & 1)) { cbf[k] = 0; continue; }
            const x265amd_tu_result r = S.tr[k];
            uint32_t c = r.num_sig != 0;
            const uint32_t latest = (uint32_t)(fB >> 15);
            if (c) fB += wave_coeff_bits(S.ctxB, S.ctxB, reinterpret_cast<const int16_t*>(S.tu[k].coeff), lg, p, 0, 0, J.sign_hide, S.step, lane);
            xa_wave_sync();
            uint32_t single = (uint32_t)(fB >> 15) - (p ? latest : 0);
            const chain_sse_t zeroDist = (chain_sse_t)r.zero_dist;
            const uint32_t zeroEnergy = J.psy_rd ? r.zero_energy : 0;
            chain_sse_t dist = zeroDist;
            uint32_t energy = zeroEnergy;
            if (c)
            {
                const uint8_t st = S.ctxB[CC_QT_CBF + (p ? 3 : 0)];                 /* estimateCbfBits one level down: ctxCbf[ttype][1] */
                const uint32_t nzCbfBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 1]) >> 15), nullBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 0]) >> 15);
                const chain_sse_t nzDist = (chain_sse_t)r.nz_dist;
                const uint32_t nzEnergy = J.psy_rd ? r.nz_energy : 0;
                const uint64_t singleCost = chain_cost(J, nzDist, nzCbfBits + single, nzEnergy), nullCost = chain_cost(J, zeroDist, nullBits, zeroEnergy);
                if (nullCost < singleCost) { c = 0; single = 0; }
                else { dist = nzDist; energy = nzEnergy; }
            }
            else single = 0;
            cbf[k] = c;
            treeBits += single; treeDist = treeDist - zeroDist + dist;
            if (!p) treeEnergy = treeEnergy - zeroEnergy + energy;
        }
        /* load(rqtRoot), resetBits(), the node's three flags: what the next node starts from */
        for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) S.ctxB[i] = S.ctxQ[i];
        xa_wave_sync();
        if (lane == 0)
        {
            uint64_t f = fN & 32767;
            f += cb_bin(S.ctxB + CC_QT_CBF + 3, cbf[4 + q]);
            f += cb_bin(S.ctxB + CC_QT_CBF + 3, cbf[8 + q]);
            f += cb_bin(S.ctxB + CC_QT_CBF + 0, cbf[q]);
            S.fracD = f;
        }
        xa_wave_sync();
        fN = S.fracD;
        for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) S.ctxQ[i] = S.ctxB[i];
        xa_wave_sync();
    }
    const uint32_t ycbf = cbf[0] | cbf[1] | cbf[2] | cbf[3], ucbf = cbf[4] | cbf[5] | cbf[6] | cbf[7], vcbf = cbf[8] | cbf[9] | cbf[10] | cbf[11];
    /* the tree's flags from the CU's start state (splitTU's tail) */
    for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) S.ctxQ[i] = S.ctx[i];
    xa_wave_sync();
    if (lane == 0)
    {
        uint64_t f = S.frac & 32767;
        f += cb_bin(S.ctxQ + CC_QT_CBF + 2, ucbf);
        f += cb_bin(S.ctxQ + CC_QT_CBF + 2, vcbf);
        for (int q = 0; q < 4; q++)
        {
            if (ucbf) f += cb_bin(S.ctxQ + CC_QT_CBF + 3, cbf[4 + q]);
            if (vcbf) f += cb_bin(S.ctxQ + CC_QT_CBF + 3, cbf[8 + q]);
            f += cb_bin(S.ctxQ + CC_QT_CBF + 0, cbf[q]);
        }
        S.fracD = f;
    }
    xa_wave_sync();
    treeBits += (uint32_t)(S.fracD >> 15);
    const uint64_t treeCost = chain_cost(J, treeDist, treeBits, treeEnergy);
    /* the cost of not signalling any residual (search.cpp:2869-2895) */
    const uint32_t cbf0Bits = (uint32_t)(((S.frac & 32767) + en_bits[S.ctx[CC_QT_ROOT_CBF] ^ 0]) >> 15);
    if (chain_cost(J, predDist, cbf0Bits, predPsy) < treeCost || !(ycbf | ucbf | vcbf)) { if (lane == 0) S.skipWins = 1; return; }       /* the residual mode has become the skip mode: not cheaper */
    /* ---- the bits of the CU coded with its residual (search.cpp:2900-2930) ---- */
    uint64_t fD = S.frac & 32767;
    if (lane == 0)
    {
        fD += cb_bin(S.ctxD + CC_SKIP + skipCtx, 0);
        fD += cb_bin(S.ctxD + CC_PRED_MODE, 0);
        fD += cb_bin(S.ctxD + CC_PART_SIZE, 1);
        fD += cb_bin(S.ctxD + CC_MERGE_FLAG, 1);
        fD += chain_merge_index(S.ctxD, (uint32_t)best, numCand);
        fD += cb_bin(S.ctxD + CC_QT_CBF + 2, ucbf);
        fD += cb_bin(S.ctxD + CC_QT_CBF + 2, vcbf);
    }
    bool dqpPending = J.use_dqp != 0;
    for (int q = 0; q < 4; q++)
    {
        const bool any = (cbf[q] | cbf[4 + q] | cbf[8 + q]) != 0;
        if (lane == 0)
        {
            if (ucbf) fD += cb_bin(S.ctxD + CC_QT_CBF + 3, cbf[4 + q]);
            if (vcbf) fD += cb_bin(S.ctxD + CC_QT_CBF + 3, cbf[8 + q]);
            fD += cb_bin(S.ctxD + CC_QT_CBF + 0, cbf[q]);
            if (any && dqpPending) fD += chain_dqp_bits(S.ctxD, S.dqp);          /* cu_qp_delta with the first node that has anything (entropy.cpp:1207-1222) */
        }
        if (any) dqpPending = false;
        xa_wave_sync();
        fD = __shfl(fD, 0, 64);
        for (int p = 0; p < 3; p++)
        {
            const int k = p * 4 + q;
            if (cbf[k]) fD += wave_coeff_bits(S.ctxD, S.ctxD, reinterpret_cast<const int16_t*>(S.tu[k].coeff), p ? C : L, p, 0, 0, J.sign_hide, S.step, lane);
            xa_wave_sync();
        }
    }
    const uint32_t bits0 = (uint32_t)(fD >> 15);
    uint32_t again = 0;
    if (J.use_dqp && S.dqpTwice)
    {
        /* Search::checkDQP inside encodeResAndCalcRdInterCU (search.cpp:3974-4003): resetBits(), codeDeltaQP, the bits added */
        if (lane == 0) S.fracD = (fD & 32767) + chain_dqp_bits(S.ctxD, S.dqp);
        xa_wave_sync();
        again = (uint32_t)(S.fracD >> 15);
    }
    chain_sse_t dist = predDist;
    uint32_t psy = predPsy;
    for (int k = 0; k < 12; k++) if (cbf[k]) dist = dist - (chain_sse_t)S.tr[k].zero_dist + (chain_sse_t)S.tr[k].nz_dist;
    if (J.psy_rd) for (int k = 0; k < 4; k++) if (cbf[k]) psy = psy - S.tr[k].zero_energy + S.tr[k].nz_energy;
    const uint64_t mergeCost = chain_cost(J, dist, bits0 + again, psy);
    if (lane == 0) S.skipWins = mergeCost < skipCost ? 0 : 1;
}

XA_DEV void block_inter_chain(const XaChainJob* jobAddr, char* smem, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    static_assert(sizeof(ChainLds) <= XA_CHAIN_HEADER, "chain header");
    static_assert(XA_CHAIN_HEADER + sizeof(CuMeasureLds) <= XA_SERVER_LDS && XA_CHAIN_HEADER + (XA_SERVER_WAVES + 1) * sizeof(TuLds) <= XA_SERVER_LDS, "LDS budget");
    ChainLds& S = *reinterpret_cast<ChainLds*>(smem);
    CuMeasureLds& ML = *reinterpret_cast<CuMeasureLds*>(smem + XA_CHAIN_HEADER);
    TuLds* TL = reinterpret_cast<TuLds*>(smem + XA_CHAIN_HEADER);
    __syncthreads();
    /* the job record and the node table: written by the host through the BAR */
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(jobAddr);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&S.job);
        for (int i = tid; i < (int)(sizeof(XaChainJob) / 8); i += NT) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(S.job.nodes);
        uint64_t* dst = reinterpret_cast<uint64_t*>(S.nodes);
        for (int i = tid; i < S.job.num_nodes; i += NT) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0)
        {
            S.count = 0; S.stop = 0; S.frac = S.job.frac; for (int k = 0; k < 8; k++) S.ticks[k] = 0; S.tprev = wall_clock64();
            S.firstQp = S.job.first_qp; S.qgDone = 0; S.dqp = 0; S.dqpTwice = 0; S.rdPad = 0;
        }
        for (int i = tid; i < 256; i += NT) S.step[i] = en_step.v[i];
        if (tid == 1) __hip_atomic_store(reinterpret_cast<uint64_t*>(&reinterpret_cast<XaChainOut*>(S.job.out)->stop), (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = tid; i < X265AMD_CTX_STRIDE; i += NT) S.ctx[i] = S.job.ctx[i];
    }
    __syncthreads();
    const XaChainJob& J = S.job;
    if (tid == 0)
        for (int p = S.nodes[J.start].parent; p != 255; p = S.nodes[p].parent) S.used[p] = J.anc_flags[(6 - S.nodes[p].log2) & 3];      /* the host's recursion came through these */
    const uint64_t* planes = reinterpret_cast<const uint64_t*>(J.planes);
    const uint64_t* srcPlanes = planes + 3 * (J.num_pics - 1);
    const uint64_t* recPlanes = planes + 3 * (J.num_pics - 2);
    XaChainOut* out = reinterpret_cast<XaChainOut*>(J.out);
    const size_t isz = sizeof(pixel);
    int node = J.start, reason = XA_CHAIN_END;
    while (node < J.end)
    {
        const XaChainNode N = S.nodes[node];
        /* topSkipMinDepth's QP test reads the CTU's first unit as the picture's records hold it when the recursion arrives here (S.firstQp) */
        const int flags = S.firstQp >= J.previous_qp ? N.flags & 3 : (N.flags >> 2) & 3;
        if (tid == 0) S.used[node] = (uint8_t)flags;
        if (!(flags & 1)) { node++; continue; }       /* not coded at this depth: into its first sub-CU */
        const int x = N.x, y = N.y, log2 = N.log2, size = 1 << log2, depth = 6 - log2;
        /* the quantisation group of the CU and the QP in force for it: lambdas, quantiser, cu_qp_delta */
        const int qg = J.max_dqp_depth == 0 ? 0 : ((y - J.ctu_y) >> 5) * 2 + ((x - J.ctu_x) >> 5);
        if (J.use_dqp)
        {
            __syncthreads();
            if (tid == 0)
            {
                const XaChainJob::QpSet q = J.qps[J.max_dqp_depth == 0 || depth == 0 ? 0 : 1 + qg];
                XaChainJob& W = S.job;
                W.lambda = q.lambda; W.lambda2 = q.lambda2; W.psy_rd = q.psy_rd; W.qp_luma = q.qp_luma; W.qp_chroma = q.qp_chroma;
                const int bd = 6 * (X265AMD_DEPTH - 8);
                S.dqp = (q.qp - chain_ref_qp(S, depth == 0 ? 0 : qg) + 78 + bd + (bd / 2)) % (52 + bd) - 26 - (bd / 2);
                S.dqpTwice = depth <= J.max_dqp_depth;
            }
            __syncthreads();
        }
        /* ---- candidates, their jobs ---- */
        if (wv == 0)
        {
            ChainCand* cand = S.cand;          /* every lane writes the same values */
            const int nc = chain_merge_candidates(J, x, y, size, cand);
            bool stop = false;
            for (int i = 0; i < nc; i++)
            {
                /* pictures coded in parallel: candidates reaching below the rows the references have finished are left out (analysis.cpp:2789-2806) */
                const bool below = J.frame_parallel && (cand[i].mv[0][1] >= (J.search_range + 1) * 4 || cand[i].mv[1][1] >= (J.search_range + 1) * 4);
                cand[i].valid = !below;
                if (!below && !chain_reach_ok(J, x, y, size, cand[i])) stop = true;
            }
            if (lane == 0)
            {
                S.nc = nc; S.stop = stop ? XA_CHAIN_GUARD : 0; S.anyLevel = 0;
                for (int i = 0; i < nc; i++)
                {
                    x265amd_mc_job& j = S.mc[i];
                    const uint64_t tile = J.tiles + (uint64_t)(depth * J.tiles_per_depth + J.cand_tile0 + i) * J.tile_bytes;
                    j = x265amd_mc_job{};
                    j.dst_y = tile; j.dst_u = tile + 4096 * isz; j.dst_v = j.dst_u + 1024 * isz;
                    j.dst_stride = 64; j.dst_cstride = 32;
                    j.x = (int16_t)x; j.y = (int16_t)y; j.cu_x = (int16_t)x; j.cu_y = (int16_t)y; j.w = (uint8_t)size; j.h = (uint8_t)size;
                    j.ref0 = (cand[i].dir & 1) && cand[i].ref_idx[0] >= 0 ? (int8_t)J.ref_pic[0][cand[i].ref_idx[0]] : -1;
                    j.ref1 = (cand[i].dir & 2) && cand[i].ref_idx[1] >= 0 ? (int8_t)J.ref_pic[1][cand[i].ref_idx[1]] : -1;
                    j.mv0[0] = cand[i].mv[0][0]; j.mv0[1] = cand[i].mv[0][1]; j.mv1[0] = cand[i].mv[1][0]; j.mv1[1] = cand[i].mv[1][1];
                    j.slice_type = (uint8_t)!J.info.is_inter_b; j.flags = 3;
                }
            }
        }
        __syncthreads();
        XA_CHAIN_T(0);
        if (S.stop) { reason = S.stop; break; }
        const int nc = S.nc;
        if (tid == 0)
        {
            /* algorithmic bytes of the CU's merge check: nine map units read; per candidate and direction the block with its interpolation border (luma 8 taps, chroma 4), the
             * prediction written; the source block read once; the winner's units: prediction and source read again */
            unsigned long long b = 9ull * sizeof(XaMapUnit) + 3ull * size * size / 2 * sizeof(pixel) * 3;
            for (int i = 0; i < nc; i++)
                if (S.cand[i].valid)
                    b += (unsigned long long)((S.cand[i].dir & 1) + ((S.cand[i].dir >> 1) & 1)) * ((unsigned)(size + 7) * (size + 7) + 2u * (unsigned)(size / 2 + 3) * (size / 2 + 3)) * sizeof(pixel) +
                         3ull * size * size / 2 * sizeof(pixel);
            XA_BYTES(b);
        }
        /* ---- every candidate still in the race: its prediction (Predict::motionCompensation) and the SA8D of it, tile by tile over the wavefronts ---- */
        for (int i = tid; i < 5 * 3 * 16; i += NT) (&S.acc[0][0][0])[i] = 0;
        __syncthreads();
        {
            const int C = size >> 1;
            const int tY = size >> 3, nY = tY * tY, tC = C >> 3, nC = tC * tC;      /* 8x8 tiles per row / in all; chroma: none when the blocks are 4x4 */
            const int perCand = nY + (nC ? 2 * nC : 1);
            /* a wavefront's tiles four at a time: the samples of all four are fetched before any is used (a tile alone is two dependent memory round trips) */
            const int total = nc * perCand;
            for (int t0 = wv; t0 < total; t0 += 4 * XA_SERVER_WAVES)
            {
                int v[4], sv[4], grp[4], pl4[4], ci4[4];
                pixel* dp[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    const int t = t0 + q * XA_SERVER_WAVES;
                    grp[q] = -1; v[q] = 0; sv[q] = 0; dp[q] = nullptr; pl4[q] = 0; ci4[q] = 0;
                    if (t >= total) continue;
                    const int ci = t / perCand, r = t - ci * perCand;
                    if (!S.cand[ci].valid) continue;
                    const x265amd_mc_job& j = S.mc[ci];
                    const McSetup su = mc_setup(j, J.info.pic_width, J.info.pic_height);
                    ci4[q] = ci;
                    if (r < nY)
                    {
                        const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, 0);
                        const int ty = r / tY, tx = r - ty * tY, xx = 8 * tx + (lane & 7), yy = 8 * ty + (lane >> 3);
                        v[q] = mc_one<8>(P, j, su.mode, su.lsel, xx, yy);
                        sv[q] = (reinterpret_cast<const pixel*>(srcPlanes[0]) + (size_t)y * J.stride + x)[(size_t)yy * J.stride + xx];
                        dp[q] = P.dst + (long)yy * P.dstStride + xx;
                        pl4[q] = 0; grp[q] = size == 8 ? 0 : (ty >> 1) * (size >> 4) + (tx >> 1);
                    }
                    else if (nC)
                    {
                        const int u = r - nY, pl = 1 + u / nC, k = u % nC, ty = k / tC, tx = k - ty * tC, xx = 8 * tx + (lane & 7), yy = 8 * ty + (lane >> 3);
                        const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, pl);
                        v[q] = mc_one<4>(P, j, su.mode, su.lsel, xx, yy);
                        sv[q] = (reinterpret_cast<const pixel*>(srcPlanes[pl]) + (size_t)(y >> 1) * J.cstride + (x >> 1))[(size_t)yy * J.cstride + xx];
                        dp[q] = P.dst + (long)yy * P.dstStride + xx;
                        pl4[q] = pl; grp[q] = C == 8 ? 0 : (ty >> 1) * (C >> 4) + (tx >> 1);
                    }
                    else
                    {
                        /* the 4x4 chroma blocks of an 8x8 CU: U on lanes 0-15, V on lanes 16-31 (cu[4x4].sa8d = satd_4x4, pixel.cpp:1171) */
                        const int pl = 1 + ((lane >> 4) & 1), l = lane & 15, xx = l & 3, yy = l >> 2;
                        if (lane < 32)
                        {
                            const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, pl);
                            v[q] = mc_one<4>(P, j, su.mode, su.lsel, xx, yy);
                            sv[q] = (reinterpret_cast<const pixel*>(srcPlanes[pl]) + (size_t)(y >> 1) * J.cstride + (x >> 1))[yy * J.cstride + xx];
                            dp[q] = P.dst + (long)yy * P.dstStride + xx;
                        }
                        grp[q] = 64;        /* marks the 4x4 pair */
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    if (grp[q] < 0) continue;
                    if (dp[q]) *dp[q] = (pixel)v[q];
                    if (grp[q] == 64)
                    {
                        const int sum = xa_row16_sum(abs(xa_lane_had4x4(sv[q] - v[q], lane)));
                        if (lane < 32 && (lane & 15) == 0) S.acc[ci4[q]][1 + ((lane >> 4) & 1)][0] = (unsigned int)sum;
                    }
                    else
                    {
                        const int raw = xa_wave_sum(abs(xa_lane_had8x8(sv[q] - v[q], lane)));
                        if (lane == 0) atomicAdd(&S.acc[ci4[q]][pl4[q]][grp[q]], (unsigned int)raw);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        XA_CHAIN_T(1);
        /* the choice: calcRdSADCost(sa8d, getTUBits(i, numCand)), the first cheapest */
        if (tid == 0)
        {
            uint64_t bestCost = ~0ull;
            int best = -1;
            const uint64_t lambda = J.lambda;
            const int C = size >> 1, gY = size == 8 ? 1 : (size >> 4) * (size >> 4), gC = C <= 8 ? 1 : (C >> 4) * (C >> 4);
            for (int i = 0; i < nc; i++)
            {
                if (!S.cand[i].valid) continue;
                unsigned int sy = 0, sc = 0;
                for (int k = 0; k < gY; k++) sy += (S.acc[i][0][k] + 2) >> 2;                  /* sa8d_8x8 / sa8d_16x16 groups (pixel.cpp:342-384) */
                for (int pl = 1; pl < 3; pl++)
                {
                    if (C == 4) sc += S.acc[i][pl][0] >> 1;
                    else for (int k = 0; k < gC; k++) sc += (S.acc[i][pl][k] + 2) >> 2;
                }
                S.meas[i].sa8d_luma = sy; S.meas[i].sa8d = sy + sc;
                const uint32_t bits = (uint32_t)(i + (i < nc - 1));            /* getTUBits */
                const uint32_t d = J.chroma_sa8d ? sy + sc : sy;
                const uint64_t c = d + ((bits * lambda + 128) >> 8);
                if (c < bestCost) { bestCost = c; best = i; }
            }
            S.best = best;
        }
        __syncthreads();
        XA_CHAIN_T(2);
        const int best = S.best;
        if (best < 0) { reason = XA_CHAIN_NOTSKIP; break; }     /* no candidate left: the host's path decides what that means */
        /* ---- the winner's residual: one transform size per plane (tu-inter-depth 1, 2Nx2N: CUData::getInterTUQtDepthRange) ---- */
        if (log2 > J.tu_log2_max)
        {
            /* the 64x64 CU (four 32x32 luma units, four 16x16 per chroma plane): skipped here when nothing quantises to a level; otherwise the host's path */
            const int L = J.tu_log2_max, C = L - 1;
            const int ntL = size >> L, ntC = (size >> 1) >> C, nLuma = ntL * ntL, nChroma = ntC * ntC;
            const pixel* tile = reinterpret_cast<const pixel*>(S.mc[best].dst_y);
            const int total = nLuma + 2 * nChroma;
            auto geo = [&](int k, const pixel*& fenc, int& fs, const pixel*& pred, int& ps, int& lg, int& tt, int& qp) {
                if (k < nLuma)
                {
                    const int ty = k / ntL, tx = k - ty * ntL, n = 1 << L;
                    fenc = reinterpret_cast<const pixel*>(srcPlanes[0]) + (size_t)(y + ty * n) * J.stride + x + tx * n; fs = J.stride;
                    pred = tile + (size_t)ty * n * 64 + tx * n; ps = 64; lg = L; tt = 0; qp = J.qp_luma;
                }
                else
                {
                    const int p = 1 + (k - nLuma) / nChroma, kk = (k - nLuma) % nChroma, ty = kk / ntC, tx = kk - ty * ntC, n = 1 << C;
                    fenc = reinterpret_cast<const pixel*>(srcPlanes[p]) + (size_t)((y >> 1) + ty * n) * J.cstride + (x >> 1) + tx * n; fs = J.cstride;
                    pred = tile + 4096 + (size_t)(p - 1) * 1024 + (size_t)ty * n * 32 + tx * n; ps = 32; lg = C; tt = p; qp = J.qp_chroma;
                }
            };
            /* A level somewhere: most of these CUs still end as skips -- every unit's flag falls to its rate-distortion check, or signalling nothing is cheaper than the
             * tree -- so a CU one size above the largest transform (four luma units) is decided here all the same: the pass below notes WHICH units have levels
             * (S.anyLevel: bit k for unit k), those units' full chains run behind the CU's measurement, and one wavefront walks the tree (chain_merge_rd64).  A residual
             * mode that WINS is the host's, as is every other tree: it gets the candidate to go on with (valid 2). */
            const bool tree = nLuma == 4 && L == 5 && XA_SERVER_WAVES >= 8 && !J.chain64_off;
            /* the large units by the whole workgroup, one after the other; the others a wavefront each */
            int small = 0;
            for (int k = 0; k < total; k++)
            {
                const pixel* fenc; const pixel* pred; int fs, ps, lg, tt, qp;
                geo(k, fenc, fs, pred, ps, lg, tt, qp);
                if (lg >= 5)
                {
                    __syncthreads();
                    const uint32_t ns = chain_tu_levels(TL[XA_SERVER_WAVES], fenc, fs, pred, ps, lg, tt, J.slice_type, qp, XaBlock{ tid, NT, S.red });
                    if (ns && tid == 0) S.anyLevel |= 1 << k;
                    __syncthreads();
                    if (S.anyLevel && !tree) break;
                }
                else
                {
                    if ((small % XA_SERVER_WAVES) == wv)
                    {
                        const uint32_t ns = chain_tu_levels(TL[wv], fenc, fs, pred, ps, lg, tt, J.slice_type, qp, XaWave{ lane });
                        if (ns && lane == 0) atomicOr(&S.anyLevel, 1 << k);
                    }
                    small++;
                }
            }
            __syncthreads();
            const uint32_t sig = (uint32_t)S.anyLevel;
            auto hand_over = [&]() {
                /* the host's to walk -- but the merge check's first half is done: the candidate it goes on with and that candidate's SA8D figures (valid 2: nothing else
                 * of the record is filled; the candidates' predictions lie in their tiles) */
                if (tid == 64)
                {
                    const ChainCand c = S.cand[best];
                    XaChainStopHead h{};
                    h.valid = 2; h.node = (uint32_t)node; h.cand = (uint8_t)best; h.dir = c.dir;
                    for (int l = 0; l < 2; l++) { const bool used = (c.dir >> l) & 1; h.ref_idx[l] = used ? c.ref_idx[l] : -1; h.mv[l][0] = used ? c.mv[l][0] : 0; h.mv[l][1] = used ? c.mv[l][1] : 0; }
                    h.sa8d = S.meas[best].sa8d; h.sa8d_luma = S.meas[best].sa8d_luma;
                    xa_st_result(reinterpret_cast<XaChainStopHead*>(&out->stop), h);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            };
            if (sig && !tree) { hand_over(); reason = XA_CHAIN_NOTSKIP; break; }
            /* the skip mode's distortion and psy energy: the prediction against the source */
            if (tid == 0)
            {
                CuMeasureJob& m = S.mj;
                m.fenc[0] = srcPlanes[0] + ((uint64_t)y * J.stride + x) * isz;
                m.fenc[1] = srcPlanes[1] + ((uint64_t)(y >> 1) * J.cstride + (x >> 1)) * isz;
                m.fenc[2] = srcPlanes[2] + ((uint64_t)(y >> 1) * J.cstride + (x >> 1)) * isz;
                m.pred = S.mc[best].dst_y;
                m.recon = J.tiles + (uint64_t)(depth * J.tiles_per_depth + J.cand_tile0 + 5) * J.tile_bytes;       /* the depth's scratch tile */
                m.resi = 0; m.sel = 0; m.fenc_stride = J.stride; m.fenc_cstride = J.cstride; m.log2_size = log2; m.assemble = 0;
            }
            __syncthreads();
            block_cu_measure_job(S.mj, &S.meas[best], ML, tid, NT);
            __syncthreads();
            if (sig)
            {
                /* the units with levels: their full chains (the luma units by the whole workgroup one after the other, the chroma units a wavefront each), then the tree */
                if (tid < 12)
                {
                    const int k = tid, p = k >> 2;
                    x265amd_tu_job& j = S.tu[k];
                    j = x265amd_tu_job{};
                    S.tr[k] = x265amd_tu_result{};
                    if ((sig >> k) & 1)
                    {
                        const pixel* fenc; const pixel* pred; int fs, ps, lg, tt, qp;
                        geo(k, fenc, fs, pred, ps, lg, tt, qp);
                        const int n = 1 << lg;
                        const uint64_t off = p ? 4096 + (uint64_t)(k - 4) * 256 : (uint64_t)k * 1024;         /* elements: four luma units, then U's four, then V's */
                        j.fenc = (uint64_t)(uintptr_t)fenc; j.pred = (uint64_t)(uintptr_t)pred;
                        j.coeff = J.scratch + off * 2;
                        j.resi = J.scratch + 6144 * 2 + off * 2;
                        j.recon = J.scratch + 6144 * 4 + off * isz;
                        j.fenc_stride = fs; j.pred_stride = ps; j.resi_stride = n; j.recon_stride = n;
                        j.log2_tr_size = (uint8_t)lg; j.ttype = (uint8_t)tt; j.intra = 0; j.dir_mode = 0; j.slice_type = (uint8_t)J.slice_type;
                        j.qp_scaled = (uint8_t)qp; j.sign_hide = (uint8_t)J.sign_hide;
                    }
                }
                __syncthreads();
                for (int k = 0; k < 4; k++)
                    if ((sig >> k) & 1)
                    {
                        grp_tu_measure<false>(TL[XA_SERVER_WAVES], nullptr, S.tu[k], nullptr, reinterpret_cast<const pixel*>(S.tu[k].pred), 64, &S.tr[k], XaBlock{ tid, NT, S.red });
                        __syncthreads();
                    }
                {
                    int idx = 0;
                    for (int k = 4; k < 12; k++)
                        if ((sig >> k) & 1)
                        {
                            if (idx == wv) wave_tu_measure<false>(TL[wv], nullptr, S.tu[k], nullptr, reinterpret_cast<const pixel*>(S.tu[k].pred), 32, &S.tr[k], lane);
                            idx++;
                        }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wv == 0) chain_merge_rd64(S, x, y, log2, best, lane, sig);
                __syncthreads();
                if (!S.skipWins) { hand_over(); reason = XA_CHAIN_NOTSKIP; break; }
            }
            if (wv == 0)
            {
                /* the skip mode's coder state (the residual mode equals it) */
                for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) S.ctxS[i] = S.ctx[i];
                xa_wave_sync();
                if (lane == 0)
                {
                    uint64_t fS = S.frac & 32767;
                    fS += cb_bin(S.ctxS + CC_SKIP + chain_skip_ctx(J, x, y), 1);
                    fS += chain_merge_index(S.ctxS, (uint32_t)best, (uint32_t)J.info.max_num_merge_cand);
                    S.fracS = fS;
                }
            }
            __syncthreads();
            XA_CHAIN_T(3);          /* (the 64x64 CU's "does anything quantise to a level" pass and its measurement: transform units' time, not the placing behind it) */
        }
        else
        {
            /* one unit per plane: the full chain of each (Quant::transformNxN, invtransformNxN, the unit's measurements: tu_dev.h), then both modes' RD by one wavefront */
            const int C = log2 - 1 < 2 ? 2 : log2 - 1;
            if (tid < 3)
            {
                const int p = tid, n = p ? 1 << C : size;
                x265amd_tu_job& j = S.tu[p];
                j = x265amd_tu_job{};
                const pixel* tile = reinterpret_cast<const pixel*>(S.mc[best].dst_y);
                j.fenc = p ? srcPlanes[p] + ((uint64_t)(y >> 1) * J.cstride + (x >> 1)) * isz : srcPlanes[0] + ((uint64_t)y * J.stride + x) * isz;
                j.pred = (uint64_t)(uintptr_t)(p ? tile + 4096 + (size_t)(p - 1) * 1024 : tile);
                const uint64_t off = p ? 1024 + (uint64_t)(p - 1) * 256 : 0;           /* elements: Y up to 32x32, U / V up to 16x16 */
                j.coeff = J.scratch + off * 2;
                j.resi = J.scratch + 1536 * 2 + off * 2;
                j.recon = J.scratch + 1536 * 4 + off * isz;
                j.fenc_stride = p ? J.cstride : J.stride; j.pred_stride = p ? 32 : 64; j.resi_stride = n; j.recon_stride = n;
                j.log2_tr_size = (uint8_t)(p ? C : log2); j.ttype = (uint8_t)p; j.intra = 0; j.dir_mode = 0; j.slice_type = (uint8_t)J.slice_type;
                j.qp_scaled = (uint8_t)(p ? J.qp_chroma : J.qp_luma); j.sign_hide = (uint8_t)J.sign_hide;
            }
            __syncthreads();
            if (log2 == 5)
            {
                grp_tu_measure<false>(TL[XA_SERVER_WAVES], nullptr, S.tu[0], nullptr, reinterpret_cast<const pixel*>(S.tu[0].pred), 64, &S.tr[0], XaBlock{ tid, NT, S.red });
                __syncthreads();
                if (wv < 2) wave_tu_measure<false>(TL[wv], nullptr, S.tu[1 + wv], nullptr, reinterpret_cast<const pixel*>(S.tu[1 + wv].pred), 32, &S.tr[1 + wv], lane);
            }
            else if (wv < 3) wave_tu_measure<false>(TL[wv], nullptr, S.tu[wv], nullptr, reinterpret_cast<const pixel*>(S.tu[wv].pred), wv ? 32 : 64, &S.tr[wv], lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            XA_CHAIN_T(3);
            if (tid == 0)
            {
                /* the prediction against the source, CU-wide: the units' own "nothing coded" figures (one unit per plane; psyCost_pp sums over 8x8 blocks) */
                x265amd_cu_measure& m = S.meas[best];
                m.sse[0] = S.tr[0].zero_dist; m.sse[1] = S.tr[1].zero_dist; m.sse[2] = S.tr[2].zero_dist; m.psy = S.tr[0].zero_energy;
                m.src_mean = 0; m.src_homo = 0; m.reserved = 0;
            }
            __syncthreads();
            if (wv == 0) chain_merge_rd(S, x, y, log2, best, lane);
            __syncthreads();
            XA_CHAIN_T(4);
            if (!S.skipWins)
            {
                /* the residual mode wins: the CU goes on on the host (sub-CUs, motion search, intra), which takes the merge check from here -- the residual mode's
                 * reconstruction (the kept units' reconstruction, the others' prediction) and the skip mode's (the prediction) into their tiles, the figures into the record */
                const pixel* tile = reinterpret_cast<const pixel*>(S.mc[best].dst_y);
                pixel* tm = reinterpret_cast<pixel*>(J.tiles + (uint64_t)(depth * J.tiles_per_depth + J.merge_recon_tile) * J.tile_bytes);
                pixel* ts = reinterpret_cast<pixel*>(J.tiles + (uint64_t)(depth * J.tiles_per_depth + J.skip_recon_tile) * J.tile_bytes);
                for (int p = 0; p < 3; p++)
                {
                    const int lg = p ? C : log2, off = p ? 4096 + (p - 1) * 1024 : 0, st = p ? 32 : 64;
                    chain_copy_plane(ts + off, st, tile + off, st, lg, p != 0, tid, NT);
                    if (S.rdCbf[p]) chain_copy_plane(tm + off, st, reinterpret_cast<const pixel*>(S.tu[p].recon), 1 << lg, lg, p != 0, tid, NT);
                    else chain_copy_plane(tm + off, st, tile + off, st, lg, p != 0, tid, NT);
                }
                XaChainStop* sp = &out->stop;
                for (int p = 0; p < 3; p++)
                {
                    const int lg = p ? C : log2, n2 = 1 << (2 * lg);
                    const int16_t* lv = reinterpret_cast<const int16_t*>(S.tu[p].coeff);
                    int16_t* dst = sp->levels + (p ? 1024 + (p - 1) * 256 : 0);
                    for (int i = tid; i < n2 / 4; i += NT)
                    {
                        const uint64_t w = *reinterpret_cast<const uint64_t*>(lv + 4 * i);     /* as saveResidualQTData leaves them: a unit's levels whether or not its flag survived */
                        __hip_atomic_store(reinterpret_cast<uint64_t*>(dst + 4 * i), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                if (tid < X265AMD_CTX_STRIDE / 8) __hip_atomic_store(reinterpret_cast<uint64_t*>(sp->ctx) + tid, reinterpret_cast<const uint64_t*>(S.ctxD)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (tid == 64)
                {
                    const ChainCand c = S.cand[best];
                    XaChainStopHead h;
                    h.valid = 1; h.node = (uint32_t)node; h.cand = (uint8_t)best; h.dir = c.dir;
                    for (int l = 0; l < 2; l++) { const bool used = (c.dir >> l) & 1; h.ref_idx[l] = used ? c.ref_idx[l] : -1; h.mv[l][0] = used ? c.mv[l][0] : 0; h.mv[l][1] = used ? c.mv[l][1] : 0; }
                    h.cbf[0] = (uint8_t)S.rdCbf[0]; h.cbf[1] = (uint8_t)S.rdCbf[1]; h.cbf[2] = (uint8_t)S.rdCbf[2]; h.reserved0 = 0;
                    h.total_bits = S.rdBits[0]; h.mv_bits = S.rdBits[1]; h.coeff_bits = S.rdBits[0] - S.rdPad - S.rdBits[1] - S.rdBits[2]; h.psy_energy = S.rdPsy;
                    h.sa8d = S.meas[best].sa8d; h.sa8d_luma = S.meas[best].sa8d_luma;
                    h.rd_cost = S.rdCost; h.luma_dist = S.rdLuma; h.chroma_dist = S.rdChroma; h.frac = S.fracD; h.meas = S.meas[best];
                    xa_st_result(reinterpret_cast<XaChainStopHead*>(sp), h);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                reason = XA_CHAIN_NOTSKIP; break;
            }
        }
        /* ---- a skipped CU: its motion into the map, its prediction into the picture and into the enclosing CUs' reconstruction tiles, the result to the host ---- */
        {
            const ChainCand c = S.cand[best];
            XaMapUnit u{};
            u.pred_mode = X265AMD_MODE_SKIP; u.inter_dir = c.dir; u.depth = (uint8_t)depth;
            for (int l = 0; l < 2; l++)
            {
                const bool used = (c.dir >> l) & 1;
                u.ref_idx[l] = used ? c.ref_idx[l] : -1; u.mv[l][0] = used ? c.mv[l][0] : 0; u.mv[l][1] = used ? c.mv[l][1] : 0;
            }
            XaMapUnit* cur = reinterpret_cast<XaMapUnit*>(J.cur);
            const int n4 = size >> 2;
            if (!(J.dbg & 1)) for (int i = tid; i < n4 * n4; i += NT) chain_st_unit(cur + ((y >> 2) + i / n4) * J.w4 + (x >> 2) + (i % n4), u);
            const pixel* s0 = reinterpret_cast<const pixel*>(S.mc[best].dst_y);
            const int half = size >> 1;
            if (!(J.dbg & 2)) {
            chain_copy_plane(reinterpret_cast<pixel*>(recPlanes[0]) + (size_t)y * J.stride + x, J.stride, s0, 64, log2, false, tid, NT);
            chain_copy_plane(reinterpret_cast<pixel*>(recPlanes[1]) + (size_t)(y >> 1) * J.cstride + (x >> 1), J.cstride, s0 + 4096, 32, log2 - 1, true, tid, NT);
            chain_copy_plane(reinterpret_cast<pixel*>(recPlanes[2]) + (size_t)(y >> 1) * J.cstride + (x >> 1), J.cstride, s0 + 5120, 32, log2 - 1, true, tid, NT);
            }
            (void)half;
            for (int da = depth - 1; da >= 0 && !(J.dbg & 4); da--)
            {
                const int asize = 64 >> da, ax = J.ctu_x + ((x - J.ctu_x) & ~(asize - 1)), ay = J.ctu_y + ((y - J.ctu_y) & ~(asize - 1));
                pixel* t = reinterpret_cast<pixel*>(J.tiles + (uint64_t)(da * J.tiles_per_depth + J.split_recon_tile) * J.tile_bytes);
                chain_copy_plane(t + (size_t)(y - ay) * 64 + (x - ax), 64, s0, 64, log2, false, tid, NT);
                chain_copy_plane(t + 4096 + (size_t)((y - ay) >> 1) * 32 + ((x - ax) >> 1), 32, s0 + 4096, 32, log2 - 1, true, tid, NT);
                chain_copy_plane(t + 5120 + (size_t)((y - ay) >> 1) * 32 + ((x - ax) >> 1), 32, s0 + 5120, 32, log2 - 1, true, tid, NT);
            }
            if (tid == 0)
            {
                /* a skipped CU put in place: its units into the map, its samples into the picture and into the enclosing CUs' tiles, its record to the host */
                XA_BYTES((unsigned long long)(size >> 2) * (size >> 2) * sizeof(XaMapUnit) + 3ull * size * size / 2 * sizeof(pixel) * (1 + depth) + sizeof(XaChainCuOut));
                XaChainCuOut o{};
                o.node = (uint32_t)node; o.cand = (uint8_t)best; o.dir = c.dir; o.ref_idx[0] = u.ref_idx[0]; o.ref_idx[1] = u.ref_idx[1];
                o.mv[0][0] = u.mv[0][0]; o.mv[0][1] = u.mv[0][1]; o.mv[1][0] = u.mv[1][0]; o.mv[1][1] = u.mv[1][1];
                o.meas = S.meas[best];
                if (!(J.dbg & 8)) xa_st_result(&out->cu[S.count], o);
                S.count++;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        XA_CHAIN_T(5);
        /* ---- the coder's state behind the CU: the skip mode's, the split flag (0) of a CU that could have split, and the split flags (1) of the enclosing CUs this
         * one completes (those that lie inside the picture but were not coded at their own depth) ---- */
        if (tid < X265AMD_CTX_STRIDE) S.ctx[tid] = S.ctxS[tid];
        __syncthreads();
        if (tid == 0)
        {
            uint64_t f = S.fracS;
            if (depth < J.max_cu_depth) chain_split_flag(J, S.ctx, f, x, y, depth, 0);
            for (int p = N.parent; p != 255 && S.nodes[p].next == N.next; p = S.nodes[p].parent)
                if (S.used[p] & 2) chain_split_flag(J, S.ctx, f, S.nodes[p].x, S.nodes[p].y, 6 - S.nodes[p].log2, 1);
            S.frac = f;
            S.ticks[6] += 1;
            if (J.use_dqp)
            {
                /* what the skipped CU leaves in the picture's QP records (Search::checkDQP: a CU without a residual at or above the groups' depth takes the predicted QP; a
                 * group whose CUs are all skipped takes it through checkDQPForSplitPred when its last CU is done) */
                const bool origin = x == J.ctu_x && y == J.ctu_y;
                if (depth <= J.max_dqp_depth)
                {
                    const int r = chain_ref_qp(S, depth == 0 ? 0 : qg);
                    if (depth == 0) { for (int k = 0; k < 4; k++) S.qgVal[k] = r; S.qgDone = 15; }
                    else { S.qgVal[qg] = r; S.qgDone |= 1u << qg; }
                    if (origin) S.firstQp = r;
                }
                else
                {
                    if (origin) S.firstQp = J.qps[J.max_dqp_depth == 0 ? 0 : 1 + qg].qp;
                    int a = N.parent;
                    while (a != 255 && 6 - S.nodes[a].log2 > J.max_dqp_depth) a = S.nodes[a].parent;
                    if (a != 255 && S.nodes[a].next == N.next)
                    {
                        const int r = chain_ref_qp(S, qg);
                        S.qgVal[qg] = r; S.qgDone |= 1u << qg;
                        if (S.nodes[a].x == J.ctu_x && S.nodes[a].y == J.ctu_y) S.firstQp = r;
                    }
                }
            }
        }
        __syncthreads();
        XA_CHAIN_T(7);
        node = N.next;
    }
    __syncthreads();
    if (tid == 0)
    {
        struct alignas(8) Head { uint32_t count, stop_node, reason, reserved; uint64_t frac; } h{ (uint32_t)S.count, (uint32_t)node, (uint32_t)reason, 0, S.frac };
        xa_st_result(reinterpret_cast<Head*>(out), h);
    }
    if (tid >= 64 && tid < 72) __hip_atomic_store(&out->ticks[tid - 64], (uint64_t)S.ticks[tid - 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < X265AMD_CTX_STRIDE / 8) __hip_atomic_store(reinterpret_cast<uint64_t*>(out->ctx) + tid, reinterpret_cast<const uint64_t*>(S.ctx)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#endif
#endif
