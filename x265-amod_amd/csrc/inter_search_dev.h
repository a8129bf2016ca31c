/* The 2Nx2N motion search of a CU of a P picture and the rate-distortion of its result as ONE device command (round 4).
 *
 * What it restates: Search::predInterSearch for one 2Nx2N prediction unit of a P slice (reference: source/encoder/search.cpp:2181-2647 -- per reference picture selectMVP
 * (:1992-2024), setSearchRange (:2724-2768), MotionEstimate::motionEstimate (motion.cpp:764-1594), the bits of the result and checkBestMVP (:2702-2713), the cheapest
 * reference), Predict::motionCompensation of the winner, its SA8D against the source (checkInter_rd0_4, analysis.cpp:3023-3085) and Search::encodeResAndCalcRdInterCU
 * (:2822-2975) for a CU with one transform unit per plane.
 *
 * Why: on the host these were four round trips per CU -- the predictors' SADs, the searches, the prediction with its measurement, the transform units with theirs -- with a
 * numeric decision between each two that needs nothing but what the device has just computed (which predictor is cheaper, which reference, which units keep their levels).
 * The host still derives what depends on the pictures' maps -- the AMVP candidates and the search's extra candidates per reference (x265amd_amvp_candidates), the reference
 * masks, the coder's state -- builds ONE record from them, and waits once; the decisions between the steps are taken here, with the host's own arithmetic (inter_search.hip).
 */
#ifndef X265AMD_INTER_SEARCH_DEV_H
#define X265AMD_INTER_SEARCH_DEV_H
#include <stdint.h>

struct XaSearchRef { int16_t amvp[2][2]; int16_t mvc[12][2]; int32_t num_mvc, ref_pic, ref_idx, reserved; };      /* one reference picture of list 0 that the CU may search */
enum { XA_SEARCH_MAX_REFS = 8 };

struct alignas(8) XaSearchJob
{
    XaSearchRef ref[XA_SEARCH_MAX_REFS];
    int32_t num_refs, x, y, log2;
    int32_t pic_w, pic_h, stride, cstride, num_pics, num_ref_idx0;
    int32_t search_method, subme, merange, me_qp, frame_parallel, search_range, lag_pixels, list_sel_bits0;
    uint64_t me_lambda;                     /* floor(256 * x265_lambda_tab[qp]): Search's getCost */
    uint64_t mvcost, bitsize, me_tables;    /* device: centre of BitCost::s_costs[qp] (uint16), BitCost::s_bitsizes (float, index |d|), all cost tables */
    uint64_t planes, luma_tab;              /* device arrays: num_pics x 3 plane addresses (the last picture is the source); num_pics luma plane addresses */
    uint64_t pred_tile, recon_tile, scratch, out;
    uint64_t lambda, lambda2;               /* RDCost::m_lambda / m_lambda2 */
    uint32_t psy_rd;
    int32_t qp_luma, qp_chroma, sign_hide, chroma_sa8d, rd_level, skip_ctx, do_rd, slice_type;
    int32_t dqp;                            /* delta QP (PPS cu_qp_delta_enabled): bits 0-7 the CU's cu_qp_delta as Entropy::codeDeltaQP wraps it (int8), bit 16 on, bit 17 the CU is a
                                             * quantisation group or above it (Search::checkDQP prices the syntax element a second time); 0: no delta QP */
    uint64_t frac;                          /* the coder's state in front of the CU (m_rqt[depth].cur) */
    uint8_t ctx[X265AMD_CTX_STRIDE];
};

struct XaSearchOut                          /* pinned host memory */
{
    uint32_t valid; int32_t best;           /* index into ref[] of the winner, -1: none */
    int16_t mv[2], mvp[2]; int32_t mvp_idx; uint32_t bits, cost, mv_cost;        /* the winner as MotionData holds it (bestME[0][0]) */
    uint32_t sa8d, sa8d_luma;               /* of its prediction (with chroma / luma alone) */
    uint8_t cbf[3], rd_done;
    uint32_t total_bits, mv_bits, coeff_bits, psy_energy, res_energy, reserved1;
    uint64_t rd_cost, luma_dist, chroma_dist, frac;
    uint8_t ctx[X265AMD_CTX_STRIDE];
    int16_t levels[1024 + 2 * 256];
};

#ifdef XA_SEARCH_DEVICE

struct SearchLds
{
    XaSearchJob job;
    x265amd_mc_job mc[2 * XA_SEARCH_MAX_REFS];
    uint32_t cost[4 * XA_SEARCH_MAX_REFS];
    XaCmd cmd;                              /* the search as the job server's own command bodies take it (xa_op_me) */
    int mvpIdx[XA_SEARCH_MAX_REFS];
    int best, bestMvpIdx; int bestMv[2], bestMvp[2]; uint32_t bestBits, bestCost, bestMvCost;
    unsigned int acc[3][16];
    uint32_t sa8d, sa8dLuma;
    x265amd_tu_job tu[3];
    x265amd_tu_result tr[3];
    uint8_t ctxB[X265AMD_CTX_STRIDE], ctxD[X265AMD_CTX_STRIDE];
    uint32_t rdBits[3], rdPsy, rdCbf[3], rdPad; uint64_t rdCost, rdLuma, rdChroma, fracD;
    uint32_t step[256];
    unsigned long long red[XA_SERVER_WAVES];
};
#define XA_SEARCH_HEADER 8192
#define XA_SEARCH_LDS_BELOW (XA_SERVER_LDS - XA_SEARCH_HEADER)       /* the header sits at the END of the workgroup's LDS: the search's windows and the transform units use what is below */

/* CUData::clipMv (cudata.cpp:1915-1928) */
XA_DEV void search_clip_mv(int& mx, int& my, int cuX, int cuY, int picW, int picH)
{
    const int xmax = (picW + 8 - cuX - 1) << 2, xmin = -((64 + 8 + cuX - 1) << 2), ymax = (picH + 8 - cuY - 1) << 2, ymin = -((64 + 8 + cuY - 1) << 2);
    mx = min(xmax, max(xmin, mx)); my = min(ymax, max(ymin, my));
}
/* BitCost::bitcost (bitcost.h:60-64) */
XA_DEV uint32_t search_bitcost(const float* bs, int mx, int my, int px, int py) { return (uint32_t)(bs[abs(mx - px)] + bs[abs(my - py)] + 0.5f); }

/* Entropy::codeRefFrmIdx + codeMvd + codeMVPIdx in counting mode (entropy.cpp:1677-1735; host form: cabac_coder.h predInfo) */
XA_DEV uint64_t search_amvp_bits(uint8_t* ctx, int refIdx, int numRefIdx, int mvdx, int mvdy, int mvpIdx)
{
    enum { C_REF_NO = 24, C_MV_RES = 26, C_MVP_IDX = 151 };
    uint64_t f = 0;
    if (numRefIdx > 1)
    {
        uint32_t ref = (uint32_t)refIdx;
        f += cb_bin(ctx + C_REF_NO, ref > 0);
        if (ref > 0)
        {
            const uint32_t refNum = (uint32_t)numRefIdx - 2;
            if (refNum)
            {
                ref--;
                f += cb_bin(ctx + C_REF_NO + 1, ref > 0);
                if (ref > 0) f += 32768ull * (ref - (ref == refNum));
            }
        }
    }
    f += cb_bin(ctx + C_MV_RES, mvdx != 0); f += cb_bin(ctx + C_MV_RES, mvdy != 0);
    const uint32_t ha = (uint32_t)abs(mvdx), va = (uint32_t)abs(mvdy);
    if (mvdx) f += cb_bin(ctx + C_MV_RES + 1, ha > 1);
    if (mvdy) f += cb_bin(ctx + C_MV_RES + 1, va > 1);
    /* writeEpExGolomb(symbol, 1) (entropy.cpp:2293-2310): the bins it codes, all bypass */
    auto egBins = [](uint32_t symbol) { uint32_t count = 1, n = 0; while (symbol >= (1u << count)) { n++; symbol -= 1u << count; count++; } return n + 1 + count; };
    if (mvdx) { if (ha > 1) f += 32768ull * egBins(ha - 2); f += 32768ull; }
    if (mvdy) { if (va > 1) f += 32768ull * egBins(va - 2); f += 32768ull; }
    f += cb_bin(ctx + C_MVP_IDX, (uint32_t)mvpIdx);
    return f;
}

template<class JOB> XA_DEV uint64_t search_cost(const JOB& J, chain_sse_t dist, uint32_t bits, uint32_t energy)
{
    return J.psy_rd ? (uint64_t)dist + ((J.lambda * J.psy_rd * energy) >> 24) + (((uint64_t)bits * J.lambda2) >> 8) : (uint64_t)dist + (((uint64_t)bits * J.lambda2 + 128) >> 8);
}

/* Search::encodeResAndCalcRdInterCU of the searched 2Nx2N mode (one transform unit per plane; host form: inter_rd_walk_impl + x265amd_inter_rd_finish, inter_rd.hip).
 * One wavefront; leaves the result in S.rd*, S.ctxD, S.fracD. */
XA_DEV void search_inter_rd(SearchLds& S, int log2, int lane)
{
    const XaSearchJob& J = S.job;
    for (int i = lane; i < X265AMD_CTX_STRIDE; i += 64) { const uint8_t v = J.ctx[i]; S.ctxB[i] = v; S.ctxD[i] = v; }
    xa_wave_sync();
    const chain_sse_t predDist = (chain_sse_t)((chain_sse_t)S.tr[0].zero_dist + (chain_sse_t)S.tr[1].zero_dist + (chain_sse_t)S.tr[2].zero_dist);
    const uint32_t predPsy = J.psy_rd ? S.tr[0].zero_energy : 0;
    const int logs[3] = { log2, log2 - 1 < 2 ? 2 : log2 - 1, log2 - 1 < 2 ? 2 : log2 - 1 };
    uint32_t cbf[3], singleBits[3];
    chain_sse_t singleDist[3];
    uint32_t energyY = 0;
    uint64_t fB = J.frac & 32767;
    bool anyLevel = false;
    for (int p = 0; p < 3; p++) anyLevel |= S.tr[p].num_sig != 0;
    if (anyLevel)
    {
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
                const uint8_t st = S.ctxB[CC_QT_CBF + (p ? 2 : 1)];
                const uint32_t nzCbfBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 1]) >> 15), nullBits = (uint32_t)(((fB & 32767) + en_bits[st ^ 0]) >> 15);
                const chain_sse_t nzDist = (chain_sse_t)r.nz_dist;
                const uint32_t nzEnergy = J.psy_rd ? r.nz_energy : 0;
                const uint64_t singleCost = search_cost(J, nzDist, nzCbfBits + singleBits[p], nzEnergy), nullCost = search_cost(J, zeroDist, nullBits, zeroEnergy);
                if (nullCost < singleCost) { cbf[p] = 0; singleBits[p] = 0; singleDist[p] = zeroDist; if (!p) energyY = zeroEnergy; }
                else { singleDist[p] = nzDist; if (!p) energyY = nzEnergy; }
            }
            else { singleBits[p] = 0; singleDist[p] = zeroDist; if (!p) energyY = zeroEnergy; }
        }
        chain_sse_t fullDist = 0;
        fullDist += singleDist[0]; fullDist += singleDist[1]; fullDist += singleDist[2];
        const uint64_t fullCost = search_cost(J, fullDist, singleBits[0] + singleBits[1] + singleBits[2], energyY);
        const uint32_t cbf0Bits = (uint32_t)(((J.frac & 32767) + en_bits[J.ctx[CC_QT_ROOT_CBF] ^ 0]) >> 15);
        if (search_cost(J, predDist, cbf0Bits, predPsy) < fullCost) cbf[0] = cbf[1] = cbf[2] = 0;
    }
    else
    {
        /* nothing quantises to a level: the walk leaves every flag at 0 (the root's cost comparison cannot set one) */
        cbf[0] = cbf[1] = cbf[2] = 0;
    }
    const uint32_t rootCbf = cbf[0] | cbf[1] | cbf[2];
    /* ---- the CU's bits (search.cpp:2900-2930): not a merged CU, so never a skip: the root flag is coded ---- */
    uint64_t fD = J.frac & 32767;
    if (lane == 0)
    {
        const XaSearchRef& R = J.ref[S.best];
        fD += cb_bin(S.ctxD + CC_SKIP + J.skip_ctx, 0);
        S.rdBits[2] = (uint32_t)(fD >> 15);
        fD += cb_bin(S.ctxD + CC_PRED_MODE, 0);
        fD += cb_bin(S.ctxD + CC_PART_SIZE, 1);
        fD += cb_bin(S.ctxD + CC_MERGE_FLAG, 0);
        fD += search_amvp_bits(S.ctxD, R.ref_idx, J.num_ref_idx0, S.bestMv[0] - S.bestMvp[0], S.bestMv[1] - S.bestMvp[1], S.bestMvpIdx);
        S.rdBits[1] = (uint32_t)(fD >> 15) - S.rdBits[2];
        fD += cb_bin(S.ctxD + CC_QT_ROOT_CBF, rootCbf);
        if (rootCbf)
        {
            fD += cb_bin(S.ctxD + CC_QT_CBF + 2, cbf[1]);
            fD += cb_bin(S.ctxD + CC_QT_CBF + 2, cbf[2]);
            if (cbf[1] | cbf[2]) fD += cb_bin(S.ctxD + CC_QT_CBF + 1, cbf[0]);
            /* cu_qp_delta with the first coded block flag (encodeTransform with bCodeDQP: any CU that has a residual, entropy.cpp:1207-1222) */
            if (J.dqp) fD += chain_dqp_bits(S.ctxD, (int)(int8_t)(J.dqp & 0xff));
        }
    }
    xa_wave_sync();
    fD = __shfl(fD, 0, 64);
    if (rootCbf)
        for (int p = 0; p < 3; p++)
        {
            if (cbf[p]) fD += wave_coeff_bits(S.ctxD, S.ctxD, reinterpret_cast<const int16_t*>(S.tu[p].coeff), logs[p], p, 0, 0, J.sign_hide, S.step, lane);
            xa_wave_sync();
        }
    chain_sse_t dist = 0;
    dist += cbf[0] ? (chain_sse_t)S.tr[0].nz_dist : (chain_sse_t)S.tr[0].zero_dist;
    chain_sse_t cd = cbf[1] ? (chain_sse_t)S.tr[1].nz_dist : (chain_sse_t)S.tr[1].zero_dist;
    cd += cbf[2] ? (chain_sse_t)S.tr[2].nz_dist : (chain_sse_t)S.tr[2].zero_dist;
    dist += cd;
    const uint32_t psy = J.psy_rd ? (cbf[0] ? S.tr[0].nz_energy : S.tr[0].zero_energy) : 0;
    if (lane == 0)
    {
        uint32_t bits = (uint32_t)(fD >> 15), again = 0;
        if (rootCbf && (J.dqp >> 17))
        {
            /* Search::checkDQP (search.cpp:3974-4003) of a CU at or above the quantisation groups' depth: resetBits(), codeDeltaQP, the bits added to the mode's */
            fD = (fD & 32767) + chain_dqp_bits(S.ctxD, (int)(int8_t)(J.dqp & 0xff));
            again = (uint32_t)(fD >> 15);
            bits += again;
        }
        S.fracD = fD; S.rdBits[0] = bits; S.rdPad = again; S.rdPsy = psy; S.rdCbf[0] = cbf[0]; S.rdCbf[1] = cbf[1]; S.rdCbf[2] = cbf[2];
        S.rdCost = search_cost(J, dist, bits, psy); S.rdLuma = (uint64_t)(dist - cd); S.rdChroma = (uint64_t)cd;
    }
}

__device__ __noinline__ void xa_op_me_call(int which, const XaCmd& c, int tid);     /* device_queue.hip: the job server's search bodies */

XA_DEV void block_inter_search(const XaSearchJob* jobAddr, char* smem, int tid)
{
    constexpr int NT = 64 * XA_SERVER_WAVES;
    const int lane = tid & 63, wv = tid >> 6;
    static_assert(sizeof(SearchLds) <= XA_SEARCH_HEADER, "search header");
    static_assert((XA_SERVER_WAVES + 1) * sizeof(TuLds) <= XA_SEARCH_LDS_BELOW, "LDS budget");
    SearchLds& S = *reinterpret_cast<SearchLds*>(smem + XA_SEARCH_LDS_BELOW);
    TuLds* TL = reinterpret_cast<TuLds*>(smem);
    __syncthreads();
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(jobAddr);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&S.job);
        for (int i = tid; i < (int)(sizeof(XaSearchJob) / 8); i += NT) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = tid; i < 256; i += NT) S.step[i] = en_step.v[i];
    }
    __syncthreads();
    const XaSearchJob& J = S.job;
    const uint64_t* planes = reinterpret_cast<const uint64_t*>(J.planes);
    const uint64_t* srcPlanes = planes + 3 * (J.num_pics - 1);
    XaSearchOut* out = reinterpret_cast<XaSearchOut*>(J.out);
    const size_t isz = sizeof(pixel);
    const int x = J.x, y = J.y, log2 = J.log2, size = 1 << log2, nr = J.num_refs;
    /* scratch: [0, 4K) search jobs, [4K, 5K) groups, [5K, 6K) results, [8K, ..) the predictors' prediction blocks, then the transform units' outputs */
    x265amd_me_job* meJobs = reinterpret_cast<x265amd_me_job*>(J.scratch);
    x265amd_me_group* meGroups = reinterpret_cast<x265amd_me_group*>(J.scratch + 4096);
    x265amd_me_result* meOut = reinterpret_cast<x265amd_me_result*>(J.scratch + 5120);
    const uint64_t predScratch = J.scratch + 8192, tuScratch = J.scratch + 8192 + (uint64_t)2 * XA_SEARCH_MAX_REFS * 4096 * isz;
    /* ---- selectMVP (search.cpp:1992-2024): the SAD of the prediction at each of the two predictors, where they differ ---- */
    if (tid == 0)
        for (int r = 0; r < nr; r++)
            for (int k = 0; k < 2; k++)
            {
                x265amd_mc_job& j = S.mc[2 * r + k];
                j = x265amd_mc_job{};
                int mx = J.ref[r].amvp[k][0], my = J.ref[r].amvp[k][1];
                search_clip_mv(mx, my, x, y, J.pic_w, J.pic_h);
                j.dst_y = predScratch + (uint64_t)(2 * r + k) * 4096 * isz; j.dst_stride = size;
                j.x = (int16_t)x; j.y = (int16_t)y; j.cu_x = (int16_t)x; j.cu_y = (int16_t)y; j.w = (uint8_t)size; j.h = (uint8_t)size;
                j.ref0 = (int8_t)J.ref[r].ref_pic; j.ref1 = -1; j.mv0[0] = (int16_t)mx; j.mv0[1] = (int16_t)my;
                j.slice_type = 1; j.flags = 1; j.metric = 1; j.chroma_cost = 0;
                S.cost[2 * (2 * r + k)] = 0;
            }
    __syncthreads();
    {
        XaArgsMc a;
        a.planes = planes; a.stride = J.stride; a.cstride = J.cstride; a.picW = J.pic_w; a.picH = J.pic_h; a.jobs = nullptr; a.n = 2 * nr;
        a.fencPlanes = srcPlanes; a.fstride = J.stride; a.fcstride = J.cstride; a.cost = S.cost;
        for (int t = wv; t < 2 * nr; t += XA_SERVER_WAVES)
        {
            const int r = t >> 1;
            const bool differ = J.ref[r].amvp[0][0] != J.ref[r].amvp[1][0] || J.ref[r].amvp[0][1] != J.ref[r].amvp[1][1];
            if (differ) wave_mc_rec<true>(a, S.mc[t], t, lane);
        }
    }
    __syncthreads();
    /* ---- the predictor, setSearchRange (search.cpp:2724-2768), the search jobs: one per reference picture, a group each (x265amd_me_plan's window) ---- */
    if (tid == 0)
    {
        int maxW = 80, maxH = 80;
        for (int r = 0; r < nr; r++)
        {
            const XaSearchRef& R = J.ref[r];
            int idx = 0;
            if (R.amvp[0][0] != R.amvp[1][0] || R.amvp[0][1] != R.amvp[1][1])
            {
                uint32_t c2[2] = { S.cost[2 * (2 * r)], S.cost[2 * (2 * r + 1)] };
                if (J.frame_parallel)
                    for (int k = 0; k < 2; k++) if (R.amvp[k][1] >= (J.search_range + 1) * 4) c2[k] = 1u << 28;
                idx = c2[0] <= c2[1] ? 0 : 1;
            }
            S.mvpIdx[r] = idx;
            const int mvpx = R.amvp[idx][0], mvpy = R.amvp[idx][1];
            int mnx = mvpx - (J.merange << 2), mny = mvpy - (J.merange << 2), mxx = mvpx + (J.merange << 2), mxy = mvpy + (J.merange << 2);
            search_clip_mv(mnx, mny, x, y, J.pic_w, J.pic_h); search_clip_mv(mxx, mxy, x, y, J.pic_w, J.pic_h);
            const int maxLen = (1 << 15) - 1;
            mnx = max(mnx, -maxLen); mny = max(mny, -maxLen); mxx = min(mxx, maxLen); mxy = min(mxy, maxLen);
            mnx >>= 2; mny >>= 2; mxx >>= 2; mxy >>= 2;
            mny = min(mny, J.lag_pixels); mxy = min(mxy, J.lag_pixels);
            mxy = max(mxy, mny);
            x265amd_me_job j{};
            j.x = (int16_t)x; j.y = (int16_t)y; j.w = (uint8_t)size; j.h = (uint8_t)size;
            j.method = (uint8_t)J.search_method; j.subme = (uint8_t)J.subme; j.qp = (uint8_t)J.me_qp; j.num_cand = (uint8_t)R.num_mvc; j.merange = (int16_t)J.merange;
            j.mvmin[0] = (int16_t)mnx; j.mvmin[1] = (int16_t)mny; j.mvmax[0] = (int16_t)mxx; j.mvmax[1] = (int16_t)mxy;
            j.mvp[0] = (int16_t)mvpx; j.mvp[1] = (int16_t)mvpy;
            for (int k = 0; k < R.num_mvc; k++) { j.mvc[k][0] = R.mvc[k][0]; j.mvc[k][1] = R.mvc[k][1]; }
            meJobs[r] = j;
            /* the window: the job's search area with the 8-tap margins and the +-2 the hexagon may step outside, clipped to 192 x 192 around its centre */
            int x0 = x + mnx - 6, y0 = y + mny - 6, x1 = x + size + mxx + 10, y1 = y + size + mxy + 7;
            int w = (x1 - x0 + 3) & ~3, h = y1 - y0;
            if (w > 192) { x0 += (w - 192) / 2; w = 192; }
            if (h > 192) { y0 += (h - 192) / 2; h = 192; }
            x265amd_me_group g{};
            g.first_job = r; g.num_jobs = 1; g.ref = R.ref_pic; g.win_x = (int16_t)x0; g.win_y = (int16_t)y0; g.win_w = (int16_t)w; g.win_h = (int16_t)h;
            g.fenc_x = (int16_t)((x >> 6) << 6); g.fenc_y = (int16_t)((y >> 6) << 6);
            meGroups[r] = g;
            maxW = max(maxW, w); maxH = max(maxH, h);
        }
        MeParams p;
        p.fenc = reinterpret_cast<const pixel*>(srcPlanes[0]); p.refs = reinterpret_cast<const uint64_t*>(J.luma_tab); p.chroma = nullptr; p.stride = J.stride; p.cstride = J.cstride;
        p.groups = meGroups; p.jobs = meJobs; p.out = meOut; p.tables = reinterpret_cast<const uint16_t*>(J.me_tables); p.maxWinW = maxW; p.maxWinH = maxH;
        S.cmd = XaCmd{};
        S.cmd.count = (uint32_t)nr;
        static_assert(sizeof(MeParams) <= sizeof(S.cmd.args), "MeParams is a command's argument record");
        *reinterpret_cast<MeParams*>(S.cmd.args) = p;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    /* ---- MotionEstimate::motionEstimate per reference picture (window-resident pass, then the pass that redoes from memory what left its window) ---- */
    xa_op_me_call((J.search_method & 0x7f) == X265AMD_ME_STAR ? 1 : 0, S.cmd, tid);
    __syncthreads();
    xa_op_me_call(2, S.cmd, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    /* ---- the results' bits and costs, checkBestMVP (search.cpp:2702-2713), the cheapest reference (the first of equals) ---- */
    if (tid == 0)
    {
        const float* bs = reinterpret_cast<const float*>(J.bitsize);
        const uint16_t* mvc = reinterpret_cast<const uint16_t*>(J.mvcost);
        auto getCost = [&](uint32_t bits) { return (uint32_t)((bits * J.me_lambda + 128) >> 8); };
        S.best = -1; S.bestCost = 0xFFFFFFFFu;
        for (int r = 0; r < nr; r++)
        {
            const XaSearchRef& R = J.ref[r];
            const x265amd_me_result res = meOut[r];
            int mvpIdx = S.mvpIdx[r];
            const int mx = res.mv[0], my = res.mv[1];
            uint32_t bits = (uint32_t)J.list_sel_bits0 + 1 + (uint32_t)(R.ref_idx + (R.ref_idx < J.num_ref_idx0 - 1));
            bits += search_bitcost(bs, mx, my, R.amvp[mvpIdx][0], R.amvp[mvpIdx][1]);
            const uint32_t mvCost = (uint32_t)mvc[mx - R.amvp[mvpIdx][0]] + (uint32_t)mvc[my - R.amvp[mvpIdx][1]];
            uint32_t cc = ((uint32_t)res.cost - mvCost) + getCost(bits);
            const int diffBits = (int)search_bitcost(bs, mx, my, R.amvp[!mvpIdx][0], R.amvp[!mvpIdx][1]) - (int)search_bitcost(bs, mx, my, R.amvp[mvpIdx][0], R.amvp[mvpIdx][1]);
            if (diffBits < 0)
            {
                mvpIdx = !mvpIdx;
                const uint32_t orig = bits;
                bits = orig + diffBits;
                cc = (cc - getCost(orig)) + getCost(bits);
            }
            if (cc < S.bestCost)
            {
                S.best = r; S.bestCost = cc; S.bestBits = bits; S.bestMvCost = mvCost; S.bestMvpIdx = mvpIdx;
                S.bestMv[0] = mx; S.bestMv[1] = my; S.bestMvp[0] = R.amvp[mvpIdx][0]; S.bestMvp[1] = R.amvp[mvpIdx][1];
            }
        }
        /* the winner's prediction: Predict::motionCompensation with chroma into the CU's prediction tile */
        if (S.best >= 0)
        {
            x265amd_mc_job& j = S.mc[0];
            j = x265amd_mc_job{};
            j.dst_y = J.pred_tile; j.dst_u = J.pred_tile + 4096 * isz; j.dst_v = j.dst_u + 1024 * isz; j.dst_stride = 64; j.dst_cstride = 32;
            j.x = (int16_t)x; j.y = (int16_t)y; j.cu_x = (int16_t)x; j.cu_y = (int16_t)y; j.w = (uint8_t)size; j.h = (uint8_t)size;
            j.ref0 = (int8_t)J.ref[S.best].ref_pic; j.ref1 = -1; j.mv0[0] = (int16_t)S.bestMv[0]; j.mv0[1] = (int16_t)S.bestMv[1];
            j.slice_type = 1; j.flags = 3;
        }
    }
    if (tid == 0)       /* algorithmic bytes besides the searches' own (windows, source tiles, records: counted by their bodies): the record, the predictors' blocks read with
                         * their interpolation border and written, the winner's prediction, source and units, the result with its levels */
        XA_BYTES(sizeof(XaSearchJob) + (unsigned long long)2 * nr * ((unsigned)(size + 7) * (size + 7) + (unsigned)size * size) * sizeof(pixel) +
                 ((unsigned long long)(size + 7) * (size + 7) + 2ull * (size / 2 + 3) * (size / 2 + 3) + 6ull * size * size) * sizeof(pixel) + 256 + 3ull * size * size);
    for (int i = tid; i < 3 * 16; i += NT) (&S.acc[0][0])[i] = 0;
    __syncthreads();
    if (S.best < 0)
    {
        if (tid == 0) { struct alignas(8) H { uint32_t valid; int32_t best; } h{ 1u, -1 }; xa_st_result(reinterpret_cast<H*>(out), h); }
        return;
    }
    /* ---- the prediction and its SA8D, tile by tile over the wavefronts (as the skip chain's candidates: inter_chain_dev.h) ---- */
    {
        const int C = size >> 1;
        const int tY = size >> 3, nY = tY * tY, tC = C >> 3, nC = tC * tC;
        const int total = nY + (nC ? 2 * nC : 1);
        const x265amd_mc_job& j = S.mc[0];
        const McSetup su = mc_setup(j, J.pic_w, J.pic_h);
        for (int t = wv; t < total; t += XA_SERVER_WAVES)
        {
            if (t < nY)
            {
                const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, 0);
                const int ty = t / tY, tx = t - ty * tY;
                const int raw = chain_pred_tile<8>(j, su, P, reinterpret_cast<const pixel*>(srcPlanes[0]) + (size_t)y * J.stride + x, J.stride, tx, ty, lane);
                if (lane == 0) atomicAdd(&S.acc[0][size == 8 ? 0 : (ty >> 1) * (size >> 4) + (tx >> 1)], (unsigned int)raw);
            }
            else if (nC)
            {
                const int u = t - nY, pl = 1 + u / nC, k = u % nC, ty = k / tC, tx = k - ty * tC;
                const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, pl);
                const int raw = chain_pred_tile<4>(j, su, P, reinterpret_cast<const pixel*>(srcPlanes[pl]) + (size_t)(y >> 1) * J.cstride + (x >> 1), J.cstride, tx, ty, lane);
                if (lane == 0) atomicAdd(&S.acc[pl][C == 8 ? 0 : (ty >> 1) * (C >> 4) + (tx >> 1)], (unsigned int)raw);
            }
            else
            {
                const int pl = 1 + ((lane >> 4) & 1), l = lane & 15, xx = l & 3, yy = l >> 2;
                int d = 0;
                if (lane < 32)
                {
                    const McPlane P = mc_plane_of(j, su, planes, J.stride, J.cstride, pl);
                    const int v = mc_one<4>(P, j, su.mode, su.lsel, xx, yy);
                    P.dst[(long)yy * P.dstStride + xx] = (pixel)v;
                    d = (int)(reinterpret_cast<const pixel*>(srcPlanes[pl]) + (size_t)(y >> 1) * J.cstride + (x >> 1))[yy * J.cstride + xx] - v;
                }
                const int sum = xa_row16_sum(abs(xa_lane_had4x4(d, lane)));
                if (lane < 32 && l == 0) S.acc[pl][0] = (unsigned int)sum;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
    {
        const int C = size >> 1, gY = size == 8 ? 1 : (size >> 4) * (size >> 4), gC = C <= 8 ? 1 : (C >> 4) * (C >> 4);
        unsigned int sy = 0, sc = 0;
        for (int k = 0; k < gY; k++) sy += (S.acc[0][k] + 2) >> 2;
        for (int pl = 1; pl < 3; pl++)
        {
            if (C == 4) sc += S.acc[pl][0] >> 1;
            else for (int k = 0; k < gC; k++) sc += (S.acc[pl][k] + 2) >> 2;
        }
        S.sa8dLuma = sy; S.sa8d = sy + sc;
    }
    const bool doRd = J.do_rd && log2 <= 5;
    if (doRd)
    {
        /* ---- the transform units (one per plane) and the CU's rate-distortion ---- */
        const int C = log2 - 1 < 2 ? 2 : log2 - 1;
        if (tid < 3)
        {
            const int p = tid, n = p ? 1 << C : size;
            x265amd_tu_job& j = S.tu[p];
            j = x265amd_tu_job{};
            const pixel* tile = reinterpret_cast<const pixel*>(J.pred_tile);
            j.fenc = p ? srcPlanes[p] + ((uint64_t)(y >> 1) * J.cstride + (x >> 1)) * isz : srcPlanes[0] + ((uint64_t)y * J.stride + x) * isz;
            j.pred = (uint64_t)(uintptr_t)(p ? tile + 4096 + (size_t)(p - 1) * 1024 : tile);
            const uint64_t off = p ? 1024 + (uint64_t)(p - 1) * 256 : 0;
            j.coeff = tuScratch + off * 2;
            j.resi = tuScratch + 1536 * 2 + off * 2;
            j.recon = tuScratch + 1536 * 4 + off * isz;
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
        if (wv == 0) search_inter_rd(S, log2, lane);
        __syncthreads();
        /* the reconstruction: the kept units' reconstruction, the others' prediction */
        const pixel* tile = reinterpret_cast<const pixel*>(J.pred_tile);
        pixel* tr = reinterpret_cast<pixel*>(J.recon_tile);
        for (int p = 0; p < 3; p++)
        {
            const int lg = p ? C : log2, off = p ? 4096 + (p - 1) * 1024 : 0, st = p ? 32 : 64;
            if (S.rdCbf[p]) chain_copy_plane(tr + off, st, reinterpret_cast<const pixel*>(S.tu[p].recon), 1 << lg, lg, p != 0, tid, NT);
            else chain_copy_plane(tr + off, st, tile + off, st, lg, p != 0, tid, NT);
        }
        for (int p = 0; p < 3; p++)
        {
            const int lg = p ? C : log2, n2 = 1 << (2 * lg);
            const int16_t* lv = reinterpret_cast<const int16_t*>(S.tu[p].coeff);
            int16_t* dst = out->levels + (p ? 1024 + (p - 1) * 256 : 0);
            for (int i = tid; i < n2 / 4; i += NT)
                __hip_atomic_store(reinterpret_cast<uint64_t*>(dst + 4 * i), *reinterpret_cast<const uint64_t*>(lv + 4 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (tid < X265AMD_CTX_STRIDE / 8) __hip_atomic_store(reinterpret_cast<uint64_t*>(out->ctx) + tid, reinterpret_cast<const uint64_t*>(S.ctxD)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (tid == 64)
    {
        struct alignas(8) Head
        {
            uint32_t valid; int32_t best; int16_t mv[2], mvp[2]; int32_t mvp_idx; uint32_t bits, cost, mv_cost; uint32_t sa8d, sa8d_luma; uint8_t cbf[3], rd_done;
            uint32_t total_bits, mv_bits, coeff_bits, psy_energy, res_energy, reserved1; uint64_t rd_cost, luma_dist, chroma_dist, frac;
        } h;
        h.valid = 1; h.best = S.best; h.mv[0] = (int16_t)S.bestMv[0]; h.mv[1] = (int16_t)S.bestMv[1]; h.mvp[0] = (int16_t)S.bestMvp[0]; h.mvp[1] = (int16_t)S.bestMvp[1];
        h.mvp_idx = S.bestMvpIdx; h.bits = S.bestBits; h.cost = S.bestCost; h.mv_cost = S.bestMvCost; h.sa8d = S.sa8d; h.sa8d_luma = S.sa8dLuma;
        h.cbf[0] = doRd ? (uint8_t)S.rdCbf[0] : 0; h.cbf[1] = doRd ? (uint8_t)S.rdCbf[1] : 0; h.cbf[2] = doRd ? (uint8_t)S.rdCbf[2] : 0; h.rd_done = doRd ? 1 : 0;
        h.total_bits = doRd ? S.rdBits[0] : 0; h.mv_bits = doRd ? S.rdBits[1] : 0; h.coeff_bits = doRd ? S.rdBits[0] - S.rdPad - S.rdBits[1] - S.rdBits[2] : 0;
        h.psy_energy = doRd ? S.rdPsy : 0; h.res_energy = doRd ? (uint32_t)S.tr[0].zero_dist : 0; h.reserved1 = 0;
        h.rd_cost = doRd ? S.rdCost : 0; h.luma_dist = doRd ? S.rdLuma : 0; h.chroma_dist = doRd ? S.rdChroma : 0; h.frac = doRd ? S.fracD : 0;
        static_assert(sizeof(Head) == offsetof(XaSearchOut, ctx), "the search result's head");
        xa_st_result(reinterpret_cast<Head*>(out), h);
    }
}

#endif
#endif
