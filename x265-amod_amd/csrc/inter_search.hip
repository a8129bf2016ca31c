/* Whole-CU inter search for a batch of CUs (include/x265amd.h: x265amd_pred_inter_search): the host side of SURVEY row a3.
 *
 * Restatement of Search::predInterSearch (reference: source/encoder/search.cpp:2181-2647) with mergeEstimation (:1891-1966), selectMVP
 * (:1992-2018), setSearchRange (:2724-2768), checkBestMVP (:2702-2713), getBlkBits (:2649-2700) for many CUs at once.  The reference
 * walks one CU and calls the primitives block by block; here the decisions stay on the host, in the reference's order, and every
 * block operation of the same step of all CUs goes to the GPU as one batch:
 *   1. candidate lists (x265amd_merge_candidates / x265amd_amvp_candidates) -> one x265amd_inter_cost launch: SAD of both AMVP
 *      candidates of every (list, reference) and SATD (+ chroma) of every merge candidate;
 *   2. one x265amd_me_search launch for every (PU, list, reference);
 *   3. B slices: one x265amd_inter_cost launch for the bi-prediction tries (found vectors, zero vectors);
 *   4. the choice merge / bi / L0 / L1, then one x265amd_motion_compensation launch for the final predictions.
 * Second PUs of two-part CUs run through the same steps afterwards, with the first PU's choice patched into the motion field.
 * Not supported (rejected or absent): weighted prediction, HME, analysis reuse, distributed ME, frame-parallel lag clipping.
 */
#include "inter_common.h"
#include "xa_queue.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace xa_inter;

namespace {

struct MeBest { Mv mv, mvp; int mvpIdx, ref, bits; uint32_t mvCost, cost; };

struct PuWork
{
    int cu;                 /* index into the CU list */
    Geo g;
    /* merge */
    x265amd_merge_cand mcand[5]; int nMerge; int mergeJob0;     /* first inter-cost job of the candidates */
    uint32_t mrgCost; int mrgBits, mrgIdx;
    /* per (list, ref) */
    int16_t amvp[2][16][2][2]; int16_t mvc[2][16][12][2]; int numMvc[2][16]; int mvpJob0[2][16]; int mvpIdx[2][16]; int meJob[2][16];
    uint32_t listSelBits[3];
    MeBest best[2];
    int bidirJob[2];        /* inter-cost jobs of the two bi-prediction tries, -1 when not made */
};

struct Dev
{
    void* p = nullptr;
    ~Dev() { xa_scratch_free(p); }
    int alloc(size_t bytes) { return xa_scratch_alloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : -1; }
};

} // namespace

extern "C" int x265amd_pred_inter_search(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                                         x265amd_mv_unit* cur, const x265amd_mv_unit* col, const uint64_t* h_planes, int num_pics, intptr_t stride, intptr_t cstride,
                                         const x265amd_inter_cu* cus, int n, x265amd_pu_result* out, int32_t* bits_out, uint64_t d_pred, size_t pred_bytes_per_cu)
{
    return x265amd_pred_inter_search_ex(me, stream, I, S, cur, col, h_planes, num_pics, stride, cstride, cus, n, out, bits_out, d_pred, pred_bytes_per_cu, nullptr, nullptr);
}

extern "C" int x265amd_pred_inter_search_ex(x265amd_me_ctx* me, void* stream, const x265amd_mvpred_info* I, const x265amd_inter_search_params* S,
                                            x265amd_mv_unit* cur, const x265amd_mv_unit* col, const uint64_t* h_planes, int num_pics, intptr_t stride, intptr_t cstride,
                                            const x265amd_inter_cu* cus, int n, x265amd_pu_result* out, int32_t* bits_out, uint64_t d_pred, size_t pred_bytes_per_cu,
                                            x265amd_me_detail* detail, const uint32_t* ref_masks)
{
    if (!me || !I || !S || !cur || !h_planes || !cus || !out || !bits_out || n < 0 || num_pics < 2)
        return xa_fail(X265AMD_EINVAL, "x265amd_pred_inter_search: bad arguments");
    if (n == 0) return X265AMD_OK;
    XA_HOSTPROF("is.pred_inter_search (all)");
    hipStream_t st = (hipStream_t)stream;
    const int srcPic = num_pics - 1, isB = I->is_inter_b, w4 = I->pic_width >> 2;
    const size_t isz = sizeof(x265amd_pixel);
    const uint64_t lambda = (uint64_t)floor(256.0 * is_lambda(S->qp));
    auto getCost = [&](uint32_t bits) { return (uint32_t)((bits * lambda + 128) >> 8); };
    /* refMasks of predInterSearch (search.cpp:2389-2400, :2469): bit r of the low half allows reference r of list 0, the high half list 1; 0 = all */
    auto allowed = [&](int cu, int pidx, int list, int ref) {
        const uint32_t m = ref_masks && ref_masks[2 * cu + pidx] ? ref_masks[2 * cu + pidx] : 0xFFFFFFFFu;
        return ((m >> (16 * list)) >> ref) & 1u;
    };
    const uint16_t* mvcostTab = x265amd_me_host_mvcost(me, S->qp) + 65536;
    auto mvcost = [&](Mv mv, Mv mvp) { return (uint32_t)mvcostTab[mv.x - mvp.x] + (uint32_t)mvcostTab[mv.y - mvp.y]; };

    /* device tables of plane addresses: MC / cost table (numPics x 3), ME luma table, ME chroma table */
    std::vector<uint64_t> lumaTab(num_pics), chromaTab(2 + 2 * num_pics);
    for (int i = 0; i < num_pics; i++) { lumaTab[i] = h_planes[3 * i]; chromaTab[2 + 2 * i] = h_planes[3 * i + 1]; chromaTab[3 + 2 * i] = h_planes[3 * i + 2]; }
    chromaTab[0] = h_planes[3 * srcPic + 1]; chromaTab[1] = h_planes[3 * srcPic + 2];
    /* Tables and job records are pushed into device memory by the host (XaMapped), results come back in pinned host memory (XaMappedOut): no copy commands.
     * The kernels read the tables and the search's group / job records with plain (partly scalar) loads, and the pool hands out reused blocks: the first
     * command behind a push carries an acquire, which also empties the scalar data cache. */
    XaMapped dPlanes, dLuma, dChroma;
    if (dPlanes.alloc(num_pics * 24) != hipSuccess || dLuma.alloc(num_pics * 8) != hipSuccess || dChroma.alloc(chromaTab.size() * 8) != hipSuccess)
        return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: out of device memory");
    memcpy(dPlanes.p, h_planes, (size_t)num_pics * 24);
    memcpy(dLuma.p, lumaTab.data(), (size_t)num_pics * 8);
    memcpy(dChroma.p, chromaTab.data(), chromaTab.size() * 8);
    xa_q_next_flags(st, XA_CMD_ACQUIRE);
    const uint64_t* dFencTab = (const uint64_t*)dPlanes.p + 3 * srcPic;

    struct CuState { int numPart, lastMode, totalBits; };
    std::vector<CuState> cs(n);
    static const uint8_t nbParts[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
    for (int i = 0; i < n; i++)
    {
        if (cus[i].part_size == 3 || cus[i].log2_size < 3 || cus[i].log2_size > 6) return xa_fail(X265AMD_EINVAL, "x265amd_pred_inter_search: unsupported CU");
        cs[i] = CuState{ nbParts[cus[i].part_size], 0, 0 };
        bits_out[i] = 0;
    }
    memset(out, 0, sizeof(x265amd_pu_result) * 2 * (size_t)n);
    std::vector<x265amd_mc_job> finalMc;

    auto clipMv = [&](Mv& mv, int cuX, int cuY) { clip_mv(mv, cuX, cuY, I->pic_width, I->pic_height); };
    const int lagPixels = S->frame_parallel ? S->search_range : I->pic_height;          /* Search::m_refLagPixels (search.cpp:92) */
    auto searchRange = [&](Mv mvp, int merange, int cuX, int cuY, Mv& mn, Mv& mx) { search_range(mvp, merange, cuX, cuY, I->pic_width, I->pic_height, lagPixels, mn, mx); };
    auto mcJob = [&](const Geo& g, int cuX, int cuY, int sliceP, int pic0, Mv mv0, int pic1, Mv mv1, int flags, int metric, int chromaCost) {
        x265amd_mc_job j;
        memset(&j, 0, sizeof(j));
        j.x = (int16_t)g.x; j.y = (int16_t)g.y; j.cu_x = (int16_t)cuX; j.cu_y = (int16_t)cuY; j.w = (uint8_t)g.w; j.h = (uint8_t)g.h;
        j.ref0 = (int8_t)pic0; j.ref1 = (int8_t)pic1;
        j.mv0[0] = (int16_t)mv0.x; j.mv0[1] = (int16_t)mv0.y; j.mv1[0] = (int16_t)mv1.x; j.mv1[1] = (int16_t)mv1.y;
        j.slice_type = (uint8_t)sliceP; j.flags = (uint8_t)flags; j.metric = (uint8_t)metric; j.chroma_cost = (uint8_t)chromaCost;
        return j;
    };
    /* a job that stands for Predict::motionCompensation in a slice with weights (predict.cpp:85-232): pps.bUseWeightPred / bUseWeightedBiPred and the table entries of the
     * references it reads (a picture is at most once in a list).  selectMVP's candidates are NOT such jobs: predInterLumaPixel on the reconstruction (search.cpp:2016) */
    auto weigh = [&](x265amd_mc_job j) {
        if (!S->weighted) return j;
        j.flags |= S->weighted == 1 ? 4 : 8;
        const int pics[2] = { j.ref0, j.ref1 };
        for (int l = 0; l < 2; l++)
            if (pics[l] >= 0)
                for (int r = 0; r < I->num_ref_idx[l]; r++)
                    if (S->ref_pic[l][r] == pics[l]) { memcpy(&j.wp[l][0], &S->wp[l][r][0], sizeof(j.wp[l])); break; }
        return j;
    };
    /* runs a batch of cost jobs; scratch prediction blocks are carved from one arena */
    auto runCost = [&](std::vector<x265amd_mc_job>& jobs, std::vector<uint32_t>& cost) -> int {
        cost.assign(2 * jobs.size(), 0);
        if (jobs.empty()) return 0;
        XA_HOSTPROF("is.runCost (incl. wait)");
        size_t arena = 0;
        std::vector<size_t> offs(jobs.size());
        for (size_t i = 0; i < jobs.size(); i++) { offs[i] = arena; arena += ((size_t)jobs[i].w * jobs[i].h * 3 / 2 * isz + 63) & ~(size_t)63; }
        Dev dArena; XaMapped dJobs; XaMappedOut dCost;
        if (dArena.alloc(arena) || dJobs.alloc(jobs.size() * sizeof(x265amd_mc_job)) != hipSuccess || dCost.alloc(cost.size() * 4) != hipSuccess) return -1;
        for (size_t i = 0; i < jobs.size(); i++)
        {
            const uint64_t b = (uint64_t)(uintptr_t)dArena.p + offs[i];
            jobs[i].dst_y = b; jobs[i].dst_u = b + (size_t)jobs[i].w * jobs[i].h * isz; jobs[i].dst_v = jobs[i].dst_u + (size_t)jobs[i].w * jobs[i].h / 4 * isz;
            jobs[i].dst_stride = jobs[i].w; jobs[i].dst_cstride = jobs[i].w / 2;
        }
        if (xa_ref_guard_mc(jobs.data(), (int)jobs.size())) return -1;
        memcpy(dJobs.p, jobs.data(), jobs.size() * sizeof(x265amd_mc_job));
        if (x265amd_inter_cost(st, (const uint64_t*)dPlanes.p, stride, cstride, I->pic_width, I->pic_height, (const x265amd_mc_job*)dJobs.p, (int)jobs.size(),
                               dFencTab, stride, cstride, (uint32_t*)dCost.p) != X265AMD_OK) return -1;
        if (xa_stream_sync(st) != hipSuccess) return -1;
        memcpy(cost.data(), dCost.p, cost.size() * 4);
        return 0;
    };

    for (int pidx = 0; pidx < 2; pidx++)
    {
        std::vector<PuWork> W;
        for (int i = 0; i < n; i++)
            if (pidx < cs[i].numPart)
            {
                PuWork w;
                memset(&w, 0, sizeof(w));
                w.cu = i; w.g = pu_geo(cus[i].x, cus[i].y, 1 << cus[i].log2_size, cus[i].part_size, pidx);
                W.push_back(w);
            }
        if (W.empty()) break;
        /* ---- step 1: candidates and their cost jobs ---- */
        std::vector<x265amd_mc_job> cj;
        { XA_HOSTPROF("is.step1 candidates");
        for (PuWork& w : W)
        {
            const x265amd_inter_cu& c = cus[w.cu];
            const int size = 1 << c.log2_size;
            const bool chromaOk = !((w.g.w >> 1) & 3) && !((w.g.h >> 1) & 3);
            const bool chromaSatd = S->subpel_refine > 2 && (S->chroma_mc != 0) && chromaOk;       /* MotionEstimate::setSourcePU (motion.cpp:234-237) */
            /* the second PU sees the first one's choice where the reference reads it from the CU under analysis */
            std::vector<x265amd_mv_unit> saved;
            Geo g0 = pu_geo(c.x, c.y, size, c.part_size, 0);
            if (pidx == 1)
            {
                const x265amd_pu_result& r0 = out[2 * w.cu];
                for (int y = g0.y; y < g0.y + g0.h; y += 4)
                    for (int x = g0.x; x < g0.x + g0.w; x += 4)
                    {
                        x265amd_mv_unit& u = cur[(y >> 2) * w4 + (x >> 2)];
                        saved.push_back(u);
                        u.pred_mode = X265AMD_MODE_INTER; u.inter_dir = r0.inter_dir;
                        for (int l = 0; l < 2; l++) { u.ref_idx[l] = r0.ref_idx[l]; u.mv[l][0] = r0.mv[l][0]; u.mv[l][1] = r0.mv[l][1]; }
                    }
            }
            w.nMerge = 0; w.mrgCost = 0xFFFFFFFFu; w.mergeJob0 = -1;
            if (cs[w.cu].numPart > 1)
            {
                w.nMerge = x265amd_merge_candidates(I, cur, col, c.x, c.y, c.log2_size, c.part_size, pidx, w.mcand);
                if (c.log2_size == 3 && c.part_size != 0)      /* isBipredRestriction: drop L1 of bi candidates (search.cpp:1900-1911) */
                    for (int k = 0; k < w.nMerge; k++)
                        if (w.mcand[k].dir == 3) { w.mcand[k].dir = 1; w.mcand[k].ref_idx[1] = -1; }
                w.mergeJob0 = (int)cj.size();
                for (int k = 0; k < w.nMerge; k++)
                {
                    const x265amd_merge_cand& m = w.mcand[k];
                    const int p0 = m.ref_idx[0] >= 0 ? S->ref_pic[0][m.ref_idx[0]] : -1, p1 = m.ref_idx[1] >= 0 ? S->ref_pic[1][m.ref_idx[1]] : -1;
                    cj.push_back(weigh(mcJob(w.g, c.x, c.y, !isB, p0, Mv{ m.mv[0][0], m.mv[0][1] }, p1, Mv{ m.mv[1][0], m.mv[1][1] }, chromaSatd ? 3 : 1, 2, chromaSatd)));
                }
            }
            /* getBlkBits (search.cpp:2649-2700) */
            {
                const int ps = c.part_size, last = cs[w.cu].lastMode;
                uint32_t* b = w.listSelBits;
                if (ps == 0) { b[0] = isB ? 3 : 1; b[1] = 3; b[2] = 5; }
                else if (!isB) { b[0] = 3; b[1] = 0; b[2] = 0; }
                else if (ps == 1 || ps == 4 || ps == 5)
                {
                    static const uint32_t t[2][3][3] = { { { 0, 0, 3 }, { 0, 0, 0 }, { 0, 0, 0 } }, { { 5, 7, 7 }, { 7, 5, 7 }, { 6, 6, 6 } } };
                    memcpy(b, t[pidx][last], sizeof(uint32_t) * 3);
                }
                else
                {
                    static const uint32_t t[2][3][3] = { { { 0, 2, 3 }, { 0, 0, 0 }, { 0, 0, 0 } }, { { 5, 7, 7 }, { 5, 5, 7 }, { 6, 6, 6 } } };
                    memcpy(b, t[pidx][last], sizeof(uint32_t) * 3);
                }
            }
            for (int list = 0; list < (isB ? 2 : 1); list++)
                for (int ref = 0; ref < I->num_ref_idx[list]; ref++)
                {
                    w.mvpJob0[list][ref] = -1; w.meJob[list][ref] = -1;
                    if (!allowed(w.cu, pidx, list, ref)) continue;
                    XA_HOSTPROF("is.amvp_candidates");
                    w.numMvc[list][ref] = x265amd_amvp_candidates(I, cur, col, c.x, c.y, c.log2_size, c.part_size, pidx, list, ref, w.amvp[list][ref], w.mvc[list][ref]);
                    if (S->lowres_mvs[list][ref])
                    {
                        /* getLowresMV: the lookahead's vector of the 16x16 block under the PU's centre, scaled up */
                        const int16_t (*lm)[2] = reinterpret_cast<const int16_t (*)[2]>((uintptr_t)S->lowres_mvs[list][ref]);
                        const size_t idx = (size_t)((w.g.y + w.g.h / 2) >> 4) * S->lowres_blocks_in_row + ((w.g.x + w.g.w / 2) >> 4);
                        const int16_t lx = (int16_t)(lm[idx][0] * 2), ly = (int16_t)(lm[idx][1] * 2);
                        if (lx || ly) { int16_t* m = w.mvc[list][ref][w.numMvc[list][ref]++]; m[0] = lx; m[1] = ly; }
                    }
                    w.mvpJob0[list][ref] = -1;
                    const int16_t (*a)[2] = w.amvp[list][ref];
                    if (!(a[0][0] == a[1][0] && a[0][1] == a[1][1]))
                    {
                        w.mvpJob0[list][ref] = (int)cj.size();
                        for (int k = 0; k < 2; k++)
                        {
                            Mv mv{ a[k][0], a[k][1] };
                            clipMv(mv, c.x, c.y);
                            cj.push_back(mcJob(w.g, c.x, c.y, 1, S->ref_pic[list][ref], mv, -1, Mv{ 0, 0 }, 1, 1, 0));
                        }
                    }
                }
            if (pidx == 1)
            {
                size_t k = 0;
                for (int y = g0.y; y < g0.y + g0.h; y += 4)
                    for (int x = g0.x; x < g0.x + g0.w; x += 4) cur[(y >> 2) * w4 + (x >> 2)] = saved[k++];
            }
        }
        }
        std::vector<uint32_t> cost;
        if (runCost(cj, cost)) return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: candidate cost launch failed");

        /* ---- step 2: merge choice, MVP choice, motion searches ---- */
        std::vector<x265amd_me_job> mj;
        std::vector<int> mjPic;
        for (PuWork& w : W)
        {
            const x265amd_inter_cu& c = cus[w.cu];
            const bool chromaOk = !((w.g.w >> 1) & 3) && !((w.g.h >> 1) & 3);
            const bool chromaSatd = S->subpel_refine > 2 && (S->chroma_mc != 0) && chromaOk;
            for (int k = 0; k < w.nMerge; k++)
            {
                /* pictures coded in parallel: no candidate that reaches below the rows the reference pictures have finished (search.cpp:1918-1937) */
                if (S->frame_parallel && (below_lag(w.mcand[k].mv[0][1], S->search_range) || below_lag(w.mcand[k].mv[1][1], S->search_range))) continue;
                const uint32_t bits = (uint32_t)(k + (k < w.nMerge - 1));                  /* getTUBits */
                const uint32_t cc = cost[2 * (w.mergeJob0 + k)] + cost[2 * (w.mergeJob0 + k) + 1] + getCost(bits);
                if (cc < w.mrgCost) { w.mrgCost = cc; w.mrgBits = (int)bits; w.mrgIdx = k; }
            }
            for (int list = 0; list < (isB ? 2 : 1); list++)
                for (int ref = 0; ref < I->num_ref_idx[list]; ref++)
                {
                    if (!allowed(w.cu, pidx, list, ref)) continue;
                    int idx = 0;
                    if (w.mvpJob0[list][ref] >= 0)
                    {
                        /* selectMVP (search.cpp:1992-2024); coded in parallel, a candidate below the lag keeps COST_MAX (:2005-2010) */
                        uint32_t c2[2] = { cost[2 * w.mvpJob0[list][ref]], cost[2 * (w.mvpJob0[list][ref] + 1)] };
                        if (S->frame_parallel)
                            for (int k = 0; k < 2; k++) if (below_lag(w.amvp[list][ref][k][1], S->search_range)) c2[k] = 1u << 28;
                        idx = c2[0] <= c2[1] ? 0 : 1;
                    }
                    w.mvpIdx[list][ref] = idx;
                    const Mv mvp{ w.amvp[list][ref][idx][0], w.amvp[list][ref][idx][1] };
                    Mv mn, mx;
                    searchRange(mvp, S->search_range, c.x, c.y, mn, mx);
                    x265amd_me_job j;
                    memset(&j, 0, sizeof(j));
                    j.x = (int16_t)w.g.x; j.y = (int16_t)w.g.y; j.w = (uint8_t)w.g.w; j.h = (uint8_t)w.g.h;
                    j.method = (uint8_t)(S->search_method | (chromaSatd ? X265AMD_ME_CHROMA_SATD : 0)); j.subme = (uint8_t)S->subpel_refine;
                    j.qp = (uint8_t)S->qp; j.num_cand = (uint8_t)w.numMvc[list][ref]; j.merange = (int16_t)S->search_range;
                    j.mvmin[0] = (int16_t)mn.x; j.mvmin[1] = (int16_t)mn.y; j.mvmax[0] = (int16_t)mx.x; j.mvmax[1] = (int16_t)mx.y;
                    j.mvp[0] = (int16_t)mvp.x; j.mvp[1] = (int16_t)mvp.y;
                    memcpy(j.mvc, w.mvc[list][ref], sizeof(int16_t) * 2 * w.numMvc[list][ref]);
                    w.meJob[list][ref] = (int)mj.size();
                    mj.push_back(j); mjPic.push_back(S->weighted ? S->me_pic[list][ref] : S->ref_pic[list][ref]);        /* MotionEstimate reads MotionReference::fpelPlane: the weighted copy where there is one */
                }
        }
        std::vector<x265amd_me_result> mres(mj.size());
        XA_HOSTPROF("is.me plan+launch+wait+rest");
        if (xa_ref_guard_me(mj.data(), mjPic.data(), (int)mj.size())) return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: a reference picture failed");
        {
            /* plan per reference picture, upload in planned order */
            std::vector<x265amd_me_job> ordered; std::vector<x265amd_me_group> groups; std::vector<int> origin;
            int maxW = 80, maxH = 80, flags = 0;
            for (int pic = 0; pic < num_pics; pic++)
            {
                std::vector<x265amd_me_job> sub; std::vector<int> idx;
                for (size_t i = 0; i < mj.size(); i++) if (mjPic[i] == pic) { sub.push_back(mj[i]); idx.push_back((int)i); }
                if (sub.empty()) continue;
                std::vector<x265amd_me_group> g(sub.size()); std::vector<int32_t> order(sub.size());
                const int ng = x265amd_me_plan(sub.data(), (int)sub.size(), pic, 192, 192, g.data(), order.data());
                if (ng < 0) return ng;
                for (int k = 0; k < ng; k++) { g[k].first_job += (int)ordered.size(); groups.push_back(g[k]); maxW = g[k].win_w > maxW ? g[k].win_w : maxW; maxH = g[k].win_h > maxH ? g[k].win_h : maxH; }
                for (size_t k = 0; k < sub.size(); k++) { ordered.push_back(sub[order[k]]); origin.push_back(idx[order[k]]); }
            }
            for (const x265amd_me_job& j : mj) { if ((j.method & 0x7f) == X265AMD_ME_STAR) flags |= X265AMD_ME_FLAG_STAR; if (j.method & X265AMD_ME_CHROMA_SATD) flags |= X265AMD_ME_FLAG_CHROMA; }
            XaMapped dJ, dG; Dev dO;           /* the results stay in device memory: the deferred pass reads what the first pass left there */
            if (dJ.alloc(ordered.size() * sizeof(x265amd_me_job)) != hipSuccess || dG.alloc(groups.size() * sizeof(x265amd_me_group)) != hipSuccess ||
                dO.alloc(ordered.size() * sizeof(x265amd_me_result)))
                return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: out of device memory");
            memcpy(dJ.p, ordered.data(), ordered.size() * sizeof(x265amd_me_job));
            memcpy(dG.p, groups.data(), groups.size() * sizeof(x265amd_me_group));
            xa_q_next_flags(st, XA_CMD_ACQUIRE);
            int rc = x265amd_me_search(me, st, (const x265amd_pixel*)(uintptr_t)h_planes[3 * srcPic], (const uint64_t*)dLuma.p, stride, (const x265amd_me_group*)dG.p, (int)groups.size(),
                                       (const x265amd_me_job*)dJ.p, (x265amd_me_result*)dO.p, maxW, maxH, flags, (const uint64_t*)dChroma.p, cstride);
            if (rc != X265AMD_OK) return rc;
            std::vector<x265amd_me_result> tmp(ordered.size());
            XA_HIP_CHECK(xa_copy_async(st, tmp.data(), dO.p, tmp.size() * sizeof(x265amd_me_result), hipMemcpyDeviceToHost));
            XA_HIP_CHECK(xa_stream_sync(st));
            for (size_t k = 0; k < tmp.size(); k++) mres[origin[k]] = tmp[k];
        }

        /* ---- step 3: uni-directional bests; bi-prediction tries ---- */
        std::vector<x265amd_mc_job> bj;
        for (PuWork& w : W)
        {
            const x265amd_inter_cu& c = cus[w.cu];
            const bool chromaOk = !((w.g.w >> 1) & 3) && !((w.g.h >> 1) & 3);
            const bool chromaSatd = S->subpel_refine > 2 && (S->chroma_mc != 0) && chromaOk;
            for (int list = 0; list < 2; list++) { w.best[list].cost = 0xFFFFFFFFu; w.best[list].ref = -1; }
            for (int list = 0; list < (isB ? 2 : 1); list++)
                for (int ref = 0; ref < I->num_ref_idx[list]; ref++)
                {
                    if (w.meJob[list][ref] < 0) continue;
                    const x265amd_me_result& r = mres[w.meJob[list][ref]];
                    int mvpIdx = w.mvpIdx[list][ref];
                    const Mv outmv{ r.mv[0], r.mv[1] };
                    const Mv a[2] = { Mv{ w.amvp[list][ref][0][0], w.amvp[list][ref][0][1] }, Mv{ w.amvp[list][ref][1][0], w.amvp[list][ref][1][1] } };
                    uint32_t bits = w.listSelBits[list] + 1 + (uint32_t)(ref + (ref < I->num_ref_idx[list] - 1));
                    bits += is_bitcost(outmv, a[mvpIdx]);
                    const uint32_t mvCost = mvcost(outmv, a[mvpIdx]);
                    uint32_t cc = ((uint32_t)r.cost - mvCost) + getCost(bits);
                    /* checkBestMVP (search.cpp:2702-2713) */
                    const int diffBits = (int)is_bitcost(outmv, a[!mvpIdx]) - (int)is_bitcost(outmv, a[mvpIdx]);
                    if (diffBits < 0)
                    {
                        mvpIdx = !mvpIdx;
                        const uint32_t orig = bits;
                        bits = orig + diffBits;
                        cc = (cc - getCost(orig)) + getCost(bits);
                    }
                    if (cc < w.best[list].cost) w.best[list] = MeBest{ outmv, a[mvpIdx], mvpIdx, ref, (int)bits, mvCost, cc };
                }
            if (detail && pidx == 0)
            {
                x265amd_me_detail& d = detail[w.cu];
                memset(&d, 0, sizeof(d));
                for (int l = 0; l < 2; l++)
                {
                    const MeBest& b = w.best[l];
                    d.cost[l] = b.cost; d.ref[l] = (int8_t)b.ref;
                    if (b.ref < 0) continue;
                    d.mv[l][0] = (int16_t)b.mv.x; d.mv[l][1] = (int16_t)b.mv.y; d.mvp[l][0] = (int16_t)b.mvp.x; d.mvp[l][1] = (int16_t)b.mvp.y;
                    d.mvp_idx[l] = (uint8_t)b.mvpIdx; d.bits[l] = (uint32_t)b.bits; d.mv_cost[l] = b.mvCost;
                    for (int k = 0; k < 2; k++) { d.amvp[l][k][0] = w.amvp[l][b.ref][k][0]; d.amvp[l][k][1] = w.amvp[l][b.ref][k][1]; }
                }
                for (int k = 0; k < 3; k++) d.list_sel_bits[k] = w.listSelBits[k];
            }
            w.bidirJob[0] = w.bidirJob[1] = -1;
            if (isB && !(c.log2_size == 3 && c.part_size != 0) && c.part_size != 0 && w.best[0].cost != 0xFFFFFFFFu && w.best[1].cost != 0xFFFFFFFFu)
            {
                const int p0 = S->ref_pic[0][w.best[0].ref], p1 = S->ref_pic[1][w.best[1].ref];
                w.bidirJob[0] = (int)bj.size();
                bj.push_back(weigh(mcJob(w.g, c.x, c.y, 0, p0, w.best[0].mv, p1, w.best[1].mv, chromaSatd ? 3 : (1 | 16), 2, chromaSatd)));
                bool tryZero = w.best[0].mv.x || w.best[0].mv.y || w.best[1].mv.x || w.best[1].mv.y;
                if (tryZero)
                {
                    Mv mn, mx;
                    searchRange(Mv{ 0, 0 }, I->pic_width > I->pic_height ? I->pic_width : I->pic_height, c.x, c.y, mn, mx);
                    mx.y += 2;
                    mn.x <<= 2; mn.y <<= 2; mx.x <<= 2; mx.y <<= 2;
                    for (int l = 0; l < 2; l++)
                        tryZero &= w.best[l].mvp.x >= mn.x && w.best[l].mvp.x <= mx.x && w.best[l].mvp.y >= mn.y && w.best[l].mvp.y <= mx.y;
                }
                if (tryZero)
                {
                    w.bidirJob[1] = (int)bj.size();
                    bj.push_back(weigh(mcJob(w.g, c.x, c.y, 0, p0, Mv{ 0, 0 }, p1, Mv{ 0, 0 }, chromaSatd ? 3 : (1 | 16), 2, chromaSatd)));
                }
            }
        }
        std::vector<uint32_t> bcost;
        if (runCost(bj, bcost)) return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: bi-prediction cost launch failed");

        /* ---- step 4: the choice ---- */
        for (PuWork& w : W)
        {
            const x265amd_inter_cu& c = cus[w.cu];
            MeBest bidir[2] = { w.best[0], w.best[1] };
            uint32_t bidirCost = 0xFFFFFFFFu; int bidirBits = 0;
            if (w.bidirJob[0] >= 0)
            {
                const uint32_t satd = bcost[2 * w.bidirJob[0]] + bcost[2 * w.bidirJob[0] + 1];
                bidirBits = w.best[0].bits + w.best[1].bits + (int)w.listSelBits[2] - (int)(w.listSelBits[0] + w.listSelBits[1]);
                bidirCost = satd + getCost((uint32_t)bidirBits);
                if (w.bidirJob[1] >= 0)
                {
                    const uint32_t satd0 = bcost[2 * w.bidirJob[1]] + bcost[2 * w.bidirJob[1] + 1];
                    const Mv zero{ 0, 0 };
                    Mv mvp0 = w.best[0].mvp, mvp1 = w.best[1].mvp;
                    int idx0 = w.best[0].mvpIdx, idx1 = w.best[1].mvpIdx;
                    uint32_t bits0 = (uint32_t)w.best[0].bits - is_bitcost(w.best[0].mv, mvp0) + is_bitcost(zero, mvp0);
                    uint32_t bits1 = (uint32_t)w.best[1].bits - is_bitcost(w.best[1].mv, mvp1) + is_bitcost(zero, mvp1);
                    uint32_t cc = satd0 + getCost(bits0) + getCost(bits1);
                    for (int l = 0; l < 2; l++)
                    {
                        const int16_t (*a)[2] = w.amvp[l][w.best[l].ref];
                        int& idx = l ? idx1 : idx0; uint32_t& bits = l ? bits1 : bits0; Mv& mvp = l ? mvp1 : mvp0;
                        const Mv am[2] = { Mv{ a[0][0], a[0][1] }, Mv{ a[1][0], a[1][1] } };
                        const int diffBits = (int)is_bitcost(zero, am[!idx]) - (int)is_bitcost(zero, am[idx]);
                        if (diffBits < 0)
                        {
                            idx = !idx;
                            const uint32_t orig = bits;
                            bits = orig + diffBits;
                            cc = (cc - getCost(orig)) + getCost(bits);
                        }
                        mvp = am[idx];
                    }
                    if (cc < bidirCost)
                    {
                        bidir[0].mv = bidir[1].mv = zero; bidir[0].mvp = mvp0; bidir[1].mvp = mvp1; bidir[0].mvpIdx = idx0; bidir[1].mvpIdx = idx1;
                        bidirCost = cc;
                        bidirBits = (int)(bits0 + bits1) + (int)w.listSelBits[2] - (int)(w.listSelBits[0] + w.listSelBits[1]);
                    }
                }
            }
            x265amd_pu_result& r = out[2 * w.cu + pidx];
            memset(&r, 0, sizeof(r));
            int pic0 = -1, pic1 = -1; Mv mv0{ 0, 0 }, mv1{ 0, 0 };
            if (w.mrgCost < bidirCost && w.mrgCost < w.best[0].cost && w.mrgCost < w.best[1].cost)
            {
                const x265amd_merge_cand& m = w.mcand[w.mrgIdx];
                r.merge_flag = 1; r.inter_dir = m.dir; r.mvp_idx[0] = (uint8_t)w.mrgIdx;
                for (int l = 0; l < 2; l++) { r.mv[l][0] = m.mv[l][0]; r.mv[l][1] = m.mv[l][1]; r.ref_idx[l] = m.ref_idx[l]; }
                cs[w.cu].totalBits += w.mrgBits;
            }
            else if (bidirCost < w.best[0].cost && bidirCost < w.best[1].cost)
            {
                cs[w.cu].lastMode = 2;
                r.inter_dir = 3;
                for (int l = 0; l < 2; l++)
                {
                    r.mv[l][0] = (int16_t)bidir[l].mv.x; r.mv[l][1] = (int16_t)bidir[l].mv.y; r.ref_idx[l] = (int8_t)w.best[l].ref;
                    r.mvd[l][0] = (int16_t)(bidir[l].mv.x - bidir[l].mvp.x); r.mvd[l][1] = (int16_t)(bidir[l].mv.y - bidir[l].mvp.y); r.mvp_idx[l] = (uint8_t)bidir[l].mvpIdx;
                }
                cs[w.cu].totalBits += bidirBits;
            }
            else
            {
                const int l = w.best[0].cost <= w.best[1].cost ? 0 : 1;
                cs[w.cu].lastMode = l;
                r.inter_dir = (uint8_t)(1 << l);
                r.mv[l][0] = (int16_t)w.best[l].mv.x; r.mv[l][1] = (int16_t)w.best[l].mv.y; r.ref_idx[l] = (int8_t)w.best[l].ref; r.ref_idx[!l] = -1;
                r.mvd[l][0] = (int16_t)(w.best[l].mv.x - w.best[l].mvp.x); r.mvd[l][1] = (int16_t)(w.best[l].mv.y - w.best[l].mvp.y); r.mvp_idx[l] = (uint8_t)w.best[l].mvpIdx;
                cs[w.cu].totalBits += w.best[l].bits;
            }
            if (r.ref_idx[0] >= 0 && (r.inter_dir & 1 || r.merge_flag)) { pic0 = S->ref_pic[0][r.ref_idx[0]]; mv0 = Mv{ r.mv[0][0], r.mv[0][1] }; }
            if (r.ref_idx[1] >= 0 && (r.inter_dir & 2 || r.merge_flag)) { pic1 = S->ref_pic[1][r.ref_idx[1]]; mv1 = Mv{ r.mv[1][0], r.mv[1][1] }; }
            if (!(r.inter_dir & 1)) pic0 = -1;
            if (!(r.inter_dir & 2)) pic1 = -1;
            /* final prediction: motionCompensation(cu, pu, *predYuv, true, bChromaMC) into the CU's prediction tile */
            x265amd_mc_job f = weigh(mcJob(w.g, c.x, c.y, !isB, pic0, mv0, pic1, mv1, (S->chroma_mc != 0) ? 3 : 1, 0, 0));
            const uint64_t base = d_pred + (size_t)w.cu * pred_bytes_per_cu;
            f.dst_y = base + ((size_t)(w.g.y - c.y) * 64 + (w.g.x - c.x)) * isz;
            f.dst_u = base + (64 * 64 + (size_t)((w.g.y - c.y) / 2) * 32 + (w.g.x - c.x) / 2) * isz;
            f.dst_v = f.dst_u + 32 * 32 * isz;
            f.dst_stride = 64; f.dst_cstride = 32;
            finalMc.push_back(f);
            bits_out[w.cu] = cs[w.cu].totalBits;
        }
    }
    /* ---- final predictions ---- */
    {
        XaMapped dJ;
        if (dJ.alloc(finalMc.size() * sizeof(x265amd_mc_job)) != hipSuccess) return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: out of device memory");
        if (xa_ref_guard_mc(finalMc.data(), (int)finalMc.size())) return xa_fail(X265AMD_EHIP, "x265amd_pred_inter_search: a reference picture failed");
        memcpy(dJ.p, finalMc.data(), finalMc.size() * sizeof(x265amd_mc_job));
        int rc = x265amd_motion_compensation(st, (const uint64_t*)dPlanes.p, stride, cstride, I->pic_width, I->pic_height, (const x265amd_mc_job*)dJ.p, (int)finalMc.size());
        if (rc != X265AMD_OK) return rc;
        /* lazy_sync: the caller's next command on this queue waits for the predictions (the decisions are final already).  What the pending command reads --
         * its jobs, the plane table -- stays out of the pool until the queue is next synchronised. */
        if (S->lazy_sync && xa_q_free_mapped_later(st, dJ.p) && xa_q_free_mapped_later(st, dPlanes.p)) { dJ.p = nullptr; dPlanes.p = nullptr; }
        else XA_HIP_CHECK(xa_stream_sync(st));
    }
    return X265AMD_OK;
}
