/* Layer 3: motion estimation on device-resident pictures (include/x265amd.h, `x265amd_me_*`).
 *
 * Device restatement of the reference's MotionEstimate::motionEstimate() (source/encoder/motion.cpp:764-1594) with
 * its DIA, HEX and STAR integer searches (StarPatternSearch, motion.cpp:387-629), the sub-pel refinement driven by
 * the `workload` table (motion.cpp:48-58), subpelCompare's luma path (motion.cpp:1596-1623) and BitCost's MV cost
 * tables (source/encoder/bitcost.cpp:30-109).  Integer arithmetic throughout: results (quarter-pel MV and cost) are
 * bit-exact with the reference for the same arguments.
 *
 * Mapping to gfx950: one workgroup per job group; the group's reference window (<= max_win_w x max_win_h samples)
 * and its 64x64 source tile are staged in LDS with coalesced row reads; each 64-lane wavefront pulls jobs from an
 * LDS counter and runs the (data dependent, serial) search for its job with all candidates' SAD/SATD sums reduced
 * across the wavefront.  8-bit SADs use v_sad_u8 on dwords assembled from the LDS window with v_alignbyte.
 */
#include <math.h>
#include <algorithm>
#include <vector>
#include "x265amd_dev.h"
#include "x265amd_host.h"

#include "me_dev.h"
#include "xa_queue.h"

template<bool STAR> __global__ __launch_bounds__(64 * ME_WAVES, ME_MIN_WAVES_PER_EU) void k_me_search(MeParams p) { block_me_search<STAR>(p, blockIdx.x, threadIdx.x, 64 * ME_WAVES); }
__global__ __launch_bounds__(64 * ME_WAVES) void k_me_deferred(MeParams p) { block_me_deferred(p, blockIdx.x, threadIdx.x, 64 * ME_WAVES); }

struct x265amd_me_ctx
{
    uint16_t* d_tables = nullptr;           /* ME_QP_COUNT tables of ME_TBL_LEN entries */
    std::vector<uint16_t> h_tables;
    float* d_bitsize = nullptr;             /* BitCost::s_bitsizes (bitcost.cpp:95-109) for |d| = 0 .. ME_TBL_HALF: the fused search command prices vectors with it (BitCost::bitcost) */
};

/* =========================================================================================================
 * host side
 * ======================================================================================================= */
/* x265_lambda_tab (constants.cpp:34-150): 2^(qp/6-2) * 2^(depth-8), tabulated to four decimals in the reference */
static double me_lambda(int qp)
{
    double v = pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265AMD_DEPTH - 8));
    return floor(v * 10000.0 + 0.5) / 10000.0;
}

/* BitCost::CalculateLogs + setQP (bitcost.cpp:30-58, :95-109) with the arithmetic of the reference build: the C `log`
 * is the double function applied to the float-converted argument, the scale 2/ln2 is a float constant, the sum is
 * rounded to float once, and the cost is double(bits) * lambda + 0.5 truncated to uint16 (capped at 2^15 - 1). */
static void me_build_table(int qp, uint16_t* t /* ME_TBL_LEN */)
{
    const double lambda = me_lambda(qp);
    const double log2_2 = (double)(float)(2.0 / log(2.0));
    uint16_t* c = t + ME_TBL_HALF;
    for (int i = 0; i <= ME_TBL_HALF; i++)
    {
        float bits = i ? (float)(log((double)(float)(i + 1)) * log2_2 + (double)1.718f) : 0.718f;
        double v = (double)bits * lambda + 0.5;
        if (v > 32767.0) v = 32767.0;
        c[i] = c[-i] = (uint16_t)v;
    }
}

extern "C" x265amd_me_ctx* x265amd_me_open(void)
{
    x265amd_me_ctx* ctx = new x265amd_me_ctx;
    ctx->h_tables.resize((size_t)ME_QP_COUNT * ME_TBL_LEN);
    for (int qp = 0; qp < ME_QP_COUNT; qp++)
        me_build_table(qp, ctx->h_tables.data() + (size_t)qp * ME_TBL_LEN);
    size_t bytes = ctx->h_tables.size() * sizeof(uint16_t);
    if (hipMalloc((void**)&ctx->d_tables, bytes) != hipSuccess ||
        hipMemcpy(ctx->d_tables, ctx->h_tables.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    {
        xa_fail(X265AMD_EHIP, "x265amd_me_open: cannot place the MV cost tables on the device");
        delete ctx;
        return nullptr;
    }
    {
        /* evaluated as the reference build does (see is_bitsize, inter_common.h) */
        std::vector<float> bs((size_t)ME_TBL_HALF + 1);
        const double log2_2 = (double)(float)(2.0 / log(2.0));
        bs[0] = 0.718f;
        for (int i = 1; i <= ME_TBL_HALF; i++) bs[i] = (float)(log((double)(float)(i + 1)) * log2_2 + (double)1.718f);
        if (hipMalloc((void**)&ctx->d_bitsize, bs.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(ctx->d_bitsize, bs.data(), bs.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        {
            xa_fail(X265AMD_EHIP, "x265amd_me_open: cannot place the vector bit-size table on the device");
            x265amd_me_close(ctx);
            return nullptr;
        }
    }
    return ctx;
}

extern "C" void x265amd_me_close(x265amd_me_ctx* ctx)
{
    if (!ctx) return;
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_bitsize) (void)hipFree(ctx->d_bitsize);
    delete ctx;
}

/* device address of the centre (MVD 0) of the MV cost table of `qp` (for the lookahead's motion search, csrc/lowres_kernels.hip) */
const uint16_t* xa_me_device_mvcost(x265amd_me_ctx* ctx, int qp)
{
    if (!ctx || qp < 0 || qp >= ME_QP_COUNT) return nullptr;
    return ctx->d_tables + (size_t)qp * ME_TBL_LEN + ME_TBL_HALF;
}

const float* xa_me_device_bitsize(x265amd_me_ctx* ctx) { return ctx ? ctx->d_bitsize : nullptr; }
const uint16_t* xa_me_device_tables(x265amd_me_ctx* ctx) { return ctx ? ctx->d_tables : nullptr; }

extern "C" const uint16_t* x265amd_me_host_mvcost(x265amd_me_ctx* ctx, int qp)
{
    if (!ctx || qp < 0 || qp >= ME_QP_COUNT) return nullptr;
    return ctx->h_tables.data() + (size_t)qp * ME_TBL_LEN;
}

extern "C" int x265amd_me_plan(const x265amd_me_job* jobs, int n, int ref, int max_win_w, int max_win_h, x265amd_me_group* groups, int32_t* order)
{
    if (!jobs || !groups || !order || n < 0 || max_win_w < 80 || max_win_h < 80 || (max_win_w & 3))
        return xa_fail(X265AMD_EINVAL, "x265amd_me_plan: bad arguments");
    std::vector<int> idx(n);
    for (int i = 0; i < n; i++)
    {
        idx[i] = i;
        if ((jobs[i].x & 63) + jobs[i].w > 64 || (jobs[i].y & 63) + jobs[i].h > 64 || (jobs[i].x & 3) || (jobs[i].w & 3) || (jobs[i].h & 3) ||
            jobs[i].num_cand > X265AMD_ME_MAX_CAND || jobs[i].subme > 7 || (jobs[i].method & 0x7f) > X265AMD_ME_FULL || jobs[i].qp >= ME_QP_COUNT)
            return xa_fail(X265AMD_EINVAL, "x265amd_me_plan: job outside the supported domain (PU must lie inside one 64x64 CTU tile, x%4==0)");
    }
    auto key = [&](int i) { return ((int64_t)(jobs[i].y >> 6) << 32) | (uint32_t)(jobs[i].x >> 6); };
    /* largest PUs first inside a tile so the long searches start early */
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
        int64_t ka = key(a), kb = key(b);
        if (ka != kb) return ka < kb;
        return jobs[a].w * jobs[a].h > jobs[b].w * jobs[b].h; });
    int ng = 0;
    for (int i = 0; i < n;)
    {
        int e = i;
        int x0 = 1 << 30, y0 = 1 << 30, x1 = -(1 << 30), y1 = -(1 << 30);
        while (e < n && key(idx[e]) == key(idx[i]))
        {
            const x265amd_me_job& j = jobs[idx[e]];
            /* search area of the job incl. the 8-tap margins and the +-2 the hexagon may step outside */
            x0 = std::min(x0, j.x + j.mvmin[0] - 6); y0 = std::min(y0, j.y + j.mvmin[1] - 6);
            x1 = std::max(x1, j.x + j.w + j.mvmax[0] + 10); y1 = std::max(y1, j.y + j.h + j.mvmax[1] + 7);
            order[e] = idx[e];
            e++;
        }
        x265amd_me_group& g = groups[ng++];
        g.first_job = i; g.num_jobs = e - i; g.ref = ref;
        g.fenc_x = (int16_t)((jobs[idx[i]].x >> 6) << 6); g.fenc_y = (int16_t)((jobs[idx[i]].y >> 6) << 6);
        int w = (x1 - x0 + 3) & ~3, h = y1 - y0;
        if (w > max_win_w) { x0 += (w - max_win_w) / 2; w = max_win_w; }
        if (h > max_win_h) { y0 += (h - max_win_h) / 2; h = max_win_h; }
        g.win_x = (int16_t)x0; g.win_y = (int16_t)y0; g.win_w = (int16_t)w; g.win_h = (int16_t)h;
        i = e;
    }
    return ng;
}

extern "C" int x265amd_me_search(x265amd_me_ctx* ctx, void* stream, const x265amd_pixel* d_fenc, const uint64_t* d_refs, intptr_t stride,
                                 const x265amd_me_group* d_groups, int num_groups, const x265amd_me_job* d_jobs, x265amd_me_result* d_out,
                                 int max_win_w, int max_win_h, int flags, const uint64_t* d_chroma, intptr_t cstride)
{
    if (!ctx || !d_fenc || !d_refs || !d_groups || !d_jobs || !d_out || num_groups < 0 || (max_win_w & 3))
        return xa_fail(X265AMD_EINVAL, "x265amd_me_search: bad arguments");
    if (num_groups == 0) return X265AMD_OK;
    const bool queue = xa_is_queue(stream);
    const int waves = queue ? XA_SERVER_WAVES : ME_WAVES;
    size_t lds = ((size_t)max_win_w * max_win_h + 16 + 64 * 64) * sizeof(pixel) + 16 + waves * sizeof(MeState);
    if (lds > (queue ? (size_t)XA_SERVER_LDS : (size_t)160 * 1024)) return xa_fail(X265AMD_EINVAL, "x265amd_me_search: window does not fit the LDS");
    const bool star = (flags & (X265AMD_ME_FLAG_STAR | X265AMD_ME_FLAG_CHROMA)) != 0;   /* the variant with star search + chroma SATD */
    if ((flags & X265AMD_ME_FLAG_CHROMA) && !d_chroma) return xa_fail(X265AMD_EINVAL, "x265amd_me_search: X265AMD_ME_FLAG_CHROMA without chroma planes");
    static thread_local size_t configured[2] = { 0, 0 };
    if (!queue && lds > configured[star])
    {
        const void* fn = star ? (const void*)k_me_search<true> : (const void*)k_me_search<false>;
        XA_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[star] = lds;
    }
    MeParams p;
    p.fenc = d_fenc; p.refs = d_refs; p.stride = (int)stride; p.groups = d_groups; p.jobs = d_jobs; p.out = d_out;
    p.chroma = d_chroma; p.cstride = (int)cstride;
    p.tables = ctx->d_tables; p.maxWinW = max_win_w; p.maxWinH = max_win_h;
    hipError_t e;
    if (star)
        XA_LAUNCH(e, stream, XA_OP_ME_SEARCH_STAR, num_groups, p, k_me_search<true>, dim3(num_groups), dim3(64 * ME_WAVES), lds, p);
    else
        XA_LAUNCH(e, stream, XA_OP_ME_SEARCH, num_groups, p, k_me_search<false>, dim3(num_groups), dim3(64 * ME_WAVES), lds, p);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    /* second pass: jobs whose search left the staged window (exactness is never traded for the fast path) */
    size_t lds2 = (size_t)64 * 64 * sizeof(pixel) + 16 + ME_WAVES * sizeof(MeState);
    XA_LAUNCH(e, stream, XA_OP_ME_DEFERRED, num_groups, p, k_me_deferred, dim3(num_groups), dim3(64 * ME_WAVES), lds2, p);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
