/* Layer-2 kernels: batched job lists, one 64-lane wavefront per job (include/x265amd.h, `x265amd_run_jobs`).
 *
 * Each family kernel is the device implementation of one group of `EncoderPrimitives` slots
 * (reference: source/common/primitives.h:239-433); the citations on the device routines name the reference C
 * primitive whose results are reproduced bit for bit.  gfx950 only.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

#define WAVES_PER_BLOCK 4

template<class T> XA_DEV T* P(uint64_t addr) { return reinterpret_cast<T*>(addr); }

/* =========================================================================================================
 * family 0: distortion (pixel.cpp)
 * ======================================================================================================= */
XA_DEV uint64_t wave_sse_ss(const int16_t* a, int sa, const int16_t* b, int sb, int size, int lane, bool self)
{
    uint64_t sum = 0;
    int n = size * size, sh = 31 - __clz(size);
    for (int i = lane; i < n; i += XA_WAVE)
    {
        int y = i >> sh, x = i & (size - 1);
        int t = self ? (int)a[y * sa + x] : (int)a[y * sa + x] - (int)b[y * sb + x];
        sum += (uint64_t)(uint32_t)(t * t);
    }
    sum = xa_wave_sum(sum);
#if XA_DEPTH <= 8
    sum = (uint32_t)sum;
#endif
    return sum;
}

__global__ __launch_bounds__(256) void k_distortion(const x265amd_job* jobs, int n)
{
    int lane = xa_lane();
    int ji = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ji >= n) return;
    const x265amd_job j = jobs[ji];
    const pixel* a = P<const pixel>(j.a);
    const pixel* b = P<const pixel>(j.b);
    uint64_t res = 0;
    switch (j.op)
    {
    case X265AMD_OP_SAD:
        res = (uint64_t)(int64_t)xa_wave_sad(a, j.sa, b, j.sb, xa_tbl.puW[j.size], xa_tbl.puH[j.size], lane);
        break;
    case X265AMD_OP_SAD_X3:
    case X265AMD_OP_SAD_X4:
    {
        int w = xa_tbl.puW[j.size], h = xa_tbl.puH[j.size];
        int nc = j.op == X265AMD_OP_SAD_X3 ? 3 : 4;
        const uint64_t cand[4] = { j.b, j.c, j.e[0], j.e[1] };
        int32_t* out = P<int32_t>(j.d);
        for (int c = 0; c < nc; c++)
        {
            int s = xa_wave_sad(a, XA_FENC_STRIDE, P<const pixel>(cand[c]), j.sb, w, h, lane);
            if (lane == 0) out[c] = s;
        }
        return;
    }
    case X265AMD_OP_SATD:
        res = (uint64_t)(int64_t)xa_wave_satd(a, j.sa, b, j.sb, xa_tbl.puW[j.size], xa_tbl.puH[j.size], lane);
        break;
    case X265AMD_OP_CHROMA_SATD:    /* 4:2:0: half the luma partition (pixel.cpp:1205-1229) */
        res = (uint64_t)(int64_t)xa_wave_satd(a, j.sa, b, j.sb, xa_tbl.puW[j.size] >> 1, xa_tbl.puH[j.size] >> 1, lane);
        break;
    case X265AMD_OP_SA8D:
        res = (uint64_t)(int64_t)xa_wave_sa8d(a, j.sa, b, j.sb, 4 << j.size, lane);
        break;
    case X265AMD_OP_CHROMA_SA8D:    /* pixel.cpp:1243-1246 */
        res = (uint64_t)(int64_t)xa_wave_sa8d(a, j.sa, b, j.sb, 2 << j.size, lane);
        break;
    case X265AMD_OP_SSE_PP:
        res = wave_sse_pp(a, j.sa, b, j.sb, 4 << j.size, lane);
        break;
    case X265AMD_OP_SSE_SS:
        res = wave_sse_ss(P<const int16_t>(j.a), j.sa, P<const int16_t>(j.b), j.sb, 4 << j.size, lane, false);
        break;
    case X265AMD_OP_SSD_S:
        res = wave_sse_ss(P<const int16_t>(j.a), j.sa, nullptr, 0, 4 << j.size, lane, true);
        break;
    case X265AMD_OP_PSY_COST_PP:
        res = (uint64_t)(int64_t)wave_psy_cost(a, j.sa, b, j.sb, j.size, lane);
        break;
    case X265AMD_OP_VAR:            /* pixel.cpp:715-733 */
    {
        int size = 4 << j.size, nn = size * size, sh = j.size + 2;
        uint32_t sum = 0, sqr = 0;
        for (int i = lane; i < nn; i += XA_WAVE)
        {
            uint32_t v = a[(i >> sh) * j.sa + (i & (size - 1))];
            sum += v; sqr += v * v;
        }
        sum = xa_wave_sum(sum); sqr = xa_wave_sum(sqr);
        res = sum + ((uint64_t)sqr << 32);
        break;
    }
    default:
        return;
    }
    if (lane == 0) *P<uint64_t>(j.d) = res;
}

/* =========================================================================================================
 * family 1: pixel / residual block ops (pixel.cpp, dct.cpp:714-742)
 * ======================================================================================================= */
__global__ __launch_bounds__(256) void k_pixel(const x265amd_job* jobs, int n)
{
    int lane = xa_lane();
    int ji = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ji >= n) return;
    const x265amd_job j = jobs[ji];
    switch (j.op)
    {
    case X265AMD_OP_SUB_PS:         /* pixel.cpp:832-844 */
    {
        int size = 4 << j.size, sh = j.size + 2;
        const pixel* s0 = P<const pixel>(j.a); const pixel* s1 = P<const pixel>(j.b); int16_t* d = P<int16_t>(j.d);
        for (int i = lane; i < size * size; i += XA_WAVE)
        {
            int y = i >> sh, x = i & (size - 1);
            d[y * j.sd + x] = (int16_t)((int)s0[y * j.sa + x] - (int)s1[y * j.sb + x]);
        }
        break;
    }
    case X265AMD_OP_ADD_PS:         /* pixel.cpp:846-858 */
    {
        int size = 4 << j.size, sh = j.size + 2;
        const pixel* s0 = P<const pixel>(j.a); const int16_t* s1 = P<const int16_t>(j.b); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < size * size; i += XA_WAVE)
        {
            int y = i >> sh, x = i & (size - 1);
            d[y * j.sd + x] = xa_clip_pixel((int)s0[y * j.sa + x] + (int)s1[y * j.sb + x]);
        }
        break;
    }
    case X265AMD_OP_PIXELAVG_PP:    /* pixel.cpp:544-556 */
    {
        int w = xa_tbl.puW[j.size], h = xa_tbl.puH[j.size];
        const pixel* s0 = P<const pixel>(j.a); const pixel* s1 = P<const pixel>(j.b); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            d[y * j.sd + x] = (pixel)(((int)s0[y * j.sa + x] + (int)s1[y * j.sb + x] + 1) >> 1);
        }
        break;
    }
    case X265AMD_OP_ADDAVG:         /* pixel.cpp:860-879; p[0]=1 -> 4:2:0 chroma (half partition) */
    {
        int w = xa_tbl.puW[j.size] >> j.p[0], h = xa_tbl.puH[j.size] >> j.p[0];
        const int shift = XA_IF_INTERNAL_PREC + 1 - XA_DEPTH;
        const int offset = (1 << (shift - 1)) + 2 * XA_IF_INTERNAL_OFFS;
        const int16_t* s0 = P<const int16_t>(j.a); const int16_t* s1 = P<const int16_t>(j.b); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            d[y * j.sd + x] = xa_clip_pixel(((int)s0[y * j.sa + x] + (int)s1[y * j.sb + x] + offset) >> shift);
        }
        break;
    }
    case X265AMD_OP_WEIGHT_PP:      /* pixel.cpp:519-542; p = width,height,w0,round,shift,offset */
    {
        int w = j.p[0], h = j.p[1];
        const int corr = XA_IF_INTERNAL_PREC - XA_DEPTH;
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int16_t val = (int16_t)(s[y * j.sa + x] << corr);
            d[y * j.sd + x] = xa_clip_pixel(((j.p[2] * val + j.p[3]) >> j.p[4]) + j.p[5]);
        }
        break;
    }
    case X265AMD_OP_WEIGHT_SP:      /* pixel.cpp:493-517 */
    {
        int w = j.p[0], h = j.p[1];
        const int16_t* s = P<const int16_t>(j.a); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            d[y * j.sd + x] = xa_clip_pixel(((j.p[2] * ((int)s[y * j.sa + x] + XA_IF_INTERNAL_OFFS) + j.p[3]) >> j.p[4]) + j.p[5]);
        }
        break;
    }
    case X265AMD_OP_SCALE2D_64TO32: /* pixel.cpp:583-600 */
    {
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < 32 * 32; i += XA_WAVE)
        {
            int y = i >> 5, x = i & 31;
            const pixel* p = s + 2 * y * j.sa + 2 * x;
            d[i] = (pixel)((p[0] + p[1] + p[j.sa] + p[j.sa + 1] + 2) >> 2);
        }
        break;
    }
    case X265AMD_OP_SCALE1D_128TO64: /* pixel.cpp:558-581 */
    {
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        d[lane] = (pixel)((s[2 * lane] + s[2 * lane + 1] + 1) >> 1);
        d[64 + lane] = (pixel)((s[128 + 2 * lane] + s[128 + 2 * lane + 1] + 1) >> 1);
        break;
    }
    case X265AMD_OP_TRANSPOSE:      /* pixel.cpp:485-491 */
    {
        int size = 4 << j.size, sh = j.size + 2;
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        for (int i = lane; i < size * size; i += XA_WAVE)
        {
            int k = i >> sh, l = i & (size - 1);
            d[i] = s[l * j.sa + k];
        }
        break;
    }
    case X265AMD_OP_CPY2DTO1D_SHL:  /* pixel.cpp:400-470; p[0]=shift */
    case X265AMD_OP_CPY2DTO1D_SHR:
    case X265AMD_OP_CPY1DTO2D_SHL:
    case X265AMD_OP_CPY1DTO2D_SHR:
    {
        int size = 4 << j.size, sh = j.size + 2, shift = j.p[0];
        bool to1d = j.op == X265AMD_OP_CPY2DTO1D_SHL || j.op == X265AMD_OP_CPY2DTO1D_SHR;
        bool shl = j.op == X265AMD_OP_CPY2DTO1D_SHL || j.op == X265AMD_OP_CPY1DTO2D_SHL;
        const int16_t* s = P<const int16_t>(j.a); int16_t* d = P<int16_t>(j.d);
        int16_t round = (int16_t)(shl ? 0 : 1 << (shift - 1));
        for (int i = lane; i < size * size; i += XA_WAVE)
        {
            int y = i >> sh, x = i & (size - 1);
            int v = to1d ? s[y * j.sa + x] : s[i];
            int16_t r = shl ? (int16_t)(v << shift) : (int16_t)((v + round) >> shift);
            if (to1d) d[i] = r; else d[y * j.sd + x] = r;
        }
        break;
    }
    case X265AMD_OP_COPY_CNT:       /* dct.cpp:729-742; numSig -> e[0] */
    {
        int size = 4 << j.size, sh = j.size + 2, cnt = 0;
        const int16_t* s = P<const int16_t>(j.a); int16_t* d = P<int16_t>(j.d);
        for (int i = lane; i < size * size; i += XA_WAVE)
        {
            int16_t v = s[(i >> sh) * j.sa + (i & (size - 1))];
            d[i] = v; cnt += v != 0;
        }
        cnt = xa_wave_sum(cnt);
        if (lane == 0) *P<uint64_t>(j.e[0]) = (uint64_t)cnt;
        break;
    }
    case X265AMD_OP_COUNT_NONZERO:  /* dct.cpp:714-727 */
    {
        int size = 4 << j.size, cnt = 0;
        const int16_t* s = P<const int16_t>(j.a);
        for (int i = lane; i < size * size; i += XA_WAVE) cnt += s[i] != 0;
        cnt = xa_wave_sum(cnt);
        if (lane == 0) *P<uint64_t>(j.d) = (uint64_t)cnt;
        break;
    }
    default:
        break;
    }
}

/* =========================================================================================================
 * family 2: transforms + quantisation (dct.cpp).  The reference's partial butterflies are exact integer
 * factorisations of the matrix product, so out[k][j] = (sum_n T[k][n] in[j][n] + add) >> shift is bit-exact.
 * ======================================================================================================= */
__global__ __launch_bounds__(256) void k_transform(const x265amd_job* jobs, int n)
{
    __shared__ int16_t lds[WAVES_PER_BLOCK][2][32 * 32];
    int lane = xa_lane(), wv = threadIdx.x >> 6;
    int ji = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (ji >= n) return;
    const x265amd_job j = jobs[ji];
    int16_t* blk = lds[wv][0];
    int16_t* tmp = lds[wv][1];
    switch (j.op)
    {
    case X265AMD_OP_DCT:            /* dct.cpp:459-525 */
    case X265AMD_OP_DST4:           /* dct.cpp:442-457 */
    {
        int cu = j.op == X265AMD_OP_DST4 ? 0 : j.size, log2N = cu + 2, N = 1 << log2N;
        const int16_t* T = j.op == X265AMD_OP_DST4 ? xa_tbl.dst4 : xa_tbl.dct[cu];
        const int16_t* s = P<const int16_t>(j.a);
        for (int i = lane; i < N * N; i += XA_WAVE)
            blk[i] = s[(i >> log2N) * j.sa + (i & (N - 1))];
        xa_wave_sync();
        wave_fwd_pass(T, log2N, blk, tmp, log2N - 1 + XA_DEPTH - 8, lane);
        xa_wave_sync();
        wave_fwd_pass(T, log2N, tmp, P<int16_t>(j.d), log2N + 6, lane);
        break;
    }
    case X265AMD_OP_IDCT:           /* dct.cpp:544-610 */
    case X265AMD_OP_IDST4:          /* dct.cpp:527-542 */
    {
        int cu = j.op == X265AMD_OP_IDST4 ? 0 : j.size, log2N = cu + 2, N = 1 << log2N;
        const int16_t* T = j.op == X265AMD_OP_IDST4 ? xa_tbl.dst4 : xa_tbl.dct[cu];
        const int16_t* s = P<const int16_t>(j.a);
        for (int i = lane; i < N * N; i += XA_WAVE)
            blk[i] = s[i];
        xa_wave_sync();
        wave_inv_pass(T, log2N, blk, tmp, N, 7, lane);
        xa_wave_sync();
        wave_inv_pass(T, log2N, tmp, P<int16_t>(j.d), j.sd, 12 - (XA_DEPTH - 8), lane);
        break;
    }
    case X265AMD_OP_QUANT:          /* dct.cpp:664-686; p = qBits, add, numCoeff; e[0]=deltaU e[1]=numSig */
    case X265AMD_OP_NQUANT:         /* dct.cpp:688-713 */
    {
        const int16_t* coef = P<const int16_t>(j.a); const int32_t* qc = P<const int32_t>(j.b);
        int16_t* q = P<int16_t>(j.d); int32_t* dU = P<int32_t>(j.e[0]);
        int qBits = j.p[0], add = j.p[1], num = j.p[2], qBits8 = qBits - 8, cnt = 0;
        bool full = j.op == X265AMD_OP_QUANT;
        for (int i = lane; i < num; i += XA_WAVE)
        {
            int level = coef[i];
            int sign = level < 0 ? -1 : 1;
            int tmplevel = abs(level) * qc[i];
            level = (tmplevel + add) >> qBits;
            if (full) dU[i] = (tmplevel - (level << qBits)) >> qBits8;
            cnt += level != 0;
            level *= sign;
            int c = xa_clip3(-32768, 32767, level);
            q[i] = (int16_t)(full ? c : abs(c));
        }
        cnt = xa_wave_sum(cnt);
        if (lane == 0) *P<uint64_t>(j.e[1]) = (uint64_t)cnt;
        break;
    }
    case X265AMD_OP_DEQUANT_NORMAL: /* dct.cpp:612-634; p = num, scale, shift */
    {
        const int16_t* q = P<const int16_t>(j.a); int16_t* d = P<int16_t>(j.d);
        int add = 1 << (j.p[2] - 1);
        for (int i = lane; i < j.p[0]; i += XA_WAVE)
            d[i] = (int16_t)xa_clip3(-32768, 32767, (q[i] * j.p[1] + add) >> j.p[2]);
        break;
    }
    case X265AMD_OP_DEQUANT_SCALING: /* dct.cpp:636-662; p = num, per, shift */
    {
        const int16_t* q = P<const int16_t>(j.a); const int32_t* dq = P<const int32_t>(j.b); int16_t* d = P<int16_t>(j.d);
        int per = j.p[1], shift = j.p[2] + 4;
        for (int i = lane; i < j.p[0]; i += XA_WAVE)
        {
            int v;
            if (shift > per)
                v = ((q[i] * dq[i]) + (1 << (shift - per - 1))) >> (shift - per);
            else
                v = (int)((unsigned)xa_clip3(-32768, 32767, q[i] * dq[i]) << (per - shift));
            d[i] = (int16_t)xa_clip3(-32768, 32767, v);
        }
        break;
    }
    default:
        break;
    }
}

/* =========================================================================================================
 * family 3: intra prediction (intrapred.cpp).  Neighbour layout (predict.cpp:600-877): s[0] top-left,
 * s[1..2N] above+above-right, s[2N+1..4N] left+below-left.
 * ======================================================================================================= */
__global__ __launch_bounds__(256) void k_intra(const x265amd_job* jobs, int n)
{
    __shared__ pixel lds[WAVES_PER_BLOCK][3][136];
    int lane = xa_lane(), wv = threadIdx.x >> 6;
    int ji = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (ji >= n) return;
    const x265amd_job j = jobs[ji];
    int N = 4 << j.size;
    pixel* nb = lds[wv][0];
    pixel* nb2 = lds[wv][1];
    pixel* sw = lds[wv][2];
    const pixel* src = P<const pixel>(j.a);
    for (int i = lane; i <= 4 * N; i += XA_WAVE) nb[i] = src[i];
    xa_wave_sync();
    switch (j.op)
    {
    case X265AMD_OP_INTRA_PRED:     /* p[0]=mode p[1]=bFilter */
        wave_intra_pred(nb, sw, j.size, j.p[0], j.p[1], P<pixel>(j.d), j.sd, false, lane);
        break;
    case X265AMD_OP_INTRA_FILTER:
        wave_intra_filter(nb, P<pixel>(j.d), N, lane);
        break;
    case X265AMD_OP_INTRA_ALLANGS:  /* a=refPix b=filtPix p[0]=bLuma */
    {
        const pixel* fsrc = P<const pixel>(j.b);
        for (int i = lane; i <= 4 * N; i += XA_WAVE) nb2[i] = fsrc[i];
        xa_wave_sync();
        pixel* d = P<pixel>(j.d);
        for (int mode = 2; mode <= 34; mode++)
            wave_intra_pred((xa_intra_filter_flags(mode) & N) ? nb2 : nb, sw, j.size, mode, j.p[0], d + (mode - 2) * N * N, N, true, lane);
        break;
    }
    default:
        break;
    }
}

/* =========================================================================================================
 * family 4: interpolation (ipfilter.cpp).  p[0]=taps p[1]=width p[2]=height p[3]=coeffIdx p[4]=isRowExt|idxY
 * ======================================================================================================= */
XA_DEV const int16_t* ip_taps(int taps, int idx) { return taps == 8 ? xa_tbl.lumaFilter[idx] : xa_tbl.chromaFilter[idx]; }

template<class S> XA_DEV int ip_dot(const S* p, int step, const int16_t* c, int taps)
{
    int sum = 0;
    for (int t = 0; t < taps; t++) sum += (int)p[t * step] * c[t];
    return sum;
}

__global__ __launch_bounds__(256) void k_interp(const x265amd_job* jobs, int n)
{
    int lane = xa_lane();
    int ji = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (ji >= n) return;
    const x265amd_job j = jobs[ji];
    int taps = j.p[0], w = j.p[1], h = j.p[2];
    const int16_t* c = ip_taps(taps, j.p[3]);
    const int half = taps / 2 - 1;
    const int headRoom = XA_IF_INTERNAL_PREC - XA_DEPTH;
    switch (j.op)
    {
    case X265AMD_OP_IP_HPP:         /* ipfilter.cpp:79-120 */
    case X265AMD_OP_IP_VPP:         /* ipfilter.cpp:169-210 */
    {
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        int step = j.op == X265AMD_OP_IP_HPP ? 1 : j.sa;
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int sum = ip_dot(s + y * j.sa + x - half * step, step, c, taps);
            int16_t val = (int16_t)((sum + (1 << (XA_IF_FILTER_PREC - 1))) >> XA_IF_FILTER_PREC);
            d[y * j.sd + x] = xa_clip_pixel(val);
        }
        break;
    }
    case X265AMD_OP_IP_HPS:         /* ipfilter.cpp:122-167 */
    case X265AMD_OP_IP_VPS:         /* ipfilter.cpp:212-248 */
    {
        const pixel* s = P<const pixel>(j.a); int16_t* d = P<int16_t>(j.d);
        bool hz = j.op == X265AMD_OP_IP_HPS;
        int step = hz ? 1 : j.sa;
        int shift = XA_IF_FILTER_PREC - headRoom;
        int offset = (int)((unsigned)-XA_IF_INTERNAL_OFFS << shift);
        if (hz && j.p[4]) { s -= half * j.sa; h += taps - 1; }
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int sum = ip_dot(s + y * j.sa + x - half * step, step, c, taps);
            d[y * j.sd + x] = (int16_t)((sum + offset) >> shift);
        }
        break;
    }
    case X265AMD_OP_IP_VSP:         /* ipfilter.cpp:250-292 */
    {
        const int16_t* s = P<const int16_t>(j.a); pixel* d = P<pixel>(j.d);
        int shift = XA_IF_FILTER_PREC + headRoom;
        int offset = (1 << (shift - 1)) + (XA_IF_INTERNAL_OFFS << XA_IF_FILTER_PREC);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int sum = ip_dot(s + (y - half) * j.sa + x, j.sa, c, taps);
            int16_t val = (int16_t)((sum + offset) >> shift);
            d[y * j.sd + x] = xa_clip_pixel(val);
        }
        break;
    }
    case X265AMD_OP_IP_VSS:         /* ipfilter.cpp:294-324 */
    {
        const int16_t* s = P<const int16_t>(j.a); int16_t* d = P<int16_t>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int sum = ip_dot(s + (y - half) * j.sa + x, j.sa, c, taps);
            d[y * j.sd + x] = (int16_t)(sum >> XA_IF_FILTER_PREC);
        }
        break;
    }
    case X265AMD_OP_IP_HVPP:        /* ipfilter.cpp:370-378: hps(rowExt) -> int16 rows -> vsp; the int16 intermediate
                                       of each of the `taps` rows is recomputed per output sample */
    {
        const pixel* s = P<const pixel>(j.a); pixel* d = P<pixel>(j.d);
        const int16_t* cy = ip_taps(taps, j.p[4]);
        int shiftH = XA_IF_FILTER_PREC - headRoom;
        int offH = (int)((unsigned)-XA_IF_INTERNAL_OFFS << shiftH);
        int shiftV = XA_IF_FILTER_PREC + headRoom;
        int offV = (1 << (shiftV - 1)) + (XA_IF_INTERNAL_OFFS << XA_IF_FILTER_PREC);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            int sum = 0;
            for (int t = 0; t < taps; t++)
            {
                int hs = ip_dot(s + (y - half + t) * j.sa + x - half, 1, c, taps);
                sum += (int)(int16_t)((hs + offH) >> shiftH) * cy[t];
            }
            int16_t val = (int16_t)((sum + offV) >> shiftV);
            d[y * j.sd + x] = xa_clip_pixel(val);
        }
        break;
    }
    case X265AMD_OP_IP_P2S:         /* ipfilter.cpp:39-56 */
    {
        const pixel* s = P<const pixel>(j.a); int16_t* d = P<int16_t>(j.d);
        for (int i = lane; i < w * h; i += XA_WAVE)
        {
            int y = i / w, x = i - y * w;
            d[y * j.sd + x] = (int16_t)((int16_t)(s[y * j.sa + x] << headRoom) - (int16_t)XA_IF_INTERNAL_OFFS);
        }
        break;
    }
    default:
        break;
    }
}

/* =========================================================================================================
 * launcher
 * ======================================================================================================= */
extern "C" int x265amd_run_jobs(void* stream, const x265amd_job* d_jobs, int n, int family)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs || family < 0 || family > 4) return xa_fail(X265AMD_EINVAL, "x265amd_run_jobs: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), block(64 * WAVES_PER_BLOCK);
    switch (family)
    {
    case 0: hipLaunchKernelGGL(k_distortion, grid, block, 0, st, d_jobs, n); break;
    case 1: hipLaunchKernelGGL(k_pixel, grid, block, 0, st, d_jobs, n); break;
    case 2: hipLaunchKernelGGL(k_transform, grid, block, 0, st, d_jobs, n); break;
    case 3: hipLaunchKernelGGL(k_intra, grid, block, 0, st, d_jobs, n); break;
    case 4: hipLaunchKernelGGL(k_interp, grid, block, 0, st, d_jobs, n); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
