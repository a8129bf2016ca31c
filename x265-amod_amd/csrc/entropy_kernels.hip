/* Entropy-side arithmetic of the residual path on the GPU (include/x265amd.h: x265amd_entropy_reset, x265amd_est_bit,
 * x265amd_coeff_bits).
 *
 *  - context initialisation: Entropy::resetEntropy / sbacInit (reference: source/encoder/entropy.cpp:1300-1355), host arithmetic;
 *  - Entropy::estBit (entropy.cpp:2220-2390): the FIX15 bit tables RDOQ reads, one 64-lane wavefront per table;
 *  - Entropy::codeCoeffNxN in bit-counting mode (entropy.cpp:1828-2200, primitives costCoeffNxN / costC1C2Flag /
 *    costCoeffRemain, source/common/dct.cpp:838-993).  Every coded bin moves its context, so the bins of one TU are
 *    strictly ordered; the kernel runs one TU per lane with the lane's context set in LDS and gets its parallelism
 *    from the batch.
 * Context initialisation values / LPS transitions: ITU-T H.265 9.3.2.2 and table 9-46, in the reference's context order
 * (source/common/contexts.h:75-106); fractional-bit constants: the reference's table (entropy.cpp:2614-2625).
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"
#include <string.h>

#include "entropy_dev.h"
#include "xa_queue.h"

static const uint8_t h_ctxInit[3][X265AMD_CTX_COUNT] = {      /* [slice type: 0 B, 1 P, 2 I][context] */
{107,139,126,197,185,201,154,137,154,139,154,154,134,183,152,139,154,154,154,95,79,63,31,31,153,153,169,198,153,111,149,92,167,154,154,224,167,122,79,121,140,61,154,170,154,139,153,139,123,123,63,124,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,170,153,138,138,122,121,122,121,167,151,183,140,151,183,140,125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,154,196,167,167,154,152,167,182,182,134,149,136,153,121,136,122,169,208,166,167,154,152,167,182,107,167,91,107,107,167,168,153,160,139,139,154},
{107,139,126,197,185,201,110,122,154,139,154,154,149,154,152,139,154,154,154,95,79,63,31,31,153,153,140,198,153,111,149,107,167,154,154,124,138,94,79,121,140,61,154,155,154,139,153,139,123,123,63,153,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,170,153,123,123,107,121,107,121,167,151,183,140,151,183,140,125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,154,196,196,167,154,152,167,182,182,134,149,136,153,121,136,137,169,194,166,167,154,167,137,182,107,167,91,122,107,167,168,153,185,139,139,154},
{139,141,157,154,154,154,154,154,184,154,154,154,154,184,63,139,154,154,154,154,154,154,154,154,154,154,154,154,111,141,94,138,182,154,154,153,138,138,154,91,171,134,141,111,111,125,110,110,94,124,108,124,107,125,141,179,153,125,107,125,141,179,153,125,107,125,141,179,153,125,140,139,182,182,152,136,152,136,153,136,139,111,136,139,111,110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,140,92,137,138,140,152,138,139,153,74,149,92,139,107,122,152,140,179,166,182,140,227,122,197,138,153,136,167,152,152,154,153,200,139,139,154},
};

extern "C" void x265amd_entropy_reset(int sliceType, int qp, uint8_t* ctx)
{
    qp = qp < 0 ? 0 : (qp > 51 ? 51 : qp);
    memset(ctx, 0, X265AMD_CTX_STRIDE);
    for (int i = 0; i < X265AMD_CTX_COUNT; i++)
    {
        const int v = h_ctxInit[sliceType][i];
        const int slope = (v >> 4) * 5 - 45, offset = ((v & 15) << 3) - 16;
        int s = ((slope * qp) >> 4) + offset;
        s = s < 1 ? 1 : (s > 126 ? 126 : s);
        const int mps = s >= 64;
        ctx[i] = (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
    }
}

__global__ __launch_bounds__(64 * EST_WAVES) void k_est_bit(const x265amd_est_job* jobs, int n)
{
    const int ji = blockIdx.x * EST_WAVES + (threadIdx.x >> 6);
    if (ji >= n) return;
    wave_est_bit_job(jobs, ji, xa_lane());
}

/* =========================================================================================================
 * bits-only coefficient coding: one TU per lane
 * ======================================================================================================= */
#define CB_LANES 64
struct CbLds { uint8_t ctx[CB_LANES][X265AMD_CTX_STRIDE]; };

XA_DEV uint32_t cb_bin(uint8_t* st, uint32_t bin) { const uint8_t s = *st; *st = en_next(s, bin); return en_bits[s ^ bin]; }

/* the 16 sample offsets of a 4x4 group in scan order `type` (g_scan4x4, constants.cpp:364-400, by rule), packed 4 bits each */
XA_DEV uint32_t cb_in_cg(int type, int k)
{
    const uint64_t t = type == 1 ? 0xFEDCBA9876543210ULL : type == 2 ? 0xFB73EA62D951C840ULL : 0xFBE7AD369C258140ULL;
    return (uint32_t)((t >> (4 * k)) & 15);
}
XA_DEV uint32_t cb_sig_ctx_inc(int log2N, uint32_t pattern, uint32_t rr)
{
    if (log2N == 2) return (uint32_t)((0x8877886654325410ULL >> (4 * rr)) & 15);
    const uint64_t t = pattern == 0 ? 0x0000000100110112ULL : pattern == 1 ? 0x0000000011112222ULL : pattern == 2 ? 0x0012001200120012ULL : 0x2222222222222222ULL;
    return (uint32_t)((t >> (4 * rr)) & 15);
}
/* raster index of group scan position g (g_scanOrderCG, constants.cpp:402-461, by rule): groups in `type` order for the 2x2 grid
 * of an 8x8 TU, up-right diagonal for the 4x4 / 8x8 grids of 16x16 / 32x32 TUs */
struct CbDiag { uint8_t d4[16], d8[64]; };
constexpr CbDiag cb_make_diag()
{
    CbDiag t = {};
    for (int n = 4; n <= 8; n += 4)
    {
        int i = 0;
        for (int d = 0; d < 2 * n - 1; d++)
            for (int y = d < n ? d : n - 1; y >= 0 && d - y < n; y--, i++)
                (n == 4 ? t.d4 : t.d8)[i] = (uint8_t)(y * n + (d - y));
    }
    return t;
}
__device__ const CbDiag cb_diag = cb_make_diag();
XA_DEV uint32_t cb_cg_blk(int type, int log2N, int g)
{
    if (log2N == 2) return 0;
    if (log2N == 3) return ((type == 1 ? 0x3210u : 0x3120u) >> (4 * g)) & 15;
    return log2N == 4 ? cb_diag.d4[g] : cb_diag.d8[g];
}

__global__ __launch_bounds__(CB_LANES) void k_coeff_bits(const x265amd_coeff_bits_job* jobs, int n, uint64_t* out)
{
    __shared__ CbLds lds;
    const int lane = threadIdx.x;
    const int ji = blockIdx.x * CB_LANES + lane;
    if (ji >= n) return;
    const x265amd_coeff_bits_job j = jobs[ji];
    uint8_t* ctx = lds.ctx[lane];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(j.ctx_in);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ctx);
        for (int i = 0; i < X265AMD_CTX_STRIDE / 4; i++) dst[i] = src[i];
    }
    const int16_t* coeff = reinterpret_cast<const int16_t*>(j.coeff);
    const int log2N = j.log2_tr_size, N = 1 << log2N, isLuma = j.ttype == 0;
    const int scanType = !j.intra ? 0 : ((log2N <= 2 || (isLuma && log2N == 3)) ? (j.dir_mode >= 22 && j.dir_mode <= 30 ? 1 : (j.dir_mode >= 6 && j.dir_mode <= 14 ? 2 : 0)) : 0);
    const int cgType = log2N >= 4 ? 0 : scanType;
    const int ncgAll = 1 << (2 * (log2N - 2));
    const uint32_t log2CG = (uint32_t)log2N - 2, cgStride = (uint32_t)N >> 2;
    uint64_t bits = 0;

    /* scanPosLast_c (dct.cpp:757-790) folded in: find the last group / position holding a level, and the group flags */
    int lastSet = -1, lastK = -1;
    uint64_t cgFlags = 0;
    for (int g = ncgAll - 1; g >= 0 && lastSet < 0; g--)
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, g);
        const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
        for (int k = 15; k >= 0; k--)
        {
            const uint32_t rr = cb_in_cg(cgType == 0 && log2N >= 4 ? 0 : scanType, k);
            if (coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)]) { lastSet = g; lastK = k; break; }
        }
    }
    if (lastSet < 0)
    {
        out[ji] = 0;
        uint32_t* dst = reinterpret_cast<uint32_t*>(j.ctx_out);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ctx);
        for (int i = 0; i < X265AMD_CTX_STRIDE / 4; i++) dst[i] = src[i];
        return;
    }
    const int inType = log2N >= 4 ? 0 : scanType;

    /* last position: context-coded prefixes, bypass suffixes (entropy.cpp:1874-1908) */
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, lastSet);
        const uint32_t rr = cb_in_cg(inType, lastK);
        uint32_t px = (blk & ((1u << log2CG) - 1)) * 4 + (rr & 3), py = (blk >> log2CG) * 4 + (rr >> 2);
        if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
        int ctxIdx = isLuma ? 3 * (log2N - 2) + (log2N == 5) : N_LAST_XY_LUMA;
        const int ctxShift = isLuma ? (log2N > 2) : log2N - 2;
        const uint32_t maxGroupIdx = ((uint32_t)log2N << 1) - 1;
        for (int i = 0; i < 2; i++, ctxIdx += N_LAST_XY)
        {
            const uint32_t pos = i ? py : px;
            uint32_t prefix = pos, suffixLen = 0;
            if (pos >= 4) { const uint32_t l = 31 - (uint32_t)__clz((int)pos); suffixLen = l - 1; prefix = 2 * l + ((pos >> (l - 1)) & 1); }
            uint8_t* c = ctx + CTX_LAST_X + ctxIdx;
            for (uint32_t k = 0; k < prefix; k++) bits += cb_bin(c + (k >> ctxShift), 1);
            if (prefix < maxGroupIdx) bits += cb_bin(c + (prefix >> ctxShift), 0);
            bits += (uint64_t)suffixLen << 15;
        }
    }
    /* groups in front of the last one that hold levels (entropy.cpp:1862-1868) */
    for (int g = 0; g < lastSet; g++)
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, g);
        const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
        bool any = false;
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) any |= coeff[base + y * N + x] != 0;
        if (any) cgFlags |= (uint64_t)1 << blk;
    }

    uint8_t* cgCtx = ctx + CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG);
    uint8_t* sigCtx = ctx + CTX_SIG + (isLuma ? 0 : N_SIG_LUMA);
    const int firstSig = log2N == 2 ? 0 : log2N == 3 ? ((scanType != 0 && isLuma) ? 15 : 9) : (isLuma ? 21 : 12);
    uint32_t c1 = 1;
    int sigOff = lastK - 1;
    uint16_t absCoeff[16];
    uint32_t numNonZero = 1;
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, lastSet);
        const uint32_t rr = cb_in_cg(inType, lastK);
        absCoeff[0] = (uint16_t)abs((int)coeff[(int)((blk >> log2CG) * 4 + (rr >> 2)) * N + (int)((blk & ((1u << log2CG) - 1)) * 4 + (rr & 3))]);
    }
    for (int sub = lastSet; sub >= 0; sub--)
    {
        const int subBase = sub << 4;
        const uint32_t cgBlk = cb_cg_blk(cgType, log2N, sub), cgY = cgBlk >> log2CG, cgX = cgBlk & ((1u << log2CG) - 1);
        const uint64_t cgMask = (uint64_t)1 << cgBlk;
        const int base = (int)(cgY * 4) * N + (int)(cgX * 4);
        uint32_t firstNZ = 16, lastNZ = 0;      /* positions (scan offsets) of the first / last level of this group */
        if (sub == lastSet || !sub) cgFlags |= cgMask;
        else
        {
            const uint32_t sigPos = cgBlk + 1 < 64 ? (uint32_t)(cgFlags >> (cgBlk + 1)) : 0;
            const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
            bits += cb_bin(cgCtx + (right | lower), (cgFlags & cgMask) != 0);
        }
        if (sub == lastSet) { firstNZ = lastNZ = (uint32_t)lastK; }
        if (sigOff >= 0 && (cgFlags & cgMask))
        {
            /* costCoeffNxN_c (dct.cpp:838-890) */
            uint32_t pattern = 0;
            if (cgStride != 1)
            {
                const uint32_t sigPos = cgBlk + 1 < 64 ? (uint32_t)(cgFlags >> (cgBlk + 1)) : 0;
                const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
                pattern = right + lower * 2;
            }
            const int offset = firstSig + ((isLuma && sub) ? 3 : 0);
            uint32_t nnz = sigOff < 15 ? 1 : 0;
            uint32_t sum = 0;
            for (int k = sigOff; k >= 0; k--)
            {
                const uint32_t rr = cb_in_cg(inType, k);
                const int v = coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)];
                const uint32_t sig = v != 0;
                if (k != 0 || subBase == 0 || nnz)
                {
                    const uint32_t ctxSig = (subBase + k) ? cb_sig_ctx_inc(log2N, pattern, rr) + (uint32_t)offset : 0;
                    sum += cb_bin(sigCtx + ctxSig, sig);
                }
                if (sig)
                {
                    absCoeff[nnz] = (uint16_t)abs(v);
                    if (firstNZ == 16 || (uint32_t)k < firstNZ) firstNZ = (uint32_t)k;
                    if (nnz == 0) lastNZ = (uint32_t)k;
                }
                nnz += sig;
            }
            bits += sum & 0xFFFFFF;
            numNonZero = nnz;
        }
        else if (sub != lastSet) numNonZero = 0;
        if (numNonZero > 0)
        {
            const bool signHidden = lastNZ - firstNZ >= 4;
            const uint32_t ctxSet = (((sub > 0) + (uint32_t)isLuma) & 2) + !(c1 & 3);
            uint8_t* oneCtx = ctx + CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + 4 * ctxSet;
            const uint32_t numC1 = numNonZero < 8 ? numNonZero : 8;
            /* costC1C2Flag_c (dct.cpp:942-993) */
            uint32_t sum = 0, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFE;
            c1 = 1;
            for (uint32_t idx = 0; idx < numC1; idx++)
            {
                const uint32_t s1 = absCoeff[idx] > 1, s2 = absCoeff[idx] > 2;
                sum += cb_bin(oneCtx + c1, s1);
                if (s1) c1Next = 0;
                if (s1 + firstC2Flag == 3) firstC2Flag = s2;
                if (s1 + firstC2Idx == 9) firstC2Idx = idx;
                c1 = c1Next & 3;
                c1Next >>= 2;
            }
            if (!c1) sum += cb_bin(ctx + CTX_ABS + (isLuma ? 0 : N_ABS_LUMA) + ctxSet, firstC2Flag);
            bits += sum & 0x00FFFFFF;
            bits += (uint64_t)(numNonZero - ((j.sign_hide && signHidden) ? 1 : 0)) << 15;
            if (numNonZero > firstC2Idx)
            {
                /* costCoeffRemain_c (dct.cpp:892-938) */
                uint32_t rice = 0, rsum = 0;
                int baseLevel = 3;
                for (uint32_t idx = firstC2Idx; idx < numNonZero; idx++)
                {
                    if (idx >= 8) baseLevel = 1;
                    int code = (int)absCoeff[idx] - baseLevel;
                    if (code >= 0)
                    {
                        code = (int)((uint32_t)code >> rice) - 3;
                        if (code >= 0)
                        {
                            const uint32_t length = 31 - (uint32_t)__clz(code + 1);
                            code = (int)(length + length);
                        }
                        rsum += (uint32_t)(3 + 1 + (int)rice + code);
                        if (absCoeff[idx] > (3u << rice)) rice = (rice + 1) - (rice >> 2);
                    }
                    baseLevel = 2;
                }
                bits += (uint64_t)rsum << 15;
            }
        }
        numNonZero = 0;
        sigOff = 15;
    }
    out[ji] = bits;
    uint32_t* dst = reinterpret_cast<uint32_t*>(j.ctx_out);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ctx);
    for (int i = 0; i < X265AMD_CTX_STRIDE / 4; i++) dst[i] = src[i];
}

/* =========================================================================================================
 * host side
 * ======================================================================================================= */
extern "C" int x265amd_est_bit(void* stream, const x265amd_est_job* d_jobs, int n)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs) return xa_fail(X265AMD_EINVAL, "x265amd_est_bit: bad arguments");
    const XaArgsJobs4 qa = { (uint64_t)(uintptr_t)d_jobs, 0, 0, 0, n };
    hipError_t e;
    XA_LAUNCH(e, stream, XA_OP_EST_BIT, n, qa, k_est_bit, dim3((n + EST_WAVES - 1) / EST_WAVES), dim3(64 * EST_WAVES), 0, d_jobs, n);
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_coeff_bits(void* stream, const x265amd_coeff_bits_job* d_jobs, int n, uint64_t* d_bits)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs || !d_bits) return xa_fail(X265AMD_EINVAL, "x265amd_coeff_bits: bad arguments");
    hipLaunchKernelGGL(k_coeff_bits, dim3((n + CB_LANES - 1) / CB_LANES), dim3(CB_LANES), 0, (hipStream_t)stream, d_jobs, n, d_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

namespace {
struct Stage
{
    char* dev = nullptr;
    ~Stage() { if (dev) (void)hipFree(dev); }
    char* get() { if (!dev) XA_HIP_FATAL(hipMalloc((void**)&dev, 16 * 1024)); return dev; }
};
thread_local Stage g_enStage;
}

extern "C" void x265amd_est_bit_host(const uint8_t* ctx, int log2TrSize, int isLuma, int32_t* est)
{
    char* d = g_enStage.get();
    uint8_t* dCtx = (uint8_t*)d; int32_t* dEst = (int32_t*)(d + 256); x265amd_est_job* dJob = (x265amd_est_job*)(d + 1024);
    XA_HIP_FATAL(hipMemcpy(dCtx, ctx, X265AMD_CTX_COUNT, hipMemcpyHostToDevice));
    XA_HIP_FATAL(hipMemcpy(dEst, est, 184 * sizeof(int32_t), hipMemcpyHostToDevice));
    x265amd_est_job j;
    memset(&j, 0, sizeof(j));
    j.ctx = (uint64_t)(uintptr_t)dCtx; j.est = (uint64_t)(uintptr_t)dEst; j.log2_tr_size = (uint8_t)log2TrSize; j.is_luma = (uint8_t)isLuma;
    XA_HIP_FATAL(hipMemcpy(dJob, &j, sizeof(j), hipMemcpyHostToDevice));
    if (x265amd_est_bit(nullptr, dJob, 1) != X265AMD_OK) { fprintf(stderr, "x265amd: fatal: %s\n", x265amd_last_error()); abort(); }
    XA_HIP_FATAL(hipMemcpy(est, dEst, 184 * sizeof(int32_t), hipMemcpyDeviceToHost));
}

extern "C" uint64_t x265amd_code_coeff_bits(const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int dirMode, int signHide, uint8_t* ctx)
{
    char* d = g_enStage.get();
    uint8_t* dCtx = (uint8_t*)d; uint64_t* dBits = (uint64_t*)(d + 256); x265amd_coeff_bits_job* dJob = (x265amd_coeff_bits_job*)(d + 512);
    int16_t* dCoeff = (int16_t*)(d + 1024);
    uint8_t tmp[X265AMD_CTX_STRIDE];
    memset(tmp, 0, sizeof(tmp)); memcpy(tmp, ctx, X265AMD_CTX_COUNT);
    XA_HIP_FATAL(hipMemcpy(dCtx, tmp, X265AMD_CTX_STRIDE, hipMemcpyHostToDevice));
    XA_HIP_FATAL(hipMemcpy(dCoeff, coeff, sizeof(int16_t) << (2 * log2TrSize), hipMemcpyHostToDevice));
    x265amd_coeff_bits_job j;
    memset(&j, 0, sizeof(j));
    j.coeff = (uint64_t)(uintptr_t)dCoeff; j.ctx_in = j.ctx_out = (uint64_t)(uintptr_t)dCtx;
    j.log2_tr_size = (uint8_t)log2TrSize; j.ttype = (uint8_t)ttype; j.intra = (uint8_t)bIntra; j.dir_mode = (uint8_t)dirMode; j.sign_hide = (uint8_t)signHide;
    XA_HIP_FATAL(hipMemcpy(dJob, &j, sizeof(j), hipMemcpyHostToDevice));
    if (x265amd_coeff_bits(nullptr, dJob, 1, dBits) != X265AMD_OK) { fprintf(stderr, "x265amd: fatal: %s\n", x265amd_last_error()); abort(); }
    uint64_t bits = 0;
    XA_HIP_FATAL(hipMemcpy(&bits, dBits, 8, hipMemcpyDeviceToHost));
    XA_HIP_FATAL(hipMemcpy(tmp, dCtx, X265AMD_CTX_STRIDE, hipMemcpyDeviceToHost));
    memcpy(ctx, tmp, X265AMD_CTX_COUNT);
    return bits;
}
