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
#include "intra_cu_dev.h"
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
    const uint64_t bits = lane_coeff_bits(ctx, reinterpret_cast<const int16_t*>(j.coeff), j.log2_tr_size, j.ttype, j.intra, j.dir_mode, j.sign_hide, EnTabs{ en_bits, en_lpsNext });
    out[ji] = bits;
    uint32_t* dst = reinterpret_cast<uint32_t*>(j.ctx_out);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(ctx);
    for (int i = 0; i < X265AMD_CTX_STRIDE / 4; i++) dst[i] = src[i];
}

/* the wavefront form (wave_coeff_bits / wave_coeff_bits_4x4, entropy_dev.h): a wavefront per job */
#define CB4_WAVES 4
__global__ __launch_bounds__(64 * CB4_WAVES) void k_coeff_bits_wave(const x265amd_coeff_bits_job* jobs, int n, uint64_t* out)
{
    __shared__ uint8_t s_ctx[CB4_WAVES][X265AMD_CTX_STRIDE];
    __shared__ int16_t s_lev[CB4_WAVES][16];
    __shared__ uint32_t s_step[256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_step[i] = en_step.v[i];
    __syncthreads();
    const int ji = blockIdx.x * CB4_WAVES + wv;
    if (ji >= n) return;
    const x265amd_coeff_bits_job j = jobs[ji];
    for (int b = lane; b < X265AMD_CTX_STRIDE; b += 64) s_ctx[wv][b] = reinterpret_cast<const uint8_t*>(j.ctx_in)[b];
    if (lane < 16) s_lev[wv][lane] = reinterpret_cast<const int16_t*>(j.coeff)[lane];
    xa_wave_sync();
    const uint64_t bits = j.log2_tr_size == 2 ? wave_coeff_bits_4x4(s_ctx[wv], s_ctx[wv], s_lev[wv], j.ttype, j.intra, j.dir_mode, j.sign_hide, s_step, lane)
                                              : wave_coeff_bits(s_ctx[wv], s_ctx[wv], reinterpret_cast<const int16_t*>(j.coeff), j.log2_tr_size, j.ttype, j.intra, j.dir_mode, j.sign_hide, s_step, lane);
    xa_wave_sync();
    if (lane == 0) out[ji] = bits;
    for (int b = lane; b < X265AMD_CTX_STRIDE; b += 64) reinterpret_cast<uint8_t*>(j.ctx_out)[b] = s_ctx[wv][b];
}

/* the bits of a whole intra CU (intra_cu_dev.h): a wavefront per CU */
__global__ __launch_bounds__(64 * CB4_WAVES) void k_intra_cu_bits(const x265amd_intra_cu_bits_job* jobs, int n, x265amd_intra_cu_bits_out* out)
{
    __shared__ uint8_t s_ctx[CB4_WAVES][X265AMD_CTX_STRIDE];
    __shared__ uint32_t s_step[256];
    __shared__ uint32_t s_enBits[128];
    __shared__ uint8_t s_enLps[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_step[i] = en_step.v[i];
    if (threadIdx.x < 128) s_enBits[threadIdx.x] = en_bits[threadIdx.x];
    if (threadIdx.x < 64) s_enLps[threadIdx.x] = en_lpsNext[threadIdx.x];
    __syncthreads();
    const int ji = blockIdx.x * CB4_WAVES + wv;
    if (ji >= n) return;
    const x265amd_intra_cu_bits_job& j = jobs[ji];
    for (int b = lane; b < X265AMD_CTX_STRIDE; b += 64) s_ctx[wv][b] = j.ctx[b];
    IntraCuBitsIn in;
    in.log2_cu = j.log2_cu; in.nxn = j.nxn; in.code_part_size = j.code_part_size; in.inter_slice = j.inter_slice; in.skip_ctx = j.skip_ctx; in.sign_hide = j.sign_hide;
    in.chroma_dir = j.chroma_dir; in.cbf_u = j.cbf_u; in.cbf_v = j.cbf_v; in.subdiv_flag = j.subdiv_flag;
    for (int k = 0; k < 4; k++)
    {
        in.luma_dir[k] = j.luma_dir[k]; in.cbf_y[k] = j.cbf_y[k]; in.lev_y[k] = reinterpret_cast<const int16_t*>(j.lev_y[k]);
        for (int i = 0; i < 3; i++) in.preds[k][i] = j.preds[k][i];
    }
    in.lev_u = reinterpret_cast<const int16_t*>(j.lev_u); in.lev_v = reinterpret_cast<const int16_t*>(j.lev_v);
    xa_wave_sync();
    uint64_t mvf = 0, skipf = 0;
    const uint64_t frac = wave_intra_cu_bits(in, s_ctx[wv], j.frac_bits, &mvf, &skipf, s_step, EnTabs{ s_enBits, s_enLps }, lane);
    xa_wave_sync();
    for (int b = lane; b < X265AMD_CTX_STRIDE; b += 64) out[ji].ctx[b] = s_ctx[wv][b];
    if (lane == 0) { out[ji].frac_bits = frac; out[ji].mv_frac = mvf; out[ji].skip_frac = skipf; }
}

/* =========================================================================================================
 * host side
 * ======================================================================================================= */
extern "C" int x265amd_intra_cu_bits(void* stream, const x265amd_intra_cu_bits_job* d_jobs, int n, x265amd_intra_cu_bits_out* d_out)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs || !d_out) return xa_fail(X265AMD_EINVAL, "x265amd_intra_cu_bits: bad arguments");
    hipLaunchKernelGGL(k_intra_cu_bits, dim3((n + CB4_WAVES - 1) / CB4_WAVES), dim3(64 * CB4_WAVES), 0, (hipStream_t)stream, d_jobs, n, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_coeff_bits_wave(void* stream, const x265amd_coeff_bits_job* d_jobs, int n, uint64_t* d_bits)
{
    if (n <= 0) return X265AMD_OK;
    if (!d_jobs || !d_bits) return xa_fail(X265AMD_EINVAL, "x265amd_coeff_bits_wave: bad arguments");
    hipLaunchKernelGGL(k_coeff_bits_wave, dim3((n + CB4_WAVES - 1) / CB4_WAVES), dim3(64 * CB4_WAVES), 0, (hipStream_t)stream, d_jobs, n, d_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

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
