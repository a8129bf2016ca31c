/* Latency of Quant::rdoQuant's device form (tu_dev.h: wave_rdo_quant) by phase, outside the encoder: one wavefront per workgroup, a few workgroups, synthetic
 * coefficient blocks with a plausible shape.  A developer tool (dbg/README.md); nothing in the product, the tests or the bench uses it.
 *
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX265AMD_DEPTH=8 -Iinclude -Ix265-amod_amd/csrc dbg/rdoq_lat.hip -o dbg/bin/rdoq_lat
 *   dbg/bin/rdoq_lat [workgroups reps amplitude rdoqLevel ttype qp]
 *       -> per size: microseconds per block, cycles per phase, a checksum of all levels and counts: build it against two versions of tu_dev.h (-I a checkout of the
 *          other one) and the checksums say whether they decide alike (dbg/rdoq_lat.sh; profiles/r05_rdoq_latency.txt)
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <chrono>

__shared__ unsigned long long rq_acc[16];
__shared__ unsigned long long rq_t0;
#define RQ_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (laneIn == 0) { rq_acc[i] += t_ - rq_t0; rq_t0 = t_; } } while (0)
#define RQ_T0() do { if (laneIn == 0) rq_t0 = __builtin_readcyclecounter(); } while (0)
#define RQ_FN wave_rdo_quant
#include "tu_dev.h"

struct Args
{
    const int16_t* dct; const int16_t* fdct; const int* est; int16_t* out; uint32_t* numSig; unsigned long long* prof;
    int log2N, nBlocks, signHide, psy, rdoqLevel, qp, ttype, reps;
};

__global__ __launch_bounds__(64) void k_rdoq_wave(Args a)
{
    __shared__ TuLds s;
    __shared__ RdoqLds r;
    const int lane = threadIdx.x, n2 = 1 << (2 * a.log2N);
    if (lane < 16) rq_acc[lane] = 0;
    RdoqParams P = { a.est, 70000, 250, a.psy, a.rdoqLevel, 0 };
    for (int rep = 0; rep < a.reps; rep++)
    for (int b = blockIdx.x; b < a.nBlocks; b += gridDim.x)
    {
        for (int i = lane; i < n2; i += 64) { s.dct[i] = a.dct[(size_t)b * n2 + i]; reinterpret_cast<int16_t*>(s.deltaU)[i] = a.fdct[(size_t)b * n2 + i]; }
        xa_wave_sync();
        const int dir = (b % 3 == 0) ? 26 : (b % 3 == 1 ? 10 : 1);
        const unsigned long long t0 = __builtin_readcyclecounter();
        const uint32_t ns = RQ_FN(s, r, P, a.log2N, a.ttype, 1, dir, a.qp, a.signHide, a.psy != 0 && a.ttype == 0, lane);
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) rq_acc[15] += t1 - t0;
        xa_wave_sync();
        for (int i = lane; i < n2; i += 64) a.out[(size_t)b * n2 + i] = s.q[i];
        if (lane == 0) a.numSig[b] = ns;
        xa_wave_sync();
    }
    if (lane < 16) atomicAdd(&a.prof[lane], rq_acc[lane]);
}

/* the sixteen-lane form: four 4x4 blocks per wavefront */
__global__ __launch_bounds__(64) void k_rdoq_grp16(Args a)
{
    __shared__ Tu16 t[4];
    __shared__ int64_t costSig[4][16], delta[4][16], costCg[4][2];
    __shared__ int32_t rateDown[4][16], sigDelta[4][16], est[184];
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
    if (lane < 16) rq_acc[lane] = 0;
    for (int i = lane; i < 184; i += 64) est[i] = a.est[i];
    xa_wave_sync();
    RdoqParams P = { est, 70000, 250, a.psy, a.rdoqLevel, 1 };
    for (int rep = 0; rep < a.reps; rep++)
    for (int b0 = blockIdx.x * 4; b0 < a.nBlocks; b0 += gridDim.x * 4)
    {
        const int b = b0 + g < a.nBlocks ? b0 + g : a.nBlocks - 1;
        t[g].dct[l] = a.dct[(size_t)b * 16 + l]; reinterpret_cast<int16_t*>(t[g].deltaU)[l] = a.fdct[(size_t)b * 16 + l];
        xa_wave_sync();
        const int dir = (b % 3 == 0) ? 26 : (b % 3 == 1 ? 10 : 1);
        RdoqRef rr{ costSig[g], delta[g], rateDown[g], sigDelta[g], costCg[g], est };
        const unsigned long long t0 = __builtin_readcyclecounter();
        const uint32_t ns = RQ_FN<RdoqRef, true, Tu16>(t[g], rr, P, 2, a.ttype, 1, dir, a.qp, a.signHide, a.psy != 0 && a.ttype == 0, lane);
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) rq_acc[15] += t1 - t0;
        xa_wave_sync();
        if (b0 + g < a.nBlocks) { a.out[(size_t)b * 16 + l] = t[g].q[l]; if (l == 0) a.numSig[b] = ns; }
        xa_wave_sync();
    }
    if (lane < 16) atomicAdd(&a.prof[lane], rq_acc[lane]);
}

static const int tu_quantScales_host[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static double rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }

int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 8;
    const int reps = argc > 2 ? atoi(argv[2]) : 4;
    const double ampMul = argc > 3 ? atof(argv[3]) : 1.0;        /* denser / larger levels */
    const int rdoqLevel = argc > 4 ? atoi(argv[4]) : 2, ttype = argc > 5 ? atoi(argv[5]) : 0, qpArg = argc > 6 ? atoi(argv[6]) : 30;
    int* estD; std::vector<int> est(184);
    for (int i = 0; i < 184; i++) est[i] = 8000 + (int)(rnd() * 90000);
    CK(hipMalloc(&estD, 184 * 4)); CK(hipMemcpy(estD, est.data(), 184 * 4, hipMemcpyHostToDevice));
    unsigned long long* profD; CK(hipMalloc(&profD, 16 * 8));
    for (int form = 0; form < 2; form++)
    for (int log2N = 2; log2N <= 5; log2N++)
    {
        if (form == 1 && log2N != 2) continue;
        const int n2 = 1 << (2 * log2N), N = 1 << log2N, nBlocks = 64 * grid;
        /* levels of about 6 at DC falling off with frequency: one level is 2^qbits / quantScale coefficient units */
        const int qp = qpArg, per = qp / 6, qbits = 14 + per + 15 - 8 - log2N;
        const double unit = (double)(1 << qbits) / (double)tu_quantScales_host[qp % 6];
        std::vector<int16_t> dct((size_t)nBlocks * n2), fdct((size_t)nBlocks * n2);
        for (int b = 0; b < nBlocks; b++)
        {
            const double amp = (2.0 + 8.0 * rnd()) * ampMul;
            for (int i = 0; i < n2; i++)
            {
                const int y = i >> log2N, x = i & (N - 1);
                const double scale = amp * unit / (1.0 + 0.9 * (x + y) * 4.0 / N * (0.5 + rnd()));
                const double u = rnd() - 0.5;
                const double v = -scale * (u < 0 ? -1 : 1) * log(1 - 2 * fabs(u) + 1e-12) * 0.6;
                dct[(size_t)b * n2 + i] = (int16_t)fmax(-32000, fmin(32000, v));
                fdct[(size_t)b * n2 + i] = (int16_t)fmax(-32000, fmin(32000, v * 1.3 + (rnd() - 0.5) * unit));
            }
        }
        int16_t *dctD, *fdctD, *outD; uint32_t* nsD;
        CK(hipMalloc(&dctD, dct.size() * 2)); CK(hipMalloc(&fdctD, dct.size() * 2)); CK(hipMalloc(&outD, dct.size() * 2)); CK(hipMalloc(&nsD, nBlocks * 4));
        CK(hipMemcpy(dctD, dct.data(), dct.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(fdctD, fdct.data(), dct.size() * 2, hipMemcpyHostToDevice));
        for (int variant = 0; variant < 2; variant++)       /* 0: psy-rdoq + sign hiding (the slow preset's), 1: neither */
        {
            Args a = { dctD, fdctD, estD, outD, nsD, profD, log2N, nBlocks, variant == 0, variant == 0 ? 256 : 0, rdoqLevel, qp, ttype, reps };
            CK(hipMemset(profD, 0, 16 * 8));
            if (form) k_rdoq_grp16<<<grid, 64>>>(a); else k_rdoq_wave<<<grid, 64>>>(a);          /* warm */
            CK(hipDeviceSynchronize());
            CK(hipMemset(profD, 0, 16 * 8));
            const auto w0 = std::chrono::steady_clock::now();
            if (form) k_rdoq_grp16<<<grid, 64>>>(a); else k_rdoq_wave<<<grid, 64>>>(a);
            CK(hipDeviceSynchronize());
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            std::vector<int16_t> out(dct.size()); std::vector<uint32_t> ns(nBlocks); unsigned long long prof[16];
            CK(hipMemcpy(out.data(), outD, out.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ns.data(), nsD, nBlocks * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(prof, profD, sizeof(prof), hipMemcpyDeviceToHost));
            uint64_t h = 1469598103934665603ull; double nnz = 0;
            for (size_t i = 0; i < out.size(); i++) { h = (h ^ (uint16_t)out[i]) * 1099511628211ull; nnz += out[i] != 0; }
            for (int b = 0; b < nBlocks; b++) h = (h ^ ns[b]) * 1099511628211ull;
            const double calls = (double)(form ? (nBlocks + 3) / 4 : nBlocks) * reps;       /* calls of wave_rdo_quant over all workgroups */
            printf("%s %2dx%-2d %s: %7.2f us per call (wall, %d workgroups side by side), %5.1f levels per block, checksum %016llx\n   cycles per call: total %.0f;",
                   form ? "grp16" : "wave ", N, N, variant == 0 ? "psy+signhide" : "plain       ", sec * 1e6 / (calls / grid), grid, nnz / nBlocks,
                   (unsigned long long)h, prof[15] / calls);
            for (int i = 0; i < 12; i++) printf(" [%d] %.0f", i, prof[i] / calls);
            printf("\n");
        }
        CK(hipFree(dctD)); CK(hipFree(fdctD)); CK(hipFree(outD)); CK(hipFree(nsD));
    }
    return 0;
}
