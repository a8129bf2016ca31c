/* Standalone reproducer for DESIGN.md section 8's fence-scope question: no encoder, no job server.
 *
 * Two RESIDENT workgroups (blockIdx 0 and 1: consecutive workgroups of a dispatch go to consecutive XCDs, so they sit on different XCDs with different L2s) hand a
 * 424-byte record back and forth through ordinary device memory (hipMalloc), exactly as the chained 8x8 CUs of an I picture do (csrc/intra_pu_dev.h: xa_chain_publish /
 * xa_chain_wait): the writer fills the record, a barrier, one lane issues a RELEASE fence and then stores the turn's number into a flag with a relaxed atomic store; the
 * reader polls the flag with relaxed atomic loads, issues an ACQUIRE fence (plus s_dcache_inv, as the encoder does), and every lane reads its word of the record with a
 * plain load.  A word that is not the turn's pattern is a lost write; the count of those is the result.
 *
 * A third variant makes every wavefront wait for its own stores (s_waitcnt vmcnt(0)) BEFORE the barrier: the release fence is issued by one lane of one wavefront, and its
 * s_waitcnt covers that wavefront's stores only -- the barrier of this execution mode does not wait for the other wavefronts' vector memory operations.
 *
 * A fourth ("shared lines + late"): the two records interleaved word by word in the same cache lines (two CUs' samples in one line of the picture), and words each side
 * writes after its release and before its next acquire, so that its L2 holds dirty lines while it invalidates (two slots by the turn's parity; checked a turn later, behind the side's next release).
 *
 * The fences' scope is the variable: "agent" (what the encoder had until round 3) or system (what it has now).  Beside the ping-pong an "atomic flood" may run on a
 * second stream, launch after launch: thousands of one-wave workgroups, each ending in an atomicAdd on one of 64 words (the first form of the lookahead's weight-cost
 * kernel, csrc/lowres_kernels.hip: k_flood) -- the condition under which the encoder's chain lost writes.
 *
 *   hipcc --offload-arch=gfx950 -O2 -o fence_scope_repro dbg/fence_scope_repro.hip
 *   ./fence_scope_repro [turns=200000] [flood workgroups=11040] [record words=106]
 * prints one line per variant: scope x {quiet, flood} x {fine placement: the two workgroups adjacent, or spread}: turns, lost words, turns with a lost word, microseconds per turn.
 * --save-temps (or llvm-objdump -d on the binary's gfx950 code object) shows the two variants' ISA; dbg/fence_scope_isa.txt holds the relevant excerpts. */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <atomic>
#include <thread>
#include <chrono>

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(2); } } while (0)

struct Shared
{
    uint64_t flag[2][16];          /* flag[k]: the last turn workgroup k has published (a cache line each) */
    uint32_t rec[2][128];          /* rec[k]: workgroup k's record (up to 128 words; 106 = 424 bytes) */
    uint32_t mix[256];             /* MIX: the two records word by word in the SAME cache lines (mix[2 i + k]: what two CUs' samples in one picture line are) */
    uint32_t late[2][2][128];         /* MIX: words a side writes AFTER its release and before its next acquire -- dirty in its L2 while it invalidates; published by its next release */
    uint32_t lostWords, lostTurns, gaveUp, xcc[2];
};

template <bool SYSTEM> __device__ __forceinline__ void fence_release()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <bool SYSTEM> __device__ __forceinline__ void fence_acquire()
{
    if (SYSTEM) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ uint32_t pattern(uint32_t turn, uint32_t who, uint32_t i) { return turn * 2654435761u + who * 40503u + i * 97u + 1u; }

/* `stride`: which two workgroups of the dispatch play (0 and `stride`); the others leave at once.  One turn = A writes, B checks and answers, A checks. */
template <bool SYSTEM, bool WAITALL, bool MIX> __global__ __launch_bounds__(128) void k_pingpong(Shared* S, uint32_t turns, uint32_t words, uint32_t stride)
{
    const uint32_t who = blockIdx.x == 0 ? 0u : (blockIdx.x == stride ? 1u : 2u);
    if (who == 2u) return;
    const uint32_t tid = threadIdx.x, other = who ^ 1u;
    __shared__ uint32_t s_ok;
    if (tid == 0)
    {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        S->xcc[who] = id & 15u;
    }
    for (uint32_t turn = 1; turn <= turns; turn++)
    {
        if (who == 0)
        {
            /* write, publish */
            if (tid < words) { if (MIX) S->mix[2 * tid] = pattern(turn, 0, tid); else S->rec[0][tid] = pattern(turn, 0, tid); }
            if (WAITALL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     /* EVERY wavefront's stores have left before the barrier lets the publishing lane go on */
            __syncthreads();
            if (tid == 0) { fence_release<SYSTEM>(); __hip_atomic_store(&S->flag[0][0], (uint64_t)turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (MIX && tid < words) S->late[0][turn & 1][tid] = pattern(turn, 2, tid);
        }
        /* wait for the other side's record of this turn (B waits for A's write; A waits for B's answer) */
        if (tid == 0)
        {
            const long long t0 = wall_clock64();
            uint32_t ok = 1;
            while (__hip_atomic_load(&S->flag[other][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < turn)
            {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 500000000ll) { ok = 0; break; }        /* five seconds: the other side is gone */
            }
            fence_acquire<SYSTEM>();
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) { if (tid == 0) atomicAdd(&S->gaveUp, 1u); return; }
        uint32_t bad = 0;
        if (tid < words) bad = (MIX ? S->mix[2 * tid + other] : S->rec[other][tid]) != pattern(turn, other, tid);
        /* the other side's late words of the turn its last release covered: B sees A's of turn - 1 (A released turn after writing them), A sees B's of turn - 1 */
        if (MIX && tid < words && turn > 1) bad |= S->late[other][(turn - 1) & 1][tid] != pattern(turn - 1, 2 + other, tid);
        const uint32_t anyBad = __syncthreads_or((int)bad);
        if (bad) atomicAdd(&S->lostWords, 1u);
        if (anyBad && tid == 0) atomicAdd(&S->lostTurns, 1u);
        if (who == 1)
        {
            if (tid < words) { if (MIX) S->mix[2 * tid + 1] = pattern(turn, 1, tid); else S->rec[1][tid] = pattern(turn, 1, tid); }
            if (WAITALL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) { fence_release<SYSTEM>(); __hip_atomic_store(&S->flag[1][0], (uint64_t)turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (MIX && tid < words) S->late[1][turn & 1][tid] = pattern(turn, 3, tid);
        }
    }
}

__global__ void k_flood(uint32_t* word);
/* Round 5: what the encoder's chain has and the variants above lack.  Workgroups of EIGHT wavefronts with 144 KB of dynamic LDS (one per compute unit, as the
 * job server's); per turn the writer's eight wavefronts write FIVE different buffers -- a 64x64 "tile" (device memory, 16-byte stores), an 8x8 block of a 2-D "picture" with a stride
 * of 2112 bytes (byte stores), the peer record (424 bytes), the chain record (its 160 context bytes as byte stores by 160 lanes, its fraction and sixteen mode bytes by one lane) and a
 * result record in PINNED HOST memory --, every wavefront waits for its own stores, a barrier, ONE lane releases and stores the flag.  The reader polls, acquires (+ s_dcache_inv) and
 * reads the chain record's fraction and modes through a wave-uniform address (the compiler makes scalar loads of them, as in nxn_chain_begin), its contexts and the peer record by
 * lanes, the picture block and the tile by lanes.  Anything that is not the turn's pattern is a lost write. */
struct Enc
{
    uint64_t flag[2][16];
    uint32_t tile[2][64 * 64 / 4];          /* bytes, written 16 at a time */
    uint8_t pic[2][8 * 2112];
    uint32_t peer[2][106];
    struct Chain { uint64_t frac; uint8_t ctx[160]; uint8_t mode[16]; } chain[2];
    uint32_t lostWords, lostTurns, gaveUp, xcc[2], lostKind[6];
};
__device__ __forceinline__ uint8_t pat8(uint32_t turn, uint32_t who, uint32_t i) { return (uint8_t)(pattern(turn, who, i) >> 7); }
template <bool SYSTEM> __global__ __launch_bounds__(512) void k_enc_like(Enc* S, uint32_t* hostRes, uint32_t turns, uint32_t stride)
{
    extern __shared__ char lds[];
    const uint32_t who = blockIdx.x == 0 ? 0u : (blockIdx.x == stride ? 1u : 2u);
    if (who == 2u) return;
    const uint32_t tid = threadIdx.x, other = who ^ 1u;
    __shared__ uint32_t s_ok;
    lds[tid * 200] = (char)tid;                 /* the LDS is really there */
    if (tid == 0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); S->xcc[who] = id & 15u; }
    auto write_all = [&](uint32_t turn) {
        /* tile: 1024 words as 256 16-byte stores by lanes 0..255; picture block: lanes 256..319, a byte each; peer record: lanes 320..425; chain contexts: lanes 0..159 (bytes) */
        if (tid < 256) { uint4 v; v.x = pattern(turn, who, 4 * tid); v.y = pattern(turn, who, 4 * tid + 1); v.z = pattern(turn, who, 4 * tid + 2); v.w = pattern(turn, who, 4 * tid + 3); reinterpret_cast<uint4*>(S->tile[who])[tid] = v; }
        else if (tid < 320) { const uint32_t i = tid - 256; S->pic[who][(i >> 3) * 2112 + (i & 7)] = pat8(turn, who, 5000 + i); }
        else if (tid < 426) S->peer[who][tid - 320] = pattern(turn, who, 6000 + tid - 320);
        if (tid < 160) S->chain[who].ctx[tid] = pat8(turn, who, 7000 + tid);
        if (tid == 511) { S->chain[who].frac = ((uint64_t)turn << 32) | pattern(turn, who, 8000); for (int k = 0; k < 16; k++) S->chain[who].mode[k] = pat8(turn, who, 8100 + k); }
        if (tid >= 448 && tid < 480) hostRes[who * 64 + tid - 448] = pattern(turn, who, 9000 + tid);        /* pinned host memory, plain stores */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) { fence_release<SYSTEM>(); __hip_atomic_store(&S->flag[who][0], (uint64_t)turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    };
    for (uint32_t turn = 1; turn <= turns; turn++)
    {
        if (who == 0) write_all(turn);
        if (tid == 0)
        {
            const long long t0 = wall_clock64();
            uint32_t ok = 1;
            while (__hip_atomic_load(&S->flag[other][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < turn)
            {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 500000000ll) { ok = 0; break; }
            }
            fence_acquire<SYSTEM>();
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) { if (tid == 0) atomicAdd(&S->gaveUp, 1u); return; }
        /* wave-uniform reads of the chain record (scalar loads), as nxn_chain_begin has them */
        const Enc::Chain* ch = &S->chain[other];
        const uint64_t frac = ch->frac;
        uint32_t kind = 0;
        if (frac != (((uint64_t)turn << 32) | pattern(turn, other, 8000))) kind = 1;
        for (int k = 0; k < 16 && !kind; k++) if (ch->mode[k] != pat8(turn, other, 8100 + k)) kind = 2;
        uint32_t bad = 0;
        if (tid == 0 && kind) bad = kind;
        if (tid < 256)
        {
            const uint4 v = reinterpret_cast<const uint4*>(S->tile[other])[tid];
            if (v.x != pattern(turn, other, 4 * tid) || v.y != pattern(turn, other, 4 * tid + 1) || v.z != pattern(turn, other, 4 * tid + 2) || v.w != pattern(turn, other, 4 * tid + 3)) bad = 3;
        }
        else if (tid < 320) { const uint32_t i = tid - 256; if (S->pic[other][(i >> 3) * 2112 + (i & 7)] != pat8(turn, other, 5000 + i)) bad = 4; }
        else if (tid < 426) { if (S->peer[other][tid - 320] != pattern(turn, other, 6000 + tid - 320)) bad = 5; }
        if (tid < 160 && S->chain[other].ctx[tid] != pat8(turn, other, 7000 + tid)) bad = 6;
        const uint32_t anyBad = __syncthreads_or((int)bad);
        if (bad) { atomicAdd(&S->lostWords, 1u); atomicAdd(&S->lostKind[bad - 1], 1u); }
        if (anyBad && tid == 0) atomicAdd(&S->lostTurns, 1u);
        if (who == 1) write_all(turn);
    }
}
template <bool SYSTEM> static void run_enc(const char* name, Enc* dE, uint32_t* hRes, uint32_t turns, uint32_t stride, int floodWgs, hipStream_t sPing, hipStream_t sFlood, uint32_t* dWord)
{
    CHECK(hipMemset(dE, 0, sizeof(Enc)));
    CHECK(hipDeviceSynchronize());
    std::atomic<bool> stop{ false };
    uint64_t floods = 0;
    std::thread flooder;
    if (floodWgs > 0)
        flooder = std::thread([&] {
            while (!stop.load())
            {
                hipLaunchKernelGGL(k_flood, dim3(floodWgs / 46 > 0 ? floodWgs / 46 : 1, 46), dim3(64), 0, sFlood, dWord);
                floods++;
                if ((floods & 7) == 0) (void)hipStreamSynchronize(sFlood);
            }
            (void)hipStreamSynchronize(sFlood);
        });
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(hipFuncSetAttribute((const void*)k_enc_like<SYSTEM>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipLaunchKernelGGL((k_enc_like<SYSTEM>), dim3(stride + 1), dim3(512), 144 * 1024, sPing, dE, hRes, turns, stride);
    CHECK(hipGetLastError());
    CHECK(hipStreamSynchronize(sPing));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    if (flooder.joinable()) flooder.join();
    Enc* h = (Enc*)malloc(sizeof(Enc));
    CHECK(hipMemcpy(h, dE, sizeof(Enc), hipMemcpyDeviceToHost));
    printf("%-44s XCC %u / %u: %u turns, %u lost in %u turns (frac %u, modes %u, tile %u, picture %u, peer %u, contexts %u)%s, %.2f us per turn, %llu flood launches\n", name, h->xcc[0], h->xcc[1], turns,
           h->lostWords, h->lostTurns, h->lostKind[0], h->lostKind[1], h->lostKind[2], h->lostKind[3], h->lostKind[4], h->lostKind[5], h->gaveUp ? " (a side gave up waiting)" : "", us / turns, (unsigned long long)floods);
    fflush(stdout);
    free(h);
}

/* the flood: a one-wave workgroup, a little LDS traffic, an atomicAdd on one of 64 words (csrc/lowres_kernels.hip: k_flood, mode 1) */
__global__ __launch_bounds__(64) void k_flood(uint32_t* word)
{
    __shared__ uint32_t t[64];
    t[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t v = t[63 - threadIdx.x];
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    if (threadIdx.x == 0) atomicAdd(word + (blockIdx.y & 63), v);
}

template <bool SYSTEM, bool WAITALL, bool MIX> static void run(const char* name, Shared* dS, uint32_t turns, uint32_t words, uint32_t stride, int floodWgs, hipStream_t sPing, hipStream_t sFlood, uint32_t* dWord)
{
    CHECK(hipMemset(dS, 0, sizeof(Shared)));
    CHECK(hipDeviceSynchronize());
    std::atomic<bool> stop{ false };
    uint64_t floods = 0;
    std::thread flooder;
    if (floodWgs > 0)
        flooder = std::thread([&] {
            while (!stop.load())
            {
                hipLaunchKernelGGL(k_flood, dim3(floodWgs / 46 > 0 ? floodWgs / 46 : 1, 46), dim3(64), 0, sFlood, dWord);
                floods++;
                if ((floods & 7) == 0) (void)hipStreamSynchronize(sFlood);
            }
            (void)hipStreamSynchronize(sFlood);
        });
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k_pingpong<SYSTEM, WAITALL, MIX>), dim3(stride + 1), dim3(128), 0, sPing, dS, turns, words, stride);
    CHECK(hipGetLastError());
    CHECK(hipStreamSynchronize(sPing));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    if (flooder.joinable()) flooder.join();
    Shared h;
    CHECK(hipMemcpy(&h, dS, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-34s XCC %u / %u: %u turns, %u lost words in %u turns%s, %.2f us per turn, %llu flood launches\n", name, h.xcc[0], h.xcc[1], turns, h.lostWords, h.lostTurns,
           h.gaveUp ? " (a side gave up waiting)" : "", us / turns, (unsigned long long)floods);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const uint32_t turns = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000u;
    const int floodWgs = argc > 2 ? atoi(argv[2]) : 11040;
    const uint32_t words = argc > 3 ? (uint32_t)atoi(argv[3]) : 106u;
    if (words > 128) { fprintf(stderr, "at most 128 words\n"); return 2; }
    Shared* dS = nullptr; uint32_t* dWord = nullptr;
    CHECK(hipMalloc((void**)&dS, sizeof(Shared)));
    CHECK(hipMalloc((void**)&dWord, 4096));
    CHECK(hipMemset(dWord, 0, 4096));
    hipStream_t sPing, sFlood;
    CHECK(hipStreamCreateWithFlags(&sPing, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sFlood, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s (%s), record %u bytes, %u turns per variant, flood %d workgroups per launch\n", prop.name, prop.gcnArchName, words * 4, turns, floodWgs);
    /* stride 1: workgroups 0 and 1 (neighbouring XCDs); stride 4: 0 and 4; stride 8: 0 and 8 (the same XCD again on an 8-XCD part: the control) */
    for (uint32_t stride : { 1u, 4u, 8u })
    {
        char name[64];
        snprintf(name, sizeof(name), "agent scope, quiet, wgs 0/%u", stride);  run<false, false, false>(name, dS, turns, words, stride, 0, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "agent scope, FLOOD, wgs 0/%u", stride);  run<false, false, false>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "agent + all waves wait, FLOOD, 0/%u", stride); run<false, true, false>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "system scope, quiet, wgs 0/%u", stride); run<true, false, false>(name, dS, turns, words, stride, 0, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "system scope, FLOOD, wgs 0/%u", stride); run<true, false, false>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "system + all waves wait, FLOOD, 0/%u", stride); run<true, true, false>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "agent, shared lines+late, quiet 0/%u", stride); run<false, false, true>(name, dS, turns, words, stride, 0, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "agent, shared lines+late, FLOOD 0/%u", stride); run<false, false, true>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "system, shared lines+late, FLOOD 0/%u", stride); run<true, false, true>(name, dS, turns, words, stride, floodWgs, sPing, sFlood, dWord);
    }
    /* round 5: the encoder-like hand-over (eight wavefronts, 144 KB LDS, five buffers, pinned host memory among them, scalar loads on the reader) */
    Enc* dE = nullptr; uint32_t* hRes = nullptr;
    CHECK(hipMalloc((void**)&dE, sizeof(Enc)));
    CHECK(hipHostMalloc((void**)&hRes, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    for (uint32_t stride : { 1u, 4u, 8u })
    {
        char name[64];
        snprintf(name, sizeof(name), "encoder-like, agent scope, quiet, wgs 0/%u", stride);  run_enc<false>(name, dE, hRes, turns, stride, 0, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "encoder-like, agent scope, FLOOD, wgs 0/%u", stride);  run_enc<false>(name, dE, hRes, turns, stride, floodWgs, sPing, sFlood, dWord);
        snprintf(name, sizeof(name), "encoder-like, system scope, FLOOD, wgs 0/%u", stride); run_enc<true>(name, dE, hRes, turns, stride, floodWgs, sPing, sFlood, dWord);
    }
    return 0;
}
