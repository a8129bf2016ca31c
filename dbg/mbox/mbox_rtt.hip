// Feasibility probe: round-trip latency of a host <-> persistent-kernel mailbox in pinned host memory, against launch + synchronise.
// build: hipcc --offload-arch=gfx950 -O2 -o mbox_rtt mbox_rtt.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Box { volatile uint64_t head; uint64_t pad0[7]; volatile uint64_t tail; uint64_t pad1[7]; volatile uint64_t quit; uint64_t pad2[7]; uint64_t payload[8]; uint64_t result[8]; };

__global__ void k_server(Box* boxes, uint64_t* devbuf, long maxIdle)
{
    Box* b = boxes + blockIdx.x;
    __shared__ uint64_t s_seq;
    uint64_t seen = 0;
    long idle = 0;
    for (;;)
    {
        if (threadIdx.x == 0)
        {
            uint64_t h;
            for (;;)
            {
                h = __hip_atomic_load(&b->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (h != seen) break;
                if (__hip_atomic_load(&b->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || ++idle > maxIdle) { h = ~0ull; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            s_seq = h;
        }
        __syncthreads();
        const uint64_t h = s_seq;
        if (h == ~0ull) return;
        idle = 0;
        // "work": every thread touches device memory, lane 0 of the block folds the payload
        devbuf[blockIdx.x * blockDim.x + threadIdx.x] += h;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            uint64_t s = 0;
            for (int i = 0; i < 8; i++) s += b->payload[i];
            b->result[0] = s + h;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_store(&b->tail, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        seen = h;
        __syncthreads();
    }
}

__global__ void k_tiny(uint64_t* devbuf, uint64_t* res, uint64_t v)
{
    devbuf[blockIdx.x * blockDim.x + threadIdx.x] += v;
    if (threadIdx.x == 0) res[0] = v;
}

int main(int argc, char** argv)
{
    const int nq = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 20000, threads = argc > 3 ? atoi(argv[3]) : 1024;
    Box* boxes; uint64_t* devbuf; uint64_t* res;
    CK(hipHostMalloc((void**)&boxes, sizeof(Box) * nq, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&res, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset((void*)boxes, 0, sizeof(Box) * nq);
    CK(hipMalloc((void**)&devbuf, sizeof(uint64_t) * nq * threads));
    CK(hipMemset(devbuf, 0, sizeof(uint64_t) * nq * threads));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // baseline: launch + synchronise
    {
        for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(threads), 0, st, devbuf, res, (uint64_t)i); CK(hipStreamSynchronize(st)); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 5000; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(threads), 0, st, devbuf, res, (uint64_t)i); CK(hipStreamSynchronize(st)); }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5000;
        printf("launch+sync (1 thread, 1 block x %d): %.2f us per round trip\n", threads, us);
    }
    hipLaunchKernelGGL(k_server, dim3(nq), dim3(threads), 0, st, boxes, devbuf, 200000000L);
    std::vector<std::thread> pool;
    std::vector<double> per(nq);
    std::atomic<int> bad(0);
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < nq; q++)
        pool.emplace_back([&, q] {
            Box* b = boxes + q;
            auto s0 = std::chrono::steady_clock::now();
            for (uint64_t i = 1; i <= (uint64_t)iters; i++)
            {
                for (int k = 0; k < 8; k++) b->payload[k] = i + k;
                std::atomic_thread_fence(std::memory_order_release);
                b->head = i;
                long spins = 0;
                while (b->tail != i) { if (++spins > 2000000000L) { bad++; return; } }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (b->result[0] != 8 * i + 28 + i) bad++;
            }
            per[q] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count() / iters;
        });
    for (auto& t : pool) t.join();
    double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    for (int q = 0; q < nq; q++) boxes[q].quit = 1;
    CK(hipStreamSynchronize(st));
    double mn = 1e9, mx = 0, sum = 0;
    for (double v : per) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; }
    printf("mailbox: %d queues x %d round trips, %d threads per block: %.2f us per round trip (min %.2f max %.2f), aggregate %.2f M round trips/s, mismatches %d\n", nq, iters, threads, sum / nq, mn, mx,
           (double)nq * iters / wall, bad.load());
    return bad.load() != 0;
}
