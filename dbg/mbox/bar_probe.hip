// Probe: which device allocations can the host CPU write / read directly (large BAR)?
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
static sigjmp_buf jb;
static void onsegv(int) { siglongjmp(jb, 1); }
__global__ void k_read(volatile uint64_t* p, uint64_t* out) { out[0] = p[0]; }
static void tryit(const char* name, volatile uint64_t* p)
{
    if (!p) { printf("%s: allocation failed\n", name); return; }
    uint64_t* out; hipHostMalloc((void**)&out, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (sigsetjmp(jb, 1) == 0)
    {
        p[0] = 0x1234567890abcdefull;
        __sync_synchronize();
        uint64_t back = p[0];
        hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, 0, p, out);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 100000; i++) p[0] = i;
        __sync_synchronize();
        double wus = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100000;
        t0 = std::chrono::steady_clock::now();
        uint64_t s = 0;
        for (int i = 0; i < 2000; i++) s += p[0];
        double rus = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        printf("%s: host write ok, host read back %llx, device saw %llx; host write %.3f us, host read %.3f us (%llu)\n", name, (unsigned long long)back, (unsigned long long)out[0], wus, rus, (unsigned long long)s);
    }
    else
        printf("%s: SIGSEGV on host access\n", name);
}
int main()
{
    signal(SIGSEGV, onsegv); signal(SIGBUS, onsegv);
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained) != hipSuccess) p = nullptr;
    tryit("hipExtMallocWithFlags(Finegrained)", (volatile uint64_t*)p);
    p = nullptr;
    if (hipExtMallocWithFlags(&p, 4096, hipDeviceMallocUncached) != hipSuccess) p = nullptr;
    tryit("hipExtMallocWithFlags(Uncached)", (volatile uint64_t*)p);
    p = nullptr;
    if (hipMallocManaged(&p, 4096) != hipSuccess) p = nullptr;
    if (p) { hipMemAdvise(p, 4096, hipMemAdviseSetPreferredLocation, 0); hipMemPrefetchAsync(p, 4096, 0, 0); hipDeviceSynchronize(); }
    tryit("hipMallocManaged(prefetched to device)", (volatile uint64_t*)p);
    p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) p = nullptr;
    tryit("hipMalloc", (volatile uint64_t*)p);
    return 0;
}
