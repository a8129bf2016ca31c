/* What one wavefront alone on a SIMD pays per instruction (gfx950): dependent and independent VALU chains, 64-bit adds, v_mad_u64_u32, SALU chains, VALU -> SALU -> VALU
 * round trips (v_cmp + s_and_saveexec), readlane, DPP, ds_bpermute, LDS read after write.  A developer tool (dbg/README.md).
 *   hipcc --offload-arch=gfx950 -O3 dbg/issue_rate.hip -o dbg/bin/issue_rate && dbg/bin/issue_rate */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 256
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
__global__ __launch_bounds__(64) void k(unsigned long long* out, int* sink, int seed)
{
    __shared__ int lds[256];
    int a = threadIdx.x + seed, b = seed * 3, c = seed * 5, d = seed * 7;
    long long w = seed; 
    unsigned long long t0, t1;
    int test = 0;
#define T_BEGIN t0 = __builtin_readcyclecounter(); asm volatile("s_nop 0" ::: "memory");
#define T_END t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) out[test] = t1 - t0; test++;
    T_BEGIN REP256(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) T_END                                   /* 0 dependent v_add */
    T_BEGIN REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) T_END  /* 1 independent x4 */
    T_BEGIN REP256(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"((long long)b));) T_END                /* 2 dependent 64-bit add */
    T_BEGIN REP256(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(a), "v"(b) : "vcc");) T_END      /* 3 dependent mad64 */
    { int s = seed; T_BEGIN REP256(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(seed) : "scc");) T_END a += s; }   /* 4 dependent s_add */
    T_BEGIN REP256(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");) T_END   /* 5 cmp+saveexec+add+restore */
    T_BEGIN REP256(asm volatile("v_readlane_b32 s20, %0, 3\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0" : "+v"(a) :: "s20", "scc");) T_END     /* 6 readlane -> salu -> valu */
    T_BEGIN REP256(asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));) T_END           /* 7 dependent dpp */
    T_BEGIN REP256(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b));) T_END             /* 8 dependent bpermute */
    { int addr = (threadIdx.x & 63) * 4; T_BEGIN REP256(asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr) : "memory");) T_END }   /* 9 LDS write+read */
    T_BEGIN REP256(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) T_END                  /* 10 dependent cndmask */
    T_BEGIN REP256(asm volatile("v_cmp_lt_i64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(w), "v"((long long)c), "v"(a), "v"(b) : "vcc");) T_END   /* 11 cmp64 + cndmask */
    T_BEGIN REP256(asm volatile("s_cbranch_scc0 1f\n s_nop 0\n1:\n s_cmp_eq_u32 %0, 12345" :: "s"(seed) : "scc");) T_END        /* 12 branch (maybe taken) + s_cmp */
    T_BEGIN REP256(asm volatile("s_branch 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n1:\n" ::: "memory");) T_END   /* 13 taken branch over 16 instrs */
    T_BEGIN REP256(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) T_END                                /* 14 dependent mul_lo */
    T_BEGIN REP256(asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w));) T_END                                          /* 15 dependent 64-bit shift */
    lds[threadIdx.x] = a; __syncthreads();
    sink[threadIdx.x] = a + b + c + d + (int)w + lds[(threadIdx.x + 1) & 63];
}
int main()
{
    unsigned long long* out; int* sink; hipMalloc(&out, 64 * 8); hipMalloc(&sink, 256);
    hipMemset(out, 0, 64 * 8);
    for (int i = 0; i < 2; i++) { k<<<1, 64>>>(out, sink, 3 + i); hipDeviceSynchronize(); }
    unsigned long long h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = { "dependent v_add_u32", "4 independent v_add_u32 (per instruction)", "dependent v_lshl_add_u64", "dependent v_mad_u64_u32", "dependent s_add_u32",
        "v_cmp + s_and_saveexec + v_add + s_or exec (per group of 4)", "v_readlane + s_add + v_add (per group of 3)", "dependent v_add_u32_dpp", "dependent ds_bpermute + wait",
        "ds_write + ds_read + wait", "dependent v_cndmask", "v_cmp_lt_i64 + v_cndmask (per pair)", "s_cbranch_scc0 (not taken / taken) + s_cmp", "s_branch over 17 instructions",
        "dependent v_mul_lo_u32", "dependent v_lshrrev_b64" };
    for (int t = 0; t < 16; t++) printf("%-62s %7.1f cycles\n", names[t], (double)h[t] / (t == 1 ? 256.0 : 256.0));
    return 0;
}
