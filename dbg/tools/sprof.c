/* Poor man's sampling profiler (no perf in the image): LD_PRELOAD=dbg/tools/libsprof.so SPROF_OUT=file program ...
 * SIGPROF every SPROF_US (default 2000) microseconds of PROCESS cpu time, delivered to whichever thread is running; the handler keeps the call chain
 * (up to 16 return addresses).  At exit: /proc/self/maps and the samples go to SPROF_OUT; dbg/tools/sprof_report.py turns them into a flat + caller profile. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <link.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
#define DEPTH 16
#define MAXS (1 << 20)
static void* (*g_buf)[DEPTH];
static volatile long g_n;
static void on_prof(int sig, siginfo_t* si, void* uc)
{
    (void)sig; (void)si; (void)uc;
    long i = __sync_fetch_and_add(&g_n, 1);
    if (i >= MAXS) return;
    void* tmp[DEPTH + 2];
    int n = backtrace(tmp, DEPTH + 2);
    for (int k = 0; k < DEPTH; k++) g_buf[i][k] = k + 2 < n ? tmp[k + 2] : 0;      /* skip the handler and the signal trampoline */
}
static int phdr_cb(struct dl_phdr_info* info, size_t size, void* data)
{
    (void)size;
    unsigned long hi = 0;
    for (int i = 0; i < info->dlpi_phnum; i++)
        if (info->dlpi_phdr[i].p_type == PT_LOAD)
        {
            unsigned long e = info->dlpi_addr + info->dlpi_phdr[i].p_vaddr + info->dlpi_phdr[i].p_memsz;
            if (e > hi) hi = e;
        }
    fprintf((FILE*)data, "M %lx %lx %s\n", (unsigned long)info->dlpi_addr, hi, info->dlpi_name && info->dlpi_name[0] ? info->dlpi_name : "/proc/self/exe");
    return 0;
}
static void dump(void)
{
    struct itimerval z; memset(&z, 0, sizeof(z)); setitimer(ITIMER_PROF, &z, 0);
    const char* out = getenv("SPROF_OUT");
    if (!out) out = "sprof.out";
    if (g_n == 0) return;               /* a wrapper process (timeout, env): leave the file to the program itself */
    FILE* f = fopen(out, "w");
    if (!f) return;
    dl_iterate_phdr(phdr_cb, f);        /* "M <load bias> <end of the highest segment> <path>": pc - bias is the address addr2line wants */
    long n = g_n < MAXS ? g_n : MAXS;
    for (long i = 0; i < n; i++)
    {
        fprintf(f, "S");
        for (int k = 0; k < DEPTH && g_buf[i][k]; k++) fprintf(f, " %lx", (unsigned long)g_buf[i][k]);
        fprintf(f, "\n");
    }
    fclose(f);
}
static void on_segv(int sig, siginfo_t* si, void* uc)
{
    void* bt[48];
    char msg[128];
    int n = snprintf(msg, sizeof(msg), "sprof: signal %d at address %p; call chain:\n", sig, si->si_addr);
    (void)!write(2, msg, n);
    n = backtrace(bt, 48);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
__attribute__((constructor)) static void init(void)
{
    if (getenv("SPROF_SEGV"))
    {
        static char alt[1 << 16];
        stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
        sigaltstack(&ss, 0);           /* the main thread only; other threads report on their own stack (not after an overflow) */
        void* warm[4]; backtrace(warm, 4);
        struct sigaction sa; memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0);
    }
    if (!getenv("SPROF_OUT")) return;
    g_buf = calloc(MAXS, sizeof(*g_buf));
    void* warm[4]; backtrace(warm, 4);          /* loads libgcc now, not inside the handler */
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, 0);
    const char* us = getenv("SPROF_US");
    long u = us ? atol(us) : 2000;
    struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = u; it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, 0);
    atexit(dump);
}
