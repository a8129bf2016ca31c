/* Row tasks on a fixed set of host threads: see csrc/xa_fiber.h.  Host C++ (no reference counterpart: the reference runs its CTU rows as jobs of its own
 * thread pool, source/common/threadpool.cpp / wavefront.cpp, with the pixel work done by the thread itself, so a row never waits for a device). */
#include "../csrc/xa_fiber.h"
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

/* void xa_ctx_switch(void** saveSp, void* loadSp): callee-saved registers, MXCSR and the x87 control word on the stack, then the stack pointers change hands */
extern "C" void xa_ctx_switch(void** saveSp, void* loadSp);
asm(".text\n"
    ".globl xa_ctx_switch\n"
    ".type xa_ctx_switch,@function\n"
    "xa_ctx_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    subq $8, %rsp\n    stmxcsr (%rsp)\n    fnstcw 4(%rsp)\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    ldmxcsr (%rsp)\n    fldcw 4(%rsp)\n    addq $8, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size xa_ctx_switch,.-xa_ctx_switch\n");

namespace {

enum { kStackBytes = 1 << 20, kMaxTasks = 1024 };
enum State { ST_EMPTY = 0, ST_NEW, ST_PARKED, ST_RUNNING, ST_DONE };

struct Group { std::mutex m; std::condition_variable cv; int left = 0; };

struct Fiber
{
    std::atomic<int> state{ ST_EMPTY };
    XaTask task;
    Group* group = nullptr;
    std::atomic<const volatile uint64_t*> waitCounter{ nullptr };      /* parked: resume when *waitCounter >= waitValue (read by every worker) */
    std::atomic<uint64_t> waitValue{ 0 };
    std::atomic<uint64_t> deadlineNs{ 0 };                              /* parked with a time limit: resume (to report the timeout) once the clock has passed this */
    XaPred pred = nullptr; void* predCtx = nullptr;                     /* parked on a general condition: evaluated by the worker that has taken the task */
    void* sp = nullptr;                 /* saved stack pointer while parked */
    char* stack = nullptr;
    void* scratchList = nullptr;        /* the task's device scratch list (travels with it) */
    uint64_t runNs = 0;                 /* X265AMD_TIMING: time spent running (not parked) */
    uint64_t stintStart = 0;            /* when it was last resumed */
    uint64_t userMark = 0;              /* for the caller's phase accounting */
    void* userSlot = nullptr;           /* a pointer of the caller's (the row's reference-picture guard) */
    uint64_t spinNs = 0;                /* xa_task_spin_ns: how long a wait polls before the task parks */
    int waitClass = 0; uint64_t parkedNs[4] = { 0, 0, 0, 0 };      /* X265AMD_TIMING: time spent in waits by the class the caller named (xa_task_wait_class) */
    struct Worker* worker = nullptr;    /* the worker running it now */
};

struct Worker { void* sp = nullptr; Fiber* cur = nullptr; };
std::atomic<uint64_t> g_busyNs{ 0 }, g_idleNs{ 0 }, g_switches{ 0 };
const bool g_stats = getenv("X265AMD_TIMING") != nullptr || getenv("X265AMD_HOSTPROF") != nullptr;
inline uint64_t now_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }

struct Sched
{
    Fiber fibers[kMaxTasks];
    std::atomic<int> live{ 0 };         /* slots not EMPTY */
    std::atomic<int> highWater{ 0 };
    std::mutex m; std::condition_variable cv;          /* workers sleep here while there is no task at all */
    std::mutex stackM; std::vector<char*> stacks;
    std::mutex placeM;                  /* one submitter places its tasks at a time */
    std::vector<std::thread> threads;
    int numWorkers = 0;
    bool started = false;
};
Sched& sched() { static Sched* s = new Sched; return *s; }
thread_local Worker* t_worker = nullptr;

__attribute__((noinline)) Worker* current_worker() { return t_worker; }

int default_workers()
{
    if (const char* e = getenv("X265AMD_WORKERS")) { const int n = atoi(e); if (n > 0) return n > 256 ? 256 : n; }
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 8;
    /* cgroup v2 CPU quota: "<quota> <period>" or "max <period>" */
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
        char q[64]; long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
        {
            const long cpus = (atol(q) + period - 1) / period;
            if (cpus > 0 && cpus < n) n = (int)cpus;
        }
        fclose(f);
    }
    /* several encoder processes on one host (one per GPU: torch.distributed.run sets LOCAL_WORLD_SIZE) share the cores */
    if (const char* l = getenv("LOCAL_WORLD_SIZE")) { const int procs = atoi(l); if (procs > 1) n /= procs; }
    n -= 2;
    return n < 2 ? 2 : (n > 64 ? 64 : n);
}

char* take_stack()
{
    Sched& S = sched();
    {
        std::lock_guard<std::mutex> g(S.stackM);
        if (!S.stacks.empty()) { char* p = S.stacks.back(); S.stacks.pop_back(); return p; }
    }
    void* p = mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (p == MAP_FAILED) { fprintf(stderr, "x265amd: fatal: no memory for a task stack\n"); abort(); }
    (void)mprotect(p, 4096, PROT_NONE);         /* guard page */
    return (char*)p;
}
void give_stack(char* p) { Sched& S = sched(); std::lock_guard<std::mutex> g(S.stackM); S.stacks.push_back(p); }

/* first frame of a task: runs on the task's own stack */
void fiber_main(Fiber* f)
{
    f->task.fn(f->task.arg);
    f->pred = nullptr;
    /* back to the worker for good; the worker recycles the stack */
    Worker* w = f->worker;
    f->state.store(ST_DONE, std::memory_order_release);
    void* dummy;
    xa_ctx_switch(&dummy, w->sp);
    __builtin_unreachable();
}
extern "C" void xa_fiber_entry();
asm(".text\n"
    ".globl xa_fiber_entry\n"
    ".type xa_fiber_entry,@function\n"
    "xa_fiber_entry:\n"
    "    movq %r12, %rdi\n"             /* the Fiber* was placed in the r12 slot of the initial frame */
    "    call xa_fiber_main_c\n"
    "    ud2\n"
    ".size xa_fiber_entry,.-xa_fiber_entry\n");
extern "C" void xa_fiber_main_c(Fiber* f) { fiber_main(f); }

void prepare_stack(Fiber* f)
{
    f->stack = take_stack();
    uintptr_t top = ((uintptr_t)f->stack + kStackBytes) & ~(uintptr_t)15;
    uint64_t* s = (uint64_t*)top;
    *--s = (uint64_t)(uintptr_t)&xa_fiber_entry; /* ret: the entry runs with a 16-byte aligned stack pointer and calls on from there */
    *--s = 0;                                   /* rbp */
    *--s = 0;                                   /* rbx */
    *--s = (uint64_t)(uintptr_t)f;              /* r12 */
    *--s = 0; *--s = 0; *--s = 0;               /* r13 r14 r15 */
    uint32_t csr[2] = { 0x1F80u, 0x037Fu };     /* MXCSR default, x87 control word default */
    *--s = (uint64_t)csr[0] | ((uint64_t)csr[1] << 32);
    f->sp = s;
}

void (*g_threadInit)(void) = nullptr;

void worker_loop()
{
    Sched& S = sched();
    Worker w;
    t_worker = &w;
    if (g_threadInit) g_threadInit();       /* e.g. hipSetDevice: HIP's current device is per thread */
    unsigned idle = 0;
    Fiber* failed[32]; int nFailed = 0;
    uint64_t tIdle0 = 0;
    for (;;)
    {
        Fiber* best = nullptr;
        uint64_t bestPrio = ~0ull, nowCached = 0;
        const int hw = S.highWater.load(std::memory_order_acquire);
        for (int i = 0; i < hw; i++)
        {
            Fiber& f = S.fibers[i];
            const int st = f.state.load(std::memory_order_acquire);
            if (st != ST_NEW && st != ST_PARKED) continue;
            if (f.task.priority >= bestPrio) continue;
            bool skip = false;
            for (int k = 0; k < nFailed; k++) skip |= failed[k] == &f;                     /* its general condition failed a moment ago */
            if (skip) continue;
            /* the counter part of the condition, without taking the task (see xa_fiber.h: the words stay mapped, a stale look wakes early at worst) */
            const volatile uint64_t* c = st == ST_NEW ? f.task.startCounter : f.waitCounter.load(std::memory_order_acquire);
            if (c && *c < (st == ST_NEW ? f.task.startValue : f.waitValue.load(std::memory_order_acquire)))
            {
                const uint64_t dl = st == ST_PARKED ? f.deadlineNs.load(std::memory_order_acquire) : 0;
                if (!dl) continue;
                if (!nowCached) nowCached = now_ns();
                if (nowCached < dl) continue;           /* past its time limit: it runs to report that */
            }
            best = &f; bestPrio = f.task.priority;
        }
        if (!best)
        {
            nFailed = 0;
            if (S.live.load(std::memory_order_acquire) == 0)
            {
                std::unique_lock<std::mutex> lk(S.m);
                S.cv.wait(lk, [&] { return S.live.load(std::memory_order_acquire) != 0; });
                idle = 0; tIdle0 = 0;
                continue;
            }
            /* tasks exist, none can run: the device answers in microseconds, other pictures' rows in milliseconds */
            if (++idle < 2000) { for (int k = 0; k < 8; k++) _mm_pause(); }
            else { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
            continue;
        }
        int expect = best->state.load(std::memory_order_acquire);
        if ((expect != ST_NEW && expect != ST_PARKED) || !best->state.compare_exchange_strong(expect, ST_RUNNING, std::memory_order_acq_rel)) continue;
        /* the task is ours: its general condition, if any, can be looked at now */
        if (expect == ST_NEW ? (best->task.ready && !best->task.ready(best->task.readyCtx)) : (best->pred && !best->pred(best->predCtx)))
        {
            best->state.store(expect, std::memory_order_release);
            if (nFailed < 32) failed[nFailed++] = best;         /* the others get their turn before it is asked again */
            else nFailed = 0;
            continue;
        }
        idle = 0; nFailed = 0;
        if (expect == ST_NEW) prepare_stack(best);
        best->worker = &w; w.cur = best;
        void* mine = xa_scratch_local_swap(best->scratchList);
        const uint64_t t0 = g_stats ? now_ns() : 0;
        if (g_stats && tIdle0) { g_idleNs += t0 - tIdle0; }
        best->stintStart = t0;
        xa_ctx_switch(&w.sp, best->sp);
        /* the task has parked or finished */
        if (g_stats) { const uint64_t t1 = now_ns(); best->runNs += t1 - t0; g_busyNs += t1 - t0; g_switches++; tIdle0 = t1; }
        best->scratchList = xa_scratch_local_swap(mine);
        w.cur = nullptr;
        if (best->state.load(std::memory_order_acquire) == ST_DONE)
        {
            give_stack(best->stack); best->stack = nullptr;
            Group* g = best->group;
            best->state.store(ST_EMPTY, std::memory_order_release);
            S.live.fetch_sub(1, std::memory_order_acq_rel);
            { std::lock_guard<std::mutex> lk(g->m); g->left--; g->cv.notify_all(); }      /* under the lock: the group lives on the submitter's stack */
        }
        else
            best->state.store(ST_PARKED, std::memory_order_release);        /* only now may another worker take it: its registers are on its stack */
    }
}

void start_workers()
{
    Sched& S = sched();
    std::lock_guard<std::mutex> g(S.m);
    if (S.started) return;
    S.numWorkers = default_workers();
    for (int i = 0; i < S.numWorkers; i++) { S.threads.emplace_back(worker_loop); S.threads.back().detach(); }
    S.started = true;
}

} // namespace

/* ---- counters: words that stay mapped for good ---- */
namespace {
enum { kCounters = 1 << 16 };
struct CounterPool { volatile uint64_t* words; std::mutex m; std::vector<int> freeList; int next = 0; };
CounterPool& counters() { static CounterPool* p = [] { CounterPool* q = new CounterPool; q->words = (volatile uint64_t*)calloc(kCounters, sizeof(uint64_t)); return q; }(); return *p; }
}
volatile uint64_t* xa_counter_alloc(void)
{
    CounterPool& P = counters();
    std::lock_guard<std::mutex> g(P.m);
    int i;
    if (!P.freeList.empty()) { i = P.freeList.back(); P.freeList.pop_back(); }
    else if (P.next < kCounters) i = P.next++;
    else { fprintf(stderr, "x265amd: fatal: out of task counters\n"); abort(); }
    P.words[i] = 0;
    return P.words + i;
}
void xa_counter_free(volatile uint64_t* c)
{
    if (!c) return;
    CounterPool& P = counters();
    std::lock_guard<std::mutex> g(P.m);
    P.freeList.push_back((int)(c - P.words));
}

int xa_worker_count(void) { start_workers(); return sched().numWorkers; }
uint64_t xa_task_run_ns(void) { Worker* w = current_worker(); return w && w->cur ? w->cur->runNs + (g_stats ? now_ns() - w->cur->stintStart : 0) : 0; }
void** xa_task_slot(void) { Worker* w = current_worker(); return w && w->cur ? &w->cur->userSlot : nullptr; }
uint64_t xa_task_run_ns_always(void) { Worker* w = current_worker(); return w && w->cur ? w->cur->runNs + (now_ns() - w->cur->stintStart) : 0; }
uint64_t* xa_task_mark(void) { Worker* w = current_worker(); return w && w->cur ? &w->cur->userMark : nullptr; }
void xa_sched_stats(uint64_t out[3]) { out[0] = g_busyNs.load(); out[1] = g_idleNs.load(); out[2] = g_switches.load(); }
int xa_in_task(void) { Worker* w = current_worker(); return w && w->cur; }
void xa_task_spin_ns(uint64_t ns) { Worker* w = current_worker(); if (w && w->cur) w->cur->spinNs = ns; }
int xa_task_wait_class(int cls) { Worker* w = current_worker(); if (!(w && w->cur)) return 0; const int old = w->cur->waitClass; w->cur->waitClass = cls & 3; return old; }
void xa_task_parked_ns(uint64_t out[4]) { Worker* w = current_worker(); for (int k = 0; k < 4; k++) out[k] = w && w->cur ? w->cur->parkedNs[k] : 0; }

void xa_tasks_run(const XaTask* tasks, int n)
{
    if (n <= 0) return;
    start_workers();
    Sched& S = sched();
    Group grp;
    grp.left = n;
    if (n > kMaxTasks) { fprintf(stderr, "x265amd: fatal: %d tasks in one submission (limit %d)\n", n, (int)kMaxTasks); abort(); }
    /* all n slots or none: rows of a later picture wait for rows of an earlier one, so a picture placed in part (the table full of later pictures' rows that
     * cannot start) would never get its remaining rows in -- wait until the whole picture fits, then place it without another submitter in between */
    std::unique_lock<std::mutex> place(S.placeM);
    while (kMaxTasks - S.live.load(std::memory_order_acquire) < n) { struct timespec ts = { 0, 1000000 }; nanosleep(&ts, nullptr); }
    for (int k = 0; k < n; k++)
    {
        /* a free slot (several pictures submit concurrently) */
        for (;;)
        {
            bool placed = false;
            for (int i = 0; i < kMaxTasks && !placed; i++)
            {
                Fiber& f = S.fibers[i];
                int expect = ST_EMPTY;
                if (f.state.load(std::memory_order_acquire) != ST_EMPTY || !f.state.compare_exchange_strong(expect, ST_RUNNING, std::memory_order_acq_rel)) continue;
                f.task = tasks[k]; f.group = &grp; f.pred = nullptr; f.predCtx = nullptr; f.sp = nullptr; f.stack = nullptr; f.scratchList = nullptr; f.worker = nullptr;
                f.waitCounter.store(nullptr); f.waitValue.store(0); f.deadlineNs.store(0); f.runNs = 0; f.userMark = 0; f.userSlot = nullptr; f.spinNs = 0;
                int hw = S.highWater.load(std::memory_order_acquire);
                while (hw < i + 1 && !S.highWater.compare_exchange_weak(hw, i + 1, std::memory_order_acq_rel)) {}
                S.live.fetch_add(1, std::memory_order_acq_rel);
                f.state.store(ST_NEW, std::memory_order_release);
                placed = true;
            }
            if (placed) break;
            struct timespec ts = { 0, 1000000 }; nanosleep(&ts, nullptr);         /* all slots taken: wait for a task to finish */
        }
    }
    place.unlock();
    { std::lock_guard<std::mutex> lk(S.m); }
    S.cv.notify_all();
    std::unique_lock<std::mutex> lk(grp.m);
    grp.cv.wait(lk, [&] { return grp.left == 0; });
}

void xa_wait_counter(const volatile uint64_t* counter, uint64_t value)
{
    if (*counter >= value) return;
    Worker* w = current_worker();
    if (w && w->cur)
    {
        Fiber* f = w->cur;
        f->pred = nullptr;
        const uint64_t tw0 = g_stats ? now_ns() : 0;
        if (f->spinNs) { const uint64_t t1 = now_ns() + f->spinNs; while (*counter < value && now_ns() < t1) _mm_pause(); if (*counter >= value) { if (g_stats) f->parkedNs[f->waitClass] += now_ns() - tw0; return; } }
        do
        {
            f->waitValue.store(value, std::memory_order_release); f->waitCounter.store(counter, std::memory_order_release);
            Worker* on = f->worker;
            xa_ctx_switch(&f->sp, on->sp);          /* parks; resumed (maybe by another worker) when the counter was seen to have reached the value */
        } while (*counter < value);
        f->waitCounter.store(nullptr, std::memory_order_release);
        if (g_stats) f->parkedNs[f->waitClass] += now_ns() - tw0;
        return;
    }
    for (unsigned spins = 0; *counter < value; spins++)
    {
        if (spins < 20000) _mm_pause();
        else { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
    }
}

int xa_wait_counter_deadline(const volatile uint64_t* counter, uint64_t value, uint64_t timeoutNs)
{
    if (*counter >= value) return 0;
    const uint64_t deadline = now_ns() + timeoutNs;
    Worker* w = current_worker();
    if (w && w->cur)
    {
        Fiber* f = w->cur;
        f->pred = nullptr;
        int rc = 0;
        const uint64_t tw0 = g_stats ? now_ns() : 0;
        if (f->spinNs) { const uint64_t t1 = now_ns() + f->spinNs; while (*counter < value && now_ns() < t1) _mm_pause(); if (*counter >= value) { if (g_stats) f->parkedNs[f->waitClass] += now_ns() - tw0; return 0; } }
        do
        {
            f->deadlineNs.store(deadline, std::memory_order_release);
            f->waitValue.store(value, std::memory_order_release); f->waitCounter.store(counter, std::memory_order_release);
            Worker* on = f->worker;
            xa_ctx_switch(&f->sp, on->sp);
            if (*counter < value && now_ns() >= deadline) { rc = -1; break; }
        } while (*counter < value);
        f->waitCounter.store(nullptr, std::memory_order_release);
        f->deadlineNs.store(0, std::memory_order_release);
        if (g_stats) f->parkedNs[f->waitClass] += now_ns() - tw0;
        return rc;
    }
    for (unsigned spins = 0; *counter < value; spins++)
    {
        if (spins < 20000) _mm_pause();
        else { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); if (now_ns() >= deadline) return *counter >= value ? 0 : -1; }
    }
    return 0;
}

void xa_fiber_set_thread_init(void (*fn)(void)) { g_threadInit = fn; }

void xa_wait_until(XaPred pred, void* ctx)
{
    if (pred(ctx)) return;
    Worker* w = current_worker();
    if (w && w->cur)
    {
        Fiber* f = w->cur;
        f->waitCounter.store(nullptr, std::memory_order_release);
        do
        {
            f->pred = pred; f->predCtx = ctx;
            Worker* on = f->worker;
            xa_ctx_switch(&f->sp, on->sp);          /* parks; the worker that takes the task next evaluates the condition before resuming it */
        } while (!pred(ctx));
        f->pred = nullptr;
        return;
    }
    for (unsigned spins = 0; !pred(ctx); spins++)
    {
        if (spins < 20000) _mm_pause();
        else { struct timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
    }
}
