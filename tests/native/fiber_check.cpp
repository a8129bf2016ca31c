#include "xa_fiber.h"
#include <atomic>
#include <cstdio>
#include <cmath>
#include <thread>
#include <vector>
static thread_local void* tl;
void* xa_scratch_local_swap(void* l) { void* o = tl; tl = l; return o; }
static std::atomic<int> counter{0};
static std::atomic<long> sum{0};
struct A { int i; };
static volatile uint64_t* gate;
static int ready(void* c) { return counter.load() >= ((A*)c)->i; }
static double work(int n) { double x = 0; for (int i = 1; i < n; i++) x += std::sqrt((double)i); return x; }
static void fn(void* c)
{
    A* a = (A*)c;
    double v = work(1000 + a->i);
    // wait for predecessor chain several times
    for (int k = 0; k < 5; k++)
    {
        struct W { int need; } w{ a->i };
        if (k & 1) xa_wait_until([](void* p) -> int { return counter.load() >= ((W*)p)->need; }, &w);
        else xa_wait_counter(gate, (uint64_t)a->i);
        v += work(200);
    }
    sum += (long)v;
    counter.fetch_add(1);
    __atomic_fetch_add((uint64_t*)gate, 1, __ATOMIC_SEQ_CST);
}
int main()
{
    printf("workers %d\n", xa_worker_count());
    for (int round = 0; round < 20; round++)
    {
        counter = 0;
        if (!gate) gate = xa_counter_alloc();
        *gate = 0;
        const int n = 300;
        std::vector<A> args(n); std::vector<XaTask> t(n);
        for (int i = 0; i < n; i++) { args[i].i = i; t[i] = XaTask{ fn, &args[i], (i % 3) ? gate : nullptr, (uint64_t)(i > 2 ? i - 2 : 0), (i & 1) ? ready : nullptr, &args[i], (uint64_t)i }; }
        // two submitters concurrently
        std::thread other([&] { std::vector<A> a2(50); std::vector<XaTask> t2(50); for (int i = 0; i < 50; i++) { a2[i].i = 0; t2[i] = XaTask{ [](void*) { sum += (long)work(5000); }, &a2[i], nullptr, 0, nullptr, nullptr, 1000u + i }; } xa_tasks_run(t2.data(), 50); });
        xa_tasks_run(t.data(), n);
        other.join();
        if (counter.load() != n) { printf("FAIL %d\n", counter.load()); return 1; }
    }
    printf("ok sum %ld\n", sum.load());
    return 0;
}
