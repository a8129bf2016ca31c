/* Row tasks on a fixed set of host threads (host/xa_fiber.cpp).
 *
 * Why: the reference's decision chain is serial per CTU, so a CTU row in flight spends most of its time waiting -- for the device to answer a command
 * (about 15 us each, hundreds per CTU), for the row above, for the reference pictures.  One OS thread per row that spins on those waits needs one core per
 * row in flight; with pictures coded in parallel that is a hundred rows and more, on a host that may grant far fewer cores (the GPU boxes of this project
 * run under a 16-core CPU quota: past it, spinning threads only slow the ones that have work).  Here a row is a TASK with a stack of its own (a fiber);
 * a task that has to wait names the condition and parks, and the worker thread that ran it resumes another task whose condition holds.  The number of
 * worker threads is the number of cores worth using (X265AMD_WORKERS; default: the CPU quota of the cgroup or the hardware threads, minus two for the
 * picture and filter threads), whatever the number of rows in flight; a switch is a few dozen instructions in user space, no system call.
 *
 * Rules for code that runs inside a task: wait only through xa_wait_until (never on a condition variable), and never park while holding a lock -- a task
 * may be resumed by another worker thread.  The calling thread's scratch list (xa_scratch_local_begin) travels with the task.
 */
#ifndef X265AMD_XA_FIBER_H
#define X265AMD_XA_FIBER_H
#include <stdint.h>

typedef int (*XaPred)(void* ctx);          /* non-zero: the condition holds */

/* What a parked task waits for is a COUNTER reaching a value: a 64-bit word that only grows (commands finished by a device queue, CTUs finished in a
 * row, rows of a picture that are final ...), at an address that stays mapped for the life of the process -- the worker threads look at the conditions of
 * all parked tasks without taking them, and a task may have moved on (and its stack with it) by the time another worker looks.  xa_counter_alloc hands out
 * such words for short-lived objects (a frame's row counters); a stale look at a recycled word can only wake a task early, and a woken task checks again. */
volatile uint64_t* xa_counter_alloc(void);          /* zeroed */
void xa_counter_free(volatile uint64_t* c);

struct XaTask
{
    void (*fn)(void* arg); void* arg;
    volatile uint64_t* startCounter; uint64_t startValue;      /* the task may start when *startCounter >= startValue (NULL: no such condition) ... */
    XaPred ready; void* readyCtx;           /* ... and this holds (NULL: at once); evaluated by one worker at a time, valid until the task has run */
    uint64_t priority;                      /* lower runs first when several tasks could: coding order of the picture, then the row */
};

/* runs the n tasks on the worker threads and returns when all of them have finished (the calling thread sleeps meanwhile) */
void xa_tasks_run(const XaTask* tasks, int n);
/* inside a task: parks it until *counter >= value.  On an ordinary thread: polls (pause; after a while short sleeps). */
void xa_wait_counter(const volatile uint64_t* counter, uint64_t value);
/* the same with a time limit: 0 when the counter got there, -1 when `timeoutNs` passed first (a device queue whose server went away) */
int xa_wait_counter_deadline(const volatile uint64_t* counter, uint64_t value, uint64_t timeoutNs);
/* run by every worker thread before its first task (set before the first xa_tasks_run: the workers start there).  HIP's current device is per thread. */
void xa_fiber_set_thread_init(void (*fn)(void));
/* general condition: an ordinary thread polls it; a task is parked and the condition is evaluated by whichever worker holds the task at that moment */
void xa_wait_until(XaPred pred, void* ctx);
int xa_in_task(void);
/* inside a task: from now on its counter waits poll for up to `ns` nanoseconds before the task parks (0: park at once).  For the few tasks everything else waits
 * for -- the cut last CTU row of a picture that others reference: hundreds of device answers per CTU, each a few microseconds away, and a parked task comes back only
 * when a worker is free to look */
void xa_task_spin_ns(uint64_t ns);
int xa_worker_count(void);
/* X265AMD_TIMING: the calling task's waits are booked under a class (0..3; returns the class in force before); xa_task_parked_ns: the totals so far */
int xa_task_wait_class(int cls);
void xa_task_parked_ns(uint64_t out[4]);
/* X265AMD_TIMING: time the calling task has spent running (up to its last resume); totals over all workers: running, looking for a task that can run, switches */
uint64_t xa_task_run_ns(void);
uint64_t xa_task_run_ns_always(void);     /* the same whether or not X265AMD_TIMING is set (X265AMD_HOSTPROF) */
void** xa_task_slot(void);                  /* a pointer slot of the calling task, NULL at its start (NULL outside a task) */
uint64_t* xa_task_mark(void);               /* a word of the calling task for the caller's own bookkeeping (NULL outside a task) */
void xa_sched_stats(uint64_t out[3]);

/* the calling thread's list of device scratch blocks (csrc/table_setup.hip): exchanged when a worker switches tasks */
void* xa_scratch_local_swap(void* list);
#endif
