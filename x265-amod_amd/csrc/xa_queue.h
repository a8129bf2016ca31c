/* Device job queues: the records shared by the host side (device_queue.hip: xa_q_*) and the resident server kernel (k_job_server).
 *
 * Why: the reference's mode decision is a serial chain of small block operations per CTU (analysis.cpp:1146-1848); run as kernel launches, every link
 * costs a launch and a stream synchronisation (about 11 us uncontended on MI355X, 30-45 us when the CTU rows of several pictures launch from their own
 * threads and serialise inside the HIP runtime).  A queue replaces launch + synchronise with a 128-byte command the host thread stores straight into
 * device memory (the BAR mapping of a fine-grained allocation) and a resident workgroup that polls for it in its own memory; completion comes back as
 * a store into pinned host memory the host thread polls.  Both sides poll local memory and write remote memory, so a round trip is two posted PCIe
 * writes: 3.4 us measured (dbg/mbox), with no runtime call in between and no lock shared by the rows.
 *
 * One queue = one workgroup of XA_SERVER_WAVES wavefronts bound to one host thread (a CTU row in flight); commands run in order, like a stream.  The
 * kernels' wave-level bodies (tu_dev.h, intra_dev.h, mc_dev.h, me_dev.h, measure_dev.h) are shared between the ordinary kernels and the server.
 */
#ifndef X265AMD_XA_QUEUE_H
#define X265AMD_XA_QUEUE_H
#include <stdint.h>

enum XaOp
{
    XA_OP_NOP = 0, XA_OP_EXIT, XA_OP_COPY, XA_OP_COPY2D, XA_OP_FILL, XA_OP_COPY_RECTS, XA_OP_MC, XA_OP_MC_COST, XA_OP_CU_MEASURE, XA_OP_TU_CHAIN, XA_OP_TU_CHAIN_RDOQ,
    XA_OP_INTRA_TU_CHAIN, XA_OP_INTRA_TU_CHAIN_RDOQ, XA_OP_INTRA_SCAN, XA_OP_ME_SEARCH, XA_OP_ME_SEARCH_STAR, XA_OP_ME_DEFERRED, XA_OP_EST_BIT, XA_OP_INTRA_PU, XA_OP_INTRA_NXN, XA_OP_INTER_CHAIN, XA_OP_INTER_SEARCH, XA_OP_WAIT, XA_OP_COUNT
};
enum
{
    XA_CMD_ACQUIRE = 1,     /* before the command: make other workgroups' / kernels' / copy engines' writes visible (agent-scope acquire) */
    XA_CMD_RELEASE = 2,     /* after the command: make this workgroup's device-memory writes visible to them (agent-scope release) */
    XA_CMD_SIGNAL = 4,      /* after the command: publish the count of finished commands to the host */
    XA_CMD_RESET = 8        /* before the command: a new owner takes the queue -- what an XA_OP_WAIT that gave up left behind (the workgroup's flag, XaRingHost::fault) is the last owner's */
};

#define XA_RING 64              /* commands per queue ring */
#define XA_CMD_ARG_WORDS 13
#define XA_SERVER_WAVES 8
#define XA_SERVER_LDS (144 * 1024)

/* `check` closes the command: the sum of its other fifteen words, each times its own odd constant (XA_CHECK_MUL * (2 i + 1)), xor XA_CHECK_MUL * (number
 * of this command, from 1) xor the server generation.  The workgroup polls the slot itself: the host's stores reach device memory through write-combining
 * buffers in no particular order, so a slot may be seen half old, half new -- it is read again until the check holds.  The weights matter: with a plain
 * xor, two address words that changed by the same bits cancelled and a half-arrived copy command ran with the previous addresses. */
struct alignas(128) XaCmd { uint32_t op, flags, count, reserved; uint64_t args[XA_CMD_ARG_WORDS]; uint64_t check; };
#define XA_CHECK_MUL 0x9E3779B97F4A7C15ull

/* device memory (fine-grained), written by the host through the BAR, polled by the workgroup */
struct alignas(128) XaRingDev
{
    XaCmd cmd[XA_RING];
    uint64_t head; uint64_t pad0[15];           /* the doorbell: the number of the command the host waits for (0: none); read with the next slot */
    uint64_t quit; uint64_t pad1[15];
    uint64_t done; uint64_t pad2[15];           /* written by the workgroup: commands finished (behind a release when the command asked for one): what another queue's
                                                   XA_OP_WAIT polls -- an order between two queues without the host in between (xa_queue_follow) */
};
/* pinned host memory, written by the workgroup, polled by the host thread */
struct alignas(128) XaRingHost
{
    uint64_t tail; uint64_t pad0[15];           /* commands finished, as of the last signalling command */
    uint64_t state;                             /* 1 while the workgroup is resident */
    uint64_t alive;                             /* hosts[0] only: bumped by the host while queues are in use; an idle workgroup leaves only when this has stood still */
    uint64_t fault;                             /* sticky: set by the workgroup when an XA_OP_WAIT gave up (the queue it follows stood for two seconds); the commands
                                                   behind it are then NOT run (they would read what the leader had not written yet), and every host wait on this queue fails */
    uint64_t pad1[13];
    uint64_t dbg[64];                           /* X265AMD_QUEUE_DEBUG & 2: what each wavefront was about to touch (dumped on abort) */
    uint64_t prof[64];                          /* per command kind [2 * op] count, [2 * op + 1] ticks of the 100 MHz clock (written when the workgroup leaves; op < 31);
                                                   [62] ticks spent polling, [63] ticks in fences */
    uint64_t bytes[32];                         /* algorithmic bytes per command kind (what each command has to read and write, from its job records: see
                                                   DESIGN.md section 5); always counted, written when the workgroup leaves */
    uint64_t resident;                          /* ticks of the 100 MHz clock between the workgroup's start and its exit, summed over server generations */
    uint64_t cycles;                            /* shader clock cycles over the same time (s_memtime): cycles / resident x 100 MHz is the clock the workgroup's CU ran at */
    uint64_t pad2[14];
    uint64_t nxn[40];                           /* X265AMD_QUEUE_PROF: the fused intra command by kind (four 4x4 units / one unit of 8 / 16 / 32) x stage: ticks */
    uint64_t stage[24];                         /* X265AMD_QUEUE_PROF: the stages of the transform chains and the fused intra steps (ticks) */
    uint64_t sized[24];                         /* X265AMD_QUEUE_DEBUG & 16: single-job commands of the hot kinds by block size: [2 b] count, [2 b + 1] ticks, b = 0..10 */
    uint64_t chain[8];                          /* X265AMD_QUEUE_PROF: chained 8x8 CUs, ticks: [0] the deciding command waiting for the chain, [1] the other command waiting for
                                                   the chain, [2] waiting for the other evaluation, [3] its record + both CUs' bits, [4] costs + the winner's samples + result, [5] publishing */
};

struct XaArgsJobs4 { uint64_t a, b, c, d; int32_t n; };      /* the (jobs, second array, results, extra, count) shapes of the job-list kernels */
struct XaArgsCopy { uint64_t dst, src, bytes; uint32_t hostDst; };
struct XaArgsCopy2D { uint64_t dst, src, dpitch, spitch, width, height; };
struct XaArgsFill { uint64_t dst, bytes; uint32_t value; };
struct XaArgsWait { uint64_t word, target; };             /* XA_OP_WAIT: until *word >= target (another queue's `done`), then an acquire */
struct XaArgsRects { uint64_t dst[4], src[4]; int16_t dst_stride[4], src_stride[4], w[4], h[4]; int32_t n; };     /* = XaRects (x265amd_host.h): 100 bytes */

#endif
