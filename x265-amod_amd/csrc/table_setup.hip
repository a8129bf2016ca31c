/* x265amd_setup_primitives(): installs the HIP-backed entries into a table laid out like the reference's
 * `EncoderPrimitives` -- the counterpart of setupAssemblyPrimitives(EncoderPrimitives&, int)
 * (reference: source/common/primitives.h:474, called from x265_setup_primitives(), primitives.cpp:248-285).
 *
 * The reference's slots carry no size argument (one function per block size), so each installed entry is a
 * thunk instantiated per size index that forwards to the layer-1 entry point with the exact typedef signature
 * of the slot (primitives.h:133-236).  Slots that are not on the north-star path are left untouched.
 */
#include <utility>
#include "x265amd_host.h"
#include "xa_fiber.h"
#include "xa_queue.h"
#include "../host/primitive_table.h"

using namespace x265amd;
typedef x265amd_pixel pixel;
enum { CSP420 = 1 };

#if X265AMD_DEPTH > 8
typedef uint64_t sse_t;     /* reference: common/common.h:142-146 */
#else
typedef uint32_t sse_t;
#endif

namespace {

template<int I> struct Thunk
{
    /* pu[I] */
    static int sad(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_sad(I, a, sa, b, sb); }
    static void sad_x3(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res) { x265amd_sad_x3(I, f, r0, r1, r2, rs, res); }
    static void sad_x4(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res) { x265amd_sad_x4(I, f, r0, r1, r2, r3, rs, res); }
    static int satd(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_satd(I, a, sa, b, sb); }
    static void luma_hpp(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_luma_hpp(I, s, ss, d, ds, c); }
    static void luma_hps(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c, int e) { x265amd_luma_hps(I, s, ss, d, ds, c, e); }
    static void luma_vpp(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_luma_vpp(I, s, ss, d, ds, c); }
    static void luma_vps(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c) { x265amd_luma_vps(I, s, ss, d, ds, c); }
    static void luma_vsp(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_luma_vsp(I, s, ss, d, ds, c); }
    static void luma_vss(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int c) { x265amd_luma_vss(I, s, ss, d, ds, c); }
    static void luma_hvpp(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int x, int y) { x265amd_luma_hvpp(I, s, ss, d, ds, x, y); }
    static void pixelavg_pp(pixel* d, intptr_t ds, const pixel* s0, intptr_t ss0, const pixel* s1, intptr_t ss1, int) { x265amd_pixelavg_pp(I, d, ds, s0, ss0, s1, ss1); }
    static void addAvg(const int16_t* s0, const int16_t* s1, pixel* d, intptr_t ss0, intptr_t ss1, intptr_t ds) { x265amd_addAvg(I, s0, s1, d, ss0, ss1, ds); }
    static void p2s(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds) { x265amd_luma_p2s(I, s, ss, d, ds); }
    /* chroma[420].pu[I] */
    static int c_satd(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_chroma_satd(CSP420, I, a, sa, b, sb); }
    static void c_hpp(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_chroma_hpp(CSP420, I, s, ss, d, ds, c); }
    static void c_hps(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c, int e) { x265amd_chroma_hps(CSP420, I, s, ss, d, ds, c, e); }
    static void c_vpp(const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_chroma_vpp(CSP420, I, s, ss, d, ds, c); }
    static void c_vps(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int c) { x265amd_chroma_vps(CSP420, I, s, ss, d, ds, c); }
    static void c_vsp(const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int c) { x265amd_chroma_vsp(CSP420, I, s, ss, d, ds, c); }
    static void c_vss(const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int c) { x265amd_chroma_vss(CSP420, I, s, ss, d, ds, c); }
    static void c_addAvg(const int16_t* s0, const int16_t* s1, pixel* d, intptr_t ss0, intptr_t ss1, intptr_t ds) { x265amd_chroma_addAvg(CSP420, I, s0, s1, d, ss0, ss1, ds); }
    static void c_p2s(const pixel* s, intptr_t ss, int16_t* d, intptr_t ds) { x265amd_chroma_p2s(CSP420, I, s, ss, d, ds); }
    /* cu[I] */
    static void dct(const int16_t* s, int16_t* d, intptr_t st) { x265amd_dct(I, s, d, st); }
    static void idct(const int16_t* s, int16_t* d, intptr_t st) { x265amd_idct(I, s, d, st); }
    static void sub_ps(int16_t* d, intptr_t ds, const pixel* s0, const pixel* s1, intptr_t ss0, intptr_t ss1) { x265amd_sub_ps(I, d, ds, s0, s1, ss0, ss1); }
    static void add_ps(pixel* d, intptr_t ds, const pixel* s0, const int16_t* s1, intptr_t ss0, intptr_t ss1) { x265amd_add_ps(I, d, ds, s0, s1, ss0, ss1); }
    static uint32_t copy_cnt(int16_t* c, const int16_t* r, intptr_t rs) { return x265amd_copy_cnt(I, c, r, rs); }
    static int count_nonzero(const int16_t* q) { return x265amd_count_nonzero(I, q); }
    static void cpy2Dto1D_shl(int16_t* d, const int16_t* s, intptr_t ss, int sh) { x265amd_cpy2Dto1D_shl(I, d, s, ss, sh); }
    static void cpy2Dto1D_shr(int16_t* d, const int16_t* s, intptr_t ss, int sh) { x265amd_cpy2Dto1D_shr(I, d, s, ss, sh); }
    static void cpy1Dto2D_shl(int16_t* d, const int16_t* s, intptr_t ds, int sh) { x265amd_cpy1Dto2D_shl(I, d, s, ds, sh); }
    static void cpy1Dto2D_shr(int16_t* d, const int16_t* s, intptr_t ds, int sh) { x265amd_cpy1Dto2D_shr(I, d, s, ds, sh); }
    static uint64_t var(const pixel* p, intptr_t st) { return x265amd_var(I, p, st); }
    static sse_t sse_pp(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return (sse_t)x265amd_sse_pp(I, a, sa, b, sb); }
    static sse_t sse_ss(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { return (sse_t)x265amd_sse_ss(I, a, sa, b, sb); }
    static sse_t ssd_s(const int16_t* a, intptr_t sa) { return (sse_t)x265amd_ssd_s(I, a, sa); }
    static int psy_cost_pp(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_psy_cost_pp(I, a, sa, b, sb); }
    static int sa8d(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_sa8d(I, a, sa, b, sb); }
    static void transpose(pixel* d, const pixel* s, intptr_t st) { x265amd_transpose(I, d, s, st); }
    static void allangs(pixel* d, pixel* r, pixel* f, int l) { x265amd_intra_allangs(I, d, r, f, l); }
    static void intra_filter(const pixel* r, pixel* f) { x265amd_intra_filter(I, r, f); }
    /* the reference's planar / DC functions ignore dirMode (intrapred.cpp:70,88) and its callers pass 0 for both
     * (search.cpp:1358,1369): the SLOT decides the mode there; only the angular slots read the argument (intrapred.cpp:103) */
    static void intra_planar(pixel* d, intptr_t ds, const pixel* s, int, int bf) { x265amd_intra_pred(I, 0, d, ds, s, bf); }
    static void intra_dc(pixel* d, intptr_t ds, const pixel* s, int, int bf) { x265amd_intra_pred(I, 1, d, ds, s, bf); }
    static void intra_ang(pixel* d, intptr_t ds, const pixel* s, int mode, int bf) { x265amd_intra_pred(I, mode, d, ds, s, bf); }
    /* chroma[420].cu[I] */
    static int c_sa8d(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return x265amd_chroma_sa8d(CSP420, I, a, sa, b, sb); }
};

/* loose members */
static void t_dst4x4(const int16_t* s, int16_t* d, intptr_t st) { x265amd_dst4x4(s, d, st); }
static void t_idst4x4(const int16_t* s, int16_t* d, intptr_t st) { x265amd_idst4x4(s, d, st); }
static void t_scale1D(pixel* d, const pixel* s) { x265amd_scale1D_128to64(d, s); }

struct Installer
{
    generic_fn* t;
    int n = 0;
    template<class F> void set(int slot, F f) { t[slot] = reinterpret_cast<generic_fn>(f); n++; }

    template<int I> void pu()
    {
        typedef Thunk<I> T;
        set(slotPU(I, PU_sad), &T::sad); set(slotPU(I, PU_sad_x3), &T::sad_x3); set(slotPU(I, PU_sad_x4), &T::sad_x4);
        set(slotPU(I, PU_satd), &T::satd);
        set(slotPU(I, PU_luma_hpp), &T::luma_hpp); set(slotPU(I, PU_luma_hps), &T::luma_hps); set(slotPU(I, PU_luma_vpp), &T::luma_vpp);
        set(slotPU(I, PU_luma_vps), &T::luma_vps); set(slotPU(I, PU_luma_vsp), &T::luma_vsp); set(slotPU(I, PU_luma_vss), &T::luma_vss);
        set(slotPU(I, PU_luma_hvpp), &T::luma_hvpp);
        for (int al = 0; al < 2; al++)
        {
            set(slotPU(I, PU_pixelavg_pp + al), &T::pixelavg_pp); set(slotPU(I, PU_addAvg + al), &T::addAvg);
            set(slotPU(I, PU_convert_p2s + al), &T::p2s);
            set(slotChromaPU(CSP420, I, CPU_addAvg + al), &T::c_addAvg);
            if (I != 0) set(slotChromaPU(CSP420, I, CPU_p2s + al), &T::c_p2s);
        }
        if (I != 0)     /* no 2x2 chroma filters (reference: ipfilter.cpp:418-466) */
        {
            set(slotChromaPU(CSP420, I, CPU_filter_hpp), &T::c_hpp); set(slotChromaPU(CSP420, I, CPU_filter_hps), &T::c_hps);
            set(slotChromaPU(CSP420, I, CPU_filter_vpp), &T::c_vpp); set(slotChromaPU(CSP420, I, CPU_filter_vps), &T::c_vps);
            set(slotChromaPU(CSP420, I, CPU_filter_vsp), &T::c_vsp); set(slotChromaPU(CSP420, I, CPU_filter_vss), &T::c_vss);
        }
        if (x265amd_chroma_satd_defined(I))
            set(slotChromaPU(CSP420, I, CPU_satd), &T::c_satd);
    }
    template<int I> void cu()
    {
        typedef Thunk<I> T;
        if constexpr (I < 4)      /* TU sizes 4..32 */
        {
            set(slotCU(I, CU_dct), &T::dct); set(slotCU(I, CU_idct), &T::idct); set(slotCU(I, CU_standard_dct), &T::dct);
            set(slotCU(I, CU_copy_cnt), &T::copy_cnt); set(slotCU(I, CU_count_nonzero), &T::count_nonzero);
            set(slotCU(I, CU_cpy2Dto1D_shl), &T::cpy2Dto1D_shl); set(slotCU(I, CU_cpy2Dto1D_shr), &T::cpy2Dto1D_shr);
            set(slotCU(I, CU_cpy1Dto2D_shl), &T::cpy1Dto2D_shl); set(slotCU(I, CU_cpy1Dto2D_shl + 1), &T::cpy1Dto2D_shl);
            set(slotCU(I, CU_cpy1Dto2D_shr), &T::cpy1Dto2D_shr);
            set(slotCU(I, CU_intra_pred_allangs), &T::allangs); set(slotCU(I, CU_intra_filter), &T::intra_filter);
            set(slotCU(I, CU_intra_pred + 0), &T::intra_planar); set(slotCU(I, CU_intra_pred + 1), &T::intra_dc);
            for (int m = 2; m < INTRA_MODES; m++) set(slotCU(I, CU_intra_pred + m), &T::intra_ang);
        }
        set(slotCU(I, CU_sub_ps), &T::sub_ps); set(slotCU(I, CU_add_ps), &T::add_ps); set(slotCU(I, CU_add_ps + 1), &T::add_ps);
        set(slotCU(I, CU_var), &T::var); set(slotCU(I, CU_sse_pp), &T::sse_pp); set(slotCU(I, CU_sse_ss), &T::sse_ss);
        set(slotCU(I, CU_ssd_s), &T::ssd_s); set(slotCU(I, CU_ssd_s + 1), &T::ssd_s);
        set(slotCU(I, CU_psy_cost_pp), &T::psy_cost_pp); set(slotCU(I, CU_sa8d), &T::sa8d); set(slotCU(I, CU_transpose), &T::transpose);
        if constexpr (I >= 1) set(slotChromaCU(CSP420, I, CCU_sa8d), &T::c_sa8d);
    }
    static bool x265amd_chroma_satd_defined(int part)
    {
        static const uint8_t w[25] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 12, 16, 4, 32, 24, 32, 8, 64, 48, 64, 16 };
        static const uint8_t h[25] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 12, 16, 4, 16, 24, 32, 8, 32, 48, 64, 16, 64 };
        return (((w[part] >> 1) | (h[part] >> 1)) & 3) == 0;
    }
    template<int... I> void allPU(std::integer_sequence<int, I...>) { (pu<I>(), ...); }
    template<int... I> void allCU(std::integer_sequence<int, I...>) { (cu<I>(), ...); }
};

} // namespace

extern "C" size_t x265amd_primitives_table_bytes(void) { return (size_t)TOTAL_SLOTS * sizeof(generic_fn); }

extern "C" int x265amd_setup_primitives(void* table, size_t table_bytes)
{
    if (!table || table_bytes != x265amd_primitives_table_bytes())
        return xa_fail(X265AMD_EINVAL, "x265amd_setup_primitives: table size does not match the EncoderPrimitives layout this library mirrors");
    if (x265amd_device_count() < 1)
        return xa_fail(X265AMD_EHIP, "x265amd_setup_primitives: no HIP device visible (there is no CPU fallback)");
    Installer ins{ (generic_fn*)table };
    ins.allPU(std::make_integer_sequence<int, NUM_PU_SIZES>());
    ins.allCU(std::make_integer_sequence<int, NUM_CU_SIZES>());
    ins.set(slotMisc(M_dst4x4), &t_dst4x4); ins.set(slotMisc(M_idst4x4), &t_idst4x4);
    ins.set(slotMisc(M_quant), &x265amd_quant); ins.set(slotMisc(M_nquant), &x265amd_nquant);
    ins.set(slotMisc(M_dequant_scaling), &x265amd_dequant_scaling); ins.set(slotMisc(M_dequant_normal), &x265amd_dequant_normal);
    ins.set(slotMisc(M_scale1D_128to64), &t_scale1D); ins.set(slotMisc(M_scale1D_128to64 + 1), &t_scale1D);
    ins.set(slotMisc(M_scale2D_64to32), &x265amd_scale2D_64to32);
    ins.set(slotMisc(M_weight_sp), &x265amd_weight_sp); ins.set(slotMisc(M_weight_pp), &x265amd_weight_pp);
    return ins.n;
}


/* ---- device scratch and mapped-record pools (x265amd_host.h) ----
 * Three pools of blocks in power-of-two size classes: device scratch, records the host writes and the device reads (device memory through the BAR), records the
 * device writes and the host reads (pinned host memory).  The CTU rows call these dozens of times per CU from as many threads as the host grants cores, so the hot
 * path takes no lock: a row task (it holds a device job queue: xa_scratch_local_begin) allocates from and frees into lists of its own, and a block's size class is
 * found in a lock-free table.  The shared pools behind a mutex serve the first allocations, the threads without lists (picture / filter threads) and take the
 * lists back when the queue is released. */
#include <atomic>
#include <map>
#include <mutex>
#include <vector>
namespace {
enum { POOL_SCRATCH = 0, POOL_MAPPED_IN = 1, POOL_MAPPED_OUT = 2, POOL_COUNT = 3, CLS_MIN_LOG2 = 8, CLS_COUNT = 28 };
inline int cls_index(size_t b) { int i = 0; size_t c = (size_t)1 << CLS_MIN_LOG2; while (c < b) { c <<= 1; i++; } return i; }
inline size_t cls_bytes(int i) { return (size_t)1 << (CLS_MIN_LOG2 + i); }
struct ScratchPool
{
    std::mutex m;
    std::vector<void*> free_[CLS_COUNT];            /* size class -> idle blocks */
    std::vector<void*> all_;                        /* every block of the pool (x265amd_release_scratch) */
    bool device_ = false;                           /* mapped pools: blocks are device memory (hipFree) rather than pinned host memory */
};
ScratchPool& pool_of(int k) { static ScratchPool* p[POOL_COUNT] = { new ScratchPool, new ScratchPool, new ScratchPool }; return *p[k]; }

/* block -> (pool, size class): open addressing, insert under the owning pool's mutex (a block enters once, when the runtime hands it out), lock-free look-up.
 * Entries go away only in x265amd_release_scratch, which runs while nothing else uses the pools. */
enum { TAB_BITS = 18, TAB_SIZE = 1 << TAB_BITS };
struct TabEntry { std::atomic<uintptr_t> key; std::atomic<uint32_t> val; };
TabEntry* g_tab = new TabEntry[TAB_SIZE]();
std::mutex g_tabM;
inline uint32_t tab_hash(uintptr_t k) { return (uint32_t)((k >> 8) * 0x9E3779B97F4A7C15ull >> (64 - TAB_BITS)); }
void tab_insert(void* p, int pool, int cls)
{
    std::lock_guard<std::mutex> g(g_tabM);
    const uintptr_t k = (uintptr_t)p;
    for (uint32_t i = tab_hash(k), n = 0; n < TAB_SIZE; i = (i + 1) & (TAB_SIZE - 1), n++)
    {
        const uintptr_t cur = g_tab[i].key.load(std::memory_order_relaxed);
        if (cur == 0 || cur == k) { g_tab[i].val.store((uint32_t)(pool << 8 | cls), std::memory_order_relaxed); g_tab[i].key.store(k, std::memory_order_release); return; }
    }
    fprintf(stderr, "x265amd: fatal: block table full\n"); abort();
}
bool tab_find(const void* p, int& pool, int& cls)
{
    const uintptr_t k = (uintptr_t)p;
    for (uint32_t i = tab_hash(k), n = 0; n < TAB_SIZE; i = (i + 1) & (TAB_SIZE - 1), n++)
    {
        const uintptr_t cur = g_tab[i].key.load(std::memory_order_acquire);
        if (cur == k) { const uint32_t v = g_tab[i].val.load(std::memory_order_relaxed); pool = (int)(v >> 8); cls = (int)(v & 255); return true; }
        if (cur == 0) return false;
    }
    return false;
}

struct LocalLists { std::vector<void*> v[POOL_COUNT][CLS_COUNT]; };
thread_local LocalLists* t_local = nullptr;

hipError_t pool_alloc(int k, void** p, size_t bytes)
{
    const int c = cls_index(bytes ? bytes : 1);
    if (c >= CLS_COUNT) return hipErrorOutOfMemory;
    if (t_local)
    {
        std::vector<void*>& v = t_local->v[k][c];
        if (!v.empty()) { *p = v.back(); v.pop_back(); return hipSuccess; }
    }
    ScratchPool& P = pool_of(k);
    {
        std::lock_guard<std::mutex> g(P.m);
        std::vector<void*>& v = P.free_[c];
        if (!v.empty()) { *p = v.back(); v.pop_back(); return hipSuccess; }
    }
    hipError_t e;
    bool device = true;
    if (k == POOL_SCRATCH) e = hipMalloc(p, cls_bytes(c));
    else
    {
        /* Job records (host writes, device reads) live in DEVICE memory the host writes through the BAR: the stores are posted and travel in order ahead of
         * the command that uses them (or are long there when a kernel launches), and the device reads them at local latency instead of pulling them over
         * PCIe (about 1.7 us per command saved in queue mode).  Uncached allocation: nothing of it lingers in the L2 between uses.  Results (device
         * writes, host reads) stay in pinned host memory: a host read over the BAR costs a microsecond per access.  X265AMD_PUSH_RECORDS=0: host memory for both. */
        static const bool push = !(getenv("X265AMD_PUSH_RECORDS") && atoi(getenv("X265AMD_PUSH_RECORDS")) == 0);
        device = k == POOL_MAPPED_IN && push;
        e = device ? hipExtMallocWithFlags(p, cls_bytes(c), hipDeviceMallocUncached) : hipHostMalloc(p, cls_bytes(c), hipHostMallocMapped | hipHostMallocCoherent);
    }
    if (e == hipSuccess)
    {
        { std::lock_guard<std::mutex> g(P.m); P.all_.push_back(*p); P.device_ = device; }
        tab_insert(*p, k, c);
    }
    return e;
}
/* returns false when the block is not one of ours */
bool pool_free(void* p)
{
    int k, c;
    if (!tab_find(p, k, c)) return false;
    if (t_local) { t_local->v[k][c].push_back(p); return true; }
    ScratchPool& P = pool_of(k);
    std::lock_guard<std::mutex> g(P.m);
    P.free_[c].push_back(p);
    return true;
}
}
void xa_scratch_local_begin() { if (!t_local) t_local = new LocalLists; }
/* row tasks (xa_fiber.h): the lists belong to the task, not to the worker thread that happens to run it */
void* xa_scratch_local_swap(void* list) { void* old = t_local; t_local = static_cast<LocalLists*>(list); return old; }
void xa_scratch_local_end()
{
    if (!t_local) return;
    for (int k = 0; k < POOL_COUNT; k++)
    {
        ScratchPool& P = pool_of(k);
        std::lock_guard<std::mutex> g(P.m);
        for (int c = 0; c < CLS_COUNT; c++) { std::vector<void*>& v = P.free_[c]; v.insert(v.end(), t_local->v[k][c].begin(), t_local->v[k][c].end()); }
    }
    delete t_local;
    t_local = nullptr;
}
hipError_t xa_scratch_alloc(void** p, size_t bytes) { return pool_alloc(POOL_SCRATCH, p, bytes); }
void xa_scratch_free(void* p)
{
    if (!p) return;
    if (!pool_free(p)) (void)hipFree(p);
}
hipError_t xa_mapped_alloc(void** p, size_t bytes, bool deviceWrites) { return pool_alloc(deviceWrites ? POOL_MAPPED_OUT : POOL_MAPPED_IN, p, bytes); }
void xa_mapped_free(void* p)
{
    if (!p) return;
    if (!pool_free(p)) (void)hipHostFree(p);
}
extern "C" void x265amd_release_scratch(void)
{
    (void)hipDeviceSynchronize();
    for (int k = 0; k < POOL_COUNT; k++)
    {
        ScratchPool& P = pool_of(k);
        std::lock_guard<std::mutex> g(P.m);
        /* only idle blocks go back to the runtime; their table entries stay (a block of the same address that comes back later is entered again with its new class) */
        for (int c = 0; c < CLS_COUNT; c++)
        {
            for (void* q : P.free_[c])
            {
                if (k == POOL_SCRATCH || P.device_) (void)hipFree(q); else (void)hipHostFree(q);
                for (size_t i = 0; i < P.all_.size(); i++) if (P.all_[i] == q) { P.all_[i] = P.all_.back(); P.all_.pop_back(); break; }
            }
            P.free_[c].clear();
        }
    }
}

/* ---- xa_copy_rects (x265amd_host.h) ---- */
__global__ __launch_bounds__(256) void k_copy_rects(XaRects r)
{
    const int k = blockIdx.x;
    if (k >= r.n) return;
    const x265amd_pixel* src = reinterpret_cast<const x265amd_pixel*>(r.src[k]);
    x265amd_pixel* dst = reinterpret_cast<x265amd_pixel*>(r.dst[k]);
    const int w = r.w[k], total = w * r.h[k];
    for (int i = threadIdx.x; i < total; i += 256)
    {
        const int y = i / w, x = i - y * w;
        dst[(size_t)y * r.dst_stride[k] + x] = src[(size_t)y * r.src_stride[k] + x];
    }
}
void xa_copy_rects(void* st, const XaRects& r)
{
    if (r.n <= 0) return;
    static_assert(sizeof(XaRects) == sizeof(XaArgsRects), "XaRects is the argument record of XA_OP_COPY_RECTS");
    hipError_t e_;
    XA_LAUNCH(e_, st, XA_OP_COPY_RECTS, 1, r, k_copy_rects, dim3(r.n), dim3(256), 0, r);
}

/* ---- X265AMD_HOSTPROF (x265amd_host.h) ---- */
#include "xa_fiber.h"
#include <atomic>
#include <string.h>
#include <time.h>
bool g_xaHostProf = getenv("X265AMD_HOSTPROF") != nullptr;
namespace {
struct HpEntry { const char* name; std::atomic<uint64_t> calls, ns; };
HpEntry g_hp[128];
std::atomic<int> g_hpN{ 0 };
const bool g_hpWall = getenv("X265AMD_HOSTPROF") && !strcmp(getenv("X265AMD_HOSTPROF"), "wall");      /* X265AMD_HOSTPROF=wall: the scopes on the wall clock (parked time included) */
inline uint64_t hp_now()
{
    if (!g_hpWall && xa_in_task()) return xa_task_run_ns_always();
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
}
int xa_hostprof_id(const char* name) { const int i = g_hpN.fetch_add(1); if (i < 128) g_hp[i].name = name; return i < 128 ? i : 127; }
XaHostProfScope::XaHostProfScope(int id_) : id(id_), t0(g_xaHostProf ? hp_now() : 0) {}
XaHostProfScope::~XaHostProfScope() { if (g_xaHostProf) { g_hp[id].calls++; g_hp[id].ns += hp_now() - t0; } }
void xa_hostprof_report(void)
{
    if (!g_xaHostProf) return;
    const int n = g_hpN.load() < 128 ? g_hpN.load() : 128;
    fprintf(stderr, "x265amd host profile (scope: calls, ms, us per call):\n");
    for (int i = 0; i < n; i++)
        if (g_hp[i].calls) fprintf(stderr, "  %-34s %9llu %9.1f %8.2f\n", g_hp[i].name, (unsigned long long)g_hp[i].calls.load(), g_hp[i].ns.load() / 1e6, g_hp[i].ns.load() / 1e3 / g_hp[i].calls.load());
}
extern "C" void x265amd_hostprof_report(void) { xa_hostprof_report(); }
