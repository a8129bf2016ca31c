/* Layer 1 of the C ABI (include/x265amd.h): per-slot entry points with HOST pointers.
 *
 * Every entry point stages its operands into a per-thread device arena (packed rows), runs the layer-2 kernel
 * of its family with a batch of ONE job (prim_kernels.hip) and copies the result back -- the kernel under test
 * is therefore exactly the one the batched path runs.  Re-entrant and stateless across calls like the reference's
 * primitives (reference: source/common/primitives.h:239-433 are called concurrently from every worker thread):
 * all scratch is thread-local.  A HIP failure aborts; there is no CPU fallback.
 */
#include <string.h>
#include <vector>
#include "x265amd_host.h"

typedef x265amd_pixel pixel;

static const uint8_t k_puW[25] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 12, 16, 4, 32, 24, 32, 8, 64, 48, 64, 16 };
static const uint8_t k_puH[25] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 12, 16, 4, 16, 24, 32, 8, 32, 48, 64, 16, 64 };

static thread_local char g_err[256] = "";

int xa_fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

extern "C" const char* x265amd_last_error(void) { return g_err; }
extern "C" int x265amd_bit_depth(void) { return X265AMD_DEPTH; }
extern "C" const char* x265amd_version(void) { return "x265amd 0.1 (gfx950) / parity target x265 3.6+1-aa7f602f7 [noasm]"; }
extern "C" int x265amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {

struct Arena
{
    char* dev = nullptr;
    size_t cap = 0, used = 0;
    x265amd_job* djob = nullptr;
    ~Arena()
    {
        if (dev) (void)hipFree(dev);
        if (djob) (void)hipFree(djob);
    }
    void ensure()
    {
        if (!dev)
        {
            cap = 8u << 20;
            XA_HIP_FATAL(hipMalloc((void**)&dev, cap));
            XA_HIP_FATAL(hipMalloc((void**)&djob, sizeof(x265amd_job)));
        }
    }
    uint64_t alloc(size_t bytes)
    {
        used = (used + 255) & ~(size_t)255;
        if (used + bytes > cap)
        {
            fprintf(stderr, "x265amd: fatal: per-slot staging arena exhausted (%zu + %zu)\n", used, bytes);
            abort();
        }
        uint64_t a = (uint64_t)(uintptr_t)(dev + used);
        used += bytes;
        return a;
    }
};

static thread_local Arena g_arena;

struct CopyBack { void* host; uint64_t dev; size_t rowBytes, hostPitch; int rows; };

struct Shim
{
    std::vector<CopyBack> outs;
    x265amd_job j;
    Shim(int op, int size)
    {
        g_arena.ensure();
        g_arena.used = 0;
        memset(&j, 0, sizeof(j));
        j.op = op; j.size = size;
    }
    /* packed device copy of a w x h region (elements of `elem` bytes); returns device address, stride = w */
    uint64_t in2d(const void* host, size_t elem, int w, int h, intptr_t stride)
    {
        uint64_t d = g_arena.alloc((size_t)w * h * elem);
        XA_HIP_FATAL(hipMemcpy2D((void*)(uintptr_t)d, w * elem, host, stride * elem, w * elem, h, hipMemcpyHostToDevice));
        return d;
    }
    uint64_t in1d(const void* host, size_t bytes)
    {
        uint64_t d = g_arena.alloc(bytes);
        XA_HIP_FATAL(hipMemcpy((void*)(uintptr_t)d, host, bytes, hipMemcpyHostToDevice));
        return d;
    }
    uint64_t out2d(void* host, size_t elem, int w, int h, intptr_t stride)
    {
        uint64_t d = g_arena.alloc((size_t)w * h * elem);
        outs.push_back({ host, d, (size_t)w * elem, (size_t)stride * elem, h });
        return d;
    }
    uint64_t out1d(void* host, size_t bytes) { return out2d(host, 1, (int)bytes, 1, (intptr_t)bytes); }
    uint64_t scalar() { return g_arena.alloc(8); }
    void run()
    {
        XA_HIP_FATAL(hipMemcpy(g_arena.djob, &j, sizeof(j), hipMemcpyHostToDevice));
        if (x265amd_run_jobs(nullptr, g_arena.djob, 1, j.op / 32) != X265AMD_OK)
        {
            fprintf(stderr, "x265amd: fatal: kernel launch failed: %s\n", x265amd_last_error());
            abort();
        }
        XA_HIP_FATAL(hipDeviceSynchronize());
        for (const CopyBack& c : outs)
            XA_HIP_FATAL(hipMemcpy2D(c.host, c.hostPitch, (void*)(uintptr_t)c.dev, c.rowBytes, c.rowBytes, c.rows, hipMemcpyDeviceToHost));
    }
    uint64_t fetch(uint64_t devAddr)
    {
        uint64_t v = 0;
        XA_HIP_FATAL(hipMemcpy(&v, (void*)(uintptr_t)devAddr, 8, hipMemcpyDeviceToHost));
        return v;
    }
};

/* distortion between two pixel blocks -> scalar */
static uint64_t dist_pp(int op, int size, int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    Shim s(op, size);
    s.j.a = s.in2d(a, sizeof(pixel), w, h, sa); s.j.sa = w;
    s.j.b = s.in2d(b, sizeof(pixel), w, h, sb); s.j.sb = w;
    s.j.d = s.scalar();
    s.run();
    return s.fetch(s.j.d);
}

} // namespace

extern "C" {

/* ---------------------------------------------------------------- distortion ---------------------------- */
int x265amd_sad(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (int)dist_pp(X265AMD_OP_SAD, part, k_puW[part], k_puH[part], a, sa, b, sb); }
int x265amd_satd(int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (int)dist_pp(X265AMD_OP_SATD, part, k_puW[part], k_puH[part], a, sa, b, sb); }
int x265amd_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (int)dist_pp(X265AMD_OP_SA8D, cu, 4 << cu, 4 << cu, a, sa, b, sb); }
int x265amd_psy_cost_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return (int)dist_pp(X265AMD_OP_PSY_COST_PP, cu, 4 << cu, 4 << cu, a, sa, b, sb); }
int x265amd_chroma_satd(int csp, int part, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    (void)csp;
    int w = k_puW[part] >> 1, h = k_puH[part] >> 1;
    if ((w | h) & 3) return -1;     /* NULL slot in the reference (pixel.cpp:1205-1229) */
    return (int)dist_pp(X265AMD_OP_CHROMA_SATD, part, w, h, a, sa, b, sb);
}
int x265amd_chroma_sa8d(int csp, int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ (void)csp; return (int)dist_pp(X265AMD_OP_CHROMA_SA8D, cu, 2 << cu, 2 << cu, a, sa, b, sb); }
uint64_t x265amd_sse_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{ return dist_pp(X265AMD_OP_SSE_PP, cu, 4 << cu, 4 << cu, a, sa, b, sb); }

static void sad_xn(int op, int part, const pixel* fenc, const pixel* const* refs, int nref, intptr_t rs, int32_t* res)
{
    int w = k_puW[part], h = k_puH[part];
    Shim s(op, part);
    s.j.a = s.in2d(fenc, sizeof(pixel), w, h, 64);
    /* the kernel reads fenc at FENC_STRIDE: stage it at that stride */
    uint64_t f = g_arena.alloc(64 * 64 * sizeof(pixel));
    XA_HIP_FATAL(hipMemcpy2D((void*)(uintptr_t)f, 64 * sizeof(pixel), fenc, 64 * sizeof(pixel), w * sizeof(pixel), h, hipMemcpyHostToDevice));
    s.j.a = f;
    uint64_t r[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < nref; i++) r[i] = s.in2d(refs[i], sizeof(pixel), w, h, rs);
    s.j.b = r[0]; s.j.c = r[1]; s.j.e[0] = r[2]; s.j.e[1] = r[3];
    s.j.sb = w;
    s.j.d = s.out1d(res, nref * sizeof(int32_t));
    s.run();
}
void x265amd_sad_x3(int part, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{ const pixel* r[3] = { r0, r1, r2 }; sad_xn(X265AMD_OP_SAD_X3, part, fenc, r, 3, rs, res); }
void x265amd_sad_x4(int part, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{ const pixel* r[4] = { r0, r1, r2, r3 }; sad_xn(X265AMD_OP_SAD_X4, part, fenc, r, 4, rs, res); }

uint64_t x265amd_sse_ss(int cu, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_SSE_SS, cu);
    s.j.a = s.in2d(a, 2, n, n, sa); s.j.sa = n;
    s.j.b = s.in2d(b, 2, n, n, sb); s.j.sb = n;
    s.j.d = s.scalar();
    s.run();
    return s.fetch(s.j.d);
}
uint64_t x265amd_ssd_s(int cu, const int16_t* a, intptr_t sa)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_SSD_S, cu);
    s.j.a = s.in2d(a, 2, n, n, sa); s.j.sa = n;
    s.j.d = s.scalar();
    s.run();
    return s.fetch(s.j.d);
}
uint64_t x265amd_var(int cu, const pixel* a, intptr_t sa)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_VAR, cu);
    s.j.a = s.in2d(a, sizeof(pixel), n, n, sa); s.j.sa = n;
    s.j.d = s.scalar();
    s.run();
    return s.fetch(s.j.d);
}

/* ---------------------------------------------------------------- pixel / residual ---------------------- */
void x265amd_sub_ps(int cu, int16_t* dst, intptr_t ds, const pixel* s0, const pixel* s1, intptr_t ss0, intptr_t ss1)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_SUB_PS, cu);
    s.j.a = s.in2d(s0, sizeof(pixel), n, n, ss0); s.j.sa = n;
    s.j.b = s.in2d(s1, sizeof(pixel), n, n, ss1); s.j.sb = n;
    s.j.d = s.out2d(dst, 2, n, n, ds); s.j.sd = n;
    s.run();
}
void x265amd_add_ps(int cu, pixel* dst, intptr_t ds, const pixel* s0, const int16_t* s1, intptr_t ss0, intptr_t ss1)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_ADD_PS, cu);
    s.j.a = s.in2d(s0, sizeof(pixel), n, n, ss0); s.j.sa = n;
    s.j.b = s.in2d(s1, 2, n, n, ss1); s.j.sb = n;
    s.j.d = s.out2d(dst, sizeof(pixel), n, n, ds); s.j.sd = n;
    s.run();
}
void x265amd_pixelavg_pp(int part, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0, const pixel* s1, intptr_t ss1)
{
    int w = k_puW[part], h = k_puH[part];
    Shim s(X265AMD_OP_PIXELAVG_PP, part);
    s.j.a = s.in2d(s0, sizeof(pixel), w, h, ss0); s.j.sa = w;
    s.j.b = s.in2d(s1, sizeof(pixel), w, h, ss1); s.j.sb = w;
    s.j.d = s.out2d(dst, sizeof(pixel), w, h, ds); s.j.sd = w;
    s.run();
}
static void addavg(int part, int chroma, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{
    int w = k_puW[part] >> chroma, h = k_puH[part] >> chroma;
    Shim s(X265AMD_OP_ADDAVG, part);
    s.j.p[0] = chroma;
    s.j.a = s.in2d(s0, 2, w, h, ss0); s.j.sa = w;
    s.j.b = s.in2d(s1, 2, w, h, ss1); s.j.sb = w;
    s.j.d = s.out2d(dst, sizeof(pixel), w, h, ds); s.j.sd = w;
    s.run();
}
void x265amd_addAvg(int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{ addavg(part, 0, s0, s1, dst, ss0, ss1, ds); }
void x265amd_chroma_addAvg(int csp, int part, const int16_t* s0, const int16_t* s1, pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{ (void)csp; addavg(part, 1, s0, s1, dst, ss0, ss1, ds); }

void x265amd_weight_pp(const pixel* src, pixel* dst, intptr_t stride, int width, int height, int w0, int round, int shift, int offset)
{
    Shim s(X265AMD_OP_WEIGHT_PP, 0);
    int p[6] = { width, height, w0, round, shift, offset };
    memcpy(s.j.p, p, sizeof(p));
    s.j.a = s.in2d(src, sizeof(pixel), width, height, stride); s.j.sa = width;
    s.j.d = s.out2d(dst, sizeof(pixel), width, height, stride); s.j.sd = width;
    s.run();
}
void x265amd_weight_sp(const int16_t* src, pixel* dst, intptr_t ss, intptr_t ds, int width, int height, int w0, int round, int shift, int offset)
{
    Shim s(X265AMD_OP_WEIGHT_SP, 0);
    int p[6] = { width, height, w0, round, shift, offset };
    memcpy(s.j.p, p, sizeof(p));
    s.j.a = s.in2d(src, 2, width, height, ss); s.j.sa = width;
    s.j.d = s.out2d(dst, sizeof(pixel), width, height, ds); s.j.sd = width;
    s.run();
}
void x265amd_scale2D_64to32(pixel* dst, const pixel* src, intptr_t stride)
{
    Shim s(X265AMD_OP_SCALE2D_64TO32, 0);
    s.j.a = s.in2d(src, sizeof(pixel), 64, 64, stride); s.j.sa = 64;
    s.j.d = s.out1d(dst, 32 * 32 * sizeof(pixel));
    s.run();
}
void x265amd_scale1D_128to64(pixel* dst, const pixel* src)
{
    Shim s(X265AMD_OP_SCALE1D_128TO64, 0);
    s.j.a = s.in1d(src, 256 * sizeof(pixel));
    s.j.d = s.out1d(dst, 128 * sizeof(pixel));
    s.run();
}
void x265amd_transpose(int cu, pixel* dst, const pixel* src, intptr_t stride)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_TRANSPOSE, cu);
    s.j.a = s.in2d(src, sizeof(pixel), n, n, stride); s.j.sa = n;
    s.j.d = s.out1d(dst, n * n * sizeof(pixel));
    s.run();
}
static void cpy_shift(int op, int cu, int16_t* dst, const int16_t* src, intptr_t stride, int shift)
{
    int n = 4 << cu;
    bool to1d = op == X265AMD_OP_CPY2DTO1D_SHL || op == X265AMD_OP_CPY2DTO1D_SHR;
    Shim s(op, cu);
    s.j.p[0] = shift;
    if (to1d)
    {
        s.j.a = s.in2d(src, 2, n, n, stride); s.j.sa = n;
        s.j.d = s.out1d(dst, n * n * 2);
    }
    else
    {
        s.j.a = s.in1d(src, n * n * 2);
        s.j.d = s.out2d(dst, 2, n, n, stride); s.j.sd = n;
    }
    s.run();
}
void x265amd_cpy2Dto1D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t ss, int shift) { cpy_shift(X265AMD_OP_CPY2DTO1D_SHL, cu, dst, src, ss, shift); }
void x265amd_cpy2Dto1D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t ss, int shift) { cpy_shift(X265AMD_OP_CPY2DTO1D_SHR, cu, dst, src, ss, shift); }
void x265amd_cpy1Dto2D_shl(int cu, int16_t* dst, const int16_t* src, intptr_t ds, int shift) { cpy_shift(X265AMD_OP_CPY1DTO2D_SHL, cu, dst, src, ds, shift); }
void x265amd_cpy1Dto2D_shr(int cu, int16_t* dst, const int16_t* src, intptr_t ds, int shift) { cpy_shift(X265AMD_OP_CPY1DTO2D_SHR, cu, dst, src, ds, shift); }
uint32_t x265amd_copy_cnt(int cu, int16_t* coeff, const int16_t* resi, intptr_t stride)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_COPY_CNT, cu);
    s.j.a = s.in2d(resi, 2, n, n, stride); s.j.sa = n;
    s.j.d = s.out1d(coeff, n * n * 2);
    s.j.e[0] = s.scalar();
    s.run();
    return (uint32_t)s.fetch(s.j.e[0]);
}
int x265amd_count_nonzero(int cu, const int16_t* q)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_COUNT_NONZERO, cu);
    s.j.a = s.in1d(q, n * n * 2);
    s.j.d = s.scalar();
    s.run();
    return (int)s.fetch(s.j.d);
}

/* ---------------------------------------------------------------- transforms ---------------------------- */
static void fwd_tr(int op, int cu, const int16_t* src, int16_t* dst, intptr_t stride)
{
    int n = 4 << cu;
    Shim s(op, cu);
    s.j.a = s.in2d(src, 2, n, n, stride); s.j.sa = n;
    s.j.d = s.out1d(dst, n * n * 2);
    s.run();
}
static void inv_tr(int op, int cu, const int16_t* src, int16_t* dst, intptr_t stride)
{
    int n = 4 << cu;
    Shim s(op, cu);
    s.j.a = s.in1d(src, n * n * 2);
    s.j.d = s.out2d(dst, 2, n, n, stride); s.j.sd = n;
    s.run();
}
void x265amd_dct(int cu, const int16_t* src, int16_t* dst, intptr_t stride) { fwd_tr(X265AMD_OP_DCT, cu, src, dst, stride); }
void x265amd_idct(int cu, const int16_t* src, int16_t* dst, intptr_t stride) { inv_tr(X265AMD_OP_IDCT, cu, src, dst, stride); }
void x265amd_dst4x4(const int16_t* src, int16_t* dst, intptr_t stride) { fwd_tr(X265AMD_OP_DST4, 0, src, dst, stride); }
void x265amd_idst4x4(const int16_t* src, int16_t* dst, intptr_t stride) { inv_tr(X265AMD_OP_IDST4, 0, src, dst, stride); }

uint32_t x265amd_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    Shim s(X265AMD_OP_QUANT, 0);
    s.j.p[0] = qBits; s.j.p[1] = add; s.j.p[2] = numCoeff;
    s.j.a = s.in1d(coef, numCoeff * 2);
    s.j.b = s.in1d(quantCoeff, numCoeff * 4);
    s.j.d = s.out1d(qCoef, numCoeff * 2);
    s.j.e[0] = s.out1d(deltaU, numCoeff * 4);
    s.j.e[1] = s.scalar();
    s.run();
    return (uint32_t)s.fetch(s.j.e[1]);
}
uint32_t x265amd_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    Shim s(X265AMD_OP_NQUANT, 0);
    s.j.p[0] = qBits; s.j.p[1] = add; s.j.p[2] = numCoeff;
    s.j.a = s.in1d(coef, numCoeff * 2);
    s.j.b = s.in1d(quantCoeff, numCoeff * 4);
    s.j.d = s.out1d(qCoef, numCoeff * 2);
    s.j.e[1] = s.scalar();
    s.run();
    return (uint32_t)s.fetch(s.j.e[1]);
}
void x265amd_dequant_normal(const int16_t* quantCoef, int16_t* coef, int num, int scale, int shift)
{
    Shim s(X265AMD_OP_DEQUANT_NORMAL, 0);
    s.j.p[0] = num; s.j.p[1] = scale; s.j.p[2] = shift;
    s.j.a = s.in1d(quantCoef, num * 2);
    s.j.d = s.out1d(coef, num * 2);
    s.run();
}
void x265amd_dequant_scaling(const int16_t* src, const int32_t* dequantCoef, int16_t* dst, int num, int per, int shift)
{
    Shim s(X265AMD_OP_DEQUANT_SCALING, 0);
    s.j.p[0] = num; s.j.p[1] = per; s.j.p[2] = shift;
    s.j.a = s.in1d(src, num * 2);
    s.j.b = s.in1d(dequantCoef, num * 4);
    s.j.d = s.out1d(dst, num * 2);
    s.run();
}

/* ---------------------------------------------------------------- intra --------------------------------- */
void x265amd_intra_pred(int cu, int mode, pixel* dst, intptr_t ds, const pixel* srcPix, int bFilter)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_INTRA_PRED, cu);
    s.j.p[0] = mode; s.j.p[1] = bFilter;
    s.j.a = s.in1d(srcPix, (4 * n + 1) * sizeof(pixel));
    s.j.d = s.out2d(dst, sizeof(pixel), n, n, ds); s.j.sd = n;
    s.run();
}
void x265amd_intra_filter(int cu, const pixel* refs, pixel* filtered)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_INTRA_FILTER, cu);
    s.j.a = s.in1d(refs, (4 * n + 1) * sizeof(pixel));
    s.j.d = s.out1d(filtered, (4 * n + 1) * sizeof(pixel));
    s.run();
}
void x265amd_intra_allangs(int cu, pixel* dst, pixel* refPix, pixel* filtPix, int bLuma)
{
    int n = 4 << cu;
    Shim s(X265AMD_OP_INTRA_ALLANGS, cu);
    s.j.p[0] = bLuma;
    s.j.a = s.in1d(refPix, (4 * n + 1) * sizeof(pixel));
    s.j.b = s.in1d(filtPix, (4 * n + 1) * sizeof(pixel));
    s.j.d = s.out1d(dst, 33 * n * n * sizeof(pixel));
    s.run();
}

/* ---------------------------------------------------------------- interpolation ------------------------- */
/* srcElem/dstElem: bytes per sample; marginX/marginTop/marginBottom: extra source samples staged around the block */
static void interp(int op, int taps, int w, int h, const void* src, size_t srcElem, intptr_t ss, void* dst, size_t dstElem, intptr_t ds,
                   int idx, int aux, bool horiz, bool vert, int outRows)
{
    int half = taps / 2 - 1;
    int mx = horiz ? half : 0, mxr = horiz ? taps - 1 - half : 0;
    int mt = vert ? half : 0, mb = vert ? taps - 1 - half : 0;
    Shim s(op, 0);
    s.j.p[0] = taps; s.j.p[1] = w; s.j.p[2] = h; s.j.p[3] = idx; s.j.p[4] = aux;
    int sw = w + mx + mxr, sh = h + mt + mb;
    const char* origin = (const char*)src - ((intptr_t)mt * ss + mx) * (intptr_t)srcElem;
    uint64_t d = s.in2d(origin, srcElem, sw, sh, ss);
    s.j.a = d + ((uint64_t)mt * sw + mx) * srcElem; s.j.sa = sw;
    s.j.d = s.out2d(dst, dstElem, w, outRows, ds); s.j.sd = w;
    s.run();
}
#define PW k_puW[part]
#define PH k_puH[part]
void x265amd_luma_hpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ interp(X265AMD_OP_IP_HPP, 8, PW, PH, s, sizeof(pixel), ss, d, sizeof(pixel), ds, idx, 0, true, false, PH); }
void x265amd_luma_hps(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int ext)
{ interp(X265AMD_OP_IP_HPS, 8, PW, PH, s, sizeof(pixel), ss, d, 2, ds, idx, ext, true, ext != 0, PH + (ext ? 7 : 0)); }
void x265amd_luma_vpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ interp(X265AMD_OP_IP_VPP, 8, PW, PH, s, sizeof(pixel), ss, d, sizeof(pixel), ds, idx, 0, false, true, PH); }
void x265amd_luma_vps(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ interp(X265AMD_OP_IP_VPS, 8, PW, PH, s, sizeof(pixel), ss, d, 2, ds, idx, 0, false, true, PH); }
void x265amd_luma_vsp(int part, const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ interp(X265AMD_OP_IP_VSP, 8, PW, PH, s, 2, ss, d, sizeof(pixel), ds, idx, 0, false, true, PH); }
void x265amd_luma_vss(int part, const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ interp(X265AMD_OP_IP_VSS, 8, PW, PH, s, 2, ss, d, 2, ds, idx, 0, false, true, PH); }
void x265amd_luma_hvpp(int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int ix, int iy)
{ interp(X265AMD_OP_IP_HVPP, 8, PW, PH, s, sizeof(pixel), ss, d, sizeof(pixel), ds, ix, iy, true, true, PH); }
void x265amd_luma_p2s(int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds)
{ interp(X265AMD_OP_IP_P2S, 8, PW, PH, s, sizeof(pixel), ss, d, 2, ds, 0, 0, false, false, PH); }
void x265amd_chroma_hpp(int csp, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (void)csp; interp(X265AMD_OP_IP_HPP, 4, PW / 2, PH / 2, s, sizeof(pixel), ss, d, sizeof(pixel), ds, idx, 0, true, false, PH / 2); }
void x265amd_chroma_hps(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx, int ext)
{ (void)csp; interp(X265AMD_OP_IP_HPS, 4, PW / 2, PH / 2, s, sizeof(pixel), ss, d, 2, ds, idx, ext, true, ext != 0, PH / 2 + (ext ? 3 : 0)); }
void x265amd_chroma_vpp(int csp, int part, const pixel* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (void)csp; interp(X265AMD_OP_IP_VPP, 4, PW / 2, PH / 2, s, sizeof(pixel), ss, d, sizeof(pixel), ds, idx, 0, false, true, PH / 2); }
void x265amd_chroma_vps(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ (void)csp; interp(X265AMD_OP_IP_VPS, 4, PW / 2, PH / 2, s, sizeof(pixel), ss, d, 2, ds, idx, 0, false, true, PH / 2); }
void x265amd_chroma_vsp(int csp, int part, const int16_t* s, intptr_t ss, pixel* d, intptr_t ds, int idx)
{ (void)csp; interp(X265AMD_OP_IP_VSP, 4, PW / 2, PH / 2, s, 2, ss, d, sizeof(pixel), ds, idx, 0, false, true, PH / 2); }
void x265amd_chroma_vss(int csp, int part, const int16_t* s, intptr_t ss, int16_t* d, intptr_t ds, int idx)
{ (void)csp; interp(X265AMD_OP_IP_VSS, 4, PW / 2, PH / 2, s, 2, ss, d, 2, ds, idx, 0, false, true, PH / 2); }
void x265amd_chroma_p2s(int csp, int part, const pixel* s, intptr_t ss, int16_t* d, intptr_t ds)
{ (void)csp; interp(X265AMD_OP_IP_P2S, 4, PW / 2, PH / 2, s, sizeof(pixel), ss, d, 2, ds, 0, 0, false, false, PH / 2); }
#undef PW
#undef PH

} /* extern "C" */
