/* TEST INFRASTRUCTURE ONLY.
 * End-to-end check of the slot-for-slot drop-in: runs the REFERENCE encoder (its own host loop, lookahead, rate control,
 * CABAC -- everything, linked from the reference objects by oracle/build_ref.sh) on a small deterministic synthetic clip,
 * either with its own C primitive table or with the table overridden by libx265amd's x265amd_setup_primitives()
 * (the hook INTEGRATION.md section 1 describes).  Prints the size and a 64-bit FNV-1a hash of the bitstream; the two
 * runs must print the same line (tests/test_hip_reference_encoder_dropin.py).
 *
 * usage: x265_dropin<8|10> <libx265amd path | none> <width> <height> <frames> <preset> [key=value ...]
 * Debug aid: X265AMD_SLOT_RANGE=lo:hi keeps the library's thunks only for table slots lo <= i < hi (pointer-sized slot
 * index) and restores the reference C pointer everywhere else -- oracle/bisect_dropin.py uses it to find a slot whose
 * result differs inside the real encoder.
 */
#include "common.h"
#include "primitives.h"
#include "x265.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace X265_NS;

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s <lib|none> <w> <h> <frames> <preset> [extra x265 options key=value ...]\n", argv[0]); return 2; }
    const char* libPath = argv[1];
    int w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]);
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, argv[5], NULL) < 0) return 3;
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = 30; p->fpsDenom = 1; p->internalCsp = X265_CSP_I420;
    p->bEmitInfoSEI = 0; p->logLevel = X265_LOG_NONE;
    p->frameNumThreads = 1;
    x265_param_parse(p, "pools", "1");
    for (int i = 6; i < argc; i++)
    {
        char* eq = strchr(argv[i], '=');
        if (eq) { *eq = 0; x265_param_parse(p, argv[i], eq + 1); }
        else x265_param_parse(p, argv[i], NULL);
    }
    x265_encoder* enc = x265_encoder_open(p);      /* fills X265_NS::primitives with the C references (primitives.cpp:248-285) */
    if (!enc) return 4;
    int installed = 0;
    if (strcmp(libPath, "none"))
    {
        void* hnd = dlopen(libPath, RTLD_NOW | RTLD_LOCAL);
        if (!hnd) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 5; }
        typedef int (*setup_fn)(void*, size_t);
        setup_fn setup = (setup_fn)dlsym(hnd, "x265amd_setup_primitives");
        EncoderPrimitives cTable = primitives;
        installed = setup ? setup(&primitives, sizeof(EncoderPrimitives)) : -1;
        if (const char* r = getenv("X265AMD_SLOT_RANGE"))
        {
            long lo = 0, hi = 0;
            sscanf(r, "%ld:%ld", &lo, &hi);
            void** dst = (void**)&primitives; void** src = (void**)&cTable;
            long n = (long)(sizeof(EncoderPrimitives) / sizeof(void*)), kept = 0;
            for (long i = 0; i < n; i++)
                if (i < lo || i >= hi) dst[i] = src[i]; else kept += dst[i] != src[i];
            fprintf(stderr, "slot range %ld:%ld -> %ld GPU slots kept\n", lo, hi, kept);
        }
        if (installed <= 0)
        {
            const char* (*err)(void) = (const char* (*)(void))dlsym(hnd, "x265amd_last_error");
            fprintf(stderr, "x265amd_setup_primitives failed: %s\n", err ? err() : "?");
            return 6;
        }
    }
    /* deterministic integer-only clip: moving texture + noise */
    x265_picture* pic = x265_picture_alloc();
    x265_picture_init(p, pic);
    const int bytes = X265_DEPTH > 8 ? 2 : 1;
    std::vector<uint8_t> bufY((size_t)w * h * bytes), bufU((size_t)w * h / 4 * bytes), bufV((size_t)w * h / 4 * bytes);
    pic->planes[0] = bufY.data(); pic->planes[1] = bufU.data(); pic->planes[2] = bufV.data();
    pic->stride[0] = w * bytes; pic->stride[1] = pic->stride[2] = w / 2 * bytes;
    pic->bitDepth = X265_DEPTH;
    uint64_t hash = 1469598103934665603ull, total = 0;
    auto absorb = [&](x265_nal* nal, uint32_t n) {
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t k = 0; k < nal[i].sizeBytes; k++) { hash = (hash ^ nal[i].payload[k]) * 1099511628211ull; total++; }
    };
    x265_nal* nal; uint32_t nnal;
    if (x265_encoder_headers(enc, &nal, &nnal) >= 0) absorb(nal, nnal);
    uint32_t lcg = 12345;
    const int shift = X265_DEPTH - 8;
    for (int f = 0; f < frames; f++)
    {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                lcg = lcg * 1664525u + 1013904223u;
                int v = 96 + (((x + 2 * f) * 5 + (y + f) * 3) & 63) + ((((x + 2 * f) >> 3) ^ ((y + f) >> 3)) & 3) * 9 + (int)((lcg >> 24) % 5);
                if (bytes == 2) ((uint16_t*)bufY.data())[y * w + x] = (uint16_t)(v << shift); else bufY[y * w + x] = (uint8_t)v;
            }
        for (int y = 0; y < h / 2; y++)
            for (int x = 0; x < w / 2; x++)
            {
                int u = 110 + ((x + f) & 15), vv = 140 - ((y + f / 2) & 15);
                if (bytes == 2) { ((uint16_t*)bufU.data())[y * (w / 2) + x] = (uint16_t)(u << shift); ((uint16_t*)bufV.data())[y * (w / 2) + x] = (uint16_t)(vv << shift); }
                else { bufU[y * (w / 2) + x] = (uint8_t)u; bufV[y * (w / 2) + x] = (uint8_t)vv; }
            }
        pic->pts = f;
        if (x265_encoder_encode(enc, &nal, &nnal, pic, NULL) < 0) return 7;
        absorb(nal, nnal);
    }
    while (x265_encoder_encode(enc, &nal, &nnal, NULL, NULL) > 0) absorb(nal, nnal);
    x265_encoder_close(enc);
    printf("bytes=%llu fnv1a=%016llx slots=%d\n", (unsigned long long)total, (unsigned long long)hash, installed > 0 ? 1 : 0);
    return 0;
}
