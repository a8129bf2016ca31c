/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the intra neighbour set and the 35-mode luma scan.
 *
 * Restates, from the algorithm (HEVC reference-sample substitution 8.4.4.2.2 and smoothing 8.4.4.2.3), the results of
 *   Predict::fillReferenceSamples                       source/common/predict.cpp:736-877
 *   Predict::initAdiPattern (incl. strong smoothing)    source/common/predict.cpp:600-649
 *   the mode scan of Search::estIntraPredQT             source/encoder/search.cpp:1566-1613 (sa8d of all 35 predictions)
 * Pinned against the reference's own Predict class and primitives through oracle/_ref/librefprims*.so
 * (tests/test_intra_oracle_vs_ref.py) and golden vectors (tests/golden/intra_golden.npz).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_DEPTH
#define ORC_DEPTH 8
#endif
#if ORC_DEPTH > 8
typedef uint16_t pixel;
#else
typedef uint8_t pixel;
#endif

void orc_intra_filter(int cu, const pixel* s, pixel* f);
void orc_intra_pred(int cu, int mode, pixel* dst, intptr_t ds, const pixel* srcPix, int bFilter);
int orc_sa8d(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_intra_filter_flags(int mode);

/* recon: top-left sample of the block inside the reconstructed plane.  flags[4*(N/4)+1]: availability per 4-sample unit in
 * the order below-left (bottom-most first) ... left, above-left, above ... above-right.  dirMode -1 = all modes.
 * outRef / outFlt: [0] above-left, [1..2N] above + above-right, [2N+1..4N] left + below-left.
 * Returns 1 when the filtered set was produced (else outFlt is left untouched, as in the reference). */
int orc_init_adi_pattern(const pixel* recon, intptr_t stride, int log2TrSize, const uint8_t* flags, int strongSmoothing, int dirMode,
                         pixel* outRef, pixel* outFlt)
{
    const int N = 1 << log2TrSize, N2 = 2 * N, units = N >> 2, leftUnits = 2 * units, total = 4 * units + 1;
    pixel line[4 * 32 + 1];         /* substitution order: index 0 = bottom-most below-left sample ... 2N = above-left ... 4N = right-most above-right */
    uint8_t avail[4 * 32 + 1];
    int any = 0;
    for (int i = 0; i <= 4 * N; i++)
    {
        int u = i < N2 ? i >> 2 : (i == N2 ? leftUnits : leftUnits + 1 + ((i - N2 - 1) >> 2));
        avail[i] = flags[u] != 0;
        any |= avail[i];
        if (i < N2) line[i] = recon[(N2 - 1 - i) * stride - 1];
        else if (i == N2) line[i] = recon[-stride - 1];
        else line[i] = recon[-stride + (i - N2 - 1)];
    }
    (void)total;
    if (!any)
        for (int i = 0; i <= 4 * N; i++) line[i] = (pixel)(1 << (ORC_DEPTH - 1));
    else
    {
        if (!avail[0])
        {
            int k = 1;
            while (!avail[k]) k++;
            line[0] = line[k];
        }
        for (int i = 1; i <= 4 * N; i++)
            if (!avail[i]) line[i] = line[i - 1];
    }
    outRef[0] = line[N2];
    for (int x = 0; x < N2; x++) outRef[1 + x] = line[N2 + 1 + x];
    for (int y = 0; y < N2; y++) outRef[N2 + 1 + y] = line[N2 - 1 - y];

    int need = dirMode < 0 ? ((8 | 16 | 32) & N) : (orc_intra_filter_flags(dirMode) & N);
    if (!need) return 0;
    if (strongSmoothing && N == 32)     /* predict.cpp:618-644 */
    {
        const int threshold = 1 << (ORC_DEPTH - 5);
        int topLeft = outRef[0], topLast = outRef[N2], leftLast = outRef[2 * N2];
        int topMiddle = outRef[32], leftMiddle = outRef[N2 + 32];
        if (abs(topLeft + topLast - 2 * topMiddle) < threshold && abs(topLeft + leftLast - 2 * leftMiddle) < threshold)
        {
            const int shift = 6;
            int init = (topLeft << shift) + N, deltaL = leftLast - topLeft, deltaR = topLast - topLeft;
            outFlt[0] = (pixel)topLeft;
            for (int i = 1; i < N2; i++)
            {
                outFlt[i + N2] = (pixel)((init + deltaL * i) >> shift);
                outFlt[i] = (pixel)((init + deltaR * i) >> shift);
            }
            outFlt[N2] = (pixel)topLast;
            outFlt[2 * N2] = (pixel)leftLast;
            return 1;
        }
    }
    orc_intra_filter(log2TrSize - 2, outRef, outFlt);
    return 1;
}

/* search.cpp:1566-1613: sa8d of the 35 luma predictions (DC filtered for N <= 16, planar from the filtered set for
 * N = 8..32, angular modes per g_intraFilterFlags with edge filters for N <= 16) against fenc */
void orc_intra_scan(const pixel* fenc, intptr_t fencStride, int log2TrSize, const pixel* refBuf, const pixel* fltBuf, int32_t* sa8d35)
{
    int cu = log2TrSize - 2, N = 1 << log2TrSize;
    pixel pred[32 * 32];
    for (int mode = 0; mode < 35; mode++)
    {
        const pixel* src = refBuf;
        if (mode == 0) src = (N >= 8 && N <= 32) ? fltBuf : refBuf;
        else if (mode >= 2) src = (orc_intra_filter_flags(mode) & N) ? fltBuf : refBuf;
        orc_intra_pred(cu, mode, pred, N, src, mode == 0 ? 0 : N <= 16);
        sa8d35[mode] = orc_sa8d(cu, fenc, fencStride, pred, N);
    }
}

typedef struct { uint64_t recon, fenc, avail; int32_t reconStride, fencStride; uint8_t log2, strong, reserved[6]; } PackedIntraJob;
int orc_intra_scan_batch(const PackedIntraJob* jobs, int n, int32_t* sa8d)
{
    pixel rb[258], fb[258];
    uint8_t flags[33];
    for (int i = 0; i < n; i++)
    {
        const PackedIntraJob* j = &jobs[i];
        int total = (1 << j->log2) + 1;
        for (int u = 0; u < total; u++) flags[u] = (uint8_t)((j->avail >> u) & 1);
        orc_init_adi_pattern((const pixel*)j->recon, j->reconStride, j->log2, flags, j->strong, -1, rb, fb);
        orc_intra_scan((const pixel*)j->fenc, j->fencStride, j->log2, rb, j->log2 >= 3 ? fb : rb, sa8d + 35 * i);
    }
    return n;
}
