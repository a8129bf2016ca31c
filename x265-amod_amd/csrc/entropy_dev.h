/* Device code of the entropy estimation tables (see entropy_kernels.hip): shared with the device job server (device_queue.hip). */
#ifndef X265AMD_ENTROPY_DEV_H
#define X265AMD_ENTROPY_DEV_H
#include "x265amd_dev.h"

enum {
    CTX_QT_CBF = 28, CTX_QT_ROOT_CBF = 38, CTX_SIG_CG = 39, CTX_SIG = 43, CTX_LAST_X = 85, CTX_ONE = 121, CTX_ABS = 145,
    N_SIG_LUMA = 27, N_LAST_XY = 18, N_LAST_XY_LUMA = 15, N_ONE_LUMA = 16, N_ABS_LUMA = 4, N_SIG_CG = 2
};

/* FIX15 bits of coding bin b in state s: en_bits[s ^ b] */
__device__ const uint32_t en_bits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb
};
__device__ const uint8_t en_lpsNext[64] = {      /* H.265 table 9-46, transIdxLps */
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63
};

/* g_nextState (entropy.cpp:2627-2645) by rule */
XA_DEV uint8_t en_next(uint8_t state, uint32_t bin)
{
    const uint32_t p = state >> 1, mps = state & 1;
    if (p == 63) return state;
    if (bin == mps) return (uint8_t)(((p < 62 ? p + 1 : 62) << 1) | mps);
    if (p == 0) return (uint8_t)(1 - mps);
    return (uint8_t)((en_lpsNext[p] << 1) | mps);
}

/* the two tables of the estimator where the caller keeps them: global memory (en_bits / en_lpsNext), or a copy in LDS -- a lane that codes bins one after the other
 * waits for every look-up, and an LDS look-up is a third of a cached global one */
struct EnTabs { const uint32_t* bits; const uint8_t* lps; };
XA_DEV uint8_t en_next_t(const EnTabs& t, uint8_t state, uint32_t bin)
{
    const uint32_t p = state >> 1, mps = state & 1;
    if (p == 63) return state;
    if (bin == mps) return (uint8_t)(((p < 62 ? p + 1 : 62) << 1) | mps);
    if (p == 0) return (uint8_t)(1 - mps);
    return (uint8_t)((t.lps[p] << 1) | mps);
}
XA_DEV uint32_t cb_bin_t(const EnTabs& t, uint8_t* st, uint32_t bin) { const uint8_t s = *st; *st = en_next_t(t, s, bin); return t.bits[s ^ bin]; }
XA_DEV uint32_t cb_bin(uint8_t* st, uint32_t bin) { const uint8_t s = *st; *st = en_next(s, bin); return en_bits[s ^ bin]; }

/* the 16 sample offsets of a 4x4 group in scan order `type` (g_scan4x4, constants.cpp:364-400, by rule), packed 4 bits each */
XA_DEV uint32_t cb_in_cg(int type, int k)
{
    const uint64_t t = type == 1 ? 0xFEDCBA9876543210ULL : type == 2 ? 0xFB73EA62D951C840ULL : 0xFBE7AD369C258140ULL;
    return (uint32_t)((t >> (4 * k)) & 15);
}
XA_DEV uint32_t cb_sig_ctx_inc(int log2N, uint32_t pattern, uint32_t rr)
{
    if (log2N == 2) return (uint32_t)((0x8877886654325410ULL >> (4 * rr)) & 15);
    const uint64_t t = pattern == 0 ? 0x0000000100110112ULL : pattern == 1 ? 0x0000000011112222ULL : pattern == 2 ? 0x0012001200120012ULL : 0x2222222222222222ULL;
    return (uint32_t)((t >> (4 * rr)) & 15);
}
/* raster index of group scan position g (g_scanOrderCG, constants.cpp:402-461, by rule): groups in `type` order for the 2x2 grid
 * of an 8x8 TU, up-right diagonal for the 4x4 / 8x8 grids of 16x16 / 32x32 TUs */
struct CbDiag { uint8_t d4[16], d8[64]; };
constexpr CbDiag cb_make_diag()
{
    CbDiag t = {};
    for (int n = 4; n <= 8; n += 4)
    {
        int i = 0;
        for (int d = 0; d < 2 * n - 1; d++)
            for (int y = d < n ? d : n - 1; y >= 0 && d - y < n; y--, i++)
                (n == 4 ? t.d4 : t.d8)[i] = (uint8_t)(y * n + (d - y));
    }
    return t;
}
__device__ const CbDiag cb_diag = cb_make_diag();
XA_DEV uint32_t cb_cg_blk(int type, int log2N, int g)
{
    if (log2N == 2) return 0;
    if (log2N == 3) return ((type == 1 ? 0x3210u : 0x3120u) >> (4 * g)) & 15;
    return log2N == 4 ? cb_diag.d4[g] : cb_diag.d8[g];
}


/* bits-only Entropy::codeCoeffNxN of ONE transform unit by the calling lane (reference: source/encoder/entropy.cpp:1828-2199 with the counting primitives of
 * source/common/dct.cpp:757-993): FIX15 bits; ctx: the lane's own copy of the context states (updated).  Serial by nature -- every coded bin moves its context. */
XA_DEV uint64_t lane_coeff_bits(uint8_t* ctx, const int16_t* coeff, int log2N, int ttype, int intra, int dir_mode, int sign_hide, const EnTabs& tabs)
{
    const int N = 1 << log2N, isLuma = ttype == 0;
    const int scanType = !intra ? 0 : ((log2N <= 2 || (isLuma && log2N == 3)) ? (dir_mode >= 22 && dir_mode <= 30 ? 1 : (dir_mode >= 6 && dir_mode <= 14 ? 2 : 0)) : 0);
    const int cgType = log2N >= 4 ? 0 : scanType;
    const int ncgAll = 1 << (2 * (log2N - 2));
    const uint32_t log2CG = (uint32_t)log2N - 2, cgStride = (uint32_t)N >> 2;
    uint64_t bits = 0;

    /* scanPosLast_c (dct.cpp:757-790) folded in: find the last group / position holding a level, and the group flags */
    int lastSet = -1, lastK = -1;
    uint64_t cgFlags = 0;
    for (int g = ncgAll - 1; g >= 0 && lastSet < 0; g--)
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, g);
        const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
        for (int k = 15; k >= 0; k--)
        {
            const uint32_t rr = cb_in_cg(cgType == 0 && log2N >= 4 ? 0 : scanType, k);
            if (coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)]) { lastSet = g; lastK = k; break; }
        }
    }
    if (lastSet < 0) return 0;
    const int inType = log2N >= 4 ? 0 : scanType;

    /* last position: context-coded prefixes, bypass suffixes (entropy.cpp:1874-1908) */
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, lastSet);
        const uint32_t rr = cb_in_cg(inType, lastK);
        uint32_t px = (blk & ((1u << log2CG) - 1)) * 4 + (rr & 3), py = (blk >> log2CG) * 4 + (rr >> 2);
        if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
        int ctxIdx = isLuma ? 3 * (log2N - 2) + (log2N == 5) : N_LAST_XY_LUMA;
        const int ctxShift = isLuma ? (log2N > 2) : log2N - 2;
        const uint32_t maxGroupIdx = ((uint32_t)log2N << 1) - 1;
        for (int i = 0; i < 2; i++, ctxIdx += N_LAST_XY)
        {
            const uint32_t pos = i ? py : px;
            uint32_t prefix = pos, suffixLen = 0;
            if (pos >= 4) { const uint32_t l = 31 - (uint32_t)__clz((int)pos); suffixLen = l - 1; prefix = 2 * l + ((pos >> (l - 1)) & 1); }
            uint8_t* c = ctx + CTX_LAST_X + ctxIdx;
            for (uint32_t k = 0; k < prefix; k++) bits += cb_bin_t(tabs, c + (k >> ctxShift), 1);
            if (prefix < maxGroupIdx) bits += cb_bin_t(tabs, c + (prefix >> ctxShift), 0);
            bits += (uint64_t)suffixLen << 15;
        }
    }
    /* groups in front of the last one that hold levels (entropy.cpp:1862-1868) */
    for (int g = 0; g < lastSet; g++)
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, g);
        const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
        bool any = false;
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) any |= coeff[base + y * N + x] != 0;
        if (any) cgFlags |= (uint64_t)1 << blk;
    }

    uint8_t* cgCtx = ctx + CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG);
    uint8_t* sigCtx = ctx + CTX_SIG + (isLuma ? 0 : N_SIG_LUMA);
    const int firstSig = log2N == 2 ? 0 : log2N == 3 ? ((scanType != 0 && isLuma) ? 15 : 9) : (isLuma ? 21 : 12);
    uint32_t c1 = 1;
    int sigOff = lastK - 1;
    uint16_t absCoeff[16];
    uint32_t numNonZero = 1;
    {
        const uint32_t blk = cb_cg_blk(cgType, log2N, lastSet);
        const uint32_t rr = cb_in_cg(inType, lastK);
        absCoeff[0] = (uint16_t)abs((int)coeff[(int)((blk >> log2CG) * 4 + (rr >> 2)) * N + (int)((blk & ((1u << log2CG) - 1)) * 4 + (rr & 3))]);
    }
    for (int sub = lastSet; sub >= 0; sub--)
    {
        const int subBase = sub << 4;
        const uint32_t cgBlk = cb_cg_blk(cgType, log2N, sub), cgY = cgBlk >> log2CG, cgX = cgBlk & ((1u << log2CG) - 1);
        const uint64_t cgMask = (uint64_t)1 << cgBlk;
        const int base = (int)(cgY * 4) * N + (int)(cgX * 4);
        uint32_t firstNZ = 16, lastNZ = 0;      /* positions (scan offsets) of the first / last level of this group */
        if (sub == lastSet || !sub) cgFlags |= cgMask;
        else
        {
            const uint32_t sigPos = cgBlk + 1 < 64 ? (uint32_t)(cgFlags >> (cgBlk + 1)) : 0;
            const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
            bits += cb_bin_t(tabs, cgCtx + (right | lower), (cgFlags & cgMask) != 0);
        }
        if (sub == lastSet) { firstNZ = lastNZ = (uint32_t)lastK; }
        if (sigOff >= 0 && (cgFlags & cgMask))
        {
            /* costCoeffNxN_c (dct.cpp:838-890) */
            uint32_t pattern = 0;
            if (cgStride != 1)
            {
                const uint32_t sigPos = cgBlk + 1 < 64 ? (uint32_t)(cgFlags >> (cgBlk + 1)) : 0;
                const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
                pattern = right + lower * 2;
            }
            const int offset = firstSig + ((isLuma && sub) ? 3 : 0);
            uint32_t nnz = sigOff < 15 ? 1 : 0;
            uint32_t sum = 0;
            for (int k = sigOff; k >= 0; k--)
            {
                const uint32_t rr = cb_in_cg(inType, k);
                const int v = coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)];
                const uint32_t sig = v != 0;
                if (k != 0 || subBase == 0 || nnz)
                {
                    const uint32_t ctxSig = (subBase + k) ? cb_sig_ctx_inc(log2N, pattern, rr) + (uint32_t)offset : 0;
                    sum += cb_bin_t(tabs, sigCtx + ctxSig, sig);
                }
                if (sig)
                {
                    absCoeff[nnz] = (uint16_t)abs(v);
                    if (firstNZ == 16 || (uint32_t)k < firstNZ) firstNZ = (uint32_t)k;
                    if (nnz == 0) lastNZ = (uint32_t)k;
                }
                nnz += sig;
            }
            bits += sum & 0xFFFFFF;
            numNonZero = nnz;
        }
        else if (sub != lastSet) numNonZero = 0;
        if (numNonZero > 0)
        {
            const bool signHidden = lastNZ - firstNZ >= 4;
            const uint32_t ctxSet = (((sub > 0) + (uint32_t)isLuma) & 2) + !(c1 & 3);
            uint8_t* oneCtx = ctx + CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + 4 * ctxSet;
            const uint32_t numC1 = numNonZero < 8 ? numNonZero : 8;
            /* costC1C2Flag_c (dct.cpp:942-993) */
            uint32_t sum = 0, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFE;
            c1 = 1;
            for (uint32_t idx = 0; idx < numC1; idx++)
            {
                const uint32_t s1 = absCoeff[idx] > 1, s2 = absCoeff[idx] > 2;
                sum += cb_bin_t(tabs, oneCtx + c1, s1);
                if (s1) c1Next = 0;
                if (s1 + firstC2Flag == 3) firstC2Flag = s2;
                if (s1 + firstC2Idx == 9) firstC2Idx = idx;
                c1 = c1Next & 3;
                c1Next >>= 2;
            }
            if (!c1) sum += cb_bin_t(tabs, ctx + CTX_ABS + (isLuma ? 0 : N_ABS_LUMA) + ctxSet, firstC2Flag);
            bits += sum & 0x00FFFFFF;
            bits += (uint64_t)(numNonZero - ((sign_hide && signHidden) ? 1 : 0)) << 15;
            if (numNonZero > firstC2Idx)
            {
                /* costCoeffRemain_c (dct.cpp:892-938) */
                uint32_t rice = 0, rsum = 0;
                int baseLevel = 3;
                for (uint32_t idx = firstC2Idx; idx < numNonZero; idx++)
                {
                    if (idx >= 8) baseLevel = 1;
                    int code = (int)absCoeff[idx] - baseLevel;
                    if (code >= 0)
                    {
                        code = (int)((uint32_t)code >> rice) - 3;
                        if (code >= 0)
                        {
                            const uint32_t length = 31 - (uint32_t)__clz(code + 1);
                            code = (int)(length + length);
                        }
                        rsum += (uint32_t)(3 + 1 + (int)rice + code);
                        if (absCoeff[idx] > (3u << rice)) rice = (rice + 1) - (rice >> 2);
                    }
                    baseLevel = 2;
                }
                bits += (uint64_t)rsum << 15;
            }
        }
        numNonZero = 0;
        sigOff = 15;
    }
    return bits;
}

/* ---- the same count for a 4x4 unit by a whole wavefront ----
 * A context's state moves only with the bins coded in THAT context, so the contexts of a unit are independent state machines: the lanes first lay out, from the
 * levels alone, which bins every context codes and in what order (ballots over the sixteen scan positions), then one lane per context walks its own few bins
 * (at most seven for a 4x4 unit) while the others walk theirs -- instead of one lane coding some forty bins one after the other, each a chain of dependent
 * look-ups.  One look-up per bin: en_step[(state << 1) | bin] = bits | next state << 24.  Same FIX15 total as lane_coeff_bits(.., log2N = 2, ..).
 * ctx: the context set (read); ctxOut: where the moved states go (may equal ctx; NULL: nowhere -- a candidate that is only priced).  All 64 lanes call. */
struct EnStep { uint32_t v[256]; };
constexpr EnStep en_make_step()
{
    constexpr uint32_t bitsTab[128] = {
        0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
        0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
        0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
        0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
        0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
        0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
        0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
        0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };
    constexpr uint8_t lps[64] = {
        0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
        24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
    EnStep t = {};
    for (uint32_t state = 0; state < 128; state++)
        for (uint32_t bin = 0; bin < 2; bin++)
        {
            const uint32_t pp = state >> 1, mps = state & 1;
            uint32_t next = state;
            if (pp != 63)
            {
                if (bin == mps) next = ((pp < 62 ? pp + 1 : 62) << 1) | mps;
                else if (pp == 0) next = 1 - mps;
                else next = ((uint32_t)lps[pp] << 1) | mps;
            }
            t.v[(state << 1) | bin] = bitsTab[state ^ bin] | (next << 24);
        }
    return t;
}
__device__ const EnStep en_step = en_make_step();

XA_DEV uint64_t wave_coeff_bits_4x4(const uint8_t* ctx, uint8_t* ctxOut, const int16_t* coeff, int ttype, int intra, int dir_mode, int sign_hide, const uint32_t* step, int lane)
{
    const int isLuma = ttype == 0;
    const int scanType = !intra ? 0 : (dir_mode >= 22 && dir_mode <= 30 ? 1 : (dir_mode >= 6 && dir_mode <= 14 ? 2 : 0));
    /* lanes 0..15: the sixteen scan positions */
    const uint32_t rrMine = cb_in_cg(scanType, lane & 15);
    const int v = lane < 16 ? (int)coeff[rrMine] : 0;
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    const uint32_t sigM = (uint32_t)__ballot(a != 0) & 0xFFFFu;
    if (!sigM) return 0;
    const uint32_t gt1M = (uint32_t)__ballot(a > 1) & 0xFFFFu, gt2M = (uint32_t)__ballot(a > 2) & 0xFFFFu;
    const int lastK = 31 - __clz((int)sigM), firstK = __ffs((int)sigM) - 1;
    const uint32_t nnz = (uint32_t)__popc(sigM);
    /* the levels that get greater-1 flags: the first eight in coding order (descending scan position) */
    uint32_t first8 = sigM;
    for (uint32_t drop = nnz > 8 ? nnz - 8 : 0; drop; drop--) first8 &= first8 - 1;
    const uint32_t g1 = gt1M & first8;
    const int s1K = g1 ? 31 - __clz((int)g1) : -1;                  /* scan position of the first level above 1, in coding order */
    /* the scan positions of every significance context (the 4x4 map of the raster position; scan position 0 is raster position 0, context 0) */
    const uint32_t myCtx = lane < 16 ? cb_sig_ctx_inc(2, 0, rrMine) : 99u;
    uint32_t todo = 0, flags = 0;
    int ci = -1;
    for (uint32_t c = 0; c < 9; c++)
    {
        const uint32_t m = (uint32_t)__ballot(myCtx == c);
        if ((uint32_t)lane == c) todo = m;
    }
    if (lane < 9)
    {
        /* significance flags below the last level, high to low */
        todo &= (1u << lastK) - 1u; flags = sigM;
        ci = CTX_SIG + (isLuma ? 0 : N_SIG_LUMA) + lane;
    }
    else if (lane >= 16 && lane < 20)
    {
        /* greater-1 flags, context set 0 (one group, nothing before it): context 1 for the first level, 2 for the second, 3 up to the first level above 1,
         * 0 behind it (costC1C2Flag_c's c1) */
        const int j = lane - 16;
        const uint32_t pos0 = 1u << lastK;
        const uint32_t rem = first8 & ~pos0;
        const uint32_t pos1 = rem ? 1u << (31 - __clz((int)rem)) : 0u;
        const uint32_t rem2 = rem & ~pos1;
        const uint32_t below = s1K >= 0 ? first8 & ((1u << s1K) - 1u) : 0u;
        todo = j == 0 ? below : (j == 1 ? pos0 : (j == 2 ? pos1 & ~below : rem2 & ~below));
        flags = gt1M;
        ci = CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + j;
    }
    else if (lane == 20)
    {
        /* the greater-2 flag of the first level above 1 */
        todo = s1K >= 0 ? 1u << s1K : 0u; flags = gt2M;
        ci = CTX_ABS + (isLuma ? 0 : N_ABS_LUMA);
    }
    else if (lane >= 24 && lane < 30)
    {
        /* last position: bin i of the x (lanes 24..26) or y (27..29) prefix, each in a context of its own */
        const uint32_t rr = cb_in_cg(scanType, lastK);
        uint32_t px = rr & 3, py = rr >> 2;
        if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
        const int isY = lane >= 27, i = lane - (isY ? 27 : 24);
        const uint32_t pos = isY ? py : px;
        const bool one = (uint32_t)i < pos, zero = (uint32_t)i == pos && pos < 3;
        todo = (one || zero) ? 1u : 0u; flags = one ? 1u : 0u;
        ci = CTX_LAST_X + (isLuma ? 0 : N_LAST_XY_LUMA) + (isY ? N_LAST_XY : 0) + i;
    }
    else todo = 0;
    uint32_t sum = 0;
    if (todo)
    {
        uint32_t st = ctx[ci];
        do
        {
            const int k = 31 - __clz((int)todo);
            todo &= ~(1u << k);
            const uint32_t e = step[(st << 1) | ((flags >> k) & 1u)];
            sum += e & 0xFFFFFFu; st = e >> 24;
        } while (todo);
        if (ctxOut) ctxOut[ci] = (uint8_t)st;
    }
    /* sign bits, and the escape codes of the levels from the first one above 1 on (costCoeffRemain_c): serial in the Rice parameter, but plain arithmetic
     * on broadcast values -- every lane runs the short loop */
    uint32_t bypass = nnz - ((sign_hide && lastK - firstK >= 4) ? 1u : 0u);
    const uint32_t startIdx = s1K >= 0 ? (uint32_t)__popc(sigM >> (s1K + 1)) : 8u;
    if (nnz > startIdx)
    {
        uint32_t rest = s1K >= 0 ? sigM & ((2u << s1K) - 1u) : sigM & ~first8;
        uint32_t idx = startIdx, rice = 0;
        int baseLevel = 3;
        while (rest)
        {
            const int k = 31 - __clz((int)rest);
            rest &= ~(1u << k);
            if (idx >= 8) baseLevel = 1;
            const uint32_t av = (uint32_t)__builtin_amdgcn_readlane((int)a, k);
            int code = (int)av - baseLevel;
            if (code >= 0)
            {
                code = (int)((uint32_t)code >> rice) - 3;
                if (code >= 0) { const uint32_t length = 31 - (uint32_t)__clz(code + 1); code = (int)(length + length); }
                bypass += (uint32_t)(3 + 1 + (int)rice + code);
                if (av > (3u << rice)) rice = (rice + 1) - (rice >> 2);
            }
            baseLevel = 2;
            idx++;
        }
    }
    return (uint64_t)xa_wave_sum(sum) + ((uint64_t)bypass << 15);
}

/* ---- and for every unit size ----
 * The same idea over 64 scan positions at a time (a lane per position, highest positions first -- the coding order), the states carried from one span to the
 * next by the lanes that own them: lanes 0..6 the significance contexts this unit can reach (context 0; three for the first group, three for the others,
 * which chroma shares with the first), 8..23 the greater-1 contexts (set x c1), 24..27 the greater-2 contexts, 28..29 the group flags, 32..36 / 40..44 the
 * last-position prefixes.  What a group needs from the groups coded before it -- the pattern of its right / lower neighbours, whether the previous group with
 * levels met a level above 1 (the context set) -- comes from the levels alone, so a first pass lays it out with a lane per group; the escape codes and sign
 * bits are per group and need no context (a lane per group).  Same FIX15 total and the same final states as lane_coeff_bits. */
XA_DEV uint64_t xa_wave_or64(uint64_t v)
{
    for (int off = 32; off; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}
XA_DEV uint64_t wave_coeff_bits(const uint8_t* ctx, uint8_t* ctxOut, const int16_t* coeff, int log2N, int ttype, int intra, int dir_mode, int sign_hide, const uint32_t* step, int lane)
{
    if (log2N == 2) return wave_coeff_bits_4x4(ctx, ctxOut, coeff, ttype, intra, dir_mode, sign_hide, step, lane);
    const int N = 1 << log2N, isLuma = ttype == 0;
    const int scanType = (intra && isLuma && log2N == 3) ? (dir_mode >= 22 && dir_mode <= 30 ? 1 : (dir_mode >= 6 && dir_mode <= 14 ? 2 : 0)) : 0;
    const int ncg = 1 << (2 * (log2N - 2)), nChunks = ncg >> 2;
    const uint32_t log2CG = (uint32_t)log2N - 2, cgStride = (uint32_t)N >> 2;
    /* ---- pass 1: a lane per group (in group scan order) ---- */
    const uint32_t myBlk = lane < ncg ? cb_cg_blk(scanType, log2N, lane) : 0;
    uint32_t myNz = 0, myG1 = 0;
    for (int c = 0; c < nChunks; c++)
    {
        const int sub = c * 4 + (lane >> 4), k = lane & 15;
        const uint32_t blk = (uint32_t)__shfl((int)myBlk, sub, 64), rr = cb_in_cg(scanType, k);
        const int v = coeff[(int)((blk >> log2CG) * 4 + (rr >> 2)) * N + (int)((blk & (cgStride - 1)) * 4 + (rr & 3))];
        const uint64_t nzB = __ballot(v != 0), g1B = __ballot(v > 1 || v < -1);
        if ((lane >> 2) == c) { myNz = (uint32_t)(nzB >> (16 * (lane & 3))) & 0xFFFFu; myG1 = (uint32_t)(g1B >> (16 * (lane & 3))) & 0xFFFFu; }
    }
    const uint64_t cgNZ = __ballot(myNz != 0);
    if (!cgNZ) return 0;
    const int lastSet = 63 - __clzll((long long)cgNZ);
    const int lastK = 31 - __clz(__builtin_amdgcn_readlane((int)myNz, lastSet));
    const int lastS = lastSet * 16 + lastK;
    const uint64_t flagsByBlk = xa_wave_or64(myNz ? (uint64_t)1 << myBlk : 0);
    uint32_t myPattern, myCgInc;
    {
        const uint32_t cgY = myBlk >> log2CG, cgX = myBlk & (cgStride - 1);
        const uint32_t sigPos = myBlk + 1 < 64 ? (uint32_t)(flagsByBlk >> (myBlk + 1)) : 0;
        const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
        myPattern = right + lower * 2; myCgInc = right | lower;
    }
    uint32_t myFirst8 = myNz;
    for (uint32_t drop = __popc(myNz) > 8 ? (uint32_t)__popc(myNz) - 8 : 0; drop; drop--) myFirst8 &= myFirst8 - 1;
    const uint32_t myS1 = myG1 & myFirst8;
    const int myS1K = myS1 ? 31 - __clz((int)myS1) : -1;
    const uint64_t s1CG = __ballot(myS1 != 0);
    uint32_t myCtxSet;
    {
        const uint64_t above = lane < 63 ? cgNZ >> (lane + 1) : 0;
        const int prevG = above ? lane + 1 + (__ffsll((long long)above) - 1) : -1;
        const uint32_t c1zero = prevG >= 0 ? (uint32_t)(s1CG >> prevG) & 1u : 0u;
        myCtxSet = ((((uint32_t)(lane > 0)) + (uint32_t)isLuma) & 2u) + c1zero;
    }
    /* sign bits and escape codes of my group (costCoeffRemain_c): no contexts */
    uint32_t bypass = 0;
    if (myNz)
    {
        const uint32_t nnz = (uint32_t)__popc(myNz);
        const int hiK = 31 - __clz((int)myNz), loK = __ffs((int)myNz) - 1;
        bypass = nnz - ((sign_hide && hiK - loK >= 4) ? 1u : 0u);
        const uint32_t startIdx = myS1K >= 0 ? (uint32_t)__popc(myNz >> (myS1K + 1)) : 8u;
        if (nnz > startIdx)
        {
            uint32_t rest = myS1K >= 0 ? myNz & ((2u << myS1K) - 1u) : myNz & ~myFirst8;
            uint32_t idx = startIdx, rice = 0;
            int baseLevel = 3;
            const int base = (int)((myBlk >> log2CG) * 4) * N + (int)((myBlk & (cgStride - 1)) * 4);
            while (rest)
            {
                const int k = 31 - __clz((int)rest);
                rest &= ~(1u << k);
                if (idx >= 8) baseLevel = 1;
                const uint32_t rr = cb_in_cg(scanType, k);
                const int v = coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)];
                const uint32_t av = (uint32_t)(v < 0 ? -v : v);
                int code = (int)av - baseLevel;
                if (code >= 0)
                {
                    code = (int)((uint32_t)code >> rice) - 3;
                    if (code >= 0) { const uint32_t length = 31 - (uint32_t)__clz(code + 1); code = (int)(length + length); }
                    bypass += (uint32_t)(3 + 1 + (int)rice + code);
                    if (av > (3u << rice)) rice = (rice + 1) - (rice >> 2);
                }
                baseLevel = 2;
                idx++;
            }
        }
    }
    /* ---- the lanes that own a context ---- */
    const int firstSig = log2N == 3 ? ((scanType != 0 && isLuma) ? 15 : 9) : (isLuma ? 21 : 12);
    int ci = -1;
    if (lane < 7) ci = CTX_SIG + (isLuma ? 0 : N_SIG_LUMA) + (lane ? firstSig + lane - 1 : 0);
    else if (lane >= 8 && lane < 24) ci = CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + (lane - 8);
    else if (lane >= 24 && lane < 28) ci = CTX_ABS + (isLuma ? 0 : N_ABS_LUMA) + (lane - 24);
    else if (lane >= 28 && lane < 30) ci = CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG) + (lane - 28);
    if (!isLuma && ((lane >= 4 && lane < 7) || (lane >= 16 && lane < 24) || (lane >= 26 && lane < 28))) ci = -1;       /* chroma has half the sets and no second significance triple */
    uint32_t st = ci >= 0 ? ctx[ci] : 0;
    bool touched = false;
    uint64_t sum = 0;
    /* ---- pass 2: 64 scan positions at a time, from the span of the last level down ---- */
    for (int c = lastSet >> 2; c >= 0; c--)
    {
        const int sub = c * 4 + (lane >> 4), k = lane & 15, S = sub * 16 + k;
        const uint32_t blk = (uint32_t)__shfl((int)myBlk, sub, 64), rr = cb_in_cg(scanType, k);
        const int v = coeff[(int)((blk >> log2CG) * 4 + (rr >> 2)) * N + (int)((blk & (cgStride - 1)) * 4 + (rr & 3))];
        const uint32_t a = (uint32_t)(v < 0 ? -v : v);
        const uint32_t nz16 = (uint32_t)__shfl((int)myNz, sub, 64), ctxSetS = (uint32_t)__shfl((int)myCtxSet, sub, 64), patt = (uint32_t)__shfl((int)myPattern, sub, 64);
        const uint32_t cgInc = (uint32_t)__shfl((int)myCgInc, sub, 64);
        const int s1Ksub = __shfl(myS1K, sub, 64);
        const bool live = sub <= lastSet;
        /* significance flags (costCoeffNxN_c): every position below the last level in a flagged group (group 0 always is), except a group's position 0
         * when nothing else in the group holds a level (then it is implied) */
        const bool sigCoded = live && S < lastS && (nz16 != 0 || sub == 0) && !(k == 0 && sub != 0 && sub != lastSet && (nz16 & 0xFFFEu) == 0);
        const uint32_t sigId = S == 0 ? 0u : 1u + cb_sig_ctx_inc(log2N, patt, rr) + ((isLuma && sub) ? 3u : 0u);
        /* greater-1 flags (costC1C2Flag_c): the first eight levels of a group; c1 = 1, 2, 3, 3, ... up to the first level above 1, 0 behind it */
        const uint32_t idxInCG = (uint32_t)__popc(nz16 >> (k + 1));
        const bool c1Coded = live && a != 0 && idxInCG < 8;
        const uint32_t c1v = (s1Ksub >= 0 && k < s1Ksub) ? 0u : (idxInCG + 1 < 3 ? idxInCG + 1 : 3u);
        const uint32_t c1Id = 4 * ctxSetS + c1v;
        const bool absCoded = live && a != 0 && k == s1Ksub;
        const bool cgCoded = k == 15 && sub > 0 && sub < lastSet;
        const uint64_t sigB = __ballot(a != 0), g1B = __ballot(a > 1), g2B = __ballot(a > 2), cgB = __ballot(k == 15 && nz16 != 0);
        uint64_t mask = 0, bins = 0;
        for (uint32_t id = 0; id < 7; id++) { const uint64_t m = __ballot(sigCoded && sigId == id); if ((uint32_t)lane == id) { mask = m; bins = sigB; } }
        for (uint32_t id = 0; id < (isLuma ? 16u : 8u); id++) { const uint64_t m = __ballot(c1Coded && c1Id == id); if ((uint32_t)lane == 8 + id) { mask = m; bins = g1B; } }
        for (uint32_t id = 0; id < (isLuma ? 4u : 2u); id++) { const uint64_t m = __ballot(absCoded && ctxSetS == id); if ((uint32_t)lane == 24 + id) { mask = m; bins = g2B; } }
        for (uint32_t id = 0; id < 2; id++) { const uint64_t m = __ballot(cgCoded && cgInc == id); if ((uint32_t)lane == 28 + id) { mask = m; bins = cgB; } }
        if (ci >= 0 && mask)
        {
            touched = true;
            do
            {
                const int kk = 63 - __clzll((long long)mask);
                mask &= ~((uint64_t)1 << kk);
                const uint32_t e = step[(st << 1) | ((uint32_t)(bins >> kk) & 1u)];
                sum += e & 0xFFFFFFu; st = e >> 24;
            } while (mask);
        }
    }
    /* ---- last position (entropy.cpp:1874-1908): a lane per prefix context, x in lanes 32.., y in lanes 40.. ---- */
    {
        const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)myBlk, lastSet), rr = cb_in_cg(scanType, lastK);
        uint32_t px = (blk & (cgStride - 1)) * 4 + (rr & 3), py = (blk >> log2CG) * 4 + (rr >> 2);
        if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
        const int ctxIdx = isLuma ? 3 * (log2N - 2) + (log2N == 5) : N_LAST_XY_LUMA;
        const int ctxShift = isLuma ? 1 : log2N - 2;
        const uint32_t maxGroupIdx = ((uint32_t)log2N << 1) - 1;
        uint32_t suffix = 0;
        for (int i = 0; i < 2; i++)
        {
            const uint32_t pos = i ? py : px;
            uint32_t prefix = pos, suffixLen = 0;
            if (pos >= 4) { const uint32_t l = 31 - (uint32_t)__clz((int)pos); suffixLen = l - 1; prefix = 2 * l + ((pos >> (l - 1)) & 1); }
            suffix += suffixLen;
            const int j = lane - (i ? 40 : 32);
            if (j >= 0 && j < 8)
            {
                /* bins k with (k >> ctxShift) == j: ones below the prefix, a zero at it unless the prefix is the largest */
                const uint32_t nBins = prefix + (prefix < maxGroupIdx ? 1u : 0u);
                const uint32_t k0 = (uint32_t)j << ctxShift, k1 = ((uint32_t)j + 1) << ctxShift;
                if (k0 < nBins)
                {
                    const int cix = CTX_LAST_X + ctxIdx + (i ? N_LAST_XY : 0) + j;
                    uint32_t s2 = ctx[cix];
                    for (uint32_t kb = k0; kb < k1 && kb < nBins; kb++)
                    {
                        const uint32_t e = step[(s2 << 1) | (kb < prefix ? 1u : 0u)];
                        sum += e & 0xFFFFFFu; s2 = e >> 24;
                    }
                    if (ctxOut) ctxOut[cix] = (uint8_t)s2;
                }
            }
        }
        if (lane == 0) bypass += suffix;
    }
    if (ctxOut && touched) ctxOut[ci] = (uint8_t)st;
    return xa_wave_sum(sum) + ((uint64_t)xa_wave_sum(bypass) << 15);
}

/* =========================================================================================================
 * estBit: one wavefront per job; lanes fan out over the table entries
 * ======================================================================================================= */
#define EST_WAVES 4
/* one table of a job list on one wavefront */
XA_DEV void wave_est_bit(const uint8_t* ctx, int32_t* e, int log2N, int isLuma, int lane);
XA_DEV void wave_est_bit_job(const x265amd_est_job* jobs, int ji, int lane)
{
    const x265amd_est_job j = xa_ld_record(jobs + ji);
    wave_est_bit(reinterpret_cast<const uint8_t*>(j.ctx), reinterpret_cast<int32_t*>(j.est), j.log2_tr_size, j.is_luma, lane);
}
/* the table entries Entropy::estBit fills for (log2N, isLuma) from the contexts at ctx (any memory: a fused command makes its tables in LDS) */
XA_DEV void wave_est_bit(const uint8_t* ctx, int32_t* e, int log2N, int isLuma, int lane)
{
    /* blockCbpBits [168..181], blockRootCbpBits [182..183], significantCoeffGroupBits [0..3] */
    if (lane < 14) e[168 + lane] = (int32_t)en_bits[ctx[CTX_QT_CBF + (lane >> 1)] ^ (lane & 1)];
    if (lane < 2) e[182 + lane] = (int32_t)en_bits[ctx[CTX_QT_ROOT_CBF] ^ lane];
    if (lane < 4) e[lane] = (int32_t)en_bits[ctx[CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG) + (lane >> 1)] ^ (lane & 1)];
    /* significantBits[bin][ctx] at 4 + bin * 42 + ctx: context 0 and the contexts of this size */
    int first = 1, num = 8;
    if (log2N >= 4) { first = isLuma ? 21 : 12; num = isLuma ? 6 : 3; }
    else if (log2N == 3) { first = 9; num = isLuma ? 12 : 3; }
    const uint8_t* sig = ctx + CTX_SIG + (isLuma ? 0 : N_SIG_LUMA);
    if (lane < 2 * (num + 1))
    {
        const int bin = lane & 1, t = lane >> 1, c = t == 0 ? 0 : first + t - 1;
        e[4 + bin * 42 + c] = (int32_t)en_bits[sig[c] ^ bin];
    }
    /* greaterOneBits [108..155], levelAbsBits [156..167] */
    const uint8_t* one = ctx + CTX_ONE + (isLuma ? 0 : N_ONE_LUMA);
    const uint8_t* ab = ctx + CTX_ABS + (isLuma ? 0 : N_ABS_LUMA);
    if (lane < (isLuma ? 32 : 16)) e[108 + lane] = (int32_t)en_bits[one[lane >> 1] ^ (lane & 1)];
    if (lane < (isLuma ? 8 : 4)) e[156 + lane] = (int32_t)en_bits[ab[lane >> 1] ^ (lane & 1)];
    /* lastBits[i][group] at 88 + i * 10 + group: prefix sums of the truncated-unary code (entropy.cpp:2287-2350) */
    if (lane < 2)
    {
        const int i = lane;
        const uint8_t* st = ctx + CTX_LAST_X + i * N_LAST_XY;
        int32_t* last = e + 88 + i * 10;
        const int maxGroupIdx = log2N * 2 - 1;
        int bits = 0;
        if (isLuma && log2N == 2)
        {
            for (int c = 0; c < 3; c++) { last[c] = bits + (int)en_bits[st[c]]; bits += (int)en_bits[st[c] ^ 1]; }
            last[maxGroupIdx] = bits;
        }
        else if (isLuma)
        {
            const int off = (log2N - 2) * 3 + (log2N == 5);
            int lastVal = 0;
            for (int c = 0; c < (maxGroupIdx >> 1) + 1; c++)
            {
                const int c0 = (int)en_bits[st[off + c]], c1 = (int)en_bits[st[off + c] ^ 1];
                last[2 * c] = bits + c0;
                lastVal = bits + c1 + c0;
                if (2 * c + 1 != maxGroupIdx) last[2 * c + 1] = lastVal;
                bits += 2 * c1;
            }
            last[maxGroupIdx] = lastVal - (int)en_bits[st[off + (maxGroupIdx >> 1)]];
        }
        else
        {
            const int shift = log2N - 2;
            for (int c = 0; c < maxGroupIdx; c++)
            {
                const int o = N_LAST_XY_LUMA + (c >> shift);
                last[c] = bits + (int)en_bits[st[o]];
                bits += (int)en_bits[st[o] ^ 1];
            }
            last[maxGroupIdx] = bits;
        }
    }
}
#endif
