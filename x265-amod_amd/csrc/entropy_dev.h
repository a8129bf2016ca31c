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

/* =========================================================================================================
 * estBit: one wavefront per job; lanes fan out over the table entries
 * ======================================================================================================= */
#define EST_WAVES 4
/* one table of a job list on one wavefront */
XA_DEV void wave_est_bit_job(const x265amd_est_job* jobs, int ji, int lane)
{
    const x265amd_est_job j = xa_ld_record(jobs + ji);
    const uint8_t* ctx = reinterpret_cast<const uint8_t*>(j.ctx);
    int32_t* e = reinterpret_cast<int32_t*>(j.est);
    const int log2N = j.log2_tr_size, isLuma = j.is_luma;
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
