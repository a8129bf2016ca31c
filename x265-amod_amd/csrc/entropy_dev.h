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
