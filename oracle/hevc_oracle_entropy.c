/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the entropy-side arithmetic the residual path needs:
 *   - CABAC context initialisation                 Entropy::resetEntropy / sbacInit   (source/encoder/entropy.cpp:1300-1355)
 *   - the fractional-bit tables of a TU            Entropy::estBit and helpers        (entropy.cpp:2220-2390)
 *   - rate-distortion optimised quantisation       Quant::rdoQuant<log2TrSize>        (source/common/quant.cpp:609-1424)
 *   - bits-only coding of a TU's levels            Entropy::codeCoeffNxN, !m_bitIf    (entropy.cpp:1828-2200) with the
 *     C primitives it calls (scanPosLast, costCoeffNxN, costC1C2Flag, costCoeffRemain: source/common/dct.cpp:757-1000)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file; the product never does.
 * Pinned against the reference itself by tests/test_entropy_oracle_vs_ref.py (oracle/_ref) and by the committed vectors
 * of tests/golden/entropy_golden.npz.
 *
 * Data tables: the context initialisation values and the LPS transition rule are those of ITU-T H.265 (9.3.2.2 tables
 * 9-5..9-37, table 9-46) arranged in the reference's context order (source/common/contexts.h:75-106); the 128
 * fractional-bit constants are the reference's own fixed-point -log2(p) table (entropy.cpp:2614-2625).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_DEPTH
#define ORC_DEPTH 8
#endif

uint32_t orc_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff);
const uint16_t* orc_tbl_scan(int scanType, int log2TrSize);
int orc_scan_type(int bIntra, int bIsLuma, int log2TrSize, int dirMode);
double orc_lambda(int qp);
double orc_lambda2(int qp);

#define QP_BD_OFFSET (6 * (ORC_DEPTH - 8))
enum { SCAN_DIAG = 0, SCAN_HOR = 1, SCAN_VER = 2 };

/* context layout (contexts.h:75-106) */
enum {
    CTX_SPLIT = 0, CTX_SKIP = 3, CTX_MERGE_FLAG = 6, CTX_MERGE_IDX = 7, CTX_PART_SIZE = 8, CTX_PRED_MODE = 12, CTX_ADI = 13,
    CTX_CHROMA_PRED = 14, CTX_DELTA_QP = 16, CTX_INTER_DIR = 19, CTX_REF_NO = 24, CTX_MV_RES = 26, CTX_QT_CBF = 28,
    CTX_TRANS_SUBDIV = 35, CTX_QT_ROOT_CBF = 38, CTX_SIG_CG = 39, CTX_SIG = 43, CTX_LAST_X = 85, CTX_LAST_Y = 103,
    CTX_ONE = 121, CTX_ABS = 145, CTX_MVP_IDX = 151, CTX_SAO_MERGE = 152, CTX_SAO_TYPE = 153, CTX_TSKIP = 154,
    CTX_TQ_BYPASS = 156, CTX_COUNT = 157
};
enum { N_SIG_LUMA = 27, N_LAST_XY = 18, N_LAST_XY_LUMA = 15, N_ONE_LUMA = 16, N_ABS_LUMA = 4, N_SIG_CG = 2 };

/* initValue per [slice type: 0 B, 1 P, 2 I][context] */
static const uint8_t k_ctxInit[3][CTX_COUNT] = {
{107,139,126,197,185,201,154,137,154,139,154,154,134,183,152,139,154,154,154,95,79,63,31,31,153,153,169,198,153,111,149,92,167,154,154,224,167,122,79,121,140,61,154,170,154,139,153,139,123,123,63,124,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,170,153,138,138,122,121,122,121,167,151,183,140,151,183,140,125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,125,110,124,110,95,94,125,111,111,79,125,126,111,111,79,108,123,93,154,196,167,167,154,152,167,182,182,134,149,136,153,121,136,122,169,208,166,167,154,152,167,182,107,167,91,107,107,167,168,153,160,139,139,154},
{107,139,126,197,185,201,110,122,154,139,154,154,149,154,152,139,154,154,154,95,79,63,31,31,153,153,140,198,153,111,149,107,167,154,154,124,138,94,79,121,140,61,154,155,154,139,153,139,123,123,63,153,166,183,140,136,153,154,166,183,140,136,153,154,166,183,140,136,153,154,170,153,123,123,107,121,107,121,167,151,183,140,151,183,140,125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,125,110,94,110,95,79,125,111,110,78,110,111,111,95,94,108,123,108,154,196,196,167,154,152,167,182,182,134,149,136,153,121,136,137,169,194,166,167,154,167,137,182,107,167,91,122,107,167,168,153,185,139,139,154},
{139,141,157,154,154,154,154,154,184,154,154,154,154,184,63,139,154,154,154,154,154,154,154,154,154,154,154,154,111,141,94,138,182,154,154,153,138,138,154,91,171,134,141,111,111,125,110,110,94,124,108,124,107,125,141,179,153,125,107,125,141,179,153,125,107,125,141,179,153,125,140,139,182,182,152,136,152,136,153,136,139,111,136,139,111,110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,110,110,124,125,140,153,125,127,140,109,111,143,127,111,79,108,123,63,140,92,137,138,140,152,138,139,153,74,149,92,139,107,122,152,140,179,166,182,140,227,122,197,138,153,136,167,152,152,154,153,200,139,139,154},
};

/* FIX15 bits of coding bin b in state s: k_bits[s ^ b]  (entropy.cpp:2614-2625) */
static const uint32_t k_bits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb
};

/* H.265 table 9-46, transIdxLps */
static const uint8_t k_lpsNext[64] = {
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63
};

/* state = (pStateIdx << 1) | valMps.  g_nextState (entropy.cpp:2627-2645) by rule */
uint8_t orc_ctx_next(uint8_t state, int bin)
{
    int p = state >> 1, mps = state & 1;
    if (p == 63) return state;                          /* terminate state never moves */
    if (bin == mps) return (uint8_t)(((p < 62 ? p + 1 : 62) << 1) | mps);
    if (p == 0) return (uint8_t)(1 - mps);
    return (uint8_t)((k_lpsNext[p] << 1) | mps);
}
uint32_t orc_ctx_bits(uint8_t state, int bin) { return k_bits[state ^ bin]; }

/* sbacInit (entropy.cpp:1300-1311) */
static uint8_t ctx_init_state(int qp, int initValue)
{
    if (qp < 0) qp = 0;
    if (qp > 51) qp = 51;
    int slope = (initValue >> 4) * 5 - 45, offset = ((initValue & 15) << 3) - 16;
    int s = ((slope * qp) >> 4) + offset;
    if (s < 1) s = 1;
    if (s > 126) s = 126;
    int mps = s >= 64;
    return (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
}

/* Entropy::resetEntropy (entropy.cpp:1321-1355): context states at slice start */
void orc_entropy_reset(int sliceType, int qp, uint8_t* ctx)
{
    for (int i = 0; i < CTX_COUNT; i++) ctx[i] = ctx_init_state(qp, k_ctxInit[sliceType][i]);
}

/* same member order and sizes as struct EstBitsSbac (entropy.h:88-97): 184 ints */
typedef struct OrcEstBits
{
    int sigCG[N_SIG_CG][2];
    int sig[2][42];
    int last[2][10];
    int greaterOne[24][2];
    int levelAbs[6][2];
    int cbf[7][2];
    int rootCbf[2];
} OrcEstBits;

/* Entropy::estBit (entropy.cpp:2220-2390).  Entries the reference leaves untouched for this (size, plane) are left
 * untouched here as well. */
void orc_est_bit(const uint8_t* ctx, int log2TrSize, int isLuma, OrcEstBits* e)
{
    for (int i = 0; i < 7; i++) for (int b = 0; b < 2; b++) e->cbf[i][b] = (int)k_bits[ctx[CTX_QT_CBF + i] ^ b];
    for (int b = 0; b < 2; b++) e->rootCbf[b] = (int)k_bits[ctx[CTX_QT_ROOT_CBF] ^ b];
    for (int i = 0; i < N_SIG_CG; i++)
        for (int b = 0; b < 2; b++) e->sigCG[i][b] = (int)k_bits[ctx[CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG) + i] ^ b];

    int first = 1, num = 8;
    if (log2TrSize >= 4) { first = isLuma ? 21 : 12; num = isLuma ? 6 : 3; }
    else if (log2TrSize == 3) { first = 9; num = isLuma ? 12 : 3; }
    const uint8_t* sig = ctx + CTX_SIG + (isLuma ? 0 : N_SIG_LUMA);
    for (int b = 0; b < 2; b++) e->sig[b][0] = (int)k_bits[sig[0] ^ b];
    for (int i = first; i < first + num; i++)
        for (int b = 0; b < 2; b++) e->sig[b][i] = (int)k_bits[sig[i] ^ b];

    const int maxGroupIdx = log2TrSize * 2 - 1;
    for (int i = 0; i < 2; i++)
    {
        const uint8_t* st = ctx + CTX_LAST_X + i * N_LAST_XY;
        int bits = 0;
        if (isLuma && log2TrSize == 2)
        {
            for (int c = 0; c < 3; c++) { e->last[i][c] = bits + (int)k_bits[st[c] ^ 0]; bits += (int)k_bits[st[c] ^ 1]; }
            e->last[i][maxGroupIdx] = bits;
        }
        else if (isLuma)
        {
            const int off = (log2TrSize - 2) * 3 + (log2TrSize == 5);
            for (int c = 0; c < (maxGroupIdx >> 1) + 1; c++)
            {
                int c0 = (int)k_bits[st[off + c] ^ 0], c1 = (int)k_bits[st[off + c] ^ 1];
                e->last[i][2 * c] = bits + c0;
                e->last[i][2 * c + 1] = bits + c1 + c0;
                bits += 2 * c1;
            }
            e->last[i][maxGroupIdx] -= (int)k_bits[st[off + (maxGroupIdx >> 1)] ^ 0];
        }
        else
        {
            const int shift = log2TrSize - 2;
            for (int c = 0; c < maxGroupIdx; c++)
            {
                int o = N_LAST_XY_LUMA + (c >> shift);
                e->last[i][c] = bits + (int)k_bits[st[o] ^ 0];
                bits += (int)k_bits[st[o] ^ 1];
            }
            e->last[i][maxGroupIdx] = bits;
        }
    }
    const uint8_t* one = ctx + CTX_ONE + (isLuma ? 0 : N_ONE_LUMA);
    const uint8_t* ab = ctx + CTX_ABS + (isLuma ? 0 : N_ABS_LUMA);
    for (int i = 0; i < (isLuma ? 16 : 8); i++) for (int b = 0; b < 2; b++) e->greaterOne[i][b] = (int)k_bits[one[i] ^ b];
    for (int i = 0; i < (isLuma ? 4 : 2); i++) for (int b = 0; b < 2; b++) e->levelAbs[i][b] = (int)k_bits[ab[i] ^ b];
}

/* ---------------------------------------------------------------------------------------------------------
 * shared pieces
 * ------------------------------------------------------------------------------------------------------- */
/* order of the 16 samples of a 4x4 group (g_scan4x4[type]): raster index of scan offset i */
static const uint16_t* cg_order(int scanType) { return orc_tbl_scan(scanType, 2); }

/* order of the 4x4 groups of a TU (g_scanOrderCG): group raster index of group scan position i */
static void cg_scan(int scanType, int log2TrSize, uint16_t* out)
{
    const uint16_t* s = orc_tbl_scan(scanType, log2TrSize);
    int N = 1 << log2TrSize, ncg = 1 << (2 * (log2TrSize - 2)), cgStride = N >> 2;
    for (int i = 0; i < ncg; i++)
    {
        int blk = s[i * 16];
        out[i] = (uint16_t)(((blk >> log2TrSize) >> 2) * cgStride + ((blk & (N - 1)) >> 2));
    }
}

/* Quant::calcPatternSigCtx / getSigCoeffGroupCtxInc (quant.h:118-146) */
static uint32_t pattern_sig_ctx(uint64_t cgFlags, uint32_t cgX, uint32_t cgY, uint32_t cgBlk, uint32_t cgStride)
{
    if (cgStride == 1) return 0;
    uint32_t sigPos = (uint32_t)(cgBlk + 1 < 64 ? cgFlags >> (cgBlk + 1) : 0);
    uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
    return right + lower * 2;
}
static uint32_t sig_cg_ctx(uint64_t cgFlags, uint32_t cgX, uint32_t cgY, uint32_t cgBlk, uint32_t cgStride)
{
    uint32_t sigPos = (uint32_t)(cgBlk + 1 < 64 ? cgFlags >> (cgBlk + 1) : 0);
    uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
    return right | lower;
}

/* significance context increment of sample `r` (raster index inside its 4x4 group): by neighbour pattern for
 * groups of 8x8 and larger TUs, by position for 4x4 TUs (quant.cpp:739-780 / H.265 9.3.4.2.5) */
static uint32_t sig_ctx_inc(int log2TrSize, uint32_t pattern, uint32_t r)
{
    static const uint8_t pos4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
    if (log2TrSize == 2) return pos4x4[r];
    uint32_t x = r & 3, y = r >> 2;
    switch (pattern)
    {
    case 0: return x + y == 0 ? 2 : (x + y < 3 ? 1 : 0);
    case 1: return y == 0 ? 2 : (y == 1 ? 1 : 0);
    case 2: return x == 0 ? 2 : (x == 1 ? 1 : 0);
    default: return 2;
    }
}

static int first_sig_ctx(int log2TrSize, int isLuma, int scanType)     /* cudata.cpp:2094-2099 */
{
    if (log2TrSize == 2) return 0;
    if (log2TrSize == 3) return (scanType != SCAN_DIAG && isLuma) ? 15 : 9;
    return isLuma ? 21 : 12;
}

/* g_lastCoeffTable (constants.cpp:473-479) by rule: prefix group index and suffix length of a last-position coordinate */
static void last_pos_code(uint32_t pos, uint32_t* prefix, uint32_t* suffixLen)
{
    if (pos < 4) { *prefix = pos; *suffixLen = 0; return; }
    uint32_t l = 31 - (uint32_t)__builtin_clz(pos);             /* pos in [2^l, 2^(l+1)) */
    *suffixLen = l - 1;
    *prefix = 2 * l + ((pos >> (l - 1)) & 1);
}

/* scanPosLast_c (dct.cpp:757-790): per-group count / flag / sign masks up to the last non-zero level */
static int scan_pos_last(const uint16_t* scan, const int16_t* coeff, uint16_t* sign, uint16_t* flag, uint8_t* num, int numSig)
{
    memset(num, 0, 64); memset(flag, 0, 64 * 2); memset(sign, 0, 64 * 2);
    int p = 0;
    do
    {
        uint32_t cg = (uint32_t)p >> 4;
        int c = coeff[scan[p++]];
        uint32_t nz = c != 0;
        numSig -= (int)nz;
        sign[cg] = (uint16_t)(sign[cg] + (((uint32_t)c >> 31) << num[cg]));
        flag[cg] = (uint16_t)((flag[cg] << 1) + nz);
        num[cg] = (uint8_t)(num[cg] + nz);
    }
    while (numSig > 0);
    return p - 1;
}

/* ---------------------------------------------------------------------------------------------------------
 * RDOQ
 * ------------------------------------------------------------------------------------------------------- */
#define IEP_RATE 32768
static const uint8_t k_goRiceRange[5] = { 7, 14, 26, 46, 78 };
static const int k_invQuantScales[6] = { 40, 45, 51, 57, 64, 72 };
static const int k_quantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
static const uint32_t k_ctxCbf[3][5] = { { 1, 0, 0, 0, 0 }, { 2, 3, 4, 5, 6 }, { 2, 3, 4, 5, 6 } };

static int imin(int a, int b) { return a < b ? a : b; }

/* getICRateCost (quant.cpp:131-165) */
static uint32_t level_rate(uint32_t absLevel, int32_t diff, const int* g1, const int* ab, uint32_t rice, uint32_t c1c2Rate)
{
    if (diff < 0)
    {
        uint32_t rate = (uint32_t)g1[absLevel == 2];
        if (absLevel == 2) rate += (uint32_t)ab[0];
        return rate;
    }
    uint32_t symbol = (uint32_t)diff, rate;
    if ((symbol >> rice) < 3) rate = ((symbol >> rice) + 1 + rice) << 15;
    else
    {
        uint32_t length = 0;
        symbol = (symbol >> rice) - 3;
        if (symbol) length = 31 - (uint32_t)__builtin_clz(symbol + 1);
        rate = (3 + length + rice + 1 + length) << 15;
    }
    return rate + c1c2Rate;
}
/* getICRate (quant.cpp:54-104) */
static int level_rate_sbh(uint32_t absLevel, int32_t diff, const int* g1, const int* ab, uint32_t rice, uint32_t maxVlc, uint32_t c1c2Rate)
{
    if (!absLevel) return 0;
    int rate = 0;
    if (diff < 0)
    {
        rate += g1[absLevel == 2];
        if (absLevel == 2) rate += ab[0];
        return rate;
    }
    uint32_t symbol = (uint32_t)diff;
    if (symbol > maxVlc)
    {
        uint32_t a = symbol - maxVlc;
        int size = 31 - __builtin_clz(a);
        rate += (size * 2 + 1) << 15;
        symbol = maxVlc + 1;
    }
    uint32_t prefLen = (symbol >> rice) + 1;
    rate += imin((int)(prefLen + rice), 8) << 15;
    rate += (int)c1c2Rate;
    return rate;
}
/* getICRateLessVlc (quant.cpp:122-139) */
static int level_rate_less_vlc(uint32_t absLevel, int32_t diff, uint32_t rice)
{
    if (!absLevel) return 0;
    uint32_t prefLen = ((uint32_t)diff >> rice) + 1;
    return imin((int)(prefLen + rice), 8) << 15;
}

/* Quant::rdoQuant (quant.cpp:609-1424), flat scaling lists.  dct / fencDct: N*N transform coefficients of the residual /
 * of the source block (the latter only read when usePsy).  Returns numSig; levels (signed) in dst. */
uint32_t orc_rdo_quant(const int16_t* dct, const int16_t* fencDct, int16_t* dst, int log2TrSize, int ttype, int bIntra, int dirMode,
                       int qpScaled, int tuDepth, int signHide, int rdoqLevel, int psyRdoqScale, int usePsy, const OrcEstBits* est)
{
    const int N = 1 << log2TrSize, numCoeff = N * N, isLuma = ttype == 0;
    const int transformShift = 15 - ORC_DEPTH - log2TrSize;
    const int rem = qpScaled % 6, per = qpScaled / 6;
    const int qbits = 14 + per + transformShift, add = 1 << (qbits - 1);
    static int32_t qc[1024];
    for (int i = 0; i < numCoeff; i++) qc[i] = k_quantScales[rem];
    uint32_t numSig = orc_nquant(dct, qc, dst, qbits, add, numCoeff);
    if (!numSig) return 0;

    const int64_t lambda2 = (int64_t)(orc_lambda2(qpScaled - QP_BD_OFFSET) * 256. + 0.5);     /* QpParam::setQpParam, quant.h:50-60 */
    const int32_t lambda = (int32_t)(orc_lambda(qpScaled - QP_BD_OFFSET) * 256. + 0.5);
    const int64_t psyScale = (int64_t)psyRdoqScale * lambda;
    const int uqScale = k_invQuantScales[rem] << per;
    const int uqShift = 20 - 14 - transformShift;
    const int uqRound = uqShift > per ? 1 << (uqShift - per - 1) : 0;
    const int scaleBits = 15 - 2 * transformShift;
    const int psyShift = 2 * transformShift + 1 > 0 ? 2 * transformShift + 1 : 0;
#define SIGCOST(bits) ((lambda2 * (int64_t)(bits)) >> 8)
#define PSYVAL(rec) ((psyScale * (int64_t)(rec)) >> psyShift)

    static int64_t costCoeff[1024], costUncoded[1024], costSig[1024];
    static int rateUp[1024], rateDown[1024], sigDelta[1024];
    int64_t costCgSig[64];
    uint64_t cgFlags = 0;
    int64_t totalUncoded = 0, totalRd = 0;

    const int scanType = orc_scan_type(bIntra, isLuma, log2TrSize, dirMode);
    const uint16_t* scan = orc_tbl_scan(scanType, log2TrSize);
    const uint16_t* inCg = cg_order(scanType);
    uint16_t scanCG[64];
    cg_scan(scanType, log2TrSize, scanCG);
    const int firstSig = first_sig_ctx(log2TrSize, isLuma, scanType);
    const uint32_t log2CG = (uint32_t)log2TrSize - 2, cgNum = 1u << (2 * log2CG), cgStride = (uint32_t)N >> 2;

    uint8_t cgCount[64]; uint16_t cgSign[64], cgFlag[64];
    const int lastScanPos = scan_pos_last(scan, dst, cgSign, cgFlag, cgCount, (int)numSig);
    const int cgLast = lastScanPos >> 4;

    /* groups behind the last level: only the uncoded distortion counts (psyRdoQuant_1p + _2p / nonPsyRdoQuant,
     * dct.cpp:985-1066; the psy form adds each sample to both totals twice, once before and once after the psy term) */
    for (int cg = cgLast + 1; cg < (int)cgNum; cg++)
    {
        for (int i = 0; i < 16; i++) { costCoeff[cg * 16 + i] = 0; costSig[cg * 16 + i] = 0; }
        uint32_t base = scan[cg * 16];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
            {
                uint32_t b = base + (uint32_t)(y * N + x);
                int64_t c = dct[b];
                costUncoded[b] = (c * c) << scaleBits;
                totalUncoded += costUncoded[b]; totalRd += costUncoded[b];
                if (usePsy)
                {
                    int64_t predicted = (int64_t)fencDct[b] - c;
                    costUncoded[b] -= (psyScale * predicted) >> psyShift;
                    totalUncoded += costUncoded[b]; totalRd += costUncoded[b];
                }
            }
    }

    uint32_t c1 = 1;
    for (int cg = cgLast; cg >= 0; cg--)
    {
        uint32_t ctxSet = (cg && isLuma) ? 2 : 0;
        const uint32_t cgBlk = scanCG[cg], cgY = cgBlk >> log2CG, cgX = cgBlk & ((1u << log2CG) - 1);
        const uint64_t cgMask = (uint64_t)1 << cgBlk;
        const uint32_t pattern = pattern_sig_ctx(cgFlags, cgX, cgY, cgBlk, cgStride);
        const int sigOff = firstSig + ((cg && isLuma) ? 3 : 0);
        if (c1 == 0) ctxSet++;
        c1 = 1;

        if (cg && cgCount[cg] == 0)
        {
            /* an empty group in front of the last level (quant.cpp:786-848) */
            uint32_t base = scan[cg * 16];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                {
                    uint32_t b = base + (uint32_t)(y * N + x);
                    int64_t c = dct[b];
                    costUncoded[b] = (c * c) << scaleBits;
                    totalUncoded += costUncoded[b]; totalRd += costUncoded[b];
                    if (usePsy)
                    {
                        int64_t predicted = (int64_t)fencDct[b] - c;
                        costUncoded[b] -= (psyScale * predicted) >> psyShift;
                        totalUncoded += costUncoded[b]; totalRd += costUncoded[b];
                    }
                }
            /* the reference pairs scan offset y*4+x with the raster sample (y, x) of the group here */
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                {
                    int o = y * 4 + x;
                    uint32_t b = base + (uint32_t)(y * N + x);
                    uint32_t ctxSig = sig_ctx_inc(log2TrSize, pattern, inCg[o]) + (uint32_t)sigOff;
                    costSig[cg * 16 + o] = SIGCOST(est->sig[0][ctxSig]);
                    costCoeff[cg * 16 + o] = costUncoded[b];
                    sigDelta[b] = est->sig[1][ctxSig] - est->sig[0][ctxSig];
                }
            uint32_t ctx = sig_cg_ctx(cgFlags, cgX, cgY, cgBlk, cgStride);
            costCgSig[cg] = SIGCOST(est->sigCG[ctx][0]);
            totalRd += costCgSig[cg];
            continue;
        }

        int nnzBeforePos0 = 0;
        int64_t codedLevelAndDist = 0, uncodedDist = 0, sigCost = 0, sigCost0;
        uint32_t flagMask = cgFlag[cg];
        uint32_t c2 = 0, rice = 0, levelThreshold = 3, c1Idx = 0, c2Idx = 0;
        int scanPos = 0;
        for (int k = 15; k >= 0; k--)
        {
            scanPos = cg * 16 + k;
            const uint32_t b = scan[scanPos];
            const uint32_t maxAbs = (uint32_t)dst[b];
            const int signCoef = dct[b];
            const int predicted = (usePsy ? fencDct[b] : 0) - signCoef;
            const int psyHere = usePsy && scanPos;

            costUncoded[b] = ((int64_t)signCoef * signCoef) << scaleBits;
            if (psyHere) costUncoded[b] -= PSYVAL(predicted);
            totalUncoded += costUncoded[b];

            const int* g1 = est->greaterOne[4 * ctxSet + c1];
            const uint32_t ctxSig = b == 0 ? 0 : sig_ctx_inc(log2TrSize, pattern, inCg[k]) + (uint32_t)sigOff;

            if (scanPos > lastScanPos)
            {
                costCoeff[scanPos] = 0; costSig[scanPos] = 0;
                totalRd += costUncoded[b];
            }
            else if (!(flagMask & 1))
            {
                costSig[scanPos] = SIGCOST(est->sig[0][ctxSig]);
                costCoeff[scanPos] = costUncoded[b] + costSig[scanPos];
                sigDelta[b] = est->sig[1][ctxSig] - est->sig[0][ctxSig];
                totalRd += costCoeff[scanPos];
                rateUp[b] = g1[0];
                flagMask >>= 1;
            }
            else
            {
                flagMask >>= 1;
                /* 0: c1 flags left, c2 flag used; 1: c1 left only... as {baseLevel}: (c1Idx<8 ? (c2Idx==0 ? 3 : 2) : 1) */
                const uint32_t hasC1 = c1Idx < 8, noC2 = c2Idx == 0;
                const uint32_t baseLevel = hasC1 ? 2 + noC2 : 1;
                const int* ab = est->levelAbs[ctxSet + c2];
                const uint32_t c1c2Rate = (hasC1 ? (uint32_t)g1[1] : 0) + ((hasC1 && noC2) ? (uint32_t)ab[1] : 0);

                uint32_t level = 0, sigBits = 0;
                costCoeff[scanPos] = INT64_MAX;
                if (scanPos == lastScanPos) sigDelta[b] = 0;
                else
                {
                    if (maxAbs < 3)
                    {
                        costSig[scanPos] = SIGCOST(est->sig[0][ctxSig]);
                        costCoeff[scanPos] = costUncoded[b] + costSig[scanPos];
                    }
                    sigDelta[b] = est->sig[1][ctxSig] - est->sig[0][ctxSig];
                    sigBits = (uint32_t)est->sig[1][ctxSig];
                }
                const uint32_t uq = maxAbs * (uint32_t)uqScale + (uint32_t)uqRound;
                const int absCoef = abs(signCoef);
                const int sgnPred = signCoef < 0 ? -predicted : predicted;      /* SIGN(predicted, signCoef) */
                for (int t = 0; t < 2; t++)     /* candidate levels maxAbs, maxAbs - 1 (the latter only when maxAbs > 1) */
                {
                    if (t == 1 && maxAbs < 2) break;
                    if (maxAbs == 0) break;
                    const uint32_t lv = maxAbs - (uint32_t)t;
                    uint32_t bits;
                    if (maxAbs == 1) bits = (hasC1 ? (uint32_t)g1[0] : ((1 + rice) << 15)) + IEP_RATE;
                    else bits = level_rate(lv, (int32_t)lv - (int32_t)baseLevel, g1, ab, rice, c1c2Rate) + IEP_RATE;
                    const int uqAbs = (int)((uq - (uint32_t)t * (uint32_t)uqScale) >> uqShift);
                    const int d = absCoef - uqAbs;
                    int64_t cost = (((int64_t)d * d) << scaleBits) + SIGCOST(sigBits + bits);
                    if (psyHere) cost -= PSYVAL(abs(uqAbs + sgnPred));
                    if (cost < costCoeff[scanPos])
                    {
                        level = lv;
                        costCoeff[scanPos] = cost;
                        costSig[scanPos] = SIGCOST(sigBits);
                    }
                }
                dst[b] = (int16_t)level;
                totalRd += costCoeff[scanPos];

                if (signHide && level)
                {
                    const int32_t diff0 = (int32_t)level - 1 - (int32_t)baseLevel, diff2 = (int32_t)level + 1 - (int32_t)baseLevel;
                    const int32_t maxVlc = k_goRiceRange[rice];
                    int r0, r1, r2;
                    if (diff0 < -2) { r0 = 0; r2 = g1[1] + ab[0]; r1 = g1[0]; }
                    else if (diff0 >= 0 && diff2 <= maxVlc)
                    {
                        r1 = level_rate_less_vlc(level, diff0 + 1, rice);
                        r2 = level_rate_less_vlc(level + 1, diff0 + 2, rice);
                        r0 = level_rate_less_vlc(level - 1, diff0, rice);
                    }
                    else
                    {
                        r1 = level_rate_sbh(level, diff0 + 1, g1, ab, rice, (uint32_t)maxVlc, c1c2Rate);
                        r2 = level_rate_sbh(level + 1, diff0 + 2, g1, ab, rice, (uint32_t)maxVlc, c1c2Rate);
                        r0 = level_rate_sbh(level - 1, diff0, g1, ab, rice, (uint32_t)maxVlc, c1c2Rate);
                    }
                    rateUp[b] = r2 - r1;
                    rateDown[b] = r0 - r1;
                }
                else { rateUp[b] = g1[0]; rateDown[b] = 0; }

                if (level >= baseLevel && rice < 4 && level > levelThreshold) { rice++; levelThreshold <<= 1; }
                const uint32_t nz = level != 0;
                c1Idx += nz;
                if (level > 1) { c1 = 0; c2 += c2 < 2; c2Idx++; }
                else if ((c1 == 1 || c1 == 2) && nz) c1++;

                if (dst[b])
                {
                    cgFlags |= cgMask;
                    codedLevelAndDist += costCoeff[scanPos] - costSig[scanPos];
                    uncodedDist += costUncoded[b];
                    nnzBeforePos0 += k;
                }
            }
            sigCost += costSig[scanPos];
        }
        sigCost0 = costSig[scanPos];        /* scanPos == cg * 16 here */
        costCgSig[cg] = 0;

        if (!cg || cg == cgLast) { /* presence of these groups is implied */ }
        else if (cgFlags & cgMask)
        {
            if (!nnzBeforePos0) { totalRd -= sigCost0; sigCost -= sigCost0; }
            uint32_t ctx = sig_cg_ctx(cgFlags, cgX, cgY, cgBlk, cgStride);
            int64_t costZeroCG = totalRd + SIGCOST(est->sigCG[ctx][0]);
            costZeroCG += uncodedDist;
            costZeroCG -= codedLevelAndDist;
            costZeroCG -= sigCost;
            costCgSig[cg] = SIGCOST(est->sigCG[ctx][1]);
            totalRd += costCgSig[cg];
            if (costZeroCG < totalRd && rdoqLevel > 1)
            {
                cgFlags &= ~cgMask;
                totalRd = costZeroCG;
                costCgSig[cg] = SIGCOST(est->sigCG[ctx][0]);
                uint32_t base = scan[cg * 16];
                for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[base + (uint32_t)(y * N + x)] = 0;
            }
        }
        else
        {
            uint32_t ctx = sig_cg_ctx(cgFlags, cgX, cgY, cgBlk, cgStride);
            costCgSig[cg] = SIGCOST(est->sigCG[ctx][0]);
            totalRd += costCgSig[cg];
            totalRd -= sigCost;
        }
    }

    /* cost of CBF = 0 against the coded block (quant.cpp:1158-1171) */
    int64_t bestCost;
    if (!bIntra && isLuma && !tuDepth)
    {
        bestCost = totalUncoded + SIGCOST(est->rootCbf[0]);
        totalRd += SIGCOST(est->rootCbf[1]);
    }
    else
    {
        uint32_t ctx = k_ctxCbf[ttype][tuDepth];
        bestCost = totalUncoded + SIGCOST(est->cbf[ctx][0]);
        totalRd += SIGCOST(est->cbf[ctx][1]);
    }

    /* choice of the last significant position (quant.cpp:1173-1254) */
    int bestLastIdx = 0, foundLast = 0;
    for (int cg = cgLast; cg >= 0 && !foundLast; cg--)
    {
        if (!cg || cg == cgLast) { }
        else if (cgFlags & ((uint64_t)1 << scanCG[cg])) totalRd -= costCgSig[cg];
        else { totalRd -= costCgSig[cg]; continue; }

        for (int k = 15; k >= 0; k--)
        {
            int scanPos = cg * 16 + k;
            if (scanPos > lastScanPos) continue;
            uint32_t b = scan[scanPos];
            if (dst[b])
            {
                uint32_t pos[2] = { b & (uint32_t)(N - 1), b >> log2TrSize };
                if (scanType == SCAN_VER) { uint32_t t = pos[0]; pos[0] = pos[1]; pos[1] = t; }
                uint32_t bitsLast = 0;
                for (int i = 0; i < 2; i++)
                {
                    uint32_t prefix, suffixLen;
                    last_pos_code(pos[i], &prefix, &suffixLen);
                    bitsLast += (uint32_t)est->last[i][prefix];
                    bitsLast += IEP_RATE * suffixLen;
                }
                int64_t asLast = totalRd - costSig[scanPos] + SIGCOST(bitsLast);
                if (asLast < bestCost) { bestLastIdx = scanPos + 1; bestCost = asLast; }
                if (dst[b] > 1 || rdoqLevel == 1) { foundLast = 1; break; }
                totalRd -= costCoeff[scanPos];
                totalRd += costUncoded[b];
            }
            else totalRd -= costSig[scanPos];
        }
    }

    /* signs back, drop everything behind the chosen last position (quant.cpp:1256-1283) */
    numSig = 0;
    for (int p = 0; p < bestLastIdx; p++)
    {
        uint32_t b = scan[p];
        int level = dst[b];
        numSig += level != 0;
        dst[b] = (int16_t)(dct[b] < 0 ? -level : level);
    }
    {
        int m = imin(lastScanPos, bestLastIdx) | 15;
        for (int p = bestLastIdx; p <= m; p++) dst[scan[p]] = 0;
        for (int p = (bestLastIdx & ~15) + 16; p <= lastScanPos; p += 16)
        {
            uint32_t base = scan[p];
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[base + (uint32_t)(y * N + x)] = 0;
        }
    }

    /* rate-distortion based sign hiding (quant.cpp:1285-1421) */
    if (signHide && numSig >= 2)
    {
        const int realLast = (bestLastIdx - 1) >> 4;
        int lastCG = 1;
        for (int sub = realLast; sub >= 0; sub--)
        {
            const int subPos = sub << 4;
            if (!(cgFlags & ((uint64_t)1 << scanCG[sub]))) continue;
            int lastNZ = -1, firstNZ = 16;
            for (int n = 15; n >= 0; n--) if (dst[scan[subPos + n]]) { lastNZ = n; break; }
            for (int n = 0; n < 16; n++) if (dst[scan[subPos + n]]) { firstNZ = n; break; }
            uint32_t absSum = 0;
            for (int n = firstNZ; n <= lastNZ; n++) absSum += (uint32_t)(int32_t)dst[scan[subPos + n]];
            if (lastNZ - firstNZ >= 4)
            {
                const int32_t signbit = dst[scan[subPos + firstNZ]];
                if (((uint32_t)signbit >> 31) != (absSum & 1))
                {
                    int64_t minCostInc = INT64_MAX, curCost = INT64_MAX;
                    uint32_t minPos = 0; int minAbs = 0;
                    int finalChange = 0, curChange = 0;
                    uint32_t lastAdjust = (uint32_t)(lastCG & (abs(dst[scan[lastNZ + subPos]]) == 1)) * 4 * IEP_RATE;
                    for (int n = lastCG ? lastNZ : 15; n >= 0; --n)
                    {
                        const uint32_t b = scan[n + subPos];
                        const int signCoef = dct[b];
                        const int absLevel = abs(dst[b]);
                        const uint32_t step = (uint32_t)uqScale;
                        const uint32_t uq = (uint32_t)absLevel * step + (uint32_t)uqRound;
                        int d = abs(signCoef) - (int)(uq >> uqShift);
                        const int64_t origDist = (int64_t)d * d;
#define DELTA_RD(dd, bits) (((((int64_t)(dd) * (dd)) - origDist) << scaleBits) + ((lambda2 * (int64_t)(bits)) >> 8))
                        const uint32_t isOne = absLevel == 1;
                        if (dst[b])
                        {
                            d = abs(signCoef) - (int)((uq + step) >> uqShift);
                            int64_t costUp = DELTA_RD(d, rateUp[b]);
                            d = abs(signCoef) - (int)((uq - step) >> uqShift);
                            int downBits = rateDown[b] - (isOne ? (IEP_RATE + sigDelta[b]) : 0);
                            int64_t costDown = DELTA_RD(d, downBits);
                            costDown -= lastAdjust;
                            curCost = ((n == firstNZ) & isOne) ? INT64_MAX : costDown;
                            curChange = 2 * (costUp < costDown) - 1;
                            curCost = (costUp < costDown) ? costUp : curCost;
                        }
                        else if ((n < firstNZ) & ((signbit ^ signCoef) < 0)) curCost = INT64_MAX;
                        else
                        {
                            d = abs(signCoef) - (int)((step + (uint32_t)uqRound) >> uqShift);
                            curCost = DELTA_RD(d, rateUp[b] + IEP_RATE + sigDelta[b]);
                            curChange = 1;
                        }
                        if (curCost < minCostInc)
                        {
                            minCostInc = curCost;
                            finalChange = curChange;
                            minPos = b; minAbs = absLevel;
                        }
                        lastAdjust = 0;
                    }
                    if (minAbs >= 32767) finalChange = -1;
                    numSig += (uint32_t)(minAbs == 0) - (uint32_t)((finalChange == -1) & (minAbs == 1));
                    dst[minPos] = (int16_t)(dst[minPos] + (dct[minPos] < 0 ? -finalChange : finalChange));
                }
            }
            lastCG = 0;
        }
    }
    return numSig;
#undef SIGCOST
#undef PSYVAL
}

/* ---------------------------------------------------------------------------------------------------------
 * bits-only coefficient coding (Entropy::codeCoeffNxN with m_bitIf == NULL, entropy.cpp:1828-2200; no transform skip,
 * no bypass).  ctx: the 157 context states, updated in place.  Returns the FIX15 bits added to m_fracBits.
 * ------------------------------------------------------------------------------------------------------- */
static uint32_t code_bin(uint8_t* st, uint32_t bin) { uint32_t b = k_bits[*st ^ bin]; *st = orc_ctx_next(*st, (int)bin); return b; }

uint64_t orc_code_coeff_bits(const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int dirMode, int signHide, uint8_t* ctx)
{
    const int N = 1 << log2TrSize, isLuma = ttype == 0;
    uint64_t bits = 0;
    int numSig = 0;
    for (int i = 0; i < N * N; i++) numSig += coeff[i] != 0;
    if (!numSig) return 0;
    const int scanType = orc_scan_type(bIntra, isLuma, log2TrSize, dirMode);
    const uint16_t* scan = orc_tbl_scan(scanType, log2TrSize);
    const uint16_t* inCg = cg_order(scanType);
    uint16_t scanCG[64];
    cg_scan(scanType, log2TrSize, scanCG);
    uint8_t cgCount[64]; uint16_t cgSign[64], cgFlag[64];
    const int scanPosLast = scan_pos_last(scan, coeff, cgSign, cgFlag, cgCount, numSig);
    const uint32_t posLast = scan[scanPosLast];
    const int lastSet = scanPosLast >> 4;
    const uint32_t log2CG = (uint32_t)log2TrSize - 2, cgStride = (uint32_t)N >> 2;
    uint64_t cgFlags = 0;
    for (int i = 0; i < lastSet; i++) if (cgCount[i]) cgFlags |= (uint64_t)1 << scanCG[i];

    /* last position: context-coded prefixes, bypass suffixes */
    {
        uint32_t pos[2] = { posLast & (uint32_t)(N - 1), posLast >> log2TrSize };
        if (scanType == SCAN_VER) { uint32_t t = pos[0]; pos[0] = pos[1]; pos[1] = t; }
        int ctxIdx = isLuma ? 3 * (log2TrSize - 2) + (log2TrSize == 5) : N_LAST_XY_LUMA;
        const int ctxShift = isLuma ? (log2TrSize > 2) : log2TrSize - 2;
        const uint32_t maxGroupIdx = ((uint32_t)log2TrSize << 1) - 1;
        uint32_t suffixTotal = 0;
        for (int i = 0; i < 2; i++, ctxIdx += N_LAST_XY)
        {
            uint32_t prefix, suffixLen;
            last_pos_code(pos[i], &prefix, &suffixLen);
            uint8_t* c = ctx + CTX_LAST_X + ctxIdx;
            for (uint32_t k = 0; k < prefix; k++) bits += code_bin(c + (k >> ctxShift), 1);
            if (prefix < maxGroupIdx) bits += code_bin(c + (prefix >> ctxShift), 0);
            suffixTotal += suffixLen;
        }
        bits += (uint64_t)IEP_RATE * suffixTotal;
    }

    uint8_t* cgCtx = ctx + CTX_SIG_CG + (isLuma ? 0 : N_SIG_CG);
    uint8_t* sigCtx = ctx + CTX_SIG + (isLuma ? 0 : N_SIG_LUMA);
    const int firstSig = first_sig_ctx(log2TrSize, isLuma, scanType);
    uint32_t c1 = 1;
    int sigOff = scanPosLast - (lastSet << 4) - 1;
    uint16_t absCoeff[17];
    uint32_t numNonZero = 1;
    absCoeff[0] = (uint16_t)abs(coeff[posLast]);

    for (int sub = lastSet; sub >= 0; sub--)
    {
        uint32_t flagMask = cgFlag[sub];
        const int subBase = sub << 4;
        if (sub == lastSet) flagMask >>= 1;
        const uint32_t cgBlk = scanCG[sub], cgY = cgBlk >> log2CG, cgX = cgBlk & ((1u << log2CG) - 1);
        const uint64_t cgMask = (uint64_t)1 << cgBlk;
        if (sub == lastSet || !sub) cgFlags |= cgMask;
        else
        {
            uint32_t sig = (cgFlags & cgMask) != 0;
            bits += code_bin(cgCtx + sig_cg_ctx(cgFlags, cgX, cgY, cgBlk, cgStride), sig);
        }
        if (sigOff >= 0 && (cgFlags & cgMask))
        {
            /* costCoeffNxN_c (dct.cpp:838-890): significance flags of the group in reverse scan order */
            const uint32_t pattern = pattern_sig_ctx(cgFlags, cgX, cgY, cgBlk, cgStride);
            const int offset = firstSig + ((isLuma && sub) ? 3 : 0);
            const uint32_t base = scan[subBase];
            uint32_t nnz = sigOff < 15 ? 1 : 0;        /* the last level of the TU was counted already */
            uint16_t* ac = absCoeff + numNonZero - nnz;
            uint32_t sum = 0;
            for (int k = sigOff; k >= 0; k--)
            {
                uint32_t r = inCg[k];
                uint32_t sig = flagMask & 1;
                flagMask >>= 1;
                if (k != 0 || subBase == 0 || nnz)
                {
                    uint32_t ctxSig = (subBase + k) ? sig_ctx_inc(log2TrSize, pattern, r) + (uint32_t)offset : 0;
                    sum += code_bin(sigCtx + ctxSig, sig);
                }
                int v = coeff[base + (r >> 2) * (uint32_t)N + (r & 3)];
                ac[nnz] = (uint16_t)abs(v);
                nnz += sig;
            }
            bits += sum & 0xFFFFFF;
        }
        numNonZero = cgCount[sub];
        if (numNonZero > 0)
        {
            const uint32_t subFlag = cgFlag[sub];
            int lastNZ = 31 - __builtin_clz(subFlag), firstNZ = __builtin_ctz(subFlag);
            const int signHidden = lastNZ - firstNZ >= 4;
            const uint32_t ctxSet = (((sub > 0) + (uint32_t)isLuma) & 2) + !(c1 & 3);
            uint8_t* oneCtx = ctx + CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + 4 * ctxSet;
            const uint32_t numC1 = numNonZero < 8 ? numNonZero : 8;
            /* costC1C2Flag_c (dct.cpp:942-993) */
            uint32_t sum = 0, firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFE;
            c1 = 1;
            for (uint32_t idx = 0; idx < numC1; idx++)
            {
                uint32_t s1 = absCoeff[idx] > 1, s2 = absCoeff[idx] > 2;
                sum += code_bin(oneCtx + c1, s1);
                if (s1) c1Next = 0;
                if (s1 + firstC2Flag == 3) firstC2Flag = s2;
                if (s1 + firstC2Idx == 9) firstC2Idx = idx;
                c1 = c1Next & 3;
                c1Next >>= 2;
            }
            if (!c1) sum += code_bin(ctx + CTX_ABS + (isLuma ? 0 : N_ABS_LUMA) + ctxSet, firstC2Flag);
            bits += sum & 0x00FFFFFF;
            bits += (uint64_t)(numNonZero - ((signHide && signHidden) ? 1 : 0)) << 15;
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
                        uint32_t length = 0;
                        code = (int)((uint32_t)code >> rice) - 3;
                        if (code >= 0)
                        {
                            length = 31 - (uint32_t)__builtin_clz((uint32_t)code + 1);
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

/* Quant::transformNxN with m_rdoqLevel != 0 (quant.cpp:397-453): transform of the residual (DST for 4x4 intra luma), transform
 * of the source block when psy-rdoq applies (luma, m_psyRdoqScale != 0), then rdoQuant.  est: EstBitsSbac filled by estBit. */
void orc_dct(int cu, const int16_t* src, int16_t* dst, intptr_t stride);
void orc_dst4x4(const int16_t* src, int16_t* dst, intptr_t stride);
#if ORC_DEPTH > 8
typedef uint16_t pixel;
#else
typedef uint8_t pixel;
#endif
uint32_t orc_transform_tu_rdoq(const pixel* fenc, intptr_t fencStride, const int16_t* resi, intptr_t resiStride, int16_t* coeff, int log2TrSize,
                               int ttype, int bIntra, int dirMode, int sliceType, int qpScaled, int signHide, int tuDepth, int rdoqLevel,
                               int psyRdoqScale, const int* est)
{
    (void)sliceType;
    int16_t dct[32 * 32], fdct[32 * 32], fsrc[32 * 32];
    const int sizeIdx = log2TrSize - 2, N = 1 << log2TrSize, isLuma = ttype == 0;
    const int usePsy = psyRdoqScale && isLuma;
    if (!sizeIdx && isLuma && bIntra) orc_dst4x4(resi, dct, resiStride);
    else orc_dct(sizeIdx, resi, dct, resiStride);
    if (usePsy)
    {
        for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) fsrc[y * N + x] = (int16_t)fenc[y * fencStride + x];
        orc_dct(sizeIdx, fsrc, fdct, N);
    }
    return orc_rdo_quant(dct, fdct, coeff, log2TrSize, ttype, bIntra, dirMode, qpScaled, tuDepth, signHide, rdoqLevel, psyRdoqScale, usePsy,
                         (const OrcEstBits*)est);
}

/* The per-TU measurement of the residual quad-tree (search.cpp:3276-3330) with RDOQ as the quantiser: as orc_tu_chain
 * (hevc_oracle_tu.c) but transformNxN runs rdoQuant. */
void orc_sub_ps(int cu, int16_t* dst, intptr_t ds, const pixel* s0, const pixel* s1, intptr_t ss0, intptr_t ss1);
void orc_add_ps(int cu, pixel* dst, intptr_t ds, const pixel* b0, const int16_t* b1, intptr_t ss0, intptr_t ss1);
uint64_t orc_sse_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
int orc_psy_cost_pp(int cu, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
void orc_invtransform_tu(int16_t* resi, intptr_t resiStride, const int16_t* coeff, int log2TrSize, int ttype, int bIntra, int qpScaled, uint32_t numSig);
void orc_tu_chain_rdoq(const pixel* fenc, intptr_t fencStride, const pixel* pred, intptr_t predStride, int log2TrSize, int ttype, int bIntra, int dirMode,
                       int sliceType, int qpScaled, int signHide, int tuDepth, int rdoqLevel, int psyRdoqScale, const int* est,
                       int16_t* coeff, int16_t* resiOut, intptr_t resiStride, pixel* recon, intptr_t reconStride, uint64_t* out)
{
    int sizeIdx = log2TrSize - 2, N = 1 << log2TrSize;
    int16_t resi[32 * 32];
    orc_sub_ps(sizeIdx, resi, N, fenc, pred, fencStride, predStride);
    uint32_t numSig = orc_transform_tu_rdoq(fenc, fencStride, resi, N, coeff, log2TrSize, ttype, bIntra, dirMode, sliceType, qpScaled, signHide,
                                            tuDepth, rdoqLevel, psyRdoqScale, est);
    out[0] = numSig;
    out[1] = orc_sse_pp(sizeIdx, fenc, fencStride, pred, predStride);
    out[2] = (uint64_t)(int64_t)orc_psy_cost_pp(sizeIdx, fenc, fencStride, pred, predStride);
    if (numSig)
    {
        orc_invtransform_tu(resiOut, resiStride, coeff, log2TrSize, ttype, bIntra, qpScaled, numSig);
        orc_add_ps(sizeIdx, recon, reconStride, pred, resiOut, predStride, resiStride);
        out[3] = orc_sse_pp(sizeIdx, fenc, fencStride, recon, reconStride);
        out[4] = (uint64_t)(int64_t)orc_psy_cost_pp(sizeIdx, fenc, fencStride, recon, reconStride);
    }
    else
    {
        for (int y = 0; y < N; y++)
            for (int x = 0; x < N; x++) { resiOut[y * resiStride + x] = 0; recon[y * reconStride + x] = pred[y * predStride + x]; }
        out[3] = out[1]; out[4] = out[2];
    }
}

/* batch form for bench.py's cpu_baseline leg (record of include/x265amd.h, x265amd_coeff_bits_job; HOST addresses) */
typedef struct { uint64_t coeff, ctxIn, ctxOut; uint8_t log2, ttype, intra, dir, signhide, reserved[3]; } PackedCoeffBitsJob;
int orc_coeff_bits_batch(const PackedCoeffBitsJob* jobs, int n, uint64_t* bits)
{
    for (int i = 0; i < n; i++)
    {
        uint8_t ctx[160];
        memcpy(ctx, (const void*)jobs[i].ctxIn, 160);
        bits[i] = orc_code_coeff_bits((const int16_t*)jobs[i].coeff, jobs[i].log2, jobs[i].ttype, jobs[i].intra, jobs[i].dir, jobs[i].signhide, ctx);
        memcpy((void*)jobs[i].ctxOut, ctx, 160);
    }
    return n;
}
