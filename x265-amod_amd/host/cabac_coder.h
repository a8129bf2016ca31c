/* The CABAC coder object shared by the final entropy pass (ctu_entropy.cpp) and the bit-counting walks of the analysis
 * (csrc/inter_rd.hip).  See ctu_entropy.cpp for the reference lines each method restates. */
#ifndef X265AMD_CABAC_CODER_H
#define X265AMD_CABAC_CODER_H
#include "x265amd.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace x265amd_host {

enum {
    C_SPLIT = 0, C_SKIP = 3, C_MERGE_FLAG = 6, C_MERGE_IDX = 7, C_PART_SIZE = 8, C_PRED_MODE = 12, C_ADI = 13, C_CHROMA_PRED = 14, C_DELTA_QP = 16,
    C_INTER_DIR = 19, C_REF_NO = 24, C_MV_RES = 26, C_QT_CBF = 28, C_TRANS_SUBDIV = 35, C_QT_ROOT_CBF = 38, C_SIG_CG = 39, C_SIG = 43, C_LAST_X = 85,
    C_ONE = 121, C_ABS = 145, C_MVP_IDX = 151, C_SAO_MERGE = 152, C_SAO_TYPE = 153, C_TQ_BYPASS = 156
};
enum { PART_2Nx2N, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };

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
static const uint8_t k_lpsNext[64] = {
    0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
    24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63
};
static const uint8_t k_rangeLps[64][4] = {     /* H.265 table 9-48 */
    128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185, 105,128,152,175, 100,122,144,166,
    95,116,137,158, 90,110,130,150, 85,104,123,142, 81,99,117,135, 77,94,111,128, 73,89,105,122, 69,85,100,116, 66,80,95,110,
    62,76,90,104, 59,72,86,99, 56,69,81,94, 53,65,77,89, 51,62,73,85, 48,59,69,80, 46,56,66,76, 43,53,63,72,
    41,50,59,69, 39,48,56,65, 37,45,54,62, 35,43,51,59, 33,41,48,56, 32,39,46,53, 30,37,43,50, 29,35,41,48,
    27,33,39,45, 26,31,37,43, 24,30,35,41, 23,28,33,39, 22,27,32,37, 21,26,30,35, 20,24,29,33, 19,23,27,31,
    18,22,26,30, 17,21,25,28, 16,20,23,27, 15,19,22,25, 14,18,21,24, 14,17,20,23, 13,16,19,22, 12,15,18,21,
    12,14,17,20, 11,14,16,19, 11,13,15,18, 10,12,15,17, 10,12,14,16, 9,11,13,15, 9,11,12,14, 8,10,12,14,
    8,9,11,13, 7,9,11,12, 7,9,10,12, 7,8,10,11, 6,8,9,11, 6,7,9,10, 6,7,8,9, 2,2,2,2
};

inline uint8_t ctxNext(uint8_t s, uint32_t bin)
{
    const uint32_t p = s >> 1, mps = s & 1;
    if (p == 63) return s;
    if (bin == mps) return (uint8_t)(((p < 62 ? p + 1 : 62) << 1) | mps);
    if (p == 0) return (uint8_t)(1 - mps);
    return (uint8_t)((k_lpsNext[p] << 1) | mps);
}

/* the 16 sample offsets of a 4x4 group in scan order, packed 4 bits each; group order of the 2x2 / 4x4 / 8x8 group grids */
inline uint32_t inCg(int type, int k)
{
    const uint64_t t = type == 1 ? 0xFEDCBA9876543210ULL : type == 2 ? 0xFB73EA62D951C840ULL : 0xFBE7AD369C258140ULL;
    return (uint32_t)((t >> (4 * k)) & 15);
}
struct Diag { uint8_t d4[16], d8[64]; Diag() { for (int n = 4; n <= 8; n += 4) { int i = 0; for (int d = 0; d < 2 * n - 1; d++) for (int y = d < n ? d : n - 1; y >= 0 && d - y < n; y--, i++) (n == 4 ? d4 : d8)[i] = (uint8_t)(y * n + (d - y)); } } };
static const Diag k_diag;
inline uint32_t cgBlk(int type, int log2N, int g)
{
    if (log2N == 2) return 0;
    if (log2N == 3) return ((type == 1 ? 0x3210u : 0x3120u) >> (4 * g)) & 15;
    return log2N == 4 ? k_diag.d4[g] : k_diag.d8[g];
}
inline uint32_t sigCtxInc(int log2N, uint32_t pattern, uint32_t rr)
{
    if (log2N == 2) return (uint32_t)((0x8877886654325410ULL >> (4 * rr)) & 15);
    const uint64_t t = pattern == 0 ? 0x0000000100110112ULL : pattern == 1 ? 0x0000000011112222ULL : pattern == 2 ? 0x0012001200120012ULL : 0x2222222222222222ULL;
    return (uint32_t)((t >> (4 * rr)) & 15);
}

} // namespace x265amd_host
using namespace x265amd_host;

struct x265amd_cabac
{
    x265amd_slice_info si;
    x265amd_cu_unit* units;
    int w4, h4, ctuW;
    bool bitsOnly;
    uint8_t ctx[X265AMD_CTX_STRIDE];
    uint64_t fracBits;
    /* arithmetic coder (entropy.cpp:2399-2612) */
    uint32_t low, range; int bitsLeft; uint32_t numBuffered; uint8_t bufferedByte;
    std::vector<uint8_t> out; uint32_t partial, partialBits;
    uint64_t ctuBits;       /* m_fracBits as it stood when the last CTU ended, before finishCU's resetBits() */

    /* ---- bit sink ---- */
    void pushByte(uint32_t v) { out.push_back((uint8_t)v); }
    void writeBits(uint32_t val, uint32_t n)        /* Bitstream::write (bitstream.cpp:45-81) */
    {
        for (int i = (int)n - 1; i >= 0; i--)
        {
            partial = (partial << 1) | ((val >> i) & 1);
            if (++partialBits == 8) { pushByte(partial); partial = 0; partialBits = 0; }
        }
    }
    void start() { low = 0; range = 510; bitsLeft = -12; numBuffered = 0; bufferedByte = 0xff; }
    void writeOut()
    {
        const uint32_t leadByte = low >> (13 + bitsLeft);
        const uint32_t lowMask = (uint32_t)(~0u) >> (11 + 8 - bitsLeft);
        bitsLeft -= 8;
        low &= lowMask;
        if (leadByte == 0xff) numBuffered++;
        else
        {
            uint32_t nb = numBuffered;
            if (nb > 0)
            {
                const uint32_t carry = leadByte >> 8;
                pushByte(bufferedByte + carry);
                const uint32_t fill = (0xff + carry) & 0xff;
                while (nb > 1) { pushByte(fill); nb--; }
            }
            numBuffered = 1;
            bufferedByte = (uint8_t)leadByte;
        }
    }
    void bin(uint32_t v, int c)
    {
        const uint32_t mstate = ctx[c];
        ctx[c] = ctxNext((uint8_t)mstate, v);
        if (bitsOnly) { fracBits += k_bits[mstate ^ v]; return; }
        const uint32_t state = mstate >> 1;
        const uint32_t lps = k_rangeLps[state][((uint8_t)range >> 6)];
        uint32_t r = range - lps;
        int numBits = (int)((uint32_t)(r - 256) >> 31);
        uint32_t l = low;
        if ((v ^ mstate) & 1)
        {
            const int idx = 31 - __builtin_clz(lps);
            numBits = 8 - idx;
            if (state >= 63) numBits = 6;
            l += r;
            r = lps;
        }
        low = l << numBits; range = r << numBits; bitsLeft += numBits;
        if (bitsLeft >= 0) writeOut();
    }
    void binEP(uint32_t v)
    {
        if (bitsOnly) { fracBits += 32768; return; }
        low <<= 1;
        if (v) low += range;
        if (++bitsLeft >= 0) writeOut();
    }
    void binsEP(uint32_t v, int n)
    {
        if (bitsOnly) { fracBits += (uint64_t)32768 * n; return; }
        while (n > 8)
        {
            n -= 8;
            const uint32_t pattern = v >> n;
            low <<= 8; low += range * pattern; v -= pattern << n; bitsLeft += 8;
            if (bitsLeft >= 0) writeOut();
        }
        low <<= n; low += range * v; bitsLeft += n;
        if (bitsLeft >= 0) writeOut();
    }
    void binTrm(uint32_t v)
    {
        if (bitsOnly) { fracBits += k_bits[126 ^ v]; return; }
        range -= 2;
        if (v) { low += range; low <<= 7; range = 2 << 7; bitsLeft += 7; }
        else if (range >= 256) return;
        else { low <<= 1; range <<= 1; bitsLeft++; }
        if (bitsLeft >= 0) writeOut();
    }
    void finish()
    {
        if (low >> (21 + bitsLeft))
        {
            pushByte(bufferedByte + 1);
            while (numBuffered > 1) { pushByte(0x00); numBuffered--; }
            low -= 1u << (21 + bitsLeft);
        }
        else
        {
            if (numBuffered > 0) pushByte(bufferedByte);
            while (numBuffered > 1) { pushByte(0xff); numBuffered--; }
        }
        writeBits(low >> 8, 13 + bitsLeft);
    }
    void epExGolomb(uint32_t symbol, uint32_t count)        /* writeEpExGolomb (:1449-1471) */
    {
        uint32_t bins = 0; int n = 0;
        while (symbol >= (1u << count)) { bins = 2 * bins + 1; n++; symbol -= 1u << count; count++; }
        bins = 2 * bins; n++;
        bins = (bins << count) | symbol; n += (int)count;
        binsEP(bins, n);
    }
    void unaryMax(uint32_t symbol, int c, int offset, uint32_t maxSymbol)       /* writeUnaryMaxSymbol (:1431-1447) */
    {
        bin(symbol ? 1 : 0, c);
        if (!symbol) return;
        const bool codeLast = maxSymbol > symbol;
        while (--symbol) bin(1, c + offset);
        if (codeLast) bin(0, c + offset);
    }

    /* ---- picture map access ---- */
    x265amd_cu_unit& U(int x4, int y4) { return units[y4 * w4 + x4]; }
    const x265amd_cu_unit* at(int x4, int y4) const { return (x4 < 0 || y4 < 0 || x4 >= w4 || y4 >= h4) ? nullptr : &units[y4 * w4 + x4]; }
    static bool coded(const x265amd_cu_unit* u) { return u && u->pred_mode != X265AMD_MODE_NONE; }

    /* ---- CU level (encodeCU, :775-847) ---- */
    void encodeCU(int x, int y, int depth, bool& dqp)
    {
        const int size = 64 >> depth;
        if (x >= si.pic_width || y >= si.pic_height) return;
        const bool inside = x + size <= si.pic_width && y + size <= si.pic_height;
        const x265amd_cu_unit& u = U(x >> 2, y >> 2);
        const bool canSplit = depth < si.max_cu_depth;
        if (!inside)
        {
            if (depth == si.max_cu_dqp_depth && si.use_dqp) dqp = true;
            for (int k = 0; k < 4; k++) encodeCU(x + (k & 1) * size / 2, y + (k >> 1) * size / 2, depth + 1, dqp);
            return;
        }
        if (canSplit)
        {
            /* codeSplitFlag with getCtxSplitFlag (cudata.cpp:955-970) */
            const x265amd_cu_unit* l = at((x >> 2) - 1, y >> 2);
            const x265amd_cu_unit* a = at(x >> 2, (y >> 2) - 1);
            const int c = (coded(l) && l->depth > depth) + (coded(a) && a->depth > depth);
            bin(u.depth > depth, C_SPLIT + c);
        }
        if (depth < u.depth && canSplit)
        {
            if (depth == si.max_cu_dqp_depth && si.use_dqp) dqp = true;
            for (int k = 0; k < 4; k++) encodeCU(x + (k & 1) * size / 2, y + (k >> 1) * size / 2, depth + 1, dqp);
            return;
        }
        if (depth <= si.max_cu_dqp_depth && si.use_dqp) dqp = true;
        if (si.tq_bypass_enabled) bin(u.tq_bypass, C_TQ_BYPASS);
        if (si.slice_type != 2)
        {
            const x265amd_cu_unit* l = at((x >> 2) - 1, y >> 2);
            const x265amd_cu_unit* a = at(x >> 2, (y >> 2) - 1);
            const int c = (coded(l) && l->pred_mode == X265AMD_MODE_SKIP) + (coded(a) && a->pred_mode == X265AMD_MODE_SKIP);
            bin(u.pred_mode == X265AMD_MODE_SKIP, C_SKIP + c);
            if (u.pred_mode == X265AMD_MODE_SKIP)
            {
                mergeIndex(u);
                finishCU(x, y, depth, dqp);
                return;
            }
            bin(u.pred_mode == X265AMD_MODE_INTRA, C_PRED_MODE);
        }
        partSize(u, depth, size);
        predInfo(x, y, size, u);
        /* getIntraTUQtDepthRange / getInterTUQtDepthRange (cudata.cpp:972-993) */
        const int log2CU = 6 - depth;
        int range[2] = { si.tu_log2_min, si.tu_log2_max };
        const bool intra = u.pred_mode == X265AMD_MODE_INTRA;
        const int maxDepth = intra ? si.tu_max_depth_intra : si.tu_max_depth_inter;
        const int splitFlag = intra ? u.part_size != PART_2Nx2N : (maxDepth == 1 && u.part_size != PART_2Nx2N);
        const uint32_t lo = (uint32_t)log2CU - (uint32_t)(maxDepth - 1 + splitFlag);      /* unsigned as in the reference: a wrap below zero clips to the maximum */
        range[0] = lo < (uint32_t)range[0] ? range[0] : (lo > (uint32_t)range[1] ? range[1] : (int)lo);
        /* codeCoeff (:1207-1222) */
        bool any = true;
        if (!intra)
        {
            const bool root = u.cbf[0] || u.cbf[1] || u.cbf[2];
            if (!(u.merge_flag && u.part_size == PART_2Nx2N)) bin(root, C_QT_ROOT_CBF);
            any = root;
        }
        if (any) transform(x, y, x, y, 0, log2CU, dqp, range);
        finishCU(x, y, depth, dqp);
    }

    void mergeIndex(const x265amd_cu_unit& u)           /* codeMergeIndex (:1572-1590) */
    {
        const uint32_t numCand = si.max_num_merge_cand;
        if (numCand > 1)
        {
            const uint32_t idx = u.mvp_idx[0];
            bin(idx != 0, C_MERGE_IDX);
            if (idx != 0)
            {
                uint32_t mask = (1u << idx) - 2;
                mask >>= (idx == numCand - 1) ? 1 : 0;
                binsEP(mask, (int)(idx - (idx == numCand - 1)));
            }
        }
    }

    void partSize(const x265amd_cu_unit& u, int depth, int size)        /* codePartSize (:1522-1570) */
    {
        const int ps = u.part_size;
        if (u.pred_mode == X265AMD_MODE_INTRA)
        {
            if (depth == si.max_cu_depth) bin(ps == PART_2Nx2N, C_PART_SIZE);
            return;
        }
        switch (ps)
        {
        case PART_2Nx2N: bin(1, C_PART_SIZE); break;
        case PART_2NxN: case PART_2NxnU: case PART_2NxnD:
            bin(0, C_PART_SIZE); bin(1, C_PART_SIZE + 1);
            if (si.max_amp_depth > depth)
            {
                bin(ps == PART_2NxN, C_PART_SIZE + 3);
                if (ps != PART_2NxN) binEP(ps == PART_2NxnU ? 0 : 1);
            }
            break;
        default:
            bin(0, C_PART_SIZE); bin(0, C_PART_SIZE + 1);
            if (depth == si.max_cu_depth && size != 8) bin(1, C_PART_SIZE + 2);
            if (si.max_amp_depth > depth)
            {
                bin(ps == PART_Nx2N, C_PART_SIZE + 3);
                if (ps != PART_Nx2N) binEP(ps == PART_nLx2N ? 0 : 1);
            }
            break;
        }
    }

    /* getIntraDirLumaPredictor (cudata.cpp:910-953): the above candidate only inside the CTU */
    void lumaPreds(int x, int y, uint32_t* p)
    {
        const x265amd_cu_unit* l = at((x >> 2) - 1, y >> 2);
        const x265amd_cu_unit* a = (y & 63) ? at(x >> 2, (y >> 2) - 1) : nullptr;
        const uint32_t left = (coded(l) && l->pred_mode == X265AMD_MODE_INTRA) ? l->luma_dir : 1;
        const uint32_t above = (coded(a) && a->pred_mode == X265AMD_MODE_INTRA) ? a->luma_dir : 1;
        if (left == above)
        {
            if (left >= 2) { p[0] = left; p[1] = ((left - 2 + 31) & 31) + 2; p[2] = ((left - 2 + 1) & 31) + 2; }
            else { p[0] = 0; p[1] = 1; p[2] = 26; }
        }
        else
        {
            p[0] = left; p[1] = above;
            p[2] = (left && above) ? 0 : ((left + above) < 2 ? 26 : 1);
        }
    }

    /* codeIntraDirLumaAng (:1592-1642) for the partitions whose first samples are (px[j], py[j]) */
    void intraDirLuma(const int* px, const int* py, int partNum)
    {
        uint32_t dir[4], preds[4][3]; int predIdx[4];
        for (int j = 0; j < partNum; j++)
        {
            dir[j] = U(px[j] >> 2, py[j] >> 2).luma_dir;
            lumaPreds(px[j], py[j], preds[j]);
            predIdx[j] = -1;
            for (int i = 0; i < 3; i++) if (dir[j] == preds[j][i]) predIdx[j] = i;
            bin(predIdx[j] != -1, C_ADI);
        }
        for (int j = 0; j < partNum; j++)
        {
            if (predIdx[j] != -1) { const int nz = !!predIdx[j]; binsEP((uint32_t)(predIdx[j] + nz), 1 + nz); }
            else
            {
                uint32_t* p = preds[j];
                if (p[0] > p[1]) { uint32_t t = p[0]; p[0] = p[1]; p[1] = t; }
                if (p[0] > p[2]) { uint32_t t = p[0]; p[0] = p[2]; p[2] = t; }
                if (p[1] > p[2]) { uint32_t t = p[1]; p[1] = p[2]; p[2] = t; }
                uint32_t d = dir[j];
                d -= d > p[2]; d -= d > p[1]; d -= d > p[0];
                binsEP(d, 5);
            }
        }
    }
    /* codeIntraDirChroma with getAllowedChromaDir (:1644-1664, cudata.cpp:889-907) */
    void intraDirChroma(const x265amd_cu_unit& u)
    {
        uint32_t c = u.chroma_dir;
        if (c == 36) bin(0, C_CHROMA_PRED);
        else
        {
            uint32_t list[4] = { 0, 26, 10, 1 };
            for (int i = 0; i < 4; i++) if (u.luma_dir == list[i]) { list[i] = 34; break; }
            for (uint32_t i = 0; i < 4; i++) if (c == list[i]) { c = i; break; }
            bin(1, C_CHROMA_PRED);
            binsEP(c, 2);
        }
    }

    /* ---- SAO syntax of a CTU (FrameEncoder::encodeSlice, frameencoder.cpp:1320-1346; Entropy::codeSaoOffset / codeSaoMaxUvlc,
     * entropy.cpp:1224-1262, :2201-2217).  flags: slice_sao_luma_flag, slice_sao_chroma_flag ---- */
    void saoMaxUvlc(uint32_t code, uint32_t maxSymbol)
    {
        binEP(code != 0);
        if (code)
        {
            const uint32_t isLast = maxSymbol > code;
            uint32_t mask = (1u << (code - 1)) - 1;
            const uint32_t len = code - 1 + isLast;
            mask <<= isLast;
            binsEP(mask, (int)len);
        }
    }
    void saoOffset(int typeIdx, int bandPos, const int8_t* offset, int plane)
    {
        const uint32_t thresh = (1u << ((X265AMD_DEPTH - 5) < 5 ? (X265AMD_DEPTH - 5) : 5)) - 1;
        if (plane != 2)
        {
            bin(typeIdx >= 0, C_SAO_TYPE);
            if (typeIdx >= 0) binEP(typeIdx < 4 ? 1 : 0);
        }
        if (typeIdx < 0) return;
        if (typeIdx == 4)
        {
            for (int i = 0; i < 4; i++) saoMaxUvlc((uint32_t)abs(offset[i]), thresh);
            for (int i = 0; i < 4; i++) if (offset[i]) binEP(offset[i] < 0);
            binsEP((uint32_t)bandPos, 5);
        }
        else
        {
            saoMaxUvlc((uint32_t)offset[0], thresh); saoMaxUvlc((uint32_t)offset[1], thresh);
            saoMaxUvlc((uint32_t)-offset[2], thresh); saoMaxUvlc((uint32_t)-offset[3], thresh);
            if (plane != 2) binsEP((uint32_t)typeIdx, 2);
        }
    }
    void saoCtu(int col, bool firstRowInSlice, const x265amd_sao_ctu& p, bool lumaFlag, bool chromaFlag)
    {
        if (!lumaFlag && !chromaFlag) return;
        const int mergeLeft = col && p.reserved[0] == 1, mergeUp = !firstRowInSlice && p.reserved[0] == 2;
        if (col) bin((uint32_t)mergeLeft, C_SAO_MERGE);
        if (!firstRowInSlice && !mergeLeft) bin((uint32_t)mergeUp, C_SAO_MERGE);
        if (!mergeLeft && !mergeUp)
        {
            if (lumaFlag) saoOffset(p.type[0], p.band_pos[0], p.offset[0], 0);
            if (chromaFlag) { saoOffset(p.type[1], p.band_pos[1], p.offset[1], 1); saoOffset(p.type[1], p.band_pos[2], p.offset[2], 2); }
        }
    }

    void predInfo(int x, int y, int size, const x265amd_cu_unit& u)     /* codePredInfo / codePUWise (:1138-1197) */
    {
        if (u.pred_mode == X265AMD_MODE_INTRA)
        {
            const int partNum = u.part_size != PART_2Nx2N ? 4 : 1;
            int px[4], py[4];
            for (int j = 0; j < partNum; j++) { px[j] = x + (j & 1) * size / 2; py[j] = y + (j >> 1) * size / 2; }
            intraDirLuma(px, py, partNum);
            intraDirChroma(u);
            return;
        }
        /* inter: every PU */
        static const uint8_t nbParts[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
        /* PU origin in quarters of the CU per part size (partAddrTable, cudata.cpp) */
        static const uint8_t puX[8][4] = { { 0 }, { 0, 0 }, { 0, 2 }, { 0, 2, 0, 2 }, { 0, 0 }, { 0, 0 }, { 0, 1 }, { 0, 3 } };
        static const uint8_t puY[8][4] = { { 0 }, { 0, 2 }, { 0, 0 }, { 0, 0, 2, 2 }, { 0, 1 }, { 0, 3 }, { 0, 0 }, { 0, 0 } };
        const int ps = u.part_size;
        for (int i = 0; i < nbParts[ps]; i++)
        {
            const x265amd_cu_unit& pu = U((x + puX[ps][i] * size / 4) >> 2, (y + puY[ps][i] * size / 4) >> 2);
            bin(pu.merge_flag, C_MERGE_FLAG);
            if (pu.merge_flag) { mergeIndex(pu); continue; }
            if (si.slice_type == 0)
            {
                /* codeInterDir (:1666-1675): context = CU depth */
                const uint32_t dir = pu.inter_dir - 1;
                if (ps == PART_2Nx2N || size != 8) bin(dir == 2, C_INTER_DIR + pu.depth);
                if (dir < 2) bin(dir, C_INTER_DIR + 4);
            }
            for (int list = 0; list < 2; list++)
                if (pu.inter_dir & (1 << list))
                {
                    if (si.num_ref_idx[list] > 1)
                    {
                        /* codeRefFrmIdx (:1677-1698) */
                        uint32_t ref = (uint32_t)pu.ref_idx[list];
                        bin(ref > 0, C_REF_NO);
                        if (ref > 0)
                        {
                            const uint32_t refNum = si.num_ref_idx[list] - 2;
                            if (refNum)
                            {
                                ref--;
                                bin(ref > 0, C_REF_NO + 1);
                                if (ref > 0)
                                {
                                    uint32_t mask = (1u << ref) - 2;
                                    mask >>= (ref == refNum) ? 1 : 0;
                                    binsEP(mask, (int)(ref - (ref == refNum)));
                                }
                            }
                        }
                    }
                    /* codeMvd (:1700-1735) */
                    const int hor = pu.mvd[list][0], ver = pu.mvd[list][1];
                    bin(hor != 0, C_MV_RES); bin(ver != 0, C_MV_RES);
                    const uint32_t ha = (uint32_t)abs(hor), va = (uint32_t)abs(ver);
                    if (hor) bin(ha > 1, C_MV_RES + 1);
                    if (ver) bin(va > 1, C_MV_RES + 1);
                    if (hor) { if (ha > 1) epExGolomb(ha - 2, 1); binEP(hor < 0); }
                    if (ver) { if (va > 1) epExGolomb(va - 2, 1); binEP(ver < 0); }
                    bin(pu.mvp_idx[list], C_MVP_IDX);
                }
        }
    }

    /* ---- transform tree (encodeTransform, :930-1063) ---- */
    bool cbfAt(int x, int y, int plane, int depth) { return (U(x >> 2, y >> 2).cbf[plane] >> depth) & 1; }
    void transform(int cuX, int cuY, int x, int y, int curDepth, int log2Size, bool& dqp, const int range[2])
    {
        const x265amd_cu_unit& u = U(x >> 2, y >> 2);
        const bool subdiv = u.tu_depth > curDepth;
        const bool intra = u.pred_mode == X265AMD_MODE_INTRA;
        if (intra && u.part_size != PART_2Nx2N && log2Size == 3) { }
        else if (!intra && u.part_size != PART_2Nx2N && !curDepth && si.tu_max_depth_inter == 1) { }
        else if (log2Size > range[1]) { }
        else if (log2Size == si.tu_log2_min || log2Size == range[0]) { }
        else bin(subdiv, C_TRANS_SUBDIV + 5 - log2Size);

        const bool smallChroma = log2Size - 1 < 2;
        if (!curDepth || !smallChroma)
        {
            /* the parent TU's chroma CBF gates the child's (a 4x4-luma quartet shares the parent's chroma block) */
            const int psize = 2 << log2Size, px = x & ~(psize - 1), py = y & ~(psize - 1);
            for (int c = 1; c < 3; c++)
                if (!curDepth || cbfAt(px, py, c, curDepth - 1))
                {
                    /* codeQtCbfChroma (:1758-1780): unsplittable TUs inherit the parent's CBF */
                    const bool canQuadSplit = log2Size - 1 > 2;
                    const int lowest = curDepth + ((subdiv && !canQuadSplit) ? 1 : 0);
                    bin(cbfAt(x, y, c, lowest), C_QT_CBF + curDepth + 2);
                }
        }
        if (subdiv)
        {
            const int half = 1 << (log2Size - 1);
            for (int k = 0; k < 4; k++) transform(cuX, cuY, x + (k & 1) * half, y + (k >> 1) * half, curDepth + 1, log2Size - 1, dqp, range);
            return;
        }
        /* a 4x4 luma TU's chroma belongs to the quartet's first TU position */
        const int xc = smallChroma ? x & ~7 : x, yc = smallChroma ? y & ~7 : y;
        const bool cbfU = cbfAt(xc, yc, 1, curDepth), cbfV = cbfAt(xc, yc, 2, curDepth);
        if (!intra && !curDepth && !cbfAt(xc, yc, 1, 0) && !cbfAt(xc, yc, 2, 0)) { /* luma CBF implied */ }
        else bin(cbfAt(x, y, 0, curDepth), C_QT_CBF + !curDepth);
        const bool cbfY = cbfAt(x, y, 0, curDepth);
        if (!(cbfY || cbfU || cbfV)) return;
        if (si.use_dqp && dqp)
        {
            deltaQP(cuX, cuY);
            dqp = false;
        }
        if (cbfY)
        {
            coeffNxN(coeffAddr(0, x, y), log2Size, 0, u);
            if (!(cbfU || cbfV)) return;
        }
        if (smallChroma)
        {
            if (!((x & 4) && (y & 4))) return;          /* (absPartIdx & 3) != 3 */
            for (int c = 1; c < 3; c++)
                if (cbfAt(xc, yc, c, curDepth)) coeffNxN(coeffAddr(c, xc, yc), 2, c, U(xc >> 2, yc >> 2));
        }
        else
            for (int c = 1; c < 3; c++)
                if (cbfAt(x, y, c, curDepth)) coeffNxN(coeffAddr(c, x, y), log2Size - 1, c, u);
    }

    /* coefficient storage of the reference: per CTU, TU blocks in z-order (coeffOffset = absPartIdx << 4, chroma >> 2) */
    const int16_t* coeffCtu[3];
    int ctuX0, ctuY0;
    static uint32_t zorder(int x4, int y4)          /* z-order index of a 4x4 unit inside its CTU */
    {
        uint32_t z = 0;
        for (int b = 0; b < 4; b++) z |= (((uint32_t)x4 >> b) & 1) << (2 * b) | (((uint32_t)y4 >> b) & 1) << (2 * b + 1);
        return z;
    }
    const int16_t* coeffAddr(int plane, int x, int y)
    {
        const uint32_t z = zorder((x - ctuX0) >> 2, (y - ctuY0) >> 2);
        return coeffCtu[plane] + (plane ? (z << 4) >> 2 : z << 4);
    }

    /* codeDeltaQP with getRefQP (:1737-1756, cudata.cpp:814-855) */
    int refQP(int x, int y)
    {
        const int qg = 64 >> si.max_cu_dqp_depth;
        const int gx = x & ~(qg - 1), gy = y & ~(qg - 1);
        const x265amd_cu_unit* l = (gx & 63) ? at((gx >> 2) - 1, gy >> 2) : nullptr;
        const x265amd_cu_unit* a = (gy & 63) ? at(gx >> 2, (gy >> 2) - 1) : nullptr;
        const int lq = l ? l->qp : lastQP(x, y), aq = a ? a->qp : lastQP(x, y);
        return (lq + aq + 1) >> 1;
    }
    /* CUData::getLastCodedQP (cudata.cpp:857-887): QP of the CU coded just before the quantisation group.  getLastValidPartIdx steps backwards over the CTU's units: a unit
     * that carries no CU (MODE_NONE) is stepped over together with the block its m_cuDepth names.  Units outside the picture are such units, and what the reference's CTU
     * arrays hold for them depends on how far the CTU has come: CUData::initCTU clears the depths (a step of the whole CTU), and the copyToPic of the smallest CU that has
     * the unit's block as an absent sub-CU writes that sub-CU's depth (setEmptyPart, cudata.cpp:422-427) -- so in a finished CTU the unit's depth is that of the largest
     * block around it whose corner lies outside the picture, and in the CTU under analysis (ctuInProgress) that only holds once the CU with the absent sub-CU is complete,
     * i.e. when it does not contain the quantisation group the question is asked for. */
    bool ctuInProgress = false;     /* the analysis asks (Search::checkDQP / checkDQPForSplitPred on the CTU being compressed); false: the CTU is complete (Entropy::encodeCTU) */
    int lastQP(int x, int y)
    {
        const int qg = 64 >> si.max_cu_dqp_depth;
        const int gx = x & ~(qg - 1), gy = y & ~(qg - 1);
        const int ctuAddr = (gy >> 6) * ctuW + (gx >> 6);
        int z = (int)zorder((gx & 63) >> 2, (gy & 63) >> 2) - 1;
        for (int addr = ctuAddr;; )
        {
            const int bx = (addr % ctuW) * 64, by = (addr / ctuW) * 64;
            const bool inProgress = ctuInProgress && addr == ctuAddr;
            while (z >= 0)
            {
                /* unit z of this CTU */
                int ux = 0, uy = 0;
                for (int b = 0; b < 4; b++) { ux |= ((z >> (2 * b)) & 1) << b; uy |= ((z >> (2 * b + 1)) & 1) << b; }
                const int px = bx + ux * 4, py = by + uy * 4;
                int depth = 0;
                if (px < si.pic_width && py < si.pic_height)
                {
                    const x265amd_cu_unit* u = at(px >> 2, py >> 2);
                    if (coded(u)) return u->qp;
                    depth = u->depth;
                }
                else
                {
                    int d = 1;
                    for (; d < 4; d++)
                    {
                        const int sz = 64 >> d;
                        if (bx + ((ux * 4) & ~(sz - 1)) >= si.pic_width || by + ((uy * 4) & ~(sz - 1)) >= si.pic_height) break;
                    }
                    depth = d;
                    if (inProgress)
                    {
                        /* the CU that has this block as an absent sub-CU: complete unless the group asked for lies in it */
                        const int psz = 64 >> (d - 1);
                        const int ox = bx + ((ux * 4) & ~(psz - 1)), oy = by + ((uy * 4) & ~(psz - 1));
                        if (gx >= ox && gx < ox + psz && gy >= oy && gy < oy + psz) depth = 0;
                    }
                }
                z -= 256 >> (2 * depth);
            }
            if (addr > 0 && !(si.wpp && !(addr % ctuW))) { addr--; z = 255; }
            else return si.slice_qp;
        }
    }
    /* where lastQP(x, y) ends when every unit of the picture in front of the group carries a CU (true of any group the analysis has reached): the unit's position, false when
     * the walk leaves the CTU (the answer is then lastQP's of the CTU's origin).  Picture geometry alone -- the device-run skip chain is told (inter_chain_dev.h: last_src) */
    bool lastQPUnitInCtu(int x, int y, int& px, int& py) const
    {
        const int qg = 64 >> si.max_cu_dqp_depth;
        const int gx = x & ~(qg - 1), gy = y & ~(qg - 1);
        const int bx = gx & ~63, by = gy & ~63;
        int z = (int)zorder((gx & 63) >> 2, (gy & 63) >> 2) - 1;
        while (z >= 0)
        {
            int ux = 0, uy = 0;
            for (int b = 0; b < 4; b++) { ux |= ((z >> (2 * b)) & 1) << b; uy |= ((z >> (2 * b + 1)) & 1) << b; }
            px = bx + ux * 4; py = by + uy * 4;
            if (px < si.pic_width && py < si.pic_height) return true;
            int d = 1;
            for (; d < 4; d++)
            {
                const int sz = 64 >> d;
                if (bx + ((ux * 4) & ~(sz - 1)) >= si.pic_width || by + ((uy * 4) & ~(sz - 1)) >= si.pic_height) break;
            }
            int depth = d;
            if (ctuInProgress)
            {
                const int psz = 64 >> (d - 1);
                const int ox = bx + ((ux * 4) & ~(psz - 1)), oy = by + ((uy * 4) & ~(psz - 1));
                if (gx >= ox && gx < ox + psz && gy >= oy && gy < oy + psz) depth = 0;
            }
            z -= 256 >> (2 * depth);
        }
        return false;
    }
    void deltaQP(int x, int y)
    {
        const x265amd_cu_unit& u = U(x >> 2, y >> 2);
        int dqp = u.qp - refQP(x, y);
        const int bd = 6 * (X265AMD_DEPTH - 8);
        dqp = (dqp + 78 + bd + (bd / 2)) % (52 + bd) - 26 - (bd / 2);
        const uint32_t a = (uint32_t)abs(dqp);
        unaryMax(a < 5 ? a : 5, C_DELTA_QP, 1, 5);
        if (a >= 5) epExGolomb(a - 5, 0);
        if (a > 0) binEP(dqp > 0 ? 0 : 1);
    }

    /* finishCU (:897-928) */
    void finishCU(int x, int y, int depth, bool dqp)
    {
        const int size = 64 >> depth;
        const int rpelx = x + size, bpely = y + size;
        const bool boundary = ((rpelx & 63) == 0 || rpelx == si.pic_width) && ((bpely & 63) == 0 || bpely == si.pic_height);
        if (si.use_dqp)
        {
            const int8_t q = dqp ? (int8_t)refQP(x, y) : U(x >> 2, y >> 2).qp;
            for (int yy = y >> 2; yy < (y + size) >> 2; yy++)
                for (int xx = x >> 2; xx < (x + size) >> 2; xx++) U(xx, yy).qp = q;
        }
        if (boundary)
        {
            const bool last = rpelx == si.pic_width && bpely == si.pic_height;         /* the slice ends with the picture */
            if (!last) binTrm(0);
            if (bitsOnly) { ctuBits = fracBits; fracBits &= 32767; }        /* resetBits() keeps the fraction (:2445-2455) */
        }
    }

    /* ---- coefficients (codeCoeffNxN, :1828-2199) ---- */
    void coeffNxN(const int16_t* coeff, int log2N, int ttype, const x265amd_cu_unit& u)
    {
        const int N = 1 << log2N, isLuma = ttype == 0;
        const bool intra = u.pred_mode == X265AMD_MODE_INTRA;
        int dirMode = isLuma ? u.luma_dir : (u.chroma_dir == 36 ? u.luma_dir : u.chroma_dir);
        const int scanType = !intra ? 0 : ((log2N <= 2 || (isLuma && log2N == 3)) ? (dirMode >= 22 && dirMode <= 30 ? 1 : (dirMode >= 6 && dirMode <= 14 ? 2 : 0)) : 0);
        const int gType = log2N >= 4 ? 0 : scanType;
        const int ncg = 1 << (2 * (log2N - 2));
        const uint32_t log2CG = (uint32_t)log2N - 2, cgStride = (uint32_t)N >> 2;
        const bool hideSign = si.sign_hide && !u.tq_bypass;
        int lastSet = -1, lastK = -1;
        uint64_t cgFlags = 0;
        for (int g = ncg - 1; g >= 0 && lastSet < 0; g--)
        {
            const uint32_t blk = cgBlk(gType, log2N, g);
            const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
            for (int k = 15; k >= 0; k--)
            {
                const uint32_t rr = inCg(gType, k);
                if (coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)]) { lastSet = g; lastK = k; break; }
            }
        }
        if (lastSet < 0) return;
        {
            const uint32_t blk = cgBlk(gType, log2N, lastSet), rr = inCg(gType, lastK);
            uint32_t px = (blk & ((1u << log2CG) - 1)) * 4 + (rr & 3), py = (blk >> log2CG) * 4 + (rr >> 2);
            if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
            int ctxIdx = isLuma ? 3 * (log2N - 2) + (log2N == 5) : 15;
            const int ctxShift = isLuma ? (log2N > 2) : log2N - 2;
            const uint32_t maxGroupIdx = ((uint32_t)log2N << 1) - 1;
            uint32_t sufBits = 0, sufLen = 0;
            for (int i = 0; i < 2; i++, ctxIdx += 18)
            {
                const uint32_t pos = i ? py : px;
                uint32_t prefix = pos, suffixLen = 0;
                if (pos >= 4) { const uint32_t l = 31 - (uint32_t)__builtin_clz(pos); suffixLen = l - 1; prefix = 2 * l + ((pos >> (l - 1)) & 1); }
                for (uint32_t k = 0; k < prefix; k++) bin(1, C_LAST_X + ctxIdx + (int)(k >> ctxShift));
                if (prefix < maxGroupIdx) bin(0, C_LAST_X + ctxIdx + (int)(prefix >> ctxShift));
                sufBits = (sufBits << suffixLen) | (pos & ((1u << suffixLen) - 1));
                sufLen += suffixLen;
            }
            binsEP(sufBits, (int)sufLen);
        }
        for (int g = 0; g < lastSet; g++)
        {
            const uint32_t blk = cgBlk(gType, log2N, g);
            const int base = (int)((blk >> log2CG) * 4) * N + (int)((blk & ((1u << log2CG) - 1)) * 4);
            bool any = false;
            for (int yy = 0; yy < 4; yy++) for (int xx = 0; xx < 4; xx++) any |= coeff[base + yy * N + xx] != 0;
            if (any) cgFlags |= (uint64_t)1 << blk;
        }
        const int cgCtx = C_SIG_CG + (isLuma ? 0 : 2), sigCtx = C_SIG + (isLuma ? 0 : 27);
        const int firstSig = log2N == 2 ? 0 : log2N == 3 ? ((scanType != 0 && isLuma) ? 15 : 9) : (isLuma ? 21 : 12);
        uint32_t c1 = 1;
        int sigOff = lastK - 1;
        uint16_t absCoeff[17]; uint32_t signs = 0;
        uint32_t numNonZero = 1;
        {
            const uint32_t blk = cgBlk(gType, log2N, lastSet), rr = inCg(gType, lastK);
            const int v = coeff[(int)((blk >> log2CG) * 4 + (rr >> 2)) * N + (int)((blk & ((1u << log2CG) - 1)) * 4 + (rr & 3))];
            absCoeff[0] = (uint16_t)abs(v); signs = v < 0;
        }
        for (int sub = lastSet; sub >= 0; sub--)
        {
            const int subBase = sub << 4;
            const uint32_t blk = cgBlk(gType, log2N, sub), cgY = blk >> log2CG, cgX = blk & ((1u << log2CG) - 1);
            const uint64_t cgMask = (uint64_t)1 << blk;
            const int base = (int)(cgY * 4) * N + (int)(cgX * 4);
            uint32_t firstNZ = 16, lastNZ = 0;
            const uint32_t sigPos = blk + 1 < 64 ? (uint32_t)(cgFlags >> (blk + 1)) : 0;
            const uint32_t right = (cgX != cgStride - 1) & sigPos, lower = (cgY != cgStride - 1) & (sigPos >> (cgStride - 1));
            if (sub == lastSet || !sub) cgFlags |= cgMask;
            else bin((cgFlags & cgMask) != 0, cgCtx + (int)(right | lower));
            if (sub == lastSet) firstNZ = lastNZ = (uint32_t)lastK;
            else { numNonZero = 0; signs = 0; }
            if (sigOff >= 0 && (cgFlags & cgMask))
            {
                const uint32_t pattern = cgStride == 1 ? 0 : right + lower * 2;
                const int offset = firstSig + ((isLuma && sub) ? 3 : 0);
                uint32_t nnz = sub == lastSet ? 1 : 0;
                for (int k = sigOff; k >= 0; k--)
                {
                    const uint32_t rr = inCg(gType, k);
                    const int v = coeff[base + (int)(rr >> 2) * N + (int)(rr & 3)];
                    const uint32_t sig = v != 0;
                    if (k != 0 || sub == 0 || nnz)
                    {
                        const uint32_t c = (subBase + k) ? sigCtxInc(log2N, pattern, rr) + (uint32_t)offset : 0;
                        bin(sig, sigCtx + (int)c);
                    }
                    if (sig)
                    {
                        absCoeff[nnz] = (uint16_t)abs(v);
                        signs |= (uint32_t)(v < 0) << nnz;
                        if (firstNZ == 16 || (uint32_t)k < firstNZ) firstNZ = (uint32_t)k;
                        if (nnz == 0) lastNZ = (uint32_t)k;
                    }
                    nnz += sig;
                }
                numNonZero = nnz;
            }
            if (numNonZero > 0)
            {
                const bool signHidden = lastNZ - firstNZ >= 4;
                const uint32_t ctxSet = (((sub > 0) + (uint32_t)isLuma) & 2) + !(c1 & 3);
                const int oneCtx = C_ONE + (isLuma ? 0 : 16) + 4 * (int)ctxSet;
                const uint32_t numC1 = numNonZero < 8 ? numNonZero : 8;
                uint32_t firstC2Idx = 8, firstC2Flag = 2, c1Next = 0xFFFFFFFE;
                c1 = 1;
                for (uint32_t idx = 0; idx < numC1; idx++)
                {
                    const uint32_t s1 = absCoeff[idx] > 1, s2 = absCoeff[idx] > 2;
                    bin(s1, oneCtx + (int)c1);
                    if (s1) c1Next = 0;
                    if (s1 + firstC2Flag == 3) firstC2Flag = s2;
                    if (s1 + firstC2Idx == 9) firstC2Idx = idx;
                    c1 = c1Next & 3;
                    c1Next >>= 2;
                }
                if (!c1) bin(firstC2Flag, C_ABS + (isLuma ? 0 : 4) + (int)ctxSet);
                /* signs: coeffSign holds the sign of the j-th level found in coding order in bit j (scanPosLast_c, dct.cpp:757-790, builds
                 * it from the other end: bit (count-1-j)); encodeBinsEP sends the most significant of n bits first */
                {
                    const int hidden = (hideSign && signHidden) ? 1 : 0;
                    uint32_t v = 0;
                    for (uint32_t j = 0; j < numNonZero; j++) v |= ((signs >> j) & 1) << (numNonZero - 1 - j);      /* first found = msb */
                    binsEP(v >> hidden, (int)numNonZero - hidden);
                }
                if (!c1 || numNonZero > 8)
                {
                    uint32_t rice = 0, threshold = 3;
                    int baseLevel = 3;
                    for (uint32_t idx = firstC2Idx; idx < numNonZero; idx++)
                    {
                        if (idx >= 8) baseLevel = 1;
                        if ((int)absCoeff[idx] >= baseLevel)
                        {
                            remain((uint32_t)(absCoeff[idx] - baseLevel), rice);
                            const int adjust = (absCoeff[idx] > threshold) & (rice <= 3);
                            rice += (uint32_t)adjust;
                            threshold += adjust ? threshold : 0;
                        }
                        baseLevel = 2;
                    }
                }
            }
            sigOff = 15;
        }
    }
    void remain(uint32_t codeNumber, uint32_t rice)     /* writeCoefRemainExGolomb (:1473-1503) */
    {
        const uint32_t codeRemain = codeNumber & ((1u << rice) - 1);
        if ((codeNumber >> rice) < 3)
        {
            const uint32_t length = codeNumber >> rice;
            binsEP((((1u << (length + 1)) - 2) << rice) + codeRemain, (int)(length + 1 + rice));
        }
        else
        {
            codeNumber = (codeNumber >> rice) - 3;
            const uint32_t length = 31 - (uint32_t)__builtin_clz(codeNumber + 1);
            codeNumber -= (1u << length) - 1;
            codeNumber = (codeNumber << rice) + codeRemain;
            binsEP((1u << (3 + length + 1)) - 2, (int)(3 + length + 1));
            binsEP(codeNumber, (int)(length + rice));
        }
    }
};

#endif
