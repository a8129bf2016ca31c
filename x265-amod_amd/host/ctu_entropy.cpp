/* Final entropy coding of CTUs (include/x265amd.h: x265amd_cabac_*): host C++, the CABAC write pass of SURVEY section 8f rank 1.
 *
 * Restatement of Entropy::encodeCTU / encodeCU / finishCU / encodeTransform / codePredInfo / codePUWise / codeCoeff
 * (reference: source/encoder/entropy.cpp:768-1222), the syntax-element coders (:1431-1830), codeCoeffNxN in bitstream mode
 * (:1828-2200) and the arithmetic coder start / encodeBin / encodeBinEP / encodeBinsEP / encodeBinTrm / writeOut / finish
 * (:2399-2612), with the context derivations of CUData (source/common/cudata.cpp:814-1012).  4:2:0, no transform skip.
 * In bit-counting mode (no bitstream) the same walk accumulates the FIX15 fractional bits the reference's m_fracBits holds, which is
 * how its analysis prices every syntax element.
 *
 * The reference keeps a CTU's decisions in z-ordered CUData arrays; here a picture-wide raster map of 4x4 units (x265amd_cu_unit)
 * holds the same fields, so the neighbour look-ups of the context derivations are plain (x-1, y) / (x, y-1) reads.
 * Tables: context initialisation values, LPS transitions and rangeTabLps are those of ITU-T H.265 (9.3.2.2, tables 9-46 / 9-48);
 * the fractional-bit constants are the reference's (entropy.cpp:2614-2625).
 */
#include "cabac_coder.h"

extern "C" {

x265amd_cabac* x265amd_cabac_open(const x265amd_slice_info* si, x265amd_cu_unit* units, int bitsOnly)
{
    if (!si || !units || si->pic_width <= 0 || si->pic_height <= 0 || (si->pic_width & 7) || (si->pic_height & 7)) return nullptr;
    x265amd_cabac* c = new x265amd_cabac;
    c->si = *si; c->units = units;
    c->w4 = si->pic_width >> 2; c->h4 = si->pic_height >> 2; c->ctuW = (si->pic_width + 63) >> 6;
    c->bitsOnly = bitsOnly != 0;
    c->fracBits = 0; c->ctuBits = 0; c->partial = 0; c->partialBits = 0;
    x265amd_entropy_reset(si->slice_type, si->slice_qp, c->ctx);
    c->start();
    return c;
}
void x265amd_cabac_close(x265amd_cabac* c) { delete c; }
void x265amd_cabac_set_contexts(x265amd_cabac* c, const uint8_t* ctx) { memcpy(c->ctx, ctx, X265AMD_CTX_COUNT); }
void x265amd_cabac_get_contexts(const x265amd_cabac* c, uint8_t* ctx) { memcpy(ctx, c->ctx, X265AMD_CTX_COUNT); }
uint64_t x265amd_cabac_frac_bits(const x265amd_cabac* c) { return c->fracBits; }
uint64_t x265amd_cabac_ctu_bits(const x265amd_cabac* c) { return c->ctuBits; }

int x265amd_cabac_encode_ctu(x265amd_cabac* c, int ctuAddr, const int16_t* coeffY, const int16_t* coeffU, const int16_t* coeffV)
{
    if (!c || ctuAddr < 0 || ctuAddr >= c->ctuW * ((c->si.pic_height + 63) >> 6)) return X265AMD_EINVAL;
    c->coeffCtu[0] = coeffY; c->coeffCtu[1] = coeffU; c->coeffCtu[2] = coeffV;
    c->ctuX0 = (ctuAddr % c->ctuW) * 64; c->ctuY0 = (ctuAddr / c->ctuW) * 64;
    bool dqp = c->si.use_dqp != 0;
    c->encodeCU(c->ctuX0, c->ctuY0, 0, dqp);
    return X265AMD_OK;
}

/* Entropy::finishSlice (entropy.h:154): terminating bin, flush, rbsp trailing bits.  Returns the number of bytes (copied to out when it fits). */
size_t x265amd_cabac_finish_slice(x265amd_cabac* c, uint8_t* out, size_t cap)
{
    if (!c->bitsOnly)
    {
        c->binTrm(1);
        c->finish();
        c->writeBits(1, 1);
        if (c->partialBits) { c->pushByte(c->partial << (8 - c->partialBits)); c->partial = 0; c->partialBits = 0; }
    }
    if (out && cap >= c->out.size()) memcpy(out, c->out.data(), c->out.size());
    return c->out.size();
}

} // extern "C"
