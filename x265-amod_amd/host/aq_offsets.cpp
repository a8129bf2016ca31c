/* Adaptive quantisation, the host half (include/x265amd.h: x265amd_aq_offsets): LookaheadTLD::calcAdaptiveQuantFrame (reference:
 * source/encoder/slicetype.cpp:452-640) from the block energies of x265amd_aq_energy to the per-block QP offsets, for --aq-mode 1 (variance),
 * 2 (auto-variance) and 3 (auto-variance with a bias to dark scenes); no HDR10 luma offsets, external quant offsets, hevc-aq or edge modes.
 * Double-precision arithmetic in the reference's order and operand types (float constants widen exactly where they do there), so that the
 * offsets -- which feed rounding to integer QPs and the fixed-point factors below -- come out bit for bit. */
#include "x265amd.h"
#include <math.h>

namespace {

/* x265_exp2fix8 (source/common/common.cpp:96-103); the table is 256 * (2^(i/64) - 1) rounded (source/common/constants.cpp:552-558) */
int exp2fix8(double x)
{
    static int lut[64];
    static bool init = false;
    if (!init) { for (int i = 0; i < 64; i++) lut[i] = (int)floor(256.0 * (pow(2.0, i / 64.0) - 1.0) + 0.5); init = true; }
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

}

/* numBlocks: the groups the picture loop visits (ceil(width / qg) x ceil(height / qg)); blockCount: what the reference averages over (widthInCU x heightInCU of the
 * lowres grid, x 4 for qg 8) -- the same unless the picture size is not a multiple of 16 */
extern "C" int x265amd_aq_offsets(const uint32_t* energy, int numBlocks, int blockCount, int aqMode, double aqStrength, double aqBiasStrength, int qgSize,
                                  double* qpAqOffset, double* qpCuTreeOffset, int32_t* invQscaleFactor)
{
    if (!energy || !qpAqOffset || !qpCuTreeOffset || !invQscaleFactor || blockCount <= 0 || numBlocks <= 0 || aqMode < 1 || aqMode > 3 || (qgSize != 16 && qgSize != 8) || aqStrength == 0)
        return X265AMD_EINVAL;
    const float varianceBase = qgSize == 8 ? 11.427f : 14.427f, autoBase = qgSize == 8 ? 8.f : 11.f;
    /* meanRoot / meanSquare: the picture's averages of the blocks' energy^0.1 and of its square (auto-variance); offset: the block's QP offset */
    double meanSquare = 0, meanRoot = 0, offset = 0;
    double darkBias = 0.f;
    double strength = 0.f;
    if (aqMode == 2 || aqMode == 3)
    {
        const double toEightBit = 1.f / (1 << (2 * (X265AMD_DEPTH - 8)));
        for (int b = 0; b < numBlocks; b++)
        {
            offset = pow(energy[b] * toEightBit + 1, 0.1);
            qpCuTreeOffset[b] = offset;
            meanRoot += offset;
            meanSquare += offset * offset;
        }
        meanRoot /= blockCount;
        meanSquare /= blockCount;
        strength = aqStrength * meanRoot;
        meanRoot = meanRoot - 0.5f * (meanSquare - autoBase) / meanRoot;
        darkBias = aqBiasStrength * aqStrength;
    }
    else
        strength = aqStrength * 1.0397f;
    for (int b = 0; b < numBlocks; b++)
    {
        if (aqMode == 3)
        {
            offset = qpCuTreeOffset[b];
            offset = strength * (offset - meanRoot) + darkBias * (1.f - autoBase / (offset * offset));
        }
        else if (aqMode == 2)
        {
            offset = qpCuTreeOffset[b];
            offset = strength * (offset - meanRoot);
        }
        else
        {
            const uint32_t e = energy[b] > 1 ? energy[b] : 1;
            offset = strength * (log2((double)e) - (varianceBase + 2 * (X265AMD_DEPTH - 8)));
        }
        qpAqOffset[b] = offset;
        qpCuTreeOffset[b] = offset;
        invQscaleFactor[b] = exp2fix8(offset);
    }
    return X265AMD_OK;
}
