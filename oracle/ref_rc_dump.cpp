/* TEST INFRASTRUCTURE ONLY.
 * Ground truth for the rate-control side of the encoder object (CRF + adaptive quantisation + cuTree): runs the REFERENCE encoder (linked from its own objects by
 * oracle/build_ref.sh) on a raw planar 4:2:0 clip and, for every picture it emits, writes what its lookahead and rate control decided for that picture --
 * read out of the reference's own objects (Encoder::m_exportedPic: the Frame behind the picture just returned, encoder.cpp:2108-2116):
 *
 *   slice type, POC, reference POCs, the rate control's QP (FrameData::m_avgQpRc, a double: ratecontrol.cpp:1585), the slice QP, Lowres::satdCost, the scene-cut mark,
 *   Lowres::qpAqOffset / qpCuTreeOffset / invQscaleFactor per 16x16 block (slicetype.cpp:452-713, :3750-3800), Lowres::intraCost / propagateCost per lowres block,
 *   and the QP, depth, prediction mode and coded-block flag of every 4x4 unit of the coded picture in raster order (CUData::m_qp ...).
 *
 * The byte stream goes to <out>.hevc.  tests/golden/make_golden.py turns the records into fixtures; dbg/ scripts compare them with the encoder object's own dump
 * (X265AMD_RC_DUMP).
 *
 * usage: x265_rc_dump<8|10> <clip.yuv> <width> <height> <frames> <out prefix> <preset> [key=value ...]
 *
 * record layout (little endian), one per emitted picture, in coding order:
 *   int32  magic 0x52434450, poc, sliceType (X265_TYPE_*), isReferenced, sliceQp, bScenecut, numRef[2], refPoc[2][16], w4, h4, blocks16, lowresBlocks
 *   int64  satdCost
 *   double avgQpRc, avgQpAq
 *   double qpAqOffset[blocks16], qpCuTreeOffset[blocks16];  int32 invQscaleFactor[blocks16]
 *   int32  intraCost[lowresBlocks];  uint16 propagateCost[lowresBlocks]
 *   int8   qp[w4 * h4]; uint8 depth[w4 * h4], predMode[w4 * h4], cbfY[w4 * h4]
 */
#define private public
#define protected public
#include "common.h"
#include "x265.h"
#include "encoder.h"
#include "frame.h"
#include "framedata.h"
#include "lowres.h"
#include "slice.h"
#include "cudata.h"
#include "picyuv.h"
#undef private
#undef protected
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace X265_NS;

static void dumpFrame(FILE* f, Encoder* enc, Frame* fr)
{
    const x265_param* p = enc->m_param;
    FrameData& d = *fr->m_encData;
    Slice* s = d.m_slice;
    const int W = p->sourceWidth, H = p->sourceHeight, w4 = W / 4, h4 = H / 4;
    Lowres& lr = fr->m_lowres;
    const int blocks16 = lr.maxBlocksInRow * lr.maxBlocksInCol;
    const int lowresBlocks = (((W / 2) + 7) >> 3) * (((H / 2) + 7) >> 3);
    int32_t hdr[8 + 32 + 4];
    memset(hdr, 0, sizeof(hdr));
    hdr[0] = 0x52434450; hdr[1] = fr->m_poc; hdr[2] = lr.sliceType; hdr[3] = IS_REFERENCED(fr) ? 1 : 0; hdr[4] = s->m_sliceQp; hdr[5] = lr.bScenecut ? 1 : 0;
    hdr[6] = s->m_sliceType == I_SLICE ? 0 : s->m_numRefIdx[0]; hdr[7] = s->m_sliceType == B_SLICE ? s->m_numRefIdx[1] : 0;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < hdr[6 + l] && r < 16; r++) hdr[8 + 16 * l + r] = s->m_refPOCList[l][r];
    hdr[40] = w4; hdr[41] = h4; hdr[42] = blocks16; hdr[43] = lowresBlocks;
    fwrite(hdr, sizeof(hdr), 1, f);
    int64_t satd = lr.satdCost;
    fwrite(&satd, 8, 1, f);
    double q[2] = { d.m_avgQpRc, d.m_avgQpAq };
    fwrite(q, 8, 2, f);
    std::vector<double> zeros(blocks16, 0.0); std::vector<int32_t> zi(blocks16 > lowresBlocks ? blocks16 : lowresBlocks, 0); std::vector<uint16_t> zs(lowresBlocks, 0);
    fwrite(lr.qpAqOffset ? lr.qpAqOffset : zeros.data(), 8, blocks16, f);
    fwrite(lr.qpCuTreeOffset ? lr.qpCuTreeOffset : zeros.data(), 8, blocks16, f);
    fwrite(lr.invQscaleFactor ? lr.invQscaleFactor : zi.data(), 4, blocks16, f);
    fwrite(lr.intraCost ? lr.intraCost : zi.data(), 4, lowresBlocks, f);
    fwrite(lr.propagateCost ? lr.propagateCost : zs.data(), 2, lowresBlocks, f);
    std::vector<int8_t> qp((size_t)w4 * h4, 0); std::vector<uint8_t> depth((size_t)w4 * h4, 0), mode((size_t)w4 * h4, 0), cbf((size_t)w4 * h4, 0);
    const uint32_t ctuW = (W + 63) / 64, ctuH = (H + 63) / 64;
    for (uint32_t addr = 0; addr < ctuW * ctuH; addr++)
    {
        const CUData* ctu = d.getPicCTU(addr);
        const int cx = (addr % ctuW) * 16, cy = (addr / ctuW) * 16;
        for (uint32_t z = 0; z < 256; z++)
        {
            const uint32_t r = g_zscanToRaster[z];
            const int x = cx + (int)(r & 15), y = cy + (int)(r >> 4);
            if (x >= w4 || y >= h4) continue;
            qp[(size_t)y * w4 + x] = ctu->m_qp[z]; depth[(size_t)y * w4 + x] = ctu->m_cuDepth[z]; mode[(size_t)y * w4 + x] = ctu->m_predMode[z];
            cbf[(size_t)y * w4 + x] = ctu->m_cbf[0][z];
        }
    }
    fwrite(qp.data(), 1, qp.size(), f); fwrite(depth.data(), 1, depth.size(), f); fwrite(mode.data(), 1, mode.size(), f); fwrite(cbf.data(), 1, cbf.size(), f);
}

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s <clip.yuv> <w> <h> <frames> <out prefix> <preset> [key=value ...]\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), frames = atoi(argv[4]);
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, argv[6], NULL) < 0) return 3;
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = 30; p->fpsDenom = 1; p->internalCsp = X265_CSP_I420;
    p->bEmitInfoSEI = 0; p->logLevel = X265_LOG_WARNING;
    for (int i = 7; i < argc; i++)
    {
        char* eq = strchr(argv[i], '=');
        int r;
        if (eq) { *eq = 0; r = x265_param_parse(p, argv[i], eq + 1); }
        else r = x265_param_parse(p, argv[i], NULL);
        if (r) { fprintf(stderr, "bad option %s\n", argv[i]); return 3; }
    }
    x265_encoder* enc = x265_encoder_open(p);
    if (!enc) return 4;
    Encoder* E = static_cast<Encoder*>(enc);
    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 5; }
    char path[1024];
    snprintf(path, sizeof(path), "%s.hevc", argv[5]);
    FILE* hevc = fopen(path, "wb");
    snprintf(path, sizeof(path), "%s.rc", argv[5]);
    FILE* rc = fopen(path, "wb");
    if (!hevc || !rc) return 5;
    x265_picture* pic = x265_picture_alloc();
    x265_picture* out = x265_picture_alloc();
    x265_picture_init(p, pic); x265_picture_init(p, out);
    const int bytes = X265_DEPTH > 8 ? 2 : 1;
    std::vector<uint8_t> buf((size_t)w * h * 3 / 2 * bytes);
    pic->planes[0] = buf.data(); pic->planes[1] = buf.data() + (size_t)w * h * bytes; pic->planes[2] = buf.data() + (size_t)w * h * bytes * 5 / 4;
    pic->stride[0] = w * bytes; pic->stride[1] = pic->stride[2] = w / 2 * bytes;
    pic->bitDepth = X265_DEPTH;
    x265_nal* nal; uint32_t nnal;
    auto emit = [&](int got) {
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, hevc);
        if (got > 0 && E->m_exportedPic) dumpFrame(rc, E, E->m_exportedPic);
    };
    if (x265_encoder_headers(enc, &nal, &nnal) >= 0) emit(0);
    for (int f = 0; f < frames; f++)
    {
        if (fread(buf.data(), 1, buf.size(), in) != buf.size()) { fprintf(stderr, "short clip\n"); return 6; }
        pic->pts = f;
        const int got = x265_encoder_encode(enc, &nal, &nnal, pic, out);
        if (got < 0) return 7;
        emit(got);
    }
    for (;;)
    {
        const int got = x265_encoder_encode(enc, &nal, &nnal, NULL, out);
        if (got <= 0) break;
        emit(got);
    }
    fclose(hevc); fclose(rc); fclose(in);
    x265_encoder_close(enc);
    return 0;
}
