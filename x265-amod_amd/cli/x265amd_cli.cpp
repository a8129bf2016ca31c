/* x265amd -- command line front end of the encoder object (on-disk formats).  It mirrors the part of the reference's
 * command line program that the built subset needs (reference: source/x265cli.cpp option table, source/abrEncApp.cpp:552-824 encode loop,
 * source/input/y4m.cpp:150-330 header parsing and :405-441 frame layout, source/output/raw.cpp / yuv.cpp writers):
 *
 *     x265amd --input clip.y4m -o out.hevc [--recon rec.yuv] [--qp N] [--bframes N] [--keyint N] [--ref N] [--rd 2..6] [--rect] [--amp]
 *             [--limit-modes] [--limit-refs N] [--[no-]early-skip] [--rskip 0|1] [--psy-rd F] [--[no-]b-intra] [--me dia|hex|star] [--subme N]
 *             [--merange N] [--max-merge N] [--rdoq-level N] [--psy-rdoq F] [--[no-]deblock] [--[no-]sao] [--[no-]wpp] [--frames N]
 *             [--scenecut N | --no-scenecut] [--rc-lookahead N] [--min-keyint N] [--b-adapt 0|2]
 *
 * Like the reference (source/encoder/api.cpp:1107-1182, x265_api_get) the pixel depth selects the library: libx265amd_main.so for 8-bit input,
 * libx265amd_main10.so for 10-bit, loaded with dlopen from the directory of this program's ../lib.  Host C++ only; all device work is the library's. */
#include "../../include/x265amd_encoder.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

struct Api
{
    void* h = nullptr;
    void (*param_default)(x265amd_param*) = nullptr;
    x265amd_encoder* (*open)(const x265amd_param*) = nullptr;
    int (*headers)(x265amd_encoder*, x265amd_nal**, uint32_t*) = nullptr;
    int (*encode)(x265amd_encoder*, x265amd_nal**, uint32_t*, const x265amd_picture*, x265amd_picture*) = nullptr;
    void (*close)(x265amd_encoder*) = nullptr;
    const char* (*last_error)(void) = nullptr;
};

bool loadApi(Api& a, const std::string& dir, int depth)
{
    const std::string path = dir + (depth > 8 ? "/libx265amd_main10.so" : "/libx265amd_main.so");
    a.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!a.h) { fprintf(stderr, "x265amd: cannot load %s: %s\n", path.c_str(), dlerror()); return false; }
    a.param_default = (void (*)(x265amd_param*))dlsym(a.h, "x265amd_param_default");
    a.open = (x265amd_encoder* (*)(const x265amd_param*))dlsym(a.h, "x265amd_encoder_open");
    a.headers = (int (*)(x265amd_encoder*, x265amd_nal**, uint32_t*))dlsym(a.h, "x265amd_encoder_headers");
    a.encode = (int (*)(x265amd_encoder*, x265amd_nal**, uint32_t*, const x265amd_picture*, x265amd_picture*))dlsym(a.h, "x265amd_encoder_encode");
    a.close = (void (*)(x265amd_encoder*))dlsym(a.h, "x265amd_encoder_close");
    a.last_error = (const char* (*)(void))dlsym(a.h, "x265amd_last_error");
    return a.param_default && a.open && a.headers && a.encode && a.close && a.last_error;
}

struct Y4m { FILE* f = nullptr; int width = 0, height = 0, depth = 8; uint32_t fpsNum = 25, fpsDen = 1; int sarW = 0, sarH = 0; size_t frameBytes = 0; };

/* YUV4MPEG2 stream header: space separated tags W H F I A C X (y4m.cpp:150-330); only 4:2:0 at 8 or 10 bits is accepted here */
bool openY4m(Y4m& y, const char* path)
{
    y.f = fopen(path, "rb");
    if (!y.f) { fprintf(stderr, "x265amd: cannot open %s\n", path); return false; }
    char line[512];
    if (!fgets(line, sizeof(line), y.f) || strncmp(line, "YUV4MPEG2", 9)) { fprintf(stderr, "x265amd: %s is not a YUV4MPEG2 file\n", path); return false; }
    int csp = 420;
    for (char* tok = strtok(line + 9, " \n"); tok; tok = strtok(nullptr, " \n"))
    {
        switch (tok[0])
        {
        case 'W': y.width = atoi(tok + 1); break;
        case 'H': y.height = atoi(tok + 1); break;
        case 'F': sscanf(tok + 1, "%u:%u", &y.fpsNum, &y.fpsDen); break;
        case 'A': sscanf(tok + 1, "%d:%d", &y.sarW, &y.sarH); break;
        case 'C':
        {
            csp = atoi(tok + 1);
            const char* p = strchr(tok, 'p');
            y.depth = p ? atoi(p + 1) : 8;
            break;
        }
        default: break;
        }
    }
    if (csp != 420 || (y.depth != 8 && y.depth != 10) || y.width <= 0 || y.height <= 0 || !y.fpsNum || !y.fpsDen)
    { fprintf(stderr, "x265amd: unsupported y4m stream (4:2:0, 8 or 10 bits only)\n"); return false; }
    y.frameBytes = (size_t)y.width * y.height * 3 / 2 * (y.depth > 8 ? 2 : 1);
    return true;
}

bool readFrame(Y4m& y, std::vector<uint8_t>& buf)
{
    char line[256];
    if (!fgets(line, sizeof(line), y.f)) return false;
    if (strncmp(line, "FRAME", 5)) return false;
    buf.resize(y.frameBytes);
    return fread(buf.data(), 1, y.frameBytes, y.f) == y.frameBytes;
}

}

int main(int argc, char** argv)
{
    const char* input = nullptr; const char* output = nullptr; const char* recon = nullptr;
    int frames = 0;
    std::vector<std::pair<std::string, std::string>> opts;
    for (int i = 1; i < argc; i++)
    {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "x265amd: %s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--input") input = val();
        else if (a == "-o" || a == "--output") output = val();
        else if (a == "--recon" || a == "-r") recon = val();
        else if (a == "--frames" || a == "-f") frames = atoi(val());
        else if (a == "--rect" || a == "--amp" || a == "--limit-modes" || a == "--early-skip" || a == "--no-early-skip" || a == "--b-intra" || a == "--no-b-intra" ||
                 a == "--deblock" || a == "--no-deblock" || a == "--sao" || a == "--no-sao" || a == "--wpp" || a == "--no-wpp" || a == "--no-rect" || a == "--no-amp" || a == "--fast-intra" || a == "--no-fast-intra" || a == "--no-scenecut" || a == "--open-gop" || a == "--no-open-gop" || a == "--b-pyramid" || a == "--no-b-pyramid" || a == "--weightp" || a == "--no-weightp")
            opts.push_back({ a, "" });
        else if (a.rfind("--", 0) == 0) opts.push_back({ a, val() });
        else { fprintf(stderr, "x265amd: unknown argument %s\n", a.c_str()); return 2; }
    }
    if (!input || !output) { fprintf(stderr, "usage: x265amd --input clip.y4m -o out.hevc [options]\n"); return 2; }
    Y4m y;
    if (!openY4m(y, input)) return 1;
    std::string dir = argv[0];
    const size_t slash = dir.find_last_of('/');
    dir = (slash == std::string::npos ? std::string(".") : dir.substr(0, slash)) + "/../lib";
    if (const char* e = getenv("X265AMD_LIBDIR")) dir = e;
    Api api;
    if (!loadApi(api, dir, y.depth)) return 1;

    x265amd_param p;
    api.param_default(&p);
    p.sourceWidth = y.width; p.sourceHeight = y.height; p.fpsNum = y.fpsNum; p.fpsDenom = y.fpsDen;
    p.aspectRatioIdc = (y.sarW == 1 && y.sarH == 1) ? 1 : 0;
    if (!(p.aspectRatioIdc || !y.sarW || !y.sarH)) { fprintf(stderr, "x265amd: sample aspect ratio %d:%d is not supported\n", y.sarW, y.sarH); return 1; }
    for (auto& o : opts)
    {
        const std::string& k = o.first; const char* v = o.second.c_str();
        if (k == "--qp") p.qp = atoi(v);
        else if (k == "--bframes") p.bframes = atoi(v);
        else if (k == "--keyint") p.keyframeMax = atoi(v);
        else if (k == "--min-keyint") p.keyframeMin = atoi(v);
        else if (k == "--scenecut") p.scenecutThreshold = atoi(v);
        else if (k == "--no-scenecut") p.scenecutThreshold = 0;
        else if (k == "--rc-lookahead") p.lookaheadDepth = atoi(v);
        else if (k == "--b-adapt") p.bFrameAdaptive = atoi(v);
        else if (k == "--lookahead-slices") p.lookaheadSlices = atoi(v);
        else if (k == "--open-gop") p.bOpenGOP = 1;
        else if (k == "--weightp") p.bEnableWeightedPred = 1;
        else if (k == "--no-weightp") p.bEnableWeightedPred = 0;
        else if (k == "--b-pyramid") p.bBPyramid = 1;
        else if (k == "--no-b-pyramid") p.bBPyramid = 0;
        else if (k == "--no-open-gop") p.bOpenGOP = 0;
        else if (k == "--ref") p.maxNumReferences = atoi(v);
        else if (k == "--rd") p.rdLevel = atoi(v);
        else if (k == "--rect") p.bEnableRectInter = 1;
        else if (k == "--no-rect") p.bEnableRectInter = 0;
        else if (k == "--amp") p.bEnableAMP = 1;
        else if (k == "--no-amp") p.bEnableAMP = 0;
        else if (k == "--limit-modes") p.limitModes = 1;
        else if (k == "--limit-refs") p.limitReferences = atoi(v);
        else if (k == "--early-skip") p.bEnableEarlySkip = 1;
        else if (k == "--no-early-skip") p.bEnableEarlySkip = 0;
        else if (k == "--rskip") p.recursionSkipMode = atoi(v);
        else if (k == "--psy-rd") p.psyRd = atof(v);
        else if (k == "--b-intra") p.bIntraInBFrames = 1;
        else if (k == "--no-b-intra") p.bIntraInBFrames = 0;
        else if (k == "--me") p.searchMethod = !strcmp(v, "dia") || !strcmp(v, "0") ? 0 : !strcmp(v, "hex") || !strcmp(v, "1") ? 1 : !strcmp(v, "star") || !strcmp(v, "3") ? 3 : -1;
        else if (k == "--subme") p.subpelRefine = atoi(v);
        else if (k == "--merange") p.searchRange = atoi(v);
        else if (k == "--max-merge") p.maxNumMergeCand = atoi(v);
        else if (k == "--deblock") p.bEnableLoopFilter = 1;
        else if (k == "--no-deblock") p.bEnableLoopFilter = 0;
        else if (k == "--sao") p.bEnableSAO = 1;
        else if (k == "--no-sao") p.bEnableSAO = 0;
        else if (k == "--wpp") p.bEnableWavefront = 1;
        else if (k == "--no-wpp") p.bEnableWavefront = 0;
        else if (k == "--fast-intra") p.bEnableFastIntra = 1;
        else if (k == "--no-fast-intra") p.bEnableFastIntra = 0;
        else if (k == "--rdoq-level") p.rdoqLevel = atoi(v);
        else if (k == "--psy-rdoq") p.psyRdoqFix8 = (int)(atof(v) * 256.0);
        else if (k == "--ipratio") p.ipFactor = atof(v);
        else if (k == "--pbratio") p.pbFactor = atof(v);
        else if (k == "--frame-threads" || k == "-F") p.frameNumThreads = atoi(v);        /* > 1: the reference's frame-parallel rules (its default); 1: one picture at a time */
        else { fprintf(stderr, "x265amd: unknown option %s\n", k.c_str()); return 2; }
    }
    x265amd_encoder* enc = api.open(&p);
    if (!enc) { fprintf(stderr, "x265amd: %s\n", api.last_error()); return 1; }
    FILE* out = fopen(output, "wb");
    FILE* rec = recon ? fopen(recon, "wb") : nullptr;
    if (!out || (recon && !rec)) { fprintf(stderr, "x265amd: cannot open the output\n"); return 1; }
    x265amd_nal* nal = nullptr; uint32_t nnal = 0;
    if (api.headers(enc, &nal, &nnal) < 0) { fprintf(stderr, "x265amd: %s\n", api.last_error()); return 1; }
    for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);

    const int isz = y.depth > 8 ? 2 : 1;
    std::vector<uint8_t> buf;
    /* the reconstruction file is in display order (output/yuv.cpp writes each picture at poc * frame size) */
    std::vector<uint8_t> recBuf(recon ? y.frameBytes : 0);
    x265amd_picture picOut;
    memset(&picOut, 0, sizeof(picOut));
    if (recon)
    {
        picOut.planes[0] = recBuf.data(); picOut.planes[1] = recBuf.data() + (size_t)y.width * y.height * isz;
        picOut.planes[2] = (uint8_t*)picOut.planes[1] + (size_t)y.width * y.height / 4 * isz;
        picOut.stride[0] = y.width * isz; picOut.stride[1] = picOut.stride[2] = y.width / 2 * isz;
    }
    int coded = 0, read = 0, rc = 0;
    auto emit = [&](int ret) {
        if (ret <= 0) return;
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);
        if (rec) { fseek(rec, (long)((size_t)picOut.poc * y.frameBytes), SEEK_SET); fwrite(recBuf.data(), 1, y.frameBytes, rec); }
        coded++;
    };
    while ((!frames || read < frames) && readFrame(y, buf))
    {
        x265amd_picture pic;
        memset(&pic, 0, sizeof(pic));
        pic.planes[0] = buf.data(); pic.planes[1] = buf.data() + (size_t)y.width * y.height * isz;
        pic.planes[2] = (uint8_t*)pic.planes[1] + (size_t)y.width * y.height / 4 * isz;
        pic.stride[0] = y.width * isz; pic.stride[1] = pic.stride[2] = y.width / 2 * isz;
        read++;
        rc = api.encode(enc, &nal, &nnal, &pic, recon ? &picOut : nullptr);
        if (rc < 0) break;
        emit(rc);
    }
    while (rc >= 0 && (rc = api.encode(enc, &nal, &nnal, nullptr, recon ? &picOut : nullptr)) > 0) emit(rc);
    if (rc < 0) fprintf(stderr, "x265amd: %s\n", api.last_error());
    api.close(enc);
    fclose(out);
    if (rec) fclose(rec);
    fprintf(stderr, "x265amd: encoded %d frames\n", coded);
    return rc < 0 ? 1 : 0;
}
