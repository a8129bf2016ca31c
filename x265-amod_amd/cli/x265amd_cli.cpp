/* x265amd -- the command line front end (on-disk formats), a client of the reference's own interface: it loads libx265amd_main.so /
 * libx265amd_main10.so by the input's bit depth (the reference's multilib scheme, source/encoder/api.cpp:1107-1182), asks it for the `x265_api` table (x265_api_get_209) and
 * does what the reference's program does with that table (source/x265cli.cpp option handling, source/abrEncApp.cpp:552-824 encode loop): x265_param_default_preset for
 * --preset / --tune, x265_param_parse for every other option, x265_encoder_open / headers / encode / close -- so the options, their names and their defaults are the
 * reference's, and `x265amd --preset medium --no-info --input clip.y4m -o out.hevc` writes the bytes `x265` writes for the same command line.
 *
 *     x265amd --input clip.y4m -o out.hevc [--preset|-p name] [--tune|-t name] [--recon|-r rec.yuv|rec.y4m] [--frames|-f N] [--csv log.csv] [any option x265_param_parse knows:
 *             --crf F --qp N --aq-mode N --aq-strength F --[no-]cutree --qcomp F --qg-size N --bframes N --b-adapt N --[no-]b-pyramid --[no-]open-gop --keyint N --min-keyint N
 *             --scenecut N --no-scenecut --rc-lookahead N --lookahead-slices N --ref N --limit-refs N --rd N --rdoq-level N --psy-rd F --psy-rdoq F --me name --subme N
 *             --merange N --max-merge N --[no-]rect --[no-]amp --[no-]limit-modes --[no-]early-skip --rskip N --[no-]weightp --[no-]weightb --[no-]sao --[no-]deblock --[no-]wpp
 *             --tu-intra-depth N --tu-inter-depth N --[no-]signhide --[no-]strong-intra-smoothing --[no-]temporal-mvp --[no-]b-intra --[no-]fast-intra --[no-]info
 *             --frame-threads N --pools S ...]
 *
 * Input: YUV4MPEG2, 4:2:0, 8 or 10 bits (source/input/y4m.cpp:150-330 header, :405-441 frames).  Output: the Annex-B stream (source/output/raw.cpp); the reconstruction in
 * display order as raw planar samples (source/output/yuv.cpp) or, for a name ending in .y4m, as YUV4MPEG2 (source/output/y4m.cpp: stream header once, "FRAME\n" before each
 * picture); --csv: the summary line of the reference's CSV log at its default level (x265_csvlog_open / x265_csvlog_encode through the table).
 * x265_picture / x265_nal are touched through the member offsets generated from the reference's header (host/x265_abi_layout.h: numbers only).  Host C++ only. */
#include "../host/x265_abi_layout.h"
#include "../host/x265_api_table.h"
#include <ctype.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

template<class T> T rd(const void* base, size_t off) { T v; memcpy(&v, (const char*)base + off, sizeof(T)); return v; }
template<class T> void wr(void* base, size_t off, T v) { memcpy((char*)base + off, &v, sizeof(T)); }

struct Nal { uint32_t type, sizeBytes; uint8_t* payload; };        /* x265_nal (x265.h:94-99) */
static_assert(sizeof(Nal) == X265ABI_SIZEOF_NAL, "x265_nal");

struct Api
{
    void* h = nullptr;
    const X265ApiTable* t = nullptr;
    const char* (*last_error)(void) = nullptr;
    void* param_alloc() const { return ((void* (*)(void))t->fn[X265API_PARAM_ALLOC])(); }
    void param_free(void* p) const { ((void (*)(void*))t->fn[X265API_PARAM_FREE])(p); }
    int param_default_preset(void* p, const char* preset, const char* tune) const { return ((int (*)(void*, const char*, const char*))t->fn[X265API_PARAM_DEFAULT_PRESET])(p, preset, tune); }
    int param_parse(void* p, const char* name, const char* value) const { return ((int (*)(void*, const char*, const char*))t->fn[X265API_PARAM_PARSE])(p, name, value); }
    void* picture_alloc() const { return ((void* (*)(void))t->fn[X265API_PICTURE_ALLOC])(); }
    void picture_free(void* p) const { ((void (*)(void*))t->fn[X265API_PICTURE_FREE])(p); }
    void picture_init(void* param, void* pic) const { ((void (*)(void*, void*))t->fn[X265API_PICTURE_INIT])(param, pic); }
    void* encoder_open(void* p) const { return ((void* (*)(void*))t->fn[X265API_ENCODER_OPEN])(p); }
    int encoder_headers(void* e, Nal** nal, uint32_t* n) const { return ((int (*)(void*, Nal**, uint32_t*))t->fn[X265API_ENCODER_HEADERS])(e, nal, n); }
    int encoder_encode(void* e, Nal** nal, uint32_t* n, void* in, void* out) const { return ((int (*)(void*, Nal**, uint32_t*, void*, void*))t->fn[X265API_ENCODER_ENCODE])(e, nal, n, in, out); }
    void encoder_get_stats(void* e, void* stats, uint32_t bytes) const { ((void (*)(void*, void*, uint32_t))t->fn[X265API_ENCODER_GET_STATS])(e, stats, bytes); }
    void encoder_close(void* e) const { ((void (*)(void*))t->fn[X265API_ENCODER_CLOSE])(e); }
    void cleanup() const { ((void (*)(void))t->fn[X265API_CLEANUP])(); }
    FILE* csvlog_open(const void* p) const { return ((FILE* (*)(const void*))t->fn2[4])(p); }
    void csvlog_encode(const void* p, const void* stats, int padx, int pady, int argc, char** argv) const { ((void (*)(const void*, const void*, int, int, int, char**))t->fn2[6])(p, stats, padx, pady, argc, argv); }
};

bool loadApi(Api& a, const std::string& dir, int depth)
{
    const std::string path = dir + (depth > 8 ? "/libx265amd_main10.so" : "/libx265amd_main.so");
    a.h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!a.h) { fprintf(stderr, "x265amd: cannot load %s: %s\n", path.c_str(), dlerror()); return false; }
    const void* (*get)(int) = (const void* (*)(int))dlsym(a.h, "x265_api_get_209");
    a.last_error = (const char* (*)(void))dlsym(a.h, "x265amd_last_error");
    a.t = get ? (const X265ApiTable*)get(depth) : nullptr;
    if (!a.t || !a.last_error) { fprintf(stderr, "x265amd: %s has no x265_api table for %d-bit samples\n", path.c_str(), depth); return false; }
    if (a.t->sizeof_picture != X265ABI_SIZEOF_PICTURE || a.t->sizeof_param != X265ABI_SIZEOF_PARAM) { fprintf(stderr, "x265amd: the library's x265_api is another build's\n"); return false; }
    return true;
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

bool endsWith(const std::string& s, const char* tail) { const size_t n = strlen(tail); return s.size() >= n && !s.compare(s.size() - n, n, tail); }

}

int main(int argc, char** argv)
{
    const char* input = nullptr; const char* output = nullptr; const char* recon = nullptr; const char* preset = nullptr; const char* tune = nullptr; const char* csv = nullptr;
    int frames = 0;
    std::vector<std::pair<std::string, const char*>> opts;         /* every other option, with the argument behind it if there is one */
    bool lastGeneric = false;
    for (int i = 1; i < argc; i++)
    {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "x265amd: %s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--input") input = val();
        else if (a == "-o" || a == "--output") output = val();
        else if (a == "--recon" || a == "-r") recon = val();
        else if (a == "--frames" || a == "-f") frames = atoi(val());
        else if (a == "--preset" || a == "-p") preset = val();
        else if (a == "--tune" || a == "-t") tune = val();
        else if (a == "--csv") csv = val();
        else if (a == "--input-depth" || a == "--output-depth" || a == "-D") (void)val();         /* the depth is the input file's and the library's */
        else if (a == "--no-progress" || a == "--progress") { }
        else if (a == "-F") opts.push_back({ "--frame-threads", val() });
        else if (a.rfind("--", 0) == 0) { opts.push_back({ a, nullptr }); lastGeneric = true; continue; }
        else if (lastGeneric && !opts.empty() && !opts.back().second && (a[0] != '-' || (a.size() > 1 && (isdigit((unsigned char)a[1]) || a[1] == '.'))))
            opts.back().second = argv[i];                                                          /* the argument of the option before it (a number may be negative) */
        else { fprintf(stderr, "x265amd: unknown argument %s\n", a.c_str()); return 2; }
        lastGeneric = false;
    }
    if (!input || !output) { fprintf(stderr, "usage: x265amd --input clip.y4m -o out.hevc [--preset name] [--tune name] [options]\n"); return 2; }
    Y4m y;
    if (!openY4m(y, input)) return 1;
    std::string dir = argv[0];
    const size_t slash = dir.find_last_of('/');
    dir = (slash == std::string::npos ? std::string(".") : dir.substr(0, slash)) + "/../lib";
    if (const char* e = getenv("X265AMD_LIBDIR")) dir = e;
    Api api;
    if (!loadApi(api, dir, y.depth)) return 1;

    void* p = api.param_alloc();
    if (!p || api.param_default_preset(p, preset, tune) < 0) { fprintf(stderr, "x265amd: preset %s / tune %s is not known\n", preset ? preset : "(default)", tune ? tune : "(none)"); return 2; }
    char text[64];
    snprintf(text, sizeof(text), "%dx%d", y.width, y.height);
    api.param_parse(p, "input-res", text);
    snprintf(text, sizeof(text), "%u/%u", y.fpsNum, y.fpsDen);
    api.param_parse(p, "fps", text);
    if (y.sarW == 1 && y.sarH == 1) api.param_parse(p, "sar", "1");            /* the y4m header's A tag (x265cli.cpp: setParamAspectRatio) */
    else if (y.sarW && y.sarH) { fprintf(stderr, "x265amd: sample aspect ratio %d:%d is not supported\n", y.sarW, y.sarH); return 1; }
    for (auto& o : opts)
    {
        const char* name = o.first.c_str() + 2;
        /* the argument behind an option, if there is one, is its value; an option that stands alone is a switch (x265_param_parse takes NULL as "true") */
        const int r = api.param_parse(p, name, o.second);
        if (r == -1) { fprintf(stderr, "x265amd: unknown option %s\n", o.first.c_str()); return 2; }
        if (r) { fprintf(stderr, "x265amd: bad value for %s: %s\n", o.first.c_str(), o.second ? o.second : "(none)"); return 2; }
    }
    void* enc = api.encoder_open(p);
    if (!enc) { fprintf(stderr, "x265amd: %s\n", api.last_error()); return 1; }
    FILE* out = fopen(output, "wb");
    FILE* rec = recon ? fopen(recon, "wb") : nullptr;
    if (!out || (recon && !rec)) { fprintf(stderr, "x265amd: cannot open the output\n"); return 1; }
    const bool recY4m = recon && endsWith(recon, ".y4m");
    size_t recHeader = 0;
    if (recY4m)
    {
        /* output/y4m.cpp: the stream header once; every picture is "FRAME\n" + its samples, written at its place in display order */
        recHeader = (size_t)fprintf(rec, "YUV4MPEG2 W%d H%d F%u:%u Ip C420%s\n", y.width, y.height, y.fpsNum, y.fpsDen, y.depth > 8 ? "p10" : "");
    }
    Nal* nal = nullptr; uint32_t nnal = 0;
    /* the parameter sets once in front -- unless the encoder repeats them with every keyframe (x265.cpp: the program asks x265_encoder_parameters what configure made of it) */
    ((void (*)(void*, void*))api.t->fn[X265API_ENCODER_PARAMETERS])(enc, p);
    if (!rd<int32_t>(p, X265ABI_PARAM_bRepeatHeaders))
    {
        if (api.encoder_headers(enc, &nal, &nnal) < 0) { fprintf(stderr, "x265amd: %s\n", api.last_error()); return 1; }
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);
    }

    const int isz = y.depth > 8 ? 2 : 1;
    std::vector<uint8_t> buf;
    void* picIn = api.picture_alloc();
    void* picOut = api.picture_alloc();
    api.picture_init(p, picIn); api.picture_init(p, picOut);
    int coded = 0, read = 0, rc = 0;
    auto emit = [&](int ret) {
        if (ret <= 0) return;
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);
        if (rec)
        {
            /* the reconstruction file is in display order (output/yuv.cpp, y4m.cpp write each picture at poc * frame size; x265_picture.poc counts from the first picture on) */
            const size_t place = (size_t)rd<int32_t>(picOut, X265ABI_PIC_poc), frameSize = y.frameBytes + (recY4m ? 6 : 0);
            fseek(rec, (long)(recHeader + place * frameSize), SEEK_SET);
            if (recY4m) fwrite("FRAME\n", 1, 6, rec);
            for (int k = 0; k < 3; k++)
            {
                const uint8_t* plane = rd<const uint8_t*>(picOut, X265ABI_PIC_planes + 8 * k);
                const int stride = rd<int32_t>(picOut, X265ABI_PIC_stride + 4 * k), w = (k ? y.width / 2 : y.width) * isz, h = k ? y.height / 2 : y.height;
                for (int r = 0; r < h; r++) fwrite(plane + (size_t)r * stride, 1, (size_t)w, rec);
            }
        }
        coded++;
    };
    while ((!frames || read < frames) && readFrame(y, buf))
    {
        uint8_t* planes[3] = { buf.data(), buf.data() + (size_t)y.width * y.height * isz, buf.data() + (size_t)y.width * y.height * isz * 5 / 4 };
        const int strides[3] = { y.width * isz, y.width / 2 * isz, y.width / 2 * isz };
        for (int k = 0; k < 3; k++) { wr<void*>(picIn, X265ABI_PIC_planes + 8 * k, planes[k]); wr<int32_t>(picIn, X265ABI_PIC_stride + 4 * k, strides[k]); }
        wr<int64_t>(picIn, X265ABI_PIC_pts, (int64_t)read);
        read++;
        rc = api.encoder_encode(enc, &nal, &nnal, picIn, rec ? picOut : nullptr);
        if (rc < 0) break;
        emit(rc);
    }
    while (rc >= 0 && (rc = api.encoder_encode(enc, &nal, &nnal, nullptr, rec ? picOut : nullptr)) > 0) emit(rc);
    if (rc < 0) fprintf(stderr, "x265amd: %s\n", api.last_error());
    if (csv && rc >= 0)
    {
        /* the reference's program: x265_csvlog_open(param) with param->csvfn set, then x265_csvlog_encode with the encoder's statistics (x265.cpp / abrEncApp.cpp) */
        wr<const char*>(p, X265ABI_PARAM_csvfn, csv);
        std::vector<uint8_t> stats((size_t)api.t->sizeof_stats, 0);
        api.encoder_get_stats(enc, stats.data(), (uint32_t)stats.size());
        FILE* f = api.csvlog_open(p);
        if (!f) fprintf(stderr, "x265amd: cannot open %s\n", csv);
        else
        {
            wr<FILE*>(p, X265ABI_PARAM_csvfpt, f);
            api.csvlog_encode(p, stats.data(), 0, 0, argc, argv);
            fclose(f);
        }
    }
    api.encoder_close(enc);
    api.picture_free(picIn); api.picture_free(picOut);
    api.param_free(p);
    api.cleanup();
    fclose(out);
    if (rec) fclose(rec);
    fprintf(stderr, "x265amd: encoded %d frames\n", coded);
    return rc < 0 ? 1 : 0;
}
