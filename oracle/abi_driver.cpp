/* TEST INFRASTRUCTURE ONLY (built by oracle/build_ref.sh into oracle/_ref/x265_abi_driver{8,10}).  A libx265 CLIENT: it fills an x265_param with the reference's
 * own x265_param_default_preset / x265_param_parse (linked from the reference's objects), then encodes a y4m clip through the `x265_api` table of ANOTHER library
 * -- libx265amd_main.so / libx265amd_main10.so, dlopen'ed, entered through x265_api_get_209 exactly as the reference's own multilib loader enters
 * libx265_main10.so (source/encoder/api.cpp:1107-1182).  tests/test_x265_api_abi.py compares the stream it writes with the reference encoder's.
 *   usage: x265_abi_driver <library.so> <in.y4m> <out.hevc> --preset <p> [--name [value]] ... */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "x265.h"

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s lib.so in.y4m out.hevc [options]\n", argv[0]); return 2; }
    const char* preset = "medium";
    for (int i = 4; i + 1 < argc; i++) if (!strcmp(argv[i], "--preset")) preset = argv[i + 1];
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, preset, NULL)) { fprintf(stderr, "preset\n"); return 2; }
    for (int i = 4; i < argc; i++)
    {
        if (strncmp(argv[i], "--", 2)) continue;
        const char* name = argv[i] + 2;
        if (!strcmp(name, "preset")) { i++; continue; }
        const char* value = (i + 1 < argc && strncmp(argv[i + 1], "--", 2)) ? argv[++i] : NULL;
        if (!strcmp(name, "no-info")) { p->bEmitInfoSEI = 0; continue; }
        if (x265_param_parse(p, name, value)) { fprintf(stderr, "option %s\n", name); return 2; }
    }
    FILE* in = fopen(argv[2], "rb");
    if (!in) { perror(argv[2]); return 2; }
    char hdr[256]; int n = 0, c;
    while ((c = fgetc(in)) != EOF && c != '\n' && n < 255) hdr[n++] = (char)c;
    hdr[n] = 0;
    int w = 0, h = 0, fn = 30, fd = 1, depth = 8;
    for (char* t = strtok(hdr, " "); t; t = strtok(NULL, " "))
    {
        if (t[0] == 'W') w = atoi(t + 1); else if (t[0] == 'H') h = atoi(t + 1);
        else if (t[0] == 'F') sscanf(t + 1, "%d:%d", &fn, &fd);
        else if (t[0] == 'C' && strstr(t, "p10")) depth = 10;
    }
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = fn; p->fpsDenom = fd; p->internalBitDepth = depth; p->internalCsp = X265_CSP_I420;
    p->vui.aspectRatioIdc = 1;      /* the y4m header's A1:1, as the reference's reader sets it (x265cli: sarWidth = sarHeight = 1 -> setParamAspectRatio) */

    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    typedef const x265_api* (*get_t)(int);
    get_t get = (get_t)dlsym(lib, "x265_api_get_209");
    typedef const x265_api* (*query_t)(int, int, int*);
    query_t query = (query_t)dlsym(lib, "x265_api_query");
    if (!get || !query) { fprintf(stderr, "the library does not export x265_api_get_209 / x265_api_query\n"); return 2; }
    int err = -1;
    const x265_api* api = query(depth, X265_BUILD, &err);
    if (!api || api != get(depth) || err) { fprintf(stderr, "no x265_api for %d bits (err %d)\n", depth, err); return 2; }
    if (api->api_major_version != X265_MAJOR_VERSION || api->api_build_number != X265_BUILD || api->sizeof_param != (int)sizeof(x265_param) ||
        api->sizeof_picture != (int)sizeof(x265_picture) || api->sizeof_stats != (int)sizeof(x265_stats) || api->sizeof_frame_stats != (int)sizeof(x265_frame_stats) ||
        api->bit_depth != depth)
    { fprintf(stderr, "x265_api table does not describe this build\n"); return 2; }
    fprintf(stderr, "x265_api of %s: %s %s\n", argv[1], api->version_str, api->build_info_str);
    x265_encoder* enc = api->encoder_open(p);
    if (!enc) { typedef const char* (*err_t)(void); err_t le = (err_t)dlsym(lib, "x265amd_last_error"); fprintf(stderr, "encoder_open failed: %s\n", le ? le() : "?"); return 3; }
    FILE* out = fopen(argv[3], "wb");
    x265_nal* nal = NULL; uint32_t nnal = 0;
    if (api->encoder_headers(enc, &nal, &nnal) < 0) { fprintf(stderr, "headers\n"); return 3; }
    for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);
    const size_t isz = depth > 8 ? 2 : 1, ysz = (size_t)w * h * isz, csz = ysz / 4;
    std::vector<unsigned char> buf(ysz + 2 * csz);
    x265_picture* pic = api->picture_alloc();
    x265_picture* rec = api->picture_alloc();
    int frames = 0, coded = 0;
    for (;;)
    {
        char tag[8];
        if (fread(tag, 1, 6, in) != 6 || memcmp(tag, "FRAME\n", 6)) break;
        if (fread(buf.data(), 1, buf.size(), in) != buf.size()) break;
        api->picture_init(p, pic);
        pic->planes[0] = buf.data(); pic->planes[1] = buf.data() + ysz; pic->planes[2] = buf.data() + ysz + csz;
        pic->stride[0] = (int)(w * isz); pic->stride[1] = pic->stride[2] = (int)(w / 2 * isz);
        pic->bitDepth = depth; pic->pts = 1000 + 40 * (int64_t)frames; frames++;
        const int r = api->encoder_encode(enc, &nal, &nnal, pic, rec);
        if (r < 0) { fprintf(stderr, "encode\n"); return 3; }
        if (r) { coded++; printf("pic %d pts %lld dts %lld\n", rec->poc, (long long)rec->pts, (long long)rec->dts); for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out); }
    }
    for (;;)
    {
        const int r = api->encoder_encode(enc, &nal, &nnal, NULL, rec);
        if (r < 0) { fprintf(stderr, "flush\n"); return 3; }
        if (!r) break;
        coded++;
        printf("pic %d pts %lld dts %lld\n", rec->poc, (long long)rec->pts, (long long)rec->dts);        /* what a muxing client reads: time stamps behind the B-frame reordering */
        for (uint32_t i = 0; i < nnal; i++) fwrite(nal[i].payload, 1, nal[i].sizeBytes, out);
    }
    fclose(out);
    api->encoder_close(enc);
    api->picture_free(pic); api->picture_free(rec);
    api->cleanup();
    x265_param_free(p);
    fprintf(stderr, "x265_abi_driver: %d frames in, %d coded\n", frames, coded);
    return coded == frames ? 0 : 4;
}
