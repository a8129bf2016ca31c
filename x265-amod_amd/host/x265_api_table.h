/* The layout of the reference's `struct x265_api` (source/x265.h:2561-2614) as this library fills it and as a client that has no x265.h reads it (cli/x265amd_cli.cpp):
 * member for member -- 3 + 4 ints, the bit depth, two strings, 20 function pointers, sizeof_frame_stats, 9 function pointers, zone_param_parse (ENABLE_LIBVMAF is off in the
 * reference build this library stands in for).  host/x265_api_abi.cpp pins it against the offsets generated from the reference's header (x265_abi_layout.h). */
#ifndef X265AMD_X265_API_TABLE_H
#define X265AMD_X265_API_TABLE_H
struct X265ApiTable
{
    int api_major_version, api_build_number, sizeof_param, sizeof_picture, sizeof_analysis_data, sizeof_zone, sizeof_stats;
    int bit_depth;
    const char* version_str; const char* build_info_str;
    void* fn[20];
    int sizeof_frame_stats;
    void* fn2[9];
    void* zone_param_parse;
};
/* fn[]: the entry points in the order of the reference's struct */
enum
{
    X265API_PARAM_ALLOC = 0, X265API_PARAM_FREE, X265API_PARAM_DEFAULT, X265API_PARAM_PARSE, X265API_SCENECUT_AWARE_QP_PARAM_PARSE, X265API_PARAM_APPLY_PROFILE,
    X265API_PARAM_DEFAULT_PRESET, X265API_PICTURE_ALLOC, X265API_PICTURE_FREE, X265API_PICTURE_INIT, X265API_ENCODER_OPEN, X265API_ENCODER_PARAMETERS,
    X265API_ENCODER_RECONFIG, X265API_ENCODER_RECONFIG_ZONE, X265API_ENCODER_HEADERS, X265API_ENCODER_ENCODE, X265API_ENCODER_GET_STATS, X265API_ENCODER_LOG,
    X265API_ENCODER_CLOSE, X265API_CLEANUP
};
#endif
