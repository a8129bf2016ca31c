/* Reference-plane production (include/x265amd.h: x265amd_extend_pic_border, x265amd_weight_plane).
 *
 *  - extendPicBorder (reference: source/common/pixel.cpp:1044-1058 with extendCURowColBorder, source/common/ipfilter.cpp:500-518):
 *    the margins of a padded plane repeat the nearest picture sample;
 *  - MotionReference::applyWeight over all rows (source/encoder/reference.cpp:109-185): the weighted copy of a reconstructed
 *    plane that motion estimation / compensation read when weighted prediction is on (weight_pp_c, pixel.cpp:519-538), with its
 *    margins.
 * HBM-bound: every output sample is one clamped read and one store, 16 samples per thread where alignment allows it is not
 * worth more here -- a 1080p plane is 2.6 MB, i.e. a few microseconds at HBM speed; launches are row-parallel and coalesced.
 */
#include "x265amd_dev.h"
#include "x265amd_host.h"

__global__ __launch_bounds__(256) void k_extend_border(pixel* pic, long stride, int width, int height, int marginX, int marginY, int firstRow)
{
    const int y = firstRow + (int)blockIdx.x;               /* padded row (-marginY .. height + marginY - 1) */
    const int cy = min(max(y, 0), height - 1);
    const pixel* srcRow = pic + (long)cy * stride;
    pixel* dstRow = pic + (long)y * stride;
    if (y >= 0 && y < height)
    {
        const pixel l = srcRow[0], r = srcRow[width - 1];
        for (int i = threadIdx.x; i < 2 * marginX; i += blockDim.x)
        {
            if (i < marginX) dstRow[-marginX + i] = l;
            else dstRow[width + (i - marginX)] = r;
        }
        return;
    }
    for (int x = -marginX + (int)threadIdx.x; x < width + marginX; x += blockDim.x)
        dstRow[x] = srcRow[min(max(x, 0), width - 1)];
}

/* margins of a band of picture lines whose samples become final column by column: left margin (doLeft) and right margin (doRight) beside lines y0 .. y1 - 1, and
 * when the band holds the last line the bottom margin below the sample columns xa .. xb - 1 (with the corner left of column 0 / right of the last column) */
__global__ __launch_bounds__(256) void k_extend_band(pixel* pic, long stride, int width, int height, int marginX, int marginY, int y0, int y1, int xa, int xb, int doLeft, int doRight)
{
    int y = y0 + (int)blockIdx.x;                           /* y0 .. y1 - 1: picture lines; then the bottom margin lines when y1 == height, then the top margin lines when y0 == 0 */
    const int bottomLines = y1 == height ? marginY : 0;
    if (y >= y1 + bottomLines) y = -1 - (y - y1 - bottomLines);         /* the top margin: lines -1 .. -marginY */
    if (y >= 0 && y < y1)
    {
        pixel* row = pic + (long)y * stride;
        const pixel l = row[0], r = row[width - 1];
        for (int i = threadIdx.x; i < marginX; i += blockDim.x)
        {
            if (doLeft) row[-marginX + i] = l;
            if (doRight) row[width + i] = r;
        }
        return;
    }
    const pixel* src = pic + (long)(y < 0 ? 0 : height - 1) * stride;
    pixel* dst = pic + (long)y * stride;
    const int from = (doLeft && xa == 0) ? -marginX : xa, to = (doRight && xb == width) ? width + marginX : xb;
    for (int x = from + (int)threadIdx.x; x < to; x += blockDim.x) dst[x] = src[min(max(x, 0), width - 1)];
}

/* the same for the three planes of a 4:2:0 picture in one launch (blockIdx.y: the plane; the chroma planes have half the lines, columns and margins) */
struct ExtendBand3 { pixel* pic[3]; long stride[3]; int width[3], height[3], marginX[3], marginY[3], y0[3], y1[3], xa[3], xb[3]; int doLeft, doRight; };
__global__ __launch_bounds__(256) void k_extend_band3(ExtendBand3 E)
{
    const int p = blockIdx.y;
    pixel* pic = E.pic[p];
    const long stride = E.stride[p];
    const int width = E.width[p], height = E.height[p], marginX = E.marginX[p], marginY = E.marginY[p], y0 = E.y0[p], y1 = E.y1[p], xa = E.xa[p], xb = E.xb[p];
    const int lines = (y1 - y0) + (y1 == height ? marginY : 0) + (y0 == 0 ? marginY : 0);
    if ((int)blockIdx.x >= lines) return;
    int y = y0 + (int)blockIdx.x;
    const int bottomLines = y1 == height ? marginY : 0;
    if (y >= y1 + bottomLines) y = -1 - (y - y1 - bottomLines);
    if (y >= 0 && y < y1)
    {
        pixel* row = pic + (long)y * stride;
        const pixel l = row[0], r = row[width - 1];
        for (int i = threadIdx.x; i < marginX; i += blockDim.x)
        {
            if (E.doLeft) row[-marginX + i] = l;
            if (E.doRight) row[width + i] = r;
        }
        return;
    }
    const pixel* src = pic + (long)(y < 0 ? 0 : height - 1) * stride;
    pixel* dst = pic + (long)y * stride;
    const int from = (E.doLeft && xa == 0) ? -marginX : xa, to = (E.doRight && xb == width) ? width + marginX : xb;
    for (int x = from + (int)threadIdx.x; x < to; x += blockDim.x) dst[x] = src[min(max(x, 0), width - 1)];
}

__global__ __launch_bounds__(256) void k_weight_plane(const pixel* src, pixel* dst, long stride, int width, int height, int marginX, int marginY,
                                                      int w0, int round, int shift, int offset)
{
    const int y = (int)blockIdx.x - marginY;
    const int cy = min(max(y, 0), height - 1);
    const pixel* srcRow = src + (long)cy * stride;
    pixel* dstRow = dst + (long)y * stride;
    const int correction = XA_IF_INTERNAL_PREC - XA_DEPTH;
    for (int x = -marginX + (int)threadIdx.x; x < width + marginX; x += blockDim.x)
    {
        const int v = (int)srcRow[min(max(x, 0), width - 1)] << correction;
        dstRow[x] = xa_clip_pixel(((w0 * v + round) >> shift) + offset);
    }
}

extern "C" int x265amd_extend_pic_border(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    return x265amd_extend_border_rows(stream, d_pic, stride, width, height, marginX, marginY, 0, height);
}

/* the margins beside picture lines y_begin .. y_end - 1, plus the top margin when the band holds line 0 and the bottom margin when it holds the last
 * line: PicYuv borders as FrameFilter::processPostRow extends them row by row (reference: source/encoder/framefilter.cpp:592-664) */
extern "C" int x265amd_extend_border_rows(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY, int y_begin, int y_end)
{
    if (!d_pic || width <= 0 || height <= 0 || marginX < 0 || marginY < 0 || y_begin < 0 || y_begin >= y_end || y_end > height)
        return xa_fail(X265AMD_EINVAL, "x265amd_extend_pic_border: bad arguments");
    const int first = y_begin == 0 ? -marginY : y_begin, last = y_end == height ? height + marginY : y_end;
    hipLaunchKernelGGL(k_extend_border, dim3(last - first), dim3(256), 0, (hipStream_t)stream, (pixel*)d_pic, (long)stride, width, height, marginX, marginY, first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

/* the margins of the picture lines y_begin .. y_end - 1 whose samples have become final in the columns x_begin .. x_end - 1 only: the left margin when left != 0,
 * the right margin when right != 0 (the caller says when the first / last sample of the lines is final), and, when the band holds the last picture line, the bottom
 * margin below those columns and with line 0 the top margin above them (with their corners under the same two conditions). */
extern "C" int x265amd_extend_border_band(void* stream, x265amd_pixel* d_pic, intptr_t stride, int width, int height, int marginX, int marginY, int y_begin, int y_end,
                                          int x_begin, int x_end, int left, int right)
{
    if (!d_pic || width <= 0 || height <= 0 || marginX < 0 || marginY < 0 || y_begin < 0 || y_begin >= y_end || y_end > height || x_begin < 0 || x_begin >= x_end || x_end > width)
        return xa_fail(X265AMD_EINVAL, "x265amd_extend_border_band: bad arguments");
    const int lines = (y_end - y_begin) + (y_end == height ? marginY : 0) + (y_begin == 0 ? marginY : 0);
    hipLaunchKernelGGL(k_extend_band, dim3(lines), dim3(256), 0, (hipStream_t)stream, (pixel*)d_pic, (long)stride, width, height, marginX, marginY, y_begin, y_end, x_begin, x_end,
                       left, right);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

/* x265amd_extend_border_band for the three planes of a 4:2:0 picture (luma geometry given; chroma: half of everything) in ONE launch: the filter thread of a picture
 * calls this per finished unit (csrc/encoder_api.hip: finishCols) */
int xa_extend_border_band_420(void* stream, x265amd_pixel* d_y, x265amd_pixel* d_u, x265amd_pixel* d_v, intptr_t stride, intptr_t cstride, int width, int height, int marginX, int marginY,
                              int y_begin, int y_end, int x_begin, int x_end, int left, int right)
{
    if (!d_y || !d_u || !d_v || width <= 0 || height <= 0 || marginX < 0 || marginY < 0 || y_begin < 0 || y_begin >= y_end || y_end > height || x_begin < 0 || x_begin >= x_end || x_end > width ||
        ((width | height | marginX | marginY | y_begin | y_end | x_begin | x_end) & 1))
        return xa_fail(X265AMD_EINVAL, "xa_extend_border_band_420: bad arguments");
    ExtendBand3 E;
    for (int p = 0; p < 3; p++)
    {
        const int sh = p ? 1 : 0;
        E.pic[p] = (pixel*)(p == 0 ? d_y : (p == 1 ? d_u : d_v)); E.stride[p] = (long)(p ? cstride : stride);
        E.width[p] = width >> sh; E.height[p] = height >> sh; E.marginX[p] = marginX >> sh; E.marginY[p] = marginY >> sh;
        E.y0[p] = y_begin >> sh; E.y1[p] = y_end >> sh; E.xa[p] = x_begin >> sh; E.xb[p] = x_end >> sh;
    }
    E.doLeft = left; E.doRight = right;
    const int lines = (y_end - y_begin) + (y_end == height ? marginY : 0) + (y_begin == 0 ? marginY : 0);
    hipLaunchKernelGGL(k_extend_band3, dim3(lines, 3), dim3(256), 0, (hipStream_t)stream, E);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}

extern "C" int x265amd_weight_plane(void* stream, const x265amd_pixel* d_src, x265amd_pixel* d_dst, intptr_t stride, int width, int height,
                                    int marginX, int marginY, int inputWeight, int inputOffset, int log2WeightDenom)
{
    if (!d_src || !d_dst || width <= 0 || height <= 0 || marginX < 0 || marginY < 0) return xa_fail(X265AMD_EINVAL, "x265amd_weight_plane: bad arguments");
    /* MotionReference::init (reference.cpp:98-101) and the call in applyWeight (:154-156) */
    const int correction = XA_IF_INTERNAL_PREC - XA_DEPTH;
    const int offset = inputOffset * (1 << (XA_DEPTH - 8));
    const int round = (log2WeightDenom ? 1 << (log2WeightDenom - 1) : 0) << correction;
    hipLaunchKernelGGL(k_weight_plane, dim3(height + 2 * marginY), dim3(256), 0, (hipStream_t)stream, (const pixel*)d_src, (pixel*)d_dst, (long)stride,
                       width, height, marginX, marginY, inputWeight, round, log2WeightDenom + correction, offset);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xa_fail(X265AMD_EHIP, hipGetErrorString(e));
    return X265AMD_OK;
}
