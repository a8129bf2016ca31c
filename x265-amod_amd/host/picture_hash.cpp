/* The decoded picture hash SEI's three digests of a reconstructed picture (--hash 1 / 2 / 3; reference: FrameEncoder::computePictureHash-like code in
 * source/encoder/frameencoder.cpp:1228-1296, source/common/picyuv.cpp:559-641, source/common/md5.cpp; H.265 D.3.19): MD5 of each plane's samples in raster order (two bytes
 * per sample, low byte first, above 8 bits), the 16-bit CRC with polynomial 0x1021, the 32-bit checksum with the position mask.  Host code; MD5 is RFC 1321 written out. */
#include "../../include/x265amd.h"
#include <stdint.h>
#include <string.h>

namespace {

struct Md5
{
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t bytes = 0;
    uint8_t buf[64];
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p)
    {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
            0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
            0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
            0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
        static const uint8_t S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                       4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; i++)
        {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B;
            B = B + rol(A + f + K[i] + m[g], S[i]);
            A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n)
    {
        size_t fill = (size_t)(bytes & 63);
        bytes += n;
        if (fill)
        {
            const size_t take = n < 64 - fill ? n : 64 - fill;
            memcpy(buf + fill, p, take); p += take; n -= take; fill += take;
            if (fill < 64) return;
            block(buf);
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) memcpy(buf, p, n);
    }
    void final(uint8_t out[16])
    {
        const uint64_t bits = bytes * 8;
        static const uint8_t pad[64] = { 0x80 };
        const size_t fill = (size_t)(bytes & 63);
        update(pad, fill < 56 ? 56 - fill : 120 - fill);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
        update(len, 8);
        const uint32_t v[4] = { a, b, c, d };
        for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(v[i] >> (8 * k));
    }
};

template<class PIX> uint32_t sampleAt(const void* plane, intptr_t strideBytes, int x, int y) { return ((const PIX*)((const uint8_t*)plane + (intptr_t)y * strideBytes))[x]; }

}

/* method 1 MD5 / 2 CRC / 3 checksum of the three planes (4:2:0; host pointers, strides in bytes; depth 8: one byte per sample, else two).  payload: the SEI message's payload --
 * hash_type (method - 1), then per plane 16 / 2 / 4 bytes; returns its size, 0 on bad arguments.
 * CRC: the reference starts the two chroma planes' value again with EVERY CTU row (frameencoder.cpp:1264: the reset is not under `if (!row)`), so what it sends for them covers the
 * last CTU row's chroma lines alone; restated as it is (ctu_size: 64). */
extern "C" size_t x265amd_picture_hash(int method, const void* const planes[3], const intptr_t strides[3], int width, int height, int depth, int ctu_size, uint8_t* payload, size_t cap)
{
    if (method < 1 || method > 3 || !planes || !strides || !payload || width <= 0 || height <= 0 || (depth != 8 && (depth < 9 || depth > 16)) || ctu_size <= 0) return 0;
    const size_t per = method == 1 ? 16 : method == 2 ? 2 : 4;
    if (cap < 1 + 3 * per) return 0;
    payload[0] = (uint8_t)(method - 1);
    const bool wide = depth > 8;
    for (int p = 0; p < 3; p++)
    {
        const int w = p ? width >> 1 : width, h = p ? height >> 1 : height;
        uint8_t* out = payload + 1 + per * p;
        auto at = [&](int x, int y) -> uint32_t { return wide ? sampleAt<uint16_t>(planes[p], strides[p], x, y) : sampleAt<uint8_t>(planes[p], strides[p], x, y); };
        if (method == 1)
        {
            Md5 md;
            uint8_t row[2 * 8192];
            for (int y = 0; y < h; y++)
            {
                if (!wide) md.update((const uint8_t*)planes[p] + (intptr_t)y * strides[p], (size_t)w);
                else
                {
                    for (int x = 0; x < w; x++) { const uint32_t v = at(x, y); row[2 * x] = (uint8_t)v; row[2 * x + 1] = (uint8_t)(v >> 8); }
                    md.update(row, (size_t)w * 2);
                }
            }
            md.final(out);
        }
        else if (method == 2)
        {
            uint32_t crc = 0xffff;
            const int rows = p ? ctu_size >> 1 : ctu_size;
            const int y0 = p ? ((h - 1) / rows) * rows : 0;            /* chroma: the last CTU row only (see above) */
            for (int y = y0; y < h; y++)
                for (int x = 0; x < w; x++)
                {
                    const uint32_t v = at(x, y);
                    for (int bit = 0; bit < 8; bit++) { const uint32_t msb = (crc >> 15) & 1, b = (v >> (7 - bit)) & 1; crc = (((crc << 1) + b) & 0xffff) ^ (msb * 0x1021); }
                    if (wide) for (int bit = 0; bit < 8; bit++) { const uint32_t msb = (crc >> 15) & 1, b = (v >> (15 - bit)) & 1; crc = (((crc << 1) + b) & 0xffff) ^ (msb * 0x1021); }
                }
            for (int bit = 0; bit < 16; bit++) { const uint32_t msb = (crc >> 15) & 1; crc = ((crc << 1) & 0xffff) ^ (msb * 0x1021); }
            out[0] = (uint8_t)(crc >> 8); out[1] = (uint8_t)crc;
        }
        else
        {
            uint32_t sum = 0;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++)
                {
                    const uint8_t mask = (uint8_t)((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8));
                    const uint32_t v = at(x, y);
                    sum += (v & 0xff) ^ mask;
                    if (wide) sum += (v >> 8) ^ mask;
                }
            out[0] = (uint8_t)(sum >> 24); out[1] = (uint8_t)(sum >> 16); out[2] = (uint8_t)(sum >> 8); out[3] = (uint8_t)sum;
        }
    }
    return 1 + 3 * per;
}
