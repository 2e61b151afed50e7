// Host-side PNG decoder for the input pipeline (`mmcv.imread` of the Cityscapes-VPS / VIPER frames in
// datasets/pipelines/loading.py:43-68). Why native: Python decoders hold the interpreter lock while they inflate, and beside a main
// thread that launches ~560 kernels per frame a pool of PIL decode threads delivers 6 frames/s (30 with an idle main thread; forked
// decode processes stall on the fork of a process with 30 GB of device mappings: 1.4 frames/s measured). A C-ABI call made through
// ctypes runs WITHOUT the lock, so plain threads scale. Scope: what the datasets of the path contain - 8-bit, non-interlaced, colour
// type 2 (RGB), 6 (RGBA, alpha dropped like cv2.IMREAD_COLOR) or 0 (grey, replicated); anything else returns VPS_EARG and the caller
// falls back to its general decoder. Output: BGR uint8 [H][W][3] (cv2 / mmcv channel order). zlib does the inflate.
#include <zlib.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/vps_hip.h"

#define VPS_EARG(x) (-1000 - (x))

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }


// the scanline buffer of a decoder thread, kept between calls (round 6): a fresh 6 MB malloc per frame is an mmap + 1500 page faults +
// an munmap, and the decoder threads of a process queue on the address-space lock for them (8 threads decoded no faster than 6)
struct RawBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~RawBuf() { free(p); }
    uint8_t* get(size_t n) {
        if (n > cap) {
            free(p);
            p = (uint8_t*)malloc(n);
            cap = p ? n : 0;
        }
        return p;
    }
};
thread_local RawBuf t_raw;

// Paeth / Average rows of a C-byte pixel, C compile-time: the C channel chains are independent (instruction-level parallelism across
// them; the chain along the row is inherent to the filter), predictor without the branchy three-way comparison
template <int C>
inline void unfilter_paeth(uint8_t* cur, const uint8_t* up, size_t stride) {
    int a[C], c[C];
    for (int k = 0; k < C; ++k) { a[k] = 0; c[k] = 0; }
    for (size_t i = 0; i < stride; i += C) {
        for (int k = 0; k < C; ++k) {
            const int b = up[i + k];
            const int p = b - c[k], pc0 = a[k] - c[k];
            const int pa = abs(p), pb = abs(pc0), pc = abs(p + pc0);
            int pred = pb < pa ? b : a[k];
            const int pm = pb < pa ? pb : pa;
            pred = pc < pm ? c[k] : pred;
            const int v = (cur[i + k] + pred) & 255;
            cur[i + k] = (uint8_t)v;
            a[k] = v;
            c[k] = b;
        }
    }
}
template <int C>
inline void unfilter_avg(uint8_t* cur, const uint8_t* up, size_t stride) {
    int a[C];
    for (int k = 0; k < C; ++k) a[k] = 0;
    for (size_t i = 0; i < stride; i += C)
        for (int k = 0; k < C; ++k) {
            const int v = (cur[i + k] + ((a[k] + (up ? up[i + k] : 0)) >> 1)) & 255;
            cur[i + k] = (uint8_t)v;
            a[k] = v;
        }
}
template <int C>
inline void unfilter_sub(uint8_t* cur, size_t stride) {
    int a[C];
    for (int k = 0; k < C; ++k) a[k] = 0;
    for (size_t i = 0; i < stride; i += C)
        for (int k = 0; k < C; ++k) {
            const int v = (cur[i + k] + a[k]) & 255;
            cur[i + k] = (uint8_t)v;
            a[k] = v;
        }
}

}  // namespace

extern "C" int vps_png_info(const uint8_t* file, int64_t nbytes, int32_t* H, int32_t* W, int32_t* channels) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (!file || nbytes < 33 || memcmp(file, sig, 8) != 0 || be32(file + 8) != 13 || memcmp(file + 12, "IHDR", 4) != 0) return VPS_EARG(1);
    // the IHDR chunk's CRC (type + 13 data bytes): a damaged header is refused before its fields size anything
    if ((uint32_t)crc32(0L, file + 12, 17) != be32(file + 29)) return VPS_EARG(1);
    const uint32_t w = be32(file + 16), h = be32(file + 20);
    const int depth = file[24], ctype = file[25], interlace = file[28];
    if (w == 0 || h == 0 || w > 65535 || h > 65535) return VPS_EARG(2);
    if (depth != 8 || interlace != 0 || !(ctype == 0 || ctype == 2 || ctype == 6)) return VPS_EARG(3);      // caller falls back
    if (H) *H = (int32_t)h;
    if (W) *W = (int32_t)w;
    if (channels) *channels = ctype == 0 ? 1 : (ctype == 2 ? 3 : 4);
    return 0;
}

extern "C" int vps_png_decode_bgr8(const uint8_t* file, int64_t nbytes, uint8_t* out, int64_t out_capacity) {
    int32_t H, W, C;
    const int st = vps_png_info(file, nbytes, &H, &W, &C);
    if (st) return st;
    if (!out || out_capacity < (int64_t)H * W * 3) return VPS_EARG(4);
    const size_t stride = (size_t)W * C;
    // zlib counts the output space in a 32-bit uInt: an image whose raw size does not fit (a 65535 x 65535 header passes the checks
    // above: ~17 GB) would be inflated only in part, with `avail_out == 0` reached early and uninitialised rows behind it.
    // Refused: the caller falls back to the general decoder. 1 GiB is > 100 frames of the path's size.
    if ((stride + 1) * (size_t)H > ((size_t)1 << 30)) return VPS_EARG(9);
    uint8_t* raw = t_raw.get((stride + 1) * (size_t)H);                   // filter byte + pixels per scanline (the thread's buffer)
    if (!raw) return VPS_EARG(5);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return VPS_EARG(6);
    zs.next_out = raw;
    zs.avail_out = (uInt)((stride + 1) * (size_t)H);
    int64_t pos = 8;
    int zret = Z_OK;
    bool end = false;
    while (pos + 12 <= nbytes && !end) {
        const uint32_t len = be32(file + pos);
        const uint8_t* type = file + pos + 4;
        if (pos + 12 + (int64_t)len > nbytes) break;
        if (memcmp(type, "IDAT", 4) == 0 && zret == Z_OK) {
            zs.next_in = const_cast<Bytef*>(file + pos + 8);
            zs.avail_in = len;
            zret = inflate(&zs, Z_NO_FLUSH);
            if (zret != Z_OK && zret != Z_STREAM_END) break;
        } else if (memcmp(type, "IEND", 4) == 0) {
            end = true;
        }
        pos += 12 + (int64_t)len;
    }
    const bool complete = zs.avail_out == 0 && (zret == Z_STREAM_END || zret == Z_OK);
    inflateEnd(&zs);
    if (!complete) return VPS_EARG(7);
    // un-filter in place (PNG specification 9.2: None, Sub, Up, Average, Paeth; bytes of the pixel to the left = C bytes back)
    for (int y = 0; y < H; ++y) {
        uint8_t* cur = raw + (size_t)y * (stride + 1) + 1;
        const uint8_t* up = y ? cur - (stride + 1) : nullptr;
        const int ft = cur[-1];
        switch (ft) {
            case 0: break;
            case 1:
                if (C == 3) unfilter_sub<3>(cur, stride); else if (C == 4) unfilter_sub<4>(cur, stride); else unfilter_sub<1>(cur, stride);
                break;
            case 2:
                if (up) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);
                break;
            case 3:
                if (C == 3) unfilter_avg<3>(cur, up, stride); else if (C == 4) unfilter_avg<4>(cur, up, stride); else unfilter_avg<1>(cur, up, stride);
                break;
            case 4:
                if (!up) {                                   // first row: Paeth degenerates to Sub (b = c = 0 -> predictor a)
                    if (C == 3) unfilter_sub<3>(cur, stride); else if (C == 4) unfilter_sub<4>(cur, stride); else unfilter_sub<1>(cur, stride);
                } else if (C == 3) unfilter_paeth<3>(cur, up, stride);
                else if (C == 4) unfilter_paeth<4>(cur, up, stride);
                else unfilter_paeth<1>(cur, up, stride);
                break;
            default:
                return VPS_EARG(8);
        }
        uint8_t* o = out + (size_t)y * W * 3;
        if (C == 1) {
            for (int x = 0; x < W; ++x) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = cur[x]; }
        } else {
            for (int x = 0; x < W; ++x) { o[3 * x] = cur[(size_t)C * x + 2]; o[3 * x + 1] = cur[(size_t)C * x + 1]; o[3 * x + 2] = cur[(size_t)C * x]; }
        }
    }
    return 0;
}
