// Host-side codec loops of the on-disk formats around the hot path: PNG scanline un-filtering (the index texture
// "0.png" is a 16-bit PNG written by libpng with adaptive filters, tracer_o3d_irt.py:91) and Radiance new-style RLE scanlines (the
// per-view ccm.hdr / hdr_texture.hdr files, datasets/dataset.py:480, tracer_o3d_irt.py:77).  Both are byte-serial recurrences, i.e.
// 10^8 interpreter iterations per file if left in Python; here they are plain C++ behind two C-ABI entry points (host pointers).
#include <cstdint>
#include <cstdlib>

#include "../../include/texir_hip.h"

extern "C" {

// raw [H][stride+1] (filter byte + filtered bytes) -> out [H][stride]; bpp = bytes per complete pixel (PNG spec 9.2)
int texir_png_unfilter(const uint8_t* raw, int32_t H, int32_t stride, int32_t bpp, uint8_t* out)
{
    if (!raw || !out || H < 0 || stride < 0 || bpp < 1) return TEXIR_ERR_INVALID;
    for (int y = 0; y < H; y++) {
        const uint8_t* in = raw + (size_t)y * (stride + 1);
        const int ft = in[0];
        in++;
        uint8_t* cur = out + (size_t)y * stride;
        const uint8_t* prev = y ? cur - stride : nullptr;
        if (ft > 4) return TEXIR_ERR_INVALID;
        for (int x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0, c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            cur[x] = (uint8_t)(in[x] + pred);
        }
    }
    return TEXIR_OK;
}

// Radiance scanlines after the resolution line: data [n] -> rgbe [H][W][4].  Handles flat and new-style RLE scanlines (per scanline,
// as the format allows).  Returns the number of bytes consumed, or a negative error code.
int64_t texir_hdr_decode_scanlines(const uint8_t* data, int64_t n, int32_t W, int32_t H, uint8_t* rgbe)
{
    if (!data || !rgbe || W <= 0 || H <= 0) return TEXIR_ERR_INVALID;
    int64_t p = 0;
    for (int y = 0; y < H; y++) {
        uint8_t* row = rgbe + (size_t)y * W * 4;
        if (p + 4 > n) return TEXIR_ERR_INVALID;
        if (W < 8 || W >= 32768 || data[p] != 2 || data[p + 1] != 2 || (data[p + 2] & 0x80)) {
            if (p + 4ll * W > n) return TEXIR_ERR_INVALID;
            for (int64_t i = 0; i < 4ll * W; i++) row[i] = data[p + i];
            p += 4ll * W;
            continue;
        }
        if (((int)data[p + 2] << 8 | (int)data[p + 3]) != W) return TEXIR_ERR_INVALID;
        p += 4;
        for (int c = 0; c < 4; c++) {
            int x = 0;
            while (x < W) {
                if (p >= n) return TEXIR_ERR_INVALID;
                int k = data[p++];
                if (k > 128) {
                    k -= 128;
                    if (x + k > W || p >= n) return TEXIR_ERR_INVALID;
                    const uint8_t v = data[p++];
                    for (int i = 0; i < k; i++) row[4 * (x + i) + c] = v;
                } else {
                    if (k == 0 || x + k > W || p + k > n) return TEXIR_ERR_INVALID;
                    for (int i = 0; i < k; i++) row[4 * (x + i) + c] = data[p + i];
                    p += k;
                }
                x += k;
            }
        }
    }
    return p;
}

}  // extern "C"
