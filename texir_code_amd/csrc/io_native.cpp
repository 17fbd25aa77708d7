// Host-side codec loops of the on-disk formats around the hot path: PNG scanline un-filtering (the index texture
// "0.png" is a 16-bit PNG written by libpng with adaptive filters, tracer_o3d_irt.py:91) and Radiance new-style RLE scanlines (the
// per-view ccm.hdr / hdr_texture.hdr files, datasets/dataset.py:480, tracer_o3d_irt.py:77).  Both are byte-serial recurrences, i.e.
// 10^8 interpreter iterations per file if left in Python; here they are plain C++ behind two C-ABI entry points (host pointers).
#include <locale.h>
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

// ---------------------------------------------------------------------------------------------------------------------------------------
// Wavefront OBJ text -> arrays and the Radiance RGBE pixel codec (round 5).  Both were per-element Python / boolean-mask numpy in
// io_formats.py and cost 10x the kernels of an IrrT run at c4 size (17 s per 1 M-triangle OBJ, 6 s per 4096^2 .hdr write); here they
// are plain C++ over host pointers, split across the host threads the process may use.  The Python forms stay in io_formats.py as the
// test reference (tests/test_host_cpu.py demands identical arrays / bytes).
// ---------------------------------------------------------------------------------------------------------------------------------------
#include <sched.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

int host_threads()
{
    static int n = [] {
        int k = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) k = std::min(k > 0 ? k : 1, CPU_COUNT(&set));
        if (const char* e = getenv("TEXIR_IO_THREADS")) k = atoi(e);
        return std::max(1, std::min(k, 64));
    }();
    return n;
}

template <class F>
void parallel_chunks(int64_t n, int64_t min_chunk, F&& fn)
{
    int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / std::max<int64_t>(1, min_chunk)));
    if (nt == 1) { fn(0, (int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&, t] { fn(t, n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }
inline bool is_eol(char c) { return c == '\n' || c == '\r'; }

// one whitespace-separated token of the current line; false at the end of the line
inline bool next_token(const char*& p, const char* end, const char*& tb, const char*& te)
{
    while (p < end && is_ws(*p)) p++;
    if (p >= end || is_eol(*p)) return false;
    tb = p;
    while (p < end && !is_ws(*p) && !is_eol(*p)) p++;
    te = p;
    return true;
}

// Python's float(token) followed by numpy's float64 -> float32 rounding: correctly rounded double, then one more rounding.
// Fast path (Clinger): a decimal mantissa below 2^53 with |exponent| <= 22 is one exact double times / over one exact power of ten,
// i.e. correctly rounded by a single IEEE operation; everything else (long mantissas, huge exponents, inf / nan) goes to strtod.
// (libstdc++ 11's std::from_chars<double> wraps strtod under a locale switch and is several times slower than this.)
inline bool parse_f32(const char* b, const char* e, float& out)
{
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* p = b;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; p++; }
    uint64_t m = 0;
    int nd = 0, exp10 = 0;
    bool any = false, ok = true;
    while (p < e && *p >= '0' && *p <= '9') { if (nd < 19) { m = m * 10 + (uint64_t)(*p - '0'); nd += (m != 0); } else exp10++; p++; any = true; }
    if (p < e && *p == '.') {
        p++;
        while (p < e && *p >= '0' && *p <= '9') { if (nd < 19) { m = m * 10 + (uint64_t)(*p - '0'); nd += (m != 0); exp10--; } p++; any = true; }
    }
    if (!any) ok = false;
    if (ok && p < e && (*p == 'e' || *p == 'E')) {
        p++;
        bool eneg = false;
        if (p < e && (*p == '+' || *p == '-')) { eneg = *p == '-'; p++; }
        if (p >= e) ok = false;
        int ex = 0;
        while (p < e && *p >= '0' && *p <= '9') { if (ex < 100000) ex = ex * 10 + (*p - '0'); p++; }
        exp10 += eneg ? -ex : ex;
    }
    if (ok && p == e && nd < 19 && m < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
        double d = (double)m;
        d = exp10 < 0 ? d / P10[-exp10] : d * P10[exp10];
        out = (float)(neg ? -d : d);
        return true;
    }
    // slow path: exactly what float() accepts and returns for the remaining spellings that matter in OBJ files
    char tmp[128];
    const size_t n = (size_t)(e - b);
    if (n == 0 || n >= sizeof(tmp)) return false;
    memcpy(tmp, b, n);
    tmp[n] = 0;
    // strtod in the "C" locale whatever LC_NUMERIC the embedding application has set (a comma-decimal locale would otherwise reject every vertex that
    // reaches this path), and without the spellings strtod takes but Python's float() refuses: hexadecimal floats
    for (size_t i = 0; i < n; i++) if (tmp[i] == 'x' || tmp[i] == 'X') return false;
    static const locale_t c_loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    char* endp = nullptr;
    const double d = c_loc ? strtod_l(tmp, &endp, c_loc) : strtod(tmp, &endp);
    if (endp != tmp + n) return false;
    out = (float)d;
    return true;
}

inline bool parse_int(const char* b, const char* e, long long& out)
{
    if (b < e && *b == '+') b++;
    auto r = std::from_chars(b, e, out);
    return r.ec == std::errc() && r.ptr == e;
}

struct ObjChunk {
    std::vector<float> v, vt, vn;
    std::vector<long long> fi, ft, fn;          // absolute (>= 0), -1 = absent, or relative entries listed in rel_*
    std::vector<uint32_t> rel_i, rel_t, rel_n;  // positions in fi/ft/fn that hold (local count + negative index): add the chunk's base
    int err = 0;
    long long err_line_off = 0;
};

struct ObjParse {
    std::vector<ObjChunk> chunks;
    int64_t n_v = 0, n_vt = 0, n_vn = 0, n_tri = 0;
};

void parse_chunk(const char* text, int64_t b, int64_t e, ObjChunk& c)
{
    const char* p = text + b;
    const char* end = text + e;
    std::vector<long long> ci, ct, cn;          // the corners of one face
    std::vector<char> ri, rt, rn;
    while (p < end) {
        const char* line = p;
        const char *tb, *te;
        if (next_token(p, end, tb, te)) {
            const size_t kl = te - tb;
            if (kl == 1 && tb[0] == 'v') {
                float x[3];
                for (int k = 0; k < 3; k++)
                    if (!next_token(p, end, tb, te) || !parse_f32(tb, te, x[k])) { c.err = 1; c.err_line_off = line - text; return; }
                c.v.insert(c.v.end(), x, x + 3);
            } else if (kl == 2 && tb[0] == 'v' && tb[1] == 't') {
                float x[2];
                for (int k = 0; k < 2; k++)
                    if (!next_token(p, end, tb, te) || !parse_f32(tb, te, x[k])) { c.err = 1; c.err_line_off = line - text; return; }
                c.vt.insert(c.vt.end(), x, x + 2);
            } else if (kl == 2 && tb[0] == 'v' && tb[1] == 'n') {
                float x[3];
                for (int k = 0; k < 3; k++)
                    if (!next_token(p, end, tb, te) || !parse_f32(tb, te, x[k])) { c.err = 1; c.err_line_off = line - text; return; }
                c.vn.insert(c.vn.end(), x, x + 3);
            } else if (kl == 1 && tb[0] == 'f') {
                ci.clear(); ct.clear(); cn.clear(); ri.clear(); rt.clear(); rn.clear();
                while (next_token(p, end, tb, te)) {
                    // a[/b[/c]] with empty b allowed ("a//c")
                    const char* s1 = (const char*)memchr(tb, '/', te - tb);
                    const char* s2 = s1 ? (const char*)memchr(s1 + 1, '/', te - (s1 + 1)) : nullptr;
                    const char* ae = s1 ? s1 : te;
                    long long a, bb = -1, cc = -1;
                    char ra = 0, rb = 0, rc = 0;
                    if (!parse_int(tb, ae, a)) { c.err = 2; c.err_line_off = line - text; return; }
                    if (a > 0) a -= 1; else { a += (long long)(c.v.size() / 3); ra = 1; }
                    if (s1) {
                        const char* be = s2 ? s2 : te;
                        if (be > s1 + 1) {
                            if (!parse_int(s1 + 1, be, bb)) { c.err = 2; c.err_line_off = line - text; return; }
                            if (bb > 0) bb -= 1; else { bb += (long long)(c.vt.size() / 2); rb = 1; }
                        }
                        if (s2 && te > s2 + 1) {
                            const char* s3 = (const char*)memchr(s2 + 1, '/', te - (s2 + 1));     // anything past a third slash is ignored, like parts[2]
                            if (!parse_int(s2 + 1, s3 ? s3 : te, cc)) { c.err = 2; c.err_line_off = line - text; return; }
                            if (cc > 0) cc -= 1; else { cc += (long long)(c.vn.size() / 3); rc = 1; }
                        }
                    }
                    ci.push_back(a); ct.push_back(bb); cn.push_back(cc); ri.push_back(ra); rt.push_back(rb); rn.push_back(rc);
                }
                for (size_t k = 1; k + 1 < ci.size(); k++) {                                         // fan triangulation
                    const size_t idx[3] = {0, k, k + 1};
                    for (size_t j : idx) {
                        if (ri[j]) c.rel_i.push_back((uint32_t)c.fi.size());
                        if (rt[j]) c.rel_t.push_back((uint32_t)c.ft.size());
                        if (rn[j]) c.rel_n.push_back((uint32_t)c.fn.size());
                        c.fi.push_back(ci[j]); c.ft.push_back(ct[j]); c.fn.push_back(cn[j]);
                    }
                }
            }
        }
        while (p < end && !is_eol(*p)) p++;      // rest of the line (comments, further tokens)
        while (p < end && is_eol(*p)) p++;
    }
}

}  // namespace

extern "C" {

// OBJ text [n] -> handle + counts {n_v, n_vt, n_vn, n_tri}.  Lines are classified by their first whitespace-separated token (v / vt / vn / f;
// everything else is skipped), LF, CRLF and CR line ends alike; faces of more than three corners are fan-triangulated; negative indices
// count back from the elements read so far; "a", "a/b", "a//c", "a/b/c" corners (absent = -1).  Replaces the parse half of
// o3d.io.read_triangle_mesh / pyredner.load_obj (models/tracer_o3d_irt.py:75,85,183-189; models/mat_nvdiffrast.py:87,193-199).
int texir_obj_parse(const char* text, int64_t n, void** handle, int64_t counts[4])
{
    if (!text || n < 0 || !handle || !counts) return TEXIR_ERR_INVALID;
    auto* P = new ObjParse();
    // chunk boundaries moved forward to line starts
    int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / (1 << 20)));
    std::vector<int64_t> cut(nt + 1);
    cut[0] = 0; cut[nt] = n;
    for (int t = 1; t < nt; t++) {
        int64_t q = n * t / nt;
        while (q < n && !is_eol(text[q])) q++;
        while (q < n && is_eol(text[q])) q++;
        cut[t] = std::max(q, cut[t - 1]);
    }
    P->chunks.resize(nt);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { parse_chunk(text, cut[t], cut[t + 1], P->chunks[t]); });
        for (auto& x : th) x.join();
    }
    for (auto& c : P->chunks) {
        if (c.err) { delete P; return TEXIR_ERR_INVALID; }
        P->n_v += c.v.size() / 3; P->n_vt += c.vt.size() / 2; P->n_vn += c.vn.size() / 3; P->n_tri += c.fi.size() / 3;
    }
    counts[0] = P->n_v; counts[1] = P->n_vt; counts[2] = P->n_vn; counts[3] = P->n_tri;
    *handle = P;
    return TEXIR_OK;
}

// copies the parsed arrays into caller-owned host buffers sized from the counts (vn / fn may be null when n_vn == 0) and frees the handle;
// with every output null it only frees.  v [n_v][3], vt [n_vt][2], vn [n_vn][3] float32; fi / ft / fn [n_tri][3] int32 (0-based, -1 = absent).
int texir_obj_take(void* handle, float* v, float* vt, float* vn, int32_t* fi, int32_t* ft, int32_t* fn)
{
    if (!handle) return TEXIR_ERR_INVALID;
    auto* P = (ObjParse*)handle;
    int64_t bv = 0, bvt = 0, bvn = 0, bf = 0;
    std::vector<std::thread> th;
    for (auto& c : P->chunks) {
        th.emplace_back([&c, bv, bvt, bvn, bf, v, vt, vn, fi, ft, fn] {
            if (v && !c.v.empty()) memcpy(v + bv * 3, c.v.data(), c.v.size() * 4);
            if (vt && !c.vt.empty()) memcpy(vt + bvt * 2, c.vt.data(), c.vt.size() * 4);
            if (vn && !c.vn.empty()) memcpy(vn + bvn * 3, c.vn.data(), c.vn.size() * 4);
            for (uint32_t k : c.rel_i) c.fi[k] += bv;
            for (uint32_t k : c.rel_t) c.ft[k] += bvt;
            for (uint32_t k : c.rel_n) c.fn[k] += bvn;
            for (size_t k = 0; k < c.fi.size(); k++) {
                if (fi) fi[bf * 3 + k] = (int32_t)c.fi[k];
                if (ft) ft[bf * 3 + k] = (int32_t)c.ft[k];
                if (fn) fn[bf * 3 + k] = (int32_t)c.fn[k];
            }
        });
        bv += c.v.size() / 3; bvt += c.vt.size() / 2; bvn += c.vn.size() / 3; bf += c.fi.size() / 3;
    }
    for (auto& x : th) x.join();
    delete P;
    return TEXIR_OK;
}

// float RGB [npix][3] -> Radiance RGBE [npix][4] (shared exponent of the largest channel, mantissas truncated), the pixel arithmetic of
// cv2.imwrite(".hdr") (trainer/generate_ir_texture.py:82, trainer/train_material.py:350-353): every operation in float32, as
// io_formats.rgbe_encode spells it in numpy.
int texir_rgbe_encode(const float* rgb, int64_t npix, uint8_t* out)
{
    if (npix < 0 || (npix && (!rgb || !out))) return TEXIR_ERR_INVALID;
    parallel_chunks(npix, 1 << 16, [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) {
            const float r = rgb[3 * i], g = rgb[3 * i + 1], bl = rgb[3 * i + 2];
            // numpy's max propagates NaN (and NaN > 1e-32 is false)
            float mx = r;
            mx = (g > mx || g != g) ? g : mx;
            if (mx == mx) mx = (bl > mx || bl != bl) ? bl : mx;
            uint8_t* o = out + 4 * i;
            if (!(mx > 1e-32f)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
            int ex;
            const float m = frexpf(mx, &ex);
            const float scale = m * 256.0f / mx;
            const float ch[3] = {r, g, bl};
            for (int k = 0; k < 3; k++) {
                float x = ch[k] * scale;
                x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);       // np.clip (NaN stays NaN -> cast below)
                o[k] = (x == x) ? (uint8_t)x : 0;
            }
            o[3] = (uint8_t)(ex + 128);
        }
    });
    return TEXIR_OK;
}

// RGBE bytes [H][W][4] -> new-style RLE scanlines as OpenCV's writer (and Radiance's own rgbe.c) lay them out: per scanline the marker
// 2, 2, W >> 8, W & 255, then the four channel planes, each as runs (128 + n, value; 4 <= n <= 127; a shorter run only when it is all
// there is before the next long one) and literals (n, n bytes; n <= 128).  Widths outside [8, 32767] are written flat, as that writer
// does.  Returns the bytes written, or < 0 (cap too small: 4 + 4 * (W + W / 64 + 4) per scanline always suffices).
int64_t texir_hdr_encode_rle(const uint8_t* rgbe, int32_t W, int32_t H, uint8_t* out, int64_t cap)
{
    if (W < 0 || H < 0 || ((int64_t)W * H && (!rgbe || !out))) return TEXIR_ERR_INVALID;
    if (W < 8 || W > 32767) {
        const int64_t n = 4ll * W * H;
        if (cap < n) return TEXIR_ERR_INVALID;
        memcpy(out, rgbe, (size_t)n);
        return n;
    }
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), H / 16));
    std::vector<std::vector<uint8_t>> part(nt);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] {
            auto& o = part[t];
            const int y0 = (int)((int64_t)H * t / nt), y1 = (int)((int64_t)H * (t + 1) / nt);
            o.reserve((size_t)(y1 - y0) * (4 + 4 * (size_t)W));
            std::vector<uint8_t> plane(W);
            for (int y = y0; y < y1; y++) {
                const uint8_t* row = rgbe + (size_t)y * W * 4;
                o.push_back(2); o.push_back(2); o.push_back((uint8_t)(W >> 8)); o.push_back((uint8_t)(W & 255));
                for (int c = 0; c < 4; c++) {
                    for (int x = 0; x < W; x++) plane[x] = row[4 * x + c];
                    const uint8_t* d = plane.data();
                    int cur = 0;
                    while (cur < W) {
                        int beg = cur, run = 0, old_run = 0;
                        while (run < 4 && beg < W) {
                            beg += run;
                            old_run = run;
                            run = 1;
                            while (beg + run < W && run < 127 && d[beg] == d[beg + run]) run++;
                        }
                        if (old_run > 1 && old_run == beg - cur) { o.push_back((uint8_t)(128 + old_run)); o.push_back(d[cur]); cur = beg; }
                        while (cur < beg) {
                            const int n = std::min(beg - cur, 128);
                            o.push_back((uint8_t)n);
                            o.insert(o.end(), d + cur, d + cur + n);
                            cur += n;
                        }
                        if (run >= 4) { o.push_back((uint8_t)(128 + run)); o.push_back(d[beg]); cur += run; }
                    }
                }
            }
        });
        for (auto& x : th) x.join();
    }
    int64_t total = 0;
    for (auto& o : part) total += (int64_t)o.size();
    if (total > cap) return TEXIR_ERR_INVALID;
    int64_t off = 0;
    for (auto& o : part) { if (!o.empty()) memcpy(out + off, o.data(), o.size()); off += (int64_t)o.size(); }
    return total;
}

// Radiance RGBE [npix][4] -> float RGB [npix][3]: mantissa * 2^(e - 136), zero exponent = black (decode half of cv2.imread(".hdr", -1))
int texir_rgbe_decode(const uint8_t* rgbe, int64_t npix, float* rgb)
{
    if (npix < 0 || (npix && (!rgbe || !rgb))) return TEXIR_ERR_INVALID;
    parallel_chunks(npix, 1 << 16, [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; i++) {
            const uint8_t* s = rgbe + 4 * i;
            const float f = s[3] ? ldexpf(1.0f, (int)s[3] - 136) : 0.0f;
            rgb[3 * i] = (float)s[0] * f; rgb[3 * i + 1] = (float)s[1] * f; rgb[3 * i + 2] = (float)s[2] * f;
        }
    });
    return TEXIR_OK;
}

}  // extern "C"
