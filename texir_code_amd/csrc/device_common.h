// Device-side building blocks shared by the gfx950 kernels: Hammersley/Cranley-Patterson sampling,
// local frames, hemisphere/GGX directions (utils/sample_util.py:28-146), per-lane BVH2 traversal
// with an LDS-resident stack, and the query_irf hit shader (models/tracer_o3d_irt.py:248-267).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "bvh_build.h"

namespace texir {

struct SceneDev {
    const float4* nodes4;  // GpuNode4 as 4 x 16 B (4-wide quantised tree; null when the scene was built binary-only)
    const float4* nodes4f; // GpuNode4F (float child planes of the same tree, index for index) -- read through the scalar cache by wave-uniform
                           // node steps; null = every node step per lane (TEXIR_UNIFORM_FLOAT=0)
    const float4* nodes;   // GpuNode as 4 x float4
    const float4* tris;    // GpuTri as kTriQuads x float4, by leaf-order slot
    const float4* quads;   // GpuQuad as 3 x float4, by record (TEXIR_QUAD: what the 4-wide tree's leaves name; record r owns slots 2 r, 2 r + 1)
    const float4* uvs;     // GpuTriUV as 2 x float4
    const float* tex;      // [Ht,Wt,3] row-major (layout 0), or the retiled copy the hit shader reads (layouts 1, 2: see shade_hit)
    int Ht, Wt;
    int tex_layout;        // 0 row-major; 1 = 8x8-texel tiles of 12-byte texels; 2 = overlapping 3x3 tiles at stride 2, one 128-byte line each;
                           // 3, 4 = 4-byte shared-exponent texels (pack_texel), overlapping 5x5 tiles at stride 4 / 8x4 tiles at stride 7x3, one line each
    int tiles_x;           // tiles per tile row (layouts 1, 2)
    int sched_weight;      // phase scheduler of trace_closest: weight of the lanes at inner nodes (2; 1 for scenes whose node steps run less than ~60 % full: texir_scene_tune measures it once per scene)
};

constexpr int kBlock = 256;          // 4 waves
constexpr int kLdsStack = 24;        // default: entries per lane kept in LDS (24 KiB per block)
constexpr int kSentinel = 0x7FFFFFFF;
constexpr int kStackCap = 96;        // LDS part + private overflow; the host checks the tree's worst case against it

// ------------------------------------------------------------------------------------------------
// sampling -- kept free of fused multiply-adds so that it tracks the reference's separately rounded
// float32 torch ops (utils/sample_util.py) as closely as the libm differences allow
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

// sample_util.py:41  float32(double(i)/double(N)).  For N = 2^k (and i < 2^24) the quotient is exact in float32, so the
// multiply by the exact reciprocal gives the identical value without a float64 division.
__device__ __forceinline__ float ham0(uint32_t i, uint32_t N)
{
    if ((N & (N - 1u)) == 0u && i < (1u << 24)) return (float)i * (1.0f / (float)N);
    return (float)((double)i / (double)N);
}
// :28-38  float32(double(bitrev(i)) * 2^-32): scaling by a power of two commutes with the rounding to 24 bits
__device__ __forceinline__ float ham1(uint32_t i) { return __uint2float_rn(__brev(i)) * 2.3283064365386963e-10f; }

__device__ __forceinline__ float shift_wrap_clamp(float s, float shift)
{
    s = s + shift;                       // :103
    if (s > 1.f) s = s - 1.f;            // :104-105 (strict >)
    if (s < 0.f) s = s + 1.f;            // :106-107
    return fminf(fmaxf(s, 1e-6f), (float)(1.0 - 1e-6));   // :108
}

struct Frame { float n[3], U[3], V[3]; };

__device__ __forceinline__ Frame make_frame(float nx, float ny, float nz)
{
    Frame f;
    // :84 axis choice on the RAW normal; :86-91 x/(|x|+1e-6)
    float ax = 1.f, ay = 0.f;
    if (fabsf(nx) > 0.99f) { ax = 0.f; ay = 1.f; }
    float ln = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f;
    f.n[0] = nx / ln; f.n[1] = ny / ln; f.n[2] = nz / ln;
    float c0 = ay * f.n[2] - 0.f * f.n[1], c1 = 0.f * f.n[0] - ax * f.n[2], c2 = ax * f.n[1] - ay * f.n[0];
    float lc = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + 1e-6f;
    f.U[0] = c0 / lc; f.U[1] = c1 / lc; f.U[2] = c2 / lc;
    float e0 = f.n[1] * f.U[2] - f.n[2] * f.U[1], e1 = f.n[2] * f.U[0] - f.n[0] * f.U[2], e2 = f.n[0] * f.U[1] - f.n[1] * f.U[0];
    float le = sqrtf(e0 * e0 + e1 * e1 + e2 * e2) + 1e-6f;
    f.V[0] = e0 / le; f.V[1] = e1 / le; f.V[2] = e2 / le;
    return f;
}

// cos/sin of the polar angle for the three modes (:115-143)
__device__ __forceinline__ void polar(int mode, float s0, float rough, float& ct, float& st)
{
    if (mode == 0) { ct = 1.0f - s0; st = sqrtf(1.0f - ct * ct); }
    else if (mode == 1) { ct = sqrtf(1.0f - s0); st = sqrtf(1.0f - ct * ct); }
    else {
        float a = rough * rough;
        ct = sqrtf((1.0f - s0) / (1.0f + (a * a - 1.f) * s0));
        ct = fminf(fmaxf(ct, -1.0f + 1e-6f), 1.0f - 1e-6f);
        st = fminf(fmaxf(sqrtf(1.0f - ct * ct), -1.0f + 1e-6f), 1.0f - 1e-6f);
    }
}

// FAST (the fused IrT kernels only; texir_generate_dir and the GGX kernels keep the libm form that is pinned at 5e-6 on the reference's
// own directions): sin / cos of phi = 2 pi s1 - pi from the hardware's v_sin_f32 / v_cos_f32, which take their argument in revolutions --
// sin(2 pi s1 - pi) = -sin(2 pi s1), no range reduction, 2 quarter-rate instructions instead of ~70 -- absolute error ~1e-6 per component,
// two orders of magnitude inside the 2e-5 the whole-loop golden test allows.
template <bool FAST = false>
__device__ __forceinline__ void sample_dir(int mode, float s0, float s1, float rough, const Frame& f, float* L)
{
    float ct, st;
    polar(mode, s0, rough, ct, st);
    float sp, cp;
    if constexpr (FAST) {
        sp = -__builtin_amdgcn_sinf(s1); cp = -__builtin_amdgcn_cosf(s1);
    } else {
        float phi = 6.283185307179586f * s1 - 3.141592653589793f;
        sincosf(phi, &sp, &cp);
    }
    sp = sp * st; cp = -(cp * st);
    for (int a = 0; a < 3; a++) L[a] = f.V[a] * sp + f.n[a] * ct + f.U[a] * cp;
}

// corner uvs of the triangle in leaf slot `slot`: a = (uv0, uv1), b = (uv2, -, -)
__device__ __forceinline__ void tri_uvs(const SceneDev& sc, int slot, float4& a, float4& b)
{
#if TEXIR_UV_QUAD
    // one 32-byte record per quad record: (q0, q1), (q2, q3); the even slot is triangle (q0, q1, q2), the odd one (q3, q2, q1)  (bvh_build.h)
    const size_t rec = (size_t)(slot >> 1);
    const float4 A = sc.uvs[2 * rec], B = sc.uvs[2 * rec + 1];
    const bool odd = (slot & 1) != 0;
    a = odd ? make_float4(B.z, B.w, B.x, B.y) : A;
    b = odd ? make_float4(A.z, A.w, 0.f, 0.f) : B;
#else
    a = sc.uvs[2 * (size_t)slot]; b = sc.uvs[2 * (size_t)slot + 1];
#endif
}
// primitive id (row of the caller's index array) of the triangle in leaf slot `slot`
__device__ __forceinline__ uint32_t tri_prim(const SceneDev& sc, int slot) { return __float_as_uint(sc.tris[3 * (size_t)slot].w); }

// ------------------------------------------------------------------------------------------------
// 4-byte texels (texture layouts 3, 4).  The reference's radiance texture is an RGBE file times 2^hdr_exposure
// (tracer_o3d_irt.py:77-81): every texel is three 8-bit integers times ONE power of two.  Such a texel is stored as
// m_r | m_g << 8 | m_b << 16 | E << 24 with value_c = m_c * 2^(E - 127) -- E is the float exponent field of the scale itself, so the decode
// is one shift + mask for the scale, v_cvt_f32_ubyte{0,1,2} and three multiplications by a power of two: the IDENTICAL float32 values
// (m <= 255 times a normal power of two is exact), hence bit-identical bilinear sums.  pack_texel says whether a float triple has this form.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ bool pack_texel(uint32_t br, uint32_t bg, uint32_t bb, uint32_t& word)
{
    const uint32_t b[3] = {br, bg, bb};
    int t[3]; uint32_t odd[3];
    int k = 1 << 30;
    for (int c = 0; c < 3; c++) {
        if (b[c] == 0u) { t[c] = 1 << 30; odd[c] = 0u; continue; }           // +0.0
        const uint32_t ex = b[c] >> 23;                                       // sign bit set -> ex > 254 below
        if (ex < 1u || ex > 254u) return false;                               // negative (incl. -0.0), subnormal, inf, nan
        uint32_t m = (b[c] & 0x7FFFFFu) | 0x800000u;
        int tz = 0;
        while (!(m & 1u)) { m >>= 1; tz++; }                                  // <= 23 rounds
        odd[c] = m; t[c] = (int)ex - 150 + tz;                                // value = odd * 2^t
        k = t[c] < k ? t[c] : k;
    }
    if (k == (1 << 30)) { word = 0u; return true; }                           // black texel
    const int E = k + 127;
    if (E < 1 || E > 254) return false;
    uint32_t w = (uint32_t)E << 24;
    for (int c = 0; c < 3; c++) {
        if (odd[c] == 0u) continue;
        const int sh = t[c] - k;
        if (sh > 7) return false;
        const uint32_t m = odd[c] << sh;
        if (m > 255u) return false;
        w |= m << (8 * c);
    }
    word = w;
    return true;
}

__device__ __forceinline__ void unpack_texel(uint32_t q, float& r, float& g, float& b)
{
    const float scale = __uint_as_float((q >> 1) & 0x7F800000u);
    r = (float)(q & 0xFFu) * scale; g = (float)((q >> 8) & 0xFFu) * scale; b = (float)((q >> 16) & 0xFFu) * scale;
}

// ------------------------------------------------------------------------------------------------
// hit shader: query_irf post-intersection math (tracer_o3d_irt.py:248-267)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void shade_hit(const SceneDev& sc, int tri_slot, float bu, float bv, float* rgb)
{
    float u = fminf(fmaxf(bu, 0.f), 1.f), v = fminf(fmaxf(bv, 0.f), 1.f);      // :250 np.clip
    float4 a, b;
    tri_uvs(sc, tri_slot, a, b);
    float w = 1.0f - u - v;
    // :260 (float64 in the reference; float32 here -- <=1e-7 in uv, far below a texel)
    float gx = a.x * w + a.z * u + b.x * v;
    float gy = a.y * w + a.w * u + b.y * v;
    gx = gx * 2.f - 1.f;                  // :262
    gy = -(1.f - gy * 2.f);               // :263
    // F.grid_sample(bilinear, border, align_corners=False) (:265)
    float x = ((gx + 1.f) * (float)sc.Wt - 1.f) * 0.5f, y = ((gy + 1.f) * (float)sc.Ht - 1.f) * 0.5f;
    x = fminf(fmaxf(x, 0.f), (float)(sc.Wt - 1)); y = fminf(fmaxf(y, 0.f), (float)(sc.Ht - 1));
    float x0f = floorf(x), y0f = floorf(y);
    int x0 = (int)x0f, y0 = (int)y0f;
    float wx1 = x - x0f, wx0 = 1.f - wx1, wy1 = y - y0f, wy0 = 1.f - wy1;
    int x1 = min(x0 + 1, sc.Wt - 1), y1 = min(y0 + 1, sc.Ht - 1);
    // out-of-range neighbours carry weight exactly 0 after the border clamp, so clamping their index is exact
    float w00 = wx0 * wy0, w10 = (x0 + 1 < sc.Wt) ? wx1 * wy0 : 0.f, w01 = (y0 + 1 < sc.Ht) ? wx0 * wy1 : 0.f,
          w11 = (x0 + 1 < sc.Wt && y0 + 1 < sc.Ht) ? wx1 * wy1 : 0.f;
    // Where the four taps live.  The values read are the same in every layout (the retiled copies hold the identical floats), so
    // the result is bit-identical; what changes is how many 128-byte lines a fetch touches: row-major 2 rows 12*Wt bytes apart
    // (2.4 lines on average), layout 2 exactly one line (any 2x2 footprint lies inside the 3x3 tile of (x0/2, y0/2)).
    if (sc.tex_layout >= 3) {
        // one 128-byte line of 4-byte texels holds the whole footprint: two adjacent-dword loads
        const uint32_t* t;
        int pitch;
        if (sc.tex_layout == 3) {
            t = (const uint32_t*)sc.tex + ((size_t)(y0 >> 2) * sc.tiles_x + (x0 >> 2)) * 32 + ((y0 & 3) * 5 + (x0 & 3));
            pitch = 5;
        } else {
            const int tx = (int)__umulhi((uint32_t)x0, 0x24924925u), ty = (int)__umulhi((uint32_t)y0, 0x55555556u);      // x0 / 7, y0 / 3 (exact below 2^17; the host checks the size)
            t = (const uint32_t*)sc.tex + ((size_t)ty * sc.tiles_x + tx) * 32 + ((y0 - 3 * ty) * 8 + (x0 - 7 * tx));
            pitch = 8;
        }
        const uint32_t q00 = t[0], q10 = t[1], q01 = t[pitch], q11 = t[pitch + 1];
        float a[3], b[3], c[3], d[3];
        unpack_texel(q00, a[0], a[1], a[2]); unpack_texel(q10, b[0], b[1], b[2]);
        unpack_texel(q01, c[0], c[1], c[2]); unpack_texel(q11, d[0], d[1], d[2]);
        for (int ch = 0; ch < 3; ch++) {
            float acc = a[ch] * w00;
            acc += b[ch] * w10;
            acc += c[ch] * w01;
            acc += d[ch] * w11;
            rgb[ch] = acc;
        }
        return;
    }
    const float *p00, *p10, *p01, *p11;
    if (sc.tex_layout == 2) {
        const float* t = sc.tex + ((size_t)(y0 >> 1) * sc.tiles_x + (x0 >> 1)) * 32 + ((y0 & 1) * 3 + (x0 & 1)) * 3;
        p00 = t; p10 = t + 3; p01 = t + 9; p11 = t + 12;
    } else if (sc.tex_layout == 1) {
        auto at = [&](int x, int y) { return sc.tex + (((size_t)(y >> 3) * sc.tiles_x + (x >> 3)) * 64 + ((y & 7) * 8 + (x & 7))) * 3; };
        p00 = at(x0, y0); p10 = at(x1, y0); p01 = at(x0, y1); p11 = at(x1, y1);
    } else {
        p00 = sc.tex + ((size_t)y0 * sc.Wt + x0) * 3;
        p10 = sc.tex + ((size_t)y0 * sc.Wt + x1) * 3;
        p01 = sc.tex + ((size_t)y1 * sc.Wt + x0) * 3;
        p11 = sc.tex + ((size_t)y1 * sc.Wt + x1) * 3;
    }
    for (int c = 0; c < 3; c++) {
        float acc = p00[c] * w00;
        acc += p10[c] * w10;
        acc += p01[c] * w01;
        acc += p11[c] * w11;
        rgb[c] = acc;
    }
}

// Watertight triangle test (Woop, Benthin, Wald: "Watertight Ray/Triangle Intersection", JCGT 2013 -- the algorithm class Embree's
// robust mode stands for; Open3D's RaycastingScene is an Embree scene): the triangle's vertices are moved to the ray's origin and
// sheared so that the ray becomes the +z axis; what is left is a 2D point-in-triangle test of the origin.  A vertex's sheared
// coordinates depend on the vertex and the ray only, so every triangle sharing it sees the same numbers, and the three edge
// functions  C.x*B.y - C.y*B.x  are evaluated with EXACT signs: products and difference rounded separately (rounding is monotonic:
// the sign of the rounded difference is right unless it is 0), and an exact-zero result is resolved by the error terms of the two
// products (fma(a, b, -fl(a*b)) is exact).  The set of triangles that accept a ray is therefore exactly the set whose sheared
// (rounded) 2D image contains the origin, edges and vertices included: a closed mesh cannot leak.
// Returns U, V, W (weights of v0, v1, v2 before normalisation) and whether the origin is inside (all signs agree, zeros count as inside).
__device__ __forceinline__ float edge2_exact(float bx, float by, float cx, float cy)
{
    const float p = cx * by, q = cy * bx;
    float r = p - q;
    if (r == 0.f) r = __builtin_fmaf(cx, by, -p) - __builtin_fmaf(cy, bx, -q);
    return r;
}

#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// closest-hit traversal, one ray per lane.  Stack: LSTK entries per lane in LDS laid out
// [entry][thread] (conflict-free ds_read/write), deeper entries in a private overflow array.
// Returns hit triangle slot (leaf order) or -1; t in units of |dir| (Embree semantics: t > 0).
// ------------------------------------------------------------------------------------------------
struct Hit { float t, u, v; int slot; };

// TEXIR_CHAIN_PROBE (measurement build only, tools/chain_probe.sh; 0 = shipped): the wave reads the shader clock (s_memtime) around every wave-level step of
// trace_core -- per-lane (vector) node step, wave-uniform (scalar-cache) node step, leaf step -- and the kernel around the trace, the hit shader and the whole
// pass; cycles and step counts per kind go to stats[8...] (kernels.hip irt_probe_flush).  Summed over the resident waves these cycles ARE the kernel's duration
// (a wave is always in exactly one of the measured regions or in the scheduler between them), which is what bench.py's `roofline.limits.chain` is built from.
#ifndef TEXIR_CHAIN_PROBE
#define TEXIR_CHAIN_PROBE 0
#endif
// A clock read is not free: s_memtime goes through the scalar memory path and its result is waited for (a first probe build that read it around EVERY step
// ran 3.9x slower than the shipped kernel and measured mostly itself).  The build therefore COUNTS every step (one scalar add) but TIMES only every
// kProbeEvery-th wave-level step of a wave, and times an empty region right after each timed step (`null`): mean(step) - mean(null) is the step's duration
// under the load of seven undisturbed neighbour waves.
#if TEXIR_CHAIN_PROBE
__device__ __forceinline__ uint32_t probe_clock() { return (uint32_t)__builtin_amdgcn_s_memtime(); }
// wave-uniform accumulate (keeps the counter in an SGPR)
__device__ __forceinline__ void probe_add(uint32_t& acc, uint32_t x) { acc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(acc + x)); }
#ifndef TEXIR_PROBE_EVERY
#define TEXIR_PROBE_EVERY 32
#endif
constexpr uint32_t kProbeEvery = TEXIR_PROBE_EVERY;
#endif
// wave_iters[2 + ...]: steps, timed steps, cycles of timed steps -- of per-lane node steps, wave-uniform node steps, leaf steps; then timed empty regions, their cycles
constexpr int kProbeSlots = 11;

// TEXIR_SCHED (A/B switch; 1 = default).  How the wave shares its issue slots between lanes that hold an inner node and lanes
// that hold a leaf:
//   0: "while-while" -- node steps until NO lane holds an inner node, then leaf steps until NO lane holds a leaf.  One lane on a
//      long run of inner nodes keeps 63 lanes waiting at their leaves: on cluttered scenes the wave executes 2.8x the node steps of
//      its average ray (c4_scan: 55 wave-level node steps per pass for 19.7 per ray, of which only 30 are the slowest lane's).
//   1: per step, ballot + population count of both kinds of lanes and the body with more lanes waiting runs; node lanes count
//      w-fold (a leaf step -- 1 or 2 watertight triangle tests and the culling pop -- costs ~1.5 node steps, and on coherent scenes leaf lanes
//      arrive in bursts that are worth a short wait).  Every lane still performs exactly the same node visits, triangle tests and stack
//      operations in the same order (the schedule only decides WHEN a lane's next step issues), so hits are identical bit for bit.
//      CPU replay of the kernel's schedule (tools/bvh_sim.cpp): c4_scan 55.3 -> 36.8 (w = 1) / 40.6 (w = 2) wave-level node steps per pass,
//      c4 19.5 -> 17.6 / 17.8.  Measured (profiles/r03/ab_tables.txt; Grays/s, while-while = 1): w = 1: c4_scan 1.28, c4 0.98 (the leaf
//      batches get smaller: wave-level triangle steps 5.2 -> 7.1); w = 2: c4_scan 1.24, c4 1.00; w = 3: 1.19, 1.00.
//      w comes with the scene (SceneDev::sched_weight): kSchedNodeWeight = 2 unless texir_scene_tune has measured that the scene's node steps
//      run less than 60 % full (cluttered scenes), then 1.
#ifndef TEXIR_SCHED
#define TEXIR_SCHED 1
#endif
// leaf steps specialised for the wave's common dominant ray axis (A/B switch; 0 = every leaf step selects the components per lane)
#ifndef TEXIR_LEAF_UNIFORM_KZ
#define TEXIR_LEAF_UNIFORM_KZ 1
#endif
#ifndef TEXIR_SCHED_NODE_WEIGHT
#define TEXIR_SCHED_NODE_WEIGHT 2
#endif
constexpr int kSchedNodeWeight = TEXIR_SCHED_NODE_WEIGHT;

// CULL: a stack entry also carries the child's entry distance, and an entry whose distance is not below the closest hit found
// since it was pushed is dropped when it is popped (it cannot contain a closer hit: same result, bit for bit) instead of
// costing a node fetch + a full node step that finds all four children behind the hit.  Entries are 8 bytes then (one
// ds_write_b64 / ds_read_b64, conflict-free in the [entry][thread] layout), so a kernel keeps LSTK * 2 KiB of LDS per block.
template <bool CULL> struct StackEntry { typedef int type; };
template <> struct StackEntry<true> { typedef int2 type; };

// trace_core is the traversal; trace_closest (one ray per lane, run to completion) and trace_stream (lanes take their NEXT ray while the others
// are still under way: compaction by refill, see there) are its two drivers.
struct NoNext { };
template <bool STATS, int LSTK, int WIDTH, bool CULL, bool STREAM, typename Next>
__device__ __forceinline__ Hit trace_core(const SceneDev& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                          uint32_t& n_nodes, uint32_t& n_tris, uint32_t* wave_iters, int refill_at, Next&& next)
{
    typedef typename StackEntry<CULL>::type Entry;
    // what the traversal keeps of a ray (set by begin_ray; a streamed lane overwrites them when it takes its next ray)
    float idx, idy, idz, oodx, oody, oodz;
    uint32_t lon;
#if TEXIR_TRI_WATERTIGHT
    // The ray-space shear of the watertight test: kz = dominant axis of the direction, (k1, k2) = the two axes that follow it cyclically;
    // a vertex A (relative to the origin) maps to x' = A[k1] - Sx A[kz], y' = A[k2] - Sy A[kz], z' = Sz A[kz], which sends d to (0, 0, 1).
    // (Woop et al. swap k1 / k2 for d[kz] < 0 to keep the winding: that negates U, V, W and their sum EXACTLY and changes neither the accept test --
    // no two signs differ -- nor t, u, v; it is left out.)  What a ray carries into the leaf steps is (kz, Sx, Sy, Sz): 4 registers instead of the
    // 9 of three general matrix rows.  When the rays of the wave share kz (one ~3-degree direction cell: they do unless the cell straddles a
    // |dx| = |dy|-like plane) the leaf step runs a body specialised for that axis -- 3 sub + 2 fma + 1 mul per vertex, no component selects; the
    // per-lane body selects the components and then evaluates the SAME expressions, so a ray's hit does not depend on its wave's company.
    int kz; float Sx, Sy, Sz;
#endif
    auto begin_ray = [&]() __attribute__((always_inline)) {
        const float ooeps = 8.271806e-25f;  // 2^-80
        idx = __builtin_amdgcn_rcpf(fabsf(dx) > ooeps ? dx : copysignf(ooeps, dx));
        idy = __builtin_amdgcn_rcpf(fabsf(dy) > ooeps ? dy : copysignf(ooeps, dy));
        idz = __builtin_amdgcn_rcpf(fabsf(dz) > ooeps ? dz : copysignf(ooeps, dz));
        oodx = ox * idx; oody = oy * idy; oodz = oz * idz;
        // byte offsets of the NEAR plane array of each axis inside a 128-byte float node (far = offset ^ 16), wave-uniform when the rays' signs agree
        lon = (idx < 0.f ? 16u : 0u) | ((idy < 0.f ? 48u : 32u) << 8) | ((idz < 0.f ? 80u : 64u) << 16);
#if TEXIR_TRI_WATERTIGHT
        const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
        kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        const float dkz = kz == 0 ? dx : (kz == 1 ? dy : dz);
        const float dk1 = kz == 0 ? dy : (kz == 1 ? dz : dx), dk2 = kz == 0 ? dz : (kz == 1 ? dx : dy);
        Sz = __builtin_amdgcn_rcpf(dkz); Sx = dk1 * Sz; Sy = dk2 * Sz;
#endif
    };
    if constexpr (!STREAM) begin_ray();
    else { idx = idy = idz = oodx = oody = oodz = 0.f; lon = 0u;
#if TEXIR_TRI_WATERTIGHT
           kz = 0; Sx = Sy = Sz = 0.f;
#endif
    }
    // what the lanes of the wave agree on (decided per ray batch; a streamed wave decides again after every refill, over the lanes that hold a ray)
    uint32_t son; bool signs_uniform; uint32_t son0, son1, son2;
#if TEXIR_TRI_WATERTIGHT
    int kz0; [[maybe_unused]] bool kz_uniform;
#endif
    // (`has`: this lane holds a ray under way -- always, except in a streamed wave)
    auto agree = [&](bool has) __attribute__((always_inline)) {
        if constexpr (STREAM) {
            const unsigned long long m = __ballot(has);
            const int src = m ? __ffsll((long long)m) - 1 : 0;
            son = (uint32_t)__builtin_amdgcn_readlane((int)lon, src);
            signs_uniform = WIDTH == 4 && sc.nodes4f != nullptr && !__any(has && lon != son);
#if TEXIR_TRI_WATERTIGHT
            kz0 = __builtin_amdgcn_readlane(kz, src);
            kz_uniform = !__any(has && kz != kz0);
#endif
        } else {
            son = (uint32_t)__builtin_amdgcn_readfirstlane((int)lon);
            signs_uniform = WIDTH == 4 && sc.nodes4f != nullptr && !__any(lon != son);
#if TEXIR_TRI_WATERTIGHT
            kz0 = __builtin_amdgcn_readfirstlane(kz);
            kz_uniform = !__any(kz != kz0);
#endif
        }
        son0 = son & 255u; son1 = (son >> 8) & 255u; son2 = son >> 16;
    };
    agree(true);
    Hit h;
    h.t = __builtin_inff(); h.u = 0.f; h.v = 0.f; h.slot = -1;
    int node = STREAM ? kSentinel : 0;                       // (a streamed lane gets its first ray from `next` like every later one)
    auto first_active = [&]() -> bool { unsigned long long m = __ballot(1); return (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1; };
    // the traversal stack: LSTK entries per lane in LDS ([entry][thread]), deeper ones private.  The stack pointer is kept as
    // the LDS address of the next free entry (push = ds_write + one add, no index scaling in the node step).
    __shared__ Entry lds_all[LSTK * kBlock];
    Entry ovf[kStackCap - LSTK];
    Entry* const base = lds_all + threadIdx.x;
    Entry* const lim = base + LSTK * kBlock;
    Entry* top = base;
    auto make = [](int code, float tn) -> Entry { if constexpr (CULL) return make_int2(code, __float_as_int(tn)); else return code; };
    // LDS part and private overflow are kept in separate, wave-uniformly guarded code paths: the overflow is almost never
    // touched (depth > LSTK), and hipcc must not merge the two address spaces into one flat access
    auto push = [&](int code, float tn) {
        const Entry x = make(code, tn);
        if (top < lim) *top = x;
        if (__any(top >= lim)) { if (top >= lim) ovf[(top - lim) / kBlock] = x; }
        top += kBlock;
    };
    auto pop = [&]() -> int {
        for (;;) {
            if (top == base) return kSentinel;
            top -= kBlock;
            Entry v = *(top < lim ? top : lim - kBlock);
            if (__any(top >= lim)) { Entry b = ovf[top >= lim ? (top - lim) / kBlock : 0]; v = top >= lim ? b : v; }
            if constexpr (CULL) { if (__int_as_float(v.y) < h.t) return v.x; }      // else: behind the closest hit, drop it
            else return v;
        }
    };

#if TEXIR_CHAIN_PROBE
    bool probe_scalar_path = false;
    uint32_t pn_nv = 0, pt_nv = 0, pc_nv = 0, pn_ns = 0, pt_ns = 0, pc_ns = 0, pn_lf = 0, pt_lf = 0, pc_lf = 0, pt_null = 0, pc_null = 0, probe_tick = 0;
#endif
    // ---- one node step of the 4-wide tree: four box tests, sort, push the far children, descend into the nearest ----
    auto node_step4 = [&]() __attribute__((always_inline)) {
        float key[4]; int code[4];
        // The box tests on per-lane node words (vector loads of the 64-byte quantised node): the ray's direction signs decide once
        // per node (for all four children at a time: the bytes stay packed) which of the lo/hi planes is the entry and which the exit
        // plane; the cell size is folded into the reciprocal direction, the node origin into the offset.
        auto box4 = [&](uint32_t q0x, uint32_t q0y, uint32_t q0z, uint32_t q0w, uint32_t q1x, uint32_t q1y, uint32_t q1z, uint32_t q1w,
                        uint32_t q2x, uint32_t q2y, uint32_t q2z, uint32_t q2w) __attribute__((always_inline)) {
            const float sx = __uint_as_float(q0w) * idx, sy = __uint_as_float(q2z) * idy, sz = __uint_as_float(q2w) * idz;
            const float bx = __uint_as_float(q0x) * idx - oodx, by = __uint_as_float(q0y) * idy - oody, bz = __uint_as_float(q0z) * idz - oodz;
            const uint32_t nx_ = idx < 0.f ? q1w : q1x, fx_ = idx < 0.f ? q1x : q1w;
            const uint32_t ny_ = idy < 0.f ? q2x : q1y, fy_ = idy < 0.f ? q1y : q2x;
            const uint32_t nz_ = idz < 0.f ? q2y : q1z, fz_ = idz < 0.f ? q1z : q2y;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int sh = 8 * k;
                float nxt = (float)((nx_ >> sh) & 255u) * sx + bx, fxt = (float)((fx_ >> sh) & 255u) * sx + bx;
                float nyt = (float)((ny_ >> sh) & 255u) * sy + by, fyt = (float)((fy_ >> sh) & 255u) * sy + by;
                float nzt = (float)((nz_ >> sh) & 255u) * sz + bz, fzt = (float)((fz_ >> sh) & 255u) * sz + bz;
                float tn = fmaxf(fmaxf(nxt, nyt), fmaxf(nzt, 0.f));
                float tf = fminf(fminf(fxt, fyt), fminf(fzt, h.t));
                // (unused slots: inverted box, and if ever entered they lead to a degenerate dummy triangle -- no test needed here)
                key[k] = tn <= tf ? tn : __builtin_inff();
            }
        };
        // Wave-uniform steps (the rays of a pass leave neighbouring points in nearly the same direction and walk the upper levels
        // together: 45 % of the wave-level node steps on c4): a vector fetch costs the L1 64 lanes x 64 bytes whether or not the lanes
        // agree (tools/tcp_node.hip: 60 clocks per wave fetch from 1 to 16 distinct nodes); the scalar path costs it nothing.  When the
        // wave's rays also share their direction signs (a pass is one ~2.5 degree direction cell: they do unless the cell straddles an
        // axis plane), the step reads the FLOAT planes of the node (GpuNode4F) through wave-uniform offsets that pick the near / far
        // plane arrays: no byte -> float conversion (24 per step), no sign select, no origin / cell-size set-up.
        const int n0 = __builtin_amdgcn_readfirstlane(node);
#if TEXIR_CHAIN_PROBE
        probe_scalar_path = signs_uniform && !__any(node != n0);
#endif
        if (signs_uniform && !__any(node != n0)) {
            typedef float F4 __attribute__((ext_vector_type(4)));
            typedef int32_t I4 __attribute__((ext_vector_type(4)));
            const char __attribute__((address_space(4)))* const nb =
                (const char __attribute__((address_space(4)))*)(reinterpret_cast<const char*>(sc.nodes4f) + ((uint32_t)n0 << 7));
            auto ldf = [&](uint32_t off) { return *(const F4 __attribute__((address_space(4)))*)(nb + off); };
            const F4 pnx = ldf(son0), pfx = ldf(son0 ^ 16u), pny = ldf(son1), pfy = ldf(son1 ^ 16u), pnz = ldf(son2), pfz = ldf(son2 ^ 16u);
            const I4 ch = *(const I4 __attribute__((address_space(4)))*)(nb + 96u);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float nxt = pnx[k] * idx - oodx, fxt = pfx[k] * idx - oodx, nyt = pny[k] * idy - oody, fyt = pfy[k] * idy - oody;
                const float nzt = pnz[k] * idz - oodz, fzt = pfz[k] * idz - oodz;
                const float tn = fmaxf(fmaxf(nxt, nyt), fmaxf(nzt, 0.f)), tf = fminf(fminf(fxt, fyt), fminf(fzt, h.t));
                key[k] = tn <= tf ? tn : __builtin_inff();
            }
            code[0] = ch.x; code[1] = ch.y; code[2] = ch.z; code[3] = ch.w;
        } else {
            // (uniform base + 32-bit byte offset: one VALU op of address arithmetic, saddr-form loads)
            const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes4) + ((uint32_t)node << 6));
            const uint4 v0 = np[0], v1 = np[1], v2 = np[2], v3 = np[3];
            box4(v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w);
            code[0] = (int)v3.x; code[1] = (int)v3.y; code[2] = (int)v3.z; code[3] = (int)v3.w;
        }
        if (STATS) { n_nodes++; if (wave_iters && first_active()) wave_iters[0]++; }
        // sort the four (key, code) pairs ascending: 5-comparator network
#define TEXIR_CSWAP(a, b) { bool s_ = key[b] < key[a]; float ka = s_ ? key[b] : key[a], kb = s_ ? key[a] : key[b]; int ca = s_ ? code[b] : code[a], cb = s_ ? code[a] : code[b]; key[a] = ka; key[b] = kb; code[a] = ca; code[b] = cb; }
        TEXIR_CSWAP(0, 1) TEXIR_CSWAP(2, 3) TEXIR_CSWAP(0, 2) TEXIR_CSWAP(1, 3) TEXIR_CSWAP(1, 2)
#undef TEXIR_CSWAP
        const float inf = __builtin_inff();
        if (!__any(top + 3 * kBlock > lim)) {
            // common case (one wave-uniform test per node): the whole update stays inside the LDS part of the stack
            if (key[3] < inf) { *top = make(code[3], key[3]); top += kBlock; }
            if (key[2] < inf) { *top = make(code[2], key[2]); top += kBlock; }
            if (key[1] < inf) { *top = make(code[1], key[1]); top += kBlock; }
            if (key[0] < inf) node = code[0];
            else if constexpr (CULL) {
                node = kSentinel;
                while (top != base) { top -= kBlock; const Entry e = *top; if (__int_as_float(e.y) < h.t) { node = e.x; break; } }
            }
            else if (top != base) { top -= kBlock; node = *reinterpret_cast<const int*>(top); }
            else node = kSentinel;
        } else {
            if (key[3] < inf) push(code[3], key[3]);
            if (key[2] < inf) push(code[2], key[2]);
            if (key[1] < inf) push(code[1], key[1]);
            node = key[0] < inf ? code[0] : pop();
        }
    };

    // ---- one node step of the binary tree (scenes too deep for the wide tree's stack bound, TEXIR_BVH_WIDTH=2) ----
    auto node_step2 = [&]() __attribute__((always_inline)) {
        const float4* np = sc.nodes + 4 * (size_t)node;
        float4 n0 = np[0], n1 = np[1], n2 = np[2];
        int2 ch = *reinterpret_cast<const int2*>(np + 3);
        if (STATS) { n_nodes++; if (wave_iters && first_active()) wave_iters[0]++; }
        float c0lox = n0.x * idx - oodx, c0hix = n0.y * idx - oodx, c0loy = n0.z * idy - oody, c0hiy = n0.w * idy - oody;
        float c0loz = n2.x * idz - oodz, c0hiz = n2.y * idz - oodz;
        float c1lox = n1.x * idx - oodx, c1hix = n1.y * idx - oodx, c1loy = n1.z * idy - oody, c1hiy = n1.w * idy - oody;
        float c1loz = n2.z * idz - oodz, c1hiz = n2.w * idz - oodz;
        float t0n = fmaxf(fmaxf(fminf(c0lox, c0hix), fminf(c0loy, c0hiy)), fmaxf(fminf(c0loz, c0hiz), 0.f));
        float t0f = fminf(fminf(fmaxf(c0lox, c0hix), fmaxf(c0loy, c0hiy)), fminf(fmaxf(c0loz, c0hiz), h.t));
        float t1n = fmaxf(fmaxf(fminf(c1lox, c1hix), fminf(c1loy, c1hiy)), fmaxf(fminf(c1loz, c1hiz), 0.f));
        float t1f = fminf(fminf(fmaxf(c1lox, c1hix), fmaxf(c1loy, c1hiy)), fminf(fmaxf(c1loz, c1hiz), h.t));
        bool h0 = t0n <= t0f, h1 = t1n <= t1f;
        if (h0 && h1) {
            bool swp = t1n < t0n;
            int nearc = swp ? ch.y : ch.x, farc = swp ? ch.x : ch.y;
            push(farc, swp ? t0n : t1n);
            node = nearc;
        } else if (h0) node = ch.x;
        else if (h1) node = ch.y;
        else node = pop();
    };

    // ---- one leaf: test its triangles, then pop (the pop drops what this leaf's hit has just put out of reach) ----
    // KZ = std::integral_constant<int, 0 | 1 | 2>: the wave's common dominant axis; -1: per lane
    auto leaf_body = [&](auto KZ) __attribute__((always_inline)) {
        constexpr int kzc = decltype(KZ)::value;
        const uint32_t code = ~(uint32_t)node;
#if TEXIR_QUAD
        if constexpr (WIDTH == 4) {
            // Quad records (bvh_build.h): three 16-byte words hold the four vertices of two triangles that share the edge (q1, q2): triangle 0 = (q0, q1, q2),
            // triangle 1 = (q3, q2, q1).  Four vertices are sheared instead of six, five edge functions evaluated instead of six -- the shared edge's value
            // for triangle 1 is triangle 0's negated (edge2_exact is antisymmetric bit for bit) -- and each triangle then goes through exactly the
            // arithmetic of the single-triangle test below, in its stored corner order.
            const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
            for (int r = first; r < first + cnt; r++) {
                const float4* qp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.quads) + (uint32_t)r * 48u);
                const float4 w0 = qp[0], w1 = qp[1], w2 = qp[2];
                if (STATS) { n_tris += 2; if (wave_iters && first_active()) wave_iters[1]++; }
                auto shear = [&](float p0, float p1, float p2, float& X, float& Y, float& Z) __attribute__((always_inline)) {
                    const float q0 = p0 - ox, q1 = p1 - oy, q2 = p2 - oz;
                    float qz, qx, qy;
                    if constexpr (kzc == 0) { qz = q0; qx = q1; qy = q2; }
                    else if constexpr (kzc == 1) { qz = q1; qx = q2; qy = q0; }
                    else if constexpr (kzc == 2) { qz = q2; qx = q0; qy = q1; }
                    else { qz = kz == 0 ? q0 : (kz == 1 ? q1 : q2); qx = kz == 0 ? q1 : (kz == 1 ? q2 : q0); qy = kz == 0 ? q2 : (kz == 1 ? q0 : q1); }
                    X = __builtin_fmaf(-Sx, qz, qx); Y = __builtin_fmaf(-Sy, qz, qy); Z = Sz * qz;
                };
                float X0, Y0, Z0, X1, Y1, Z1, X2, Y2, Z2, X3, Y3, Z3;
                shear(w0.x, w0.y, w0.z, X0, Y0, Z0); shear(w0.w, w1.x, w1.y, X1, Y1, Z1); shear(w1.z, w1.w, w2.x, X2, Y2, Z2); shear(w2.y, w2.z, w2.w, X3, Y3, Z3);
                const float E12 = edge2_exact(X1, Y1, X2, Y2);
                auto accept = [&](float U, float V, float W, float Az, float Bz, float Cz, int slot) __attribute__((always_inline)) {
                    const float mn = fminf(fminf(U, V), W), mxw = fmaxf(fmaxf(U, V), W);
                    const float det = U + V + W;
                    const float inv = __builtin_amdgcn_rcpf(det);
                    const float t = (U * Az + V * Bz + W * Cz) * inv;
                    const float u = V * inv, v = W * inv;
                    const bool ok = !((mn < 0.f) & (mxw > 0.f)) & (det != 0.f) & (t > 0.f) & (t < h.t);
                    if (ok) { h.t = t; h.u = u; h.v = v; h.slot = slot; }
                };
                // triangle 0 = (A, B, C) = (q0, q1, q2): U = e(B, C), V = e(C, A), W = e(A, B)
                accept(E12, edge2_exact(X2, Y2, X0, Y0), edge2_exact(X0, Y0, X1, Y1), Z0, Z1, Z2, 2 * r);
                // triangle 1 = (A, B, C) = (q3, q2, q1): U = e(q2, q1) = -e(q1, q2), V = e(q1, q3), W = e(q3, q2)
                accept(-E12, edge2_exact(X1, Y1, X3, Y3), edge2_exact(X3, Y3, X2, Y2), Z3, Z2, Z1, 2 * r + 1);
            }
            node = pop();
            return;
        }
#endif
        const int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
        for (int i = first; i < first + cnt; i++) {
            const float4* tp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.tris) + (uint32_t)i * (uint32_t)(16 * kTriQuads));
            float4 v0 = tp[0], e1 = tp[1], e2 = tp[2];
            if (STATS) { n_tris++; if (wave_iters && first_active()) wave_iters[1]++; }
#if TEXIR_TRI_WATERTIGHT
            // (tp[1], tp[2] hold the vertices v1, v2 here, not edges.)  Shear the three vertices into ray space ...
            auto shear = [&](float p0, float p1, float p2, float& X, float& Y, float& Z) __attribute__((always_inline)) {
                const float q0 = p0 - ox, q1 = p1 - oy, q2 = p2 - oz;
                float qz, qx, qy;
                if constexpr (kzc == 0) { qz = q0; qx = q1; qy = q2; }
                else if constexpr (kzc == 1) { qz = q1; qx = q2; qy = q0; }
                else if constexpr (kzc == 2) { qz = q2; qx = q0; qy = q1; }
                else { qz = kz == 0 ? q0 : (kz == 1 ? q1 : q2); qx = kz == 0 ? q1 : (kz == 1 ? q2 : q0); qy = kz == 0 ? q2 : (kz == 1 ? q0 : q1); }
                X = __builtin_fmaf(-Sx, qz, qx); Y = __builtin_fmaf(-Sy, qz, qy); Z = Sz * qz;
            };
            float Ax, Ay, Az, Bx, By, Bz, Cx, Cy, Cz;
            shear(v0.x, v0.y, v0.z, Ax, Ay, Az); shear(e1.x, e1.y, e1.z, Bx, By, Bz); shear(e2.x, e2.y, e2.z, Cx, Cy, Cz);
            // ... 2D edge functions with exact signs (edge2_exact): U, V, W = unnormalised weights of v0, v1, v2
            const float U = edge2_exact(Bx, By, Cx, Cy), V = edge2_exact(Cx, Cy, Ax, Ay), W = edge2_exact(Ax, Ay, Bx, By);
            const float mn = fminf(fminf(U, V), W), mxw = fmaxf(fmaxf(U, V), W);
            const float det = U + V + W;
            const float inv = __builtin_amdgcn_rcpf(det);
            const float t = (U * Az + V * Bz + W * Cz) * inv;
            const float u = V * inv, v = W * inv;
            // inside <=> no two of the signs differ (zeros -- the origin exactly on an edge or vertex -- count as inside for BOTH neighbours)
            const bool ok = !((mn < 0.f) & (mxw > 0.f)) & (det != 0.f) & (t > 0.f) & (t < h.t);
#else
            // Moeller-Trumbore, same operation order as the oracle
            float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
            float det = e1.x * px + e1.y * py + e1.z * pz;
            float inv = __builtin_amdgcn_rcpf(det);
            float tx = ox - v0.x, ty = oy - v0.y, tz = oz - v0.z;
            float u = (tx * px + ty * py + tz * pz) * inv;
            float qx = ty * e1.z - tz * e1.y, qy = tz * e1.x - tx * e1.z, qz = tx * e1.y - ty * e1.x;
            float v = (dx * qx + dy * qy + dz * qz) * inv;
            float t = (e2.x * qx + e2.y * qy + e2.z * qz) * inv;
            bool ok = (det != 0.f) & (u >= 0.f) & (u <= 1.f) & (v >= 0.f) & (u + v <= 1.f) & (t > 0.f) & (t < h.t);
#endif
            if (ok) { h.t = t; h.u = u; h.v = v; h.slot = i; }
        }
        node = pop();
    };
    auto leaf_step = [&]() __attribute__((always_inline)) {
#if TEXIR_TRI_WATERTIGHT && TEXIR_LEAF_UNIFORM_KZ
        if (kz_uniform) {
            if (kz0 == 0) leaf_body(std::integral_constant<int, 0>{});
            else if (kz0 == 1) leaf_body(std::integral_constant<int, 1>{});
            else leaf_body(std::integral_constant<int, 2>{});
        } else
#endif
        leaf_body(std::integral_constant<int, -1>{});
    };
    auto node_step = [&]() __attribute__((always_inline)) { if constexpr (WIDTH == 4) node_step4(); else node_step2(); };

    const int sched_w = sc.sched_weight > 0 ? sc.sched_weight : kSchedNodeWeight;      // (wave-uniform: an SGPR)
    if constexpr (STREAM) {
        // Compaction by refill (SURVEY / north_star: "wavefront ballot / prefix-sum ray compaction").  A lane whose ray is finished does not wait for
        // the slowest ray of the batch: once `refill_at` lanes of the wave are idle (ballot + s_bcnt1) they hand their hit to `next`, which shades and
        // accumulates it and gives the lane its NEXT ray (the caller's next direction cell of the same texel) -- the wave stays full while rays of
        // very different lengths pass through it.  next(finished, h, dx, dy, dz) -> true if the lane has a new ray in (dx, dy, dz).
        // Every ray still performs exactly the same node visits and triangle tests, and a lane accumulates its samples in its own cell order: the
        // sums are the same bits as the lock-step kernel's.
        bool holds = false;                                  // this lane holds a finished ray that has not been handed over yet
        bool more = true;                                    // `next` may still have rays for this lane
        for (;;) {
            const bool at_node = (uint32_t)node < (uint32_t)kSentinel;
            const unsigned long long m_node = __ballot(at_node), m_leaf = __ballot(node < 0);
            const bool idle = node == kSentinel && (holds || more);
            const unsigned long long m_idle = __ballot(idle);
            const bool drained = !(m_node | m_leaf);
            if (drained && !m_idle) break;
            if (drained || __popcll(m_idle) >= refill_at) {
                if (idle) {
                    const bool got = next(holds, h, dx, dy, dz);
                    holds = got; more = got;
                    if (got) { begin_ray(); h.t = __builtin_inff(); h.u = 0.f; h.v = 0.f; h.slot = -1; node = 0; top = base; }
                }
                agree(node != kSentinel);                    // (over the lanes that hold a ray now; lanes without one take part in no step)
                continue;
            }
            if (sched_w * __popcll(m_node) >= __popcll(m_leaf)) { if (at_node) node_step(); }
            else if (node < 0) leaf_step();
        }
        return h;
    } else {
#if TEXIR_SCHED
    for (;;) {
        const bool at_node = (uint32_t)node < (uint32_t)kSentinel;        // an inner node (>= 0 and not the sentinel)
        const unsigned long long m_node = __ballot(at_node), m_leaf = __ballot(node < 0);
        if (!(m_node | m_leaf)) break;
#if TEXIR_CHAIN_PROBE
        probe_add(probe_tick, 1u);
        const bool timed = (probe_tick & (kProbeEvery - 1u)) == 0u;              // wave-uniform
        uint32_t c0 = 0;
        if (timed) c0 = probe_clock();
        if (sched_w * __popcll(m_node) >= __popcll(m_leaf)) {
            if (at_node) node_step();
            const bool scalar = __ballot(at_node && probe_scalar_path) != 0ull;
            if (scalar) probe_add(pn_ns, 1u); else probe_add(pn_nv, 1u);
            if (timed) {
                const uint32_t dt = probe_clock() - c0;
                if (scalar) { probe_add(pc_ns, dt); probe_add(pt_ns, 1u); } else { probe_add(pc_nv, dt); probe_add(pt_nv, 1u); }
            }
        } else {
            if (node < 0) leaf_step();
            probe_add(pn_lf, 1u);
            if (timed) { probe_add(pc_lf, probe_clock() - c0); probe_add(pt_lf, 1u); }
        }
        if (timed) { const uint32_t n0 = probe_clock(); probe_add(pc_null, probe_clock() - n0); probe_add(pt_null, 1u); }
#else
        if (sched_w * __popcll(m_node) >= __popcll(m_leaf)) { if (at_node) node_step(); }
        else if (node < 0) leaf_step();
#endif
    }
#if TEXIR_CHAIN_PROBE
    if (wave_iters) {
        const uint32_t v[kProbeSlots] = {pn_nv, pt_nv, pc_nv, pn_ns, pt_ns, pc_ns, pn_lf, pt_lf, pc_lf, pt_null, pc_null};
        for (int q = 0; q < kProbeSlots; q++) probe_add(wave_iters[2 + q], v[q]);
    }
#endif
#else
    while (node != kSentinel) {
        while (node >= 0 && node != kSentinel) node_step();
        while (node < 0) leaf_step();
    }
#endif
    return h;
    }
}

// closest hit of one ray per lane, run to completion (all lanes of the wave enter and leave together).  One instance per kernel: it
// owns the LDS part of the stacks.  STATS: n_nodes / n_tris count this lane's node fetches and triangle tests; wave_iters[0/1] (if
// given) count, on the first active lane, how many times the wave executed the node-step and the triangle-test bodies.
template <bool STATS, int LSTK = kLdsStack, int WIDTH = 2, bool CULL = false>
__device__ __forceinline__ Hit trace_closest(const SceneDev& sc, float ox, float oy, float oz, float dx, float dy, float dz,
                                             uint32_t& n_nodes, uint32_t& n_tris, uint32_t* wave_iters = nullptr)
{
    return trace_core<STATS, LSTK, WIDTH, CULL, false>(sc, ox, oy, oz, dx, dy, dz, n_nodes, n_tris, wave_iters, 64, NoNext{});
}

// Streamed form: every ray of the lane, the first one included, comes from `next`, which is called for the idle lanes once `refill_at` of them have
// gathered (or the wave has run dry): next(finished, hit, dx, dy, dz) first consumes the finished ray's hit (finished = false on a lane's first call),
// then returns true with the lane's next direction, or false when the lane has no ray left.
template <bool STATS, int LSTK, int WIDTH, bool CULL, typename Next>
__device__ __forceinline__ void trace_stream(const SceneDev& sc, float ox, float oy, float oz, uint32_t& n_nodes, uint32_t& n_tris, uint32_t* wave_iters,
                                             int refill_at, Next&& next)
{
    (void)trace_core<STATS, LSTK, WIDTH, CULL, true>(sc, ox, oy, oz, 0.f, 0.f, 0.f, n_nodes, n_tris, wave_iters, refill_at, next);
}

__device__ __forceinline__ float wave_sum(float x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

}  // namespace texir
