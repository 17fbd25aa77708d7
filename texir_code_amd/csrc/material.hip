// Material-side kernels of libtexir_hip.so:
//   gbuffer_kernel     cube-map G-buffer (position, normal, uv, uv pixel-differentials, mask) by PRIMARY-RAY CASTING through
//                      the same BVH -- replaces nvdiffrast rasterize + interpolate (models/mat_nvdiffrast.py:119-128,
//                      models/tracer_o3d_irt.py:102-108).  Geometry and cameras are constant, so a view's G-buffer is
//                      computed once and cached by the host.
//   mip_build / tex_fetch_fwd / tex_fetch_bwd / mip_fold
//                      nvdiffrast `texture` semantics restated (mat_nvdiffrast.py:131-139): 'linear' = bilinear with wrap,
//                      'linear-mipmap-linear' = 2x2-box mip stack, LOD = log2 of the major axis of the uv_da footprint in
//                      texels, clamped to [0, max level], trilinear; backward scatters into every touched mip and folds
//                      the stack down to level 0.   (nvdiffrast is un-vendored: parity unpinned, see DESIGN.md.)
//   adam_kernel        torch.optim.Adam step fused with the trainer's post-step clamp (train_material.py:448-458).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/texir_hip.h"
#include "device_common.h"
#include "env.h"
#include "kernels.h"

namespace texir {

// ------------------------------------------------------------------------------------------------------------------
// G-buffer by ray casting.  Row-vector convention of the reference: clip = [x,y,z,1] @ mvp (datasets/dataset.py:464-465).
// Pixel (row i, col j) of a face has ndc = ((j+.5)/c*2-1, (i+.5)/c*2-1) (nvdiffrast: row 0 is clip y = -1).
// The ray direction is LINEAR in (x,y) (see launch_gbuffer), which gives exact analytic pixel derivatives of the
// barycentrics -- what nvdiffrast's rast_db / interpolate(diff_attrs='all') provide.
// ------------------------------------------------------------------------------------------------------------------
struct FaceBasis { float eye[3], d0[3], dx[3], dy[3]; };   // dir(ndc x, y) = d0 + x*dx + y*dy, from `eye`
struct GbufArgs {
    FaceBasis face[6];
    const float4* cnrm;     // leaf-ordered corner normals, 3 x float4 per triangle (may be null -> geometric normal)
    int c; int flip_v;
    float *pos, *nrm, *mask, *uv, *uvda; int32_t* tri;
};

template <int WIDTH>
__global__ __launch_bounds__(kBlock) void gbuffer_kernel(SceneDev sc, GbufArgs g)
{
    const int64_t P = (int64_t)6 * g.c * g.c;
    uint32_t cn = 0, ct = 0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < P; p += (int64_t)gridDim.x * kBlock) {
        const int face = (int)(p / ((int64_t)g.c * g.c));
        const int rem = (int)(p - (int64_t)face * g.c * g.c);
        const int i = rem / g.c, j = rem - i * g.c;
        const FaceBasis& fb = g.face[face];
        const float x = ((float)j + 0.5f) / (float)g.c * 2.f - 1.f, y = ((float)i + 0.5f) / (float)g.c * 2.f - 1.f;
        const float ex = fb.eye[0], ey = fb.eye[1], ez = fb.eye[2];
        float dx = fb.d0[0] + x * fb.dx[0] + y * fb.dy[0], dy = fb.d0[1] + x * fb.dx[1] + y * fb.dy[1], dz = fb.d0[2] + x * fb.dx[2] + y * fb.dy[2];
        // d(dir)/dX, d(dir)/dY in PIXEL units (d ndc / d pixel = 2/c)
        const float s2 = 2.f / (float)g.c;
        float dXx = s2 * fb.dx[0], dXy = s2 * fb.dx[1], dXz = s2 * fb.dx[2];
        float dYx = s2 * fb.dy[0], dYy = s2 * fb.dy[1], dYz = s2 * fb.dy[2];
        Hit h = trace_closest<false, kLdsStack / 2, WIDTH, true>(sc, ex, ey, ez, dx, dy, dz, cn, ct);
        float o_pos[3] = {1.f, 0.f, 0.f}, o_n[3] = {1.f, 0.f, 0.f}, o_uv[2] = {0.f, 0.f}, o_da[4] = {0.f, 0.f, 0.f, 0.f};   // bg (mat_nvdiffrast.py:125)
        float m = 0.f; int32_t tri = 0;
        if (h.slot >= 0) {
            const float4* tp = sc.tris + kTriQuads * (size_t)h.slot;
            float4 v0 = tp[0], e1 = tp[1], e2 = tp[2];
#if TEXIR_TRI_WATERTIGHT
            e1.x -= v0.x; e1.y -= v0.y; e1.z -= v0.z; e2.x -= v0.x; e2.y -= v0.y; e2.z -= v0.z;      // the record holds v1, v2
#endif
            m = 1.f; tri = (int32_t)tri_prim(sc, h.slot) + 1;
            const float u = h.u, v = h.v, w = 1.f - u - v;
            o_pos[0] = v0.x + u * e1.x + v * e2.x; o_pos[1] = v0.y + u * e1.y + v * e2.y; o_pos[2] = v0.z + u * e1.z + v * e2.z;
            if (g.cnrm) {
                float4 n0 = g.cnrm[3 * (size_t)h.slot], n1 = g.cnrm[3 * (size_t)h.slot + 1], n2 = g.cnrm[3 * (size_t)h.slot + 2];
                o_n[0] = n0.x * w + n1.x * u + n2.x * v; o_n[1] = n0.y * w + n1.y * u + n2.y * v; o_n[2] = n0.z * w + n1.z * u + n2.z * v;
            } else {
                float nx = e1.y * e2.z - e1.z * e2.y, ny = e1.z * e2.x - e1.x * e2.z, nz = e1.x * e2.y - e1.y * e2.x;
                float il = rsqrtf(nx * nx + ny * ny + nz * nz);
                o_n[0] = nx * il; o_n[1] = ny * il; o_n[2] = nz * il;
            }
            // analytic d(u,v)/d(pixel): [e1 e2 -d] [u' v' t']^T = t * d'  (same Moeller-Trumbore solve, rhs b = t*d')
            float px = dy * e2.z - dz * e2.y, py = dz * e2.x - dx * e2.z, pz = dx * e2.y - dy * e2.x;
            float det = e1.x * px + e1.y * py + e1.z * pz;
            float inv = 1.f / det;
            float du[2], dv[2];
            const float bx[2] = {h.t * dXx, h.t * dYx}, by[2] = {h.t * dXy, h.t * dYy}, bz[2] = {h.t * dXz, h.t * dYz};
            for (int k = 0; k < 2; k++) {
                du[k] = (bx[k] * px + by[k] * py + bz[k] * pz) * inv;
                float qx = by[k] * e1.z - bz[k] * e1.y, qy = bz[k] * e1.x - bx[k] * e1.z, qz = bx[k] * e1.y - by[k] * e1.x;
                dv[k] = (dx * qx + dy * qy + dz * qz) * inv;
            }
            float4 a, b;
            tri_uvs(sc, h.slot, a, b);
            float a1x = a.z - a.x, a1y = a.w - a.y, a2x = b.x - a.x, a2y = b.y - a.y;
            o_uv[0] = a.x * w + a.z * u + b.x * v; o_uv[1] = a.y * w + a.w * u + b.y * v;
            o_da[0] = a1x * du[0] + a2x * dv[0]; o_da[1] = a1x * du[1] + a2x * dv[1];        // du/dX, du/dY
            o_da[2] = a1y * du[0] + a2y * dv[0]; o_da[3] = a1y * du[1] + a2y * dv[1];        // dv/dX, dv/dY
            if (g.flip_v) { o_uv[1] = 1.f - o_uv[1]; o_da[2] = -o_da[2]; o_da[3] = -o_da[3]; }
        }
        for (int k = 0; k < 3; k++) { g.pos[3 * p + k] = o_pos[k]; g.nrm[3 * p + k] = o_n[k]; }
        g.mask[p] = m; g.tri[p] = tri;
        g.uv[2 * p] = o_uv[0]; g.uv[2 * p + 1] = o_uv[1];
        for (int k = 0; k < 4; k++) g.uvda[4 * p + k] = o_da[k];
    }
}

static bool invert4(const double* m, double* out)
{
    double a[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = m[4 * r + c]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col; double best = fabs(a[col][col]);
        for (int r = col + 1; r < 4; r++) if (fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
        if (best == 0.0) return false;
        if (piv != col) for (int c = 0; c < 8; c++) { double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
        double d = a[col][col];
        for (int c = 0; c < 8; c++) a[col][c] /= d;
        for (int r = 0; r < 4; r++) if (r != col) { double f = a[r][col]; if (f != 0.0) for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[4 * r + c] = a[r][4 + c];
    return true;
}

// mvp [6,4,4] HOST, row-vector convention.  With Minv = inverse(mvp) (rows r0..r3): eye_h = r2 (the pre-image of clip
// (0,0,1,0)), and for a pixel at ndc (x,y) the world point at ANY depth z is X_h = x r0 + y r1 + z r2 + r3, so
//   dir(x,y) = X_h.xyz - eye * X_h.w = x A0 + y A1 + A3,   Ak = rk.xyz - eye * rk.w     (the z term cancels exactly).
// Evaluated in double on the host: in float32 X_h.w suffers catastrophic cancellation for n = 1e-4, f = 100.
hipError_t launch_gbuffer(const SceneDev& sc, const float* mvp_host, const float4* cnrm, int c, int flip_v, float* pos, float* nrm, float* mask,
                          float* uv, float* uvda, int32_t* tri, hipStream_t st)
{
    GbufArgs g;
    for (int f = 0; f < 6; f++) {
        double m[16], mi[16];
        for (int k = 0; k < 16; k++) m[k] = (double)mvp_host[16 * f + k];
        if (!invert4(m, mi)) return hipErrorInvalidValue;
        const double* r0 = mi, *r1 = mi + 4, *r2 = mi + 8, *r3 = mi + 12;
        if (r2[3] == 0.0) return hipErrorInvalidValue;               // not a perspective camera
        double eye[3] = {r2[0] / r2[3], r2[1] / r2[3], r2[2] / r2[3]};
        double wfar = r2[3] + r3[3];                                 // X_h.w of the face centre on the far plane
        double sgn = wfar < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < 3; k++) {
            g.face[f].eye[k] = (float)eye[k];
            g.face[f].dx[k] = (float)(sgn * (r0[k] - eye[k] * r0[3]));
            g.face[f].dy[k] = (float)(sgn * (r1[k] - eye[k] * r1[3]));
            g.face[f].d0[k] = (float)(sgn * (r3[k] - eye[k] * r3[3]));
        }
        // normalise the scale of the basis (direction length is irrelevant; keeps t in sane float range)
        double len = sqrt((double)g.face[f].d0[0] * g.face[f].d0[0] + (double)g.face[f].d0[1] * g.face[f].d0[1] + (double)g.face[f].d0[2] * g.face[f].d0[2]);
        if (len > 0.0) for (int k = 0; k < 3; k++) { g.face[f].d0[k] = (float)(g.face[f].d0[k] / len); g.face[f].dx[k] = (float)(g.face[f].dx[k] / len); g.face[f].dy[k] = (float)(g.face[f].dy[k] / len); }
    }
    g.cnrm = cnrm; g.c = c; g.flip_v = flip_v; g.pos = pos; g.nrm = nrm; g.mask = mask; g.uv = uv; g.uvda = uvda; g.tri = tri;
    int64_t P = (int64_t)6 * c * c;
    int64_t nb = (P + kBlock - 1) / kBlock;
    if (sc.nodes4) hipLaunchKernelGGL(gbuffer_kernel<4>, dim3((int)(nb > 2048 ? 2048 : nb)), dim3(kBlock), 0, st, sc, g);
    else hipLaunchKernelGGL(gbuffer_kernel<2>, dim3((int)(nb > 2048 ? 2048 : nb)), dim3(kBlock), 0, st, sc, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// mip stack + texture fetch
// ------------------------------------------------------------------------------------------------------------------
// Level 0 IS the caller's texture (no copy); levels 1.. live in a separate "rest" buffer: level l at rest + off[l].
struct MipDesc { int H, W, C, levels; int64_t off[16]; };

static MipDesc make_desc(int H, int W, int C, int levels)
{
    MipDesc d; d.H = H; d.W = W; d.C = C; d.levels = levels;
    int64_t o = 0;
    d.off[0] = 0;
    for (int l = 1; l < 16; l++) { d.off[l] = o; if (l < levels) o += (int64_t)(H >> l) * (W >> l) * C; }
    return d;
}

int mip_levels(int H, int W, int max_mip_level)
{
    int l = 1;   // level 0
    while (l <= max_mip_level && l < 16 && ((H >> (l - 1)) % 2 == 0) && ((W >> (l - 1)) % 2 == 0) && (H >> l) >= 1 && (W >> l) >= 1) l++;
    return l;
}

// elements of levels 1 .. levels-1 (the "rest" buffer); at least 1 so that callers can always allocate
int64_t mip_total_elems(int H, int W, int C, int levels)
{
    int64_t o = 0;
    for (int l = 1; l < levels; l++) o += (int64_t)(H >> l) * (W >> l) * C;
    return o > 0 ? o : 1;
}

// one thread per output float; rows come from blockIdx.y, so the index math is 32-bit with a compile-time C (no 64-bit div/mod)
// and consecutive lanes read consecutive floats of the two source rows
template <int C>
__global__ __launch_bounds__(256) void mip_down_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hd, int Wd)
{
    const int e = blockIdx.x * 256 + threadIdx.x;           // element inside an output row: texel * C + channel
    if (e >= Wd * C) return;
    const int tx = e / C, ch = e - tx * C;
    for (int y = blockIdx.y; y < Hd; y += gridDim.y) {
        const float* r0 = src + ((size_t)(2 * y) * (2 * Wd) + 2 * tx) * C + ch;
        const float* r1 = r0 + (size_t)(2 * Wd) * C;
        dst[(size_t)y * Wd * C + e] = 0.25f * (r0[0] + r0[C] + r1[0] + r1[C]);
    }
}

// all remaining (small) levels in one single-block launch: level l from level l-1, block barrier in between
__device__ __forceinline__ void mip_down_tail_body(float* __restrict__ rest, const MipDesc& d, int l_begin)
{
    for (int l = l_begin; l < d.levels; l++) {
        const int Hd = d.H >> l, Wd = d.W >> l, C = d.C, Ws = Wd * 2;
        const float* src = rest + d.off[l - 1];
        float* dst = rest + d.off[l];
        const int n = Hd * Wd * C;
        for (int g = threadIdx.x; g < n; g += 1024) {
            int ch = g % C, t = g / C, x = t % Wd, y = t / Wd;
            const float* s = src + ((2 * y) * Ws + 2 * x) * C + ch;
            dst[g] = 0.25f * (s[0] + s[C] + s[Ws * C] + s[Ws * C + C]);
        }
        __threadfence_block();
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void mip_down_tail_kernel(float* __restrict__ rest, MipDesc d, int l_begin) { mip_down_tail_body(rest, d, l_begin); }

// fold: grad[l-1][2y+a][2x+b] += 0.25 * grad[l][y][x].  One thread per FINE float (coalesced read-modify-write of the fine row),
// rows from blockIdx.y, 32-bit index math with compile-time C.
template <int C>
__global__ __launch_bounds__(256) void mip_fold_kernel(float* __restrict__ fine, const float* __restrict__ coarse, int Hf, int Wf)
{
    const int e = blockIdx.x * 256 + threadIdx.x;           // element inside a fine row
    if (e >= Wf * C) return;
    const int tx = e / C, ch = e - tx * C;
    for (int y = blockIdx.y; y < Hf; y += gridDim.y)
        fine[(size_t)y * Wf * C + e] = __builtin_fmaf(0.25f, coarse[((size_t)(y >> 1) * (Wf >> 1) + (tx >> 1)) * C + ch], fine[(size_t)y * Wf * C + e]);
}

__global__ __launch_bounds__(1024) void mip_fold_tail_kernel(float* __restrict__ rest, MipDesc d, int l_end /* fold levels-1 .. l_end+1 into l_end */)
{
    for (int l = d.levels - 1; l > l_end; l--) {
        const int Hf = d.H >> (l - 1), Wf = d.W >> (l - 1), C = d.C;
        float* fine = rest + d.off[l - 1];
        const float* coarse = rest + d.off[l];
        const int n = Hf * Wf * C;
        for (int g = threadIdx.x; g < n; g += 1024) {
            int ch = g % C, t = g / C, x = t % Wf, y = t / Wf;
            fine[g] += 0.25f * coarse[((y >> 1) * (Wf >> 1) + (x >> 1)) * C + ch];
        }
        __threadfence_block();
        __syncthreads();
    }
}


// Several mip levels per launch.  A block owns a 32x32-texel tile of the SOURCE level s and produces its 16x16 / 8x8 / ... / 1x1
// descendants (levels s+1 .. s+5) through LDS; the levels above (at most 32^2 texels) are left to the single-block tail kernel.
// Same arithmetic as one mip_down_kernel launch per level (0.25 * (((a + b) + c) + d)), so the stack is bit-identical.
template <int C>
__device__ __forceinline__ void mip_pyr_down_body(const float* __restrict__ src, float* __restrict__ rest, const MipDesc& d, int s, int n_out, int bid,
                                                  float (*buf)[16 * 16 * C] /* LDS: 2 x 16*16*C floats */)
{
    const int Hs = d.H >> s, Ws = d.W >> s;
    const int tiles_x = (Ws + 31) / 32;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    // level s+1: 16x16 outputs straight from global memory, one thread per output texel
    {
        const int oy = threadIdx.x >> 4, ox = threadIdx.x & 15;
        const int Y = ty * 16 + oy, X = tx * 16 + ox, Hd = Hs >> 1, Wd = Ws >> 1;
        if (Y < Hd && X < Wd) {
            const float* r0 = src + ((size_t)(2 * Y) * Ws + 2 * X) * C;
            const float* r1 = r0 + (size_t)Ws * C;
            float* o = rest + d.off[s + 1] + ((size_t)Y * Wd + X) * C;
#pragma unroll
            for (int c = 0; c < C; c++) { const float v = 0.25f * (r0[c] + r0[C + c] + r1[c] + r1[C + c]); o[c] = v; buf[0][(oy * 16 + ox) * C + c] = v; }
        }
    }
    int cur = 0;
    for (int k = 2; k <= n_out; k++) {
        __syncthreads();
        const int n = 32 >> k;                                  // outputs per tile side at level s+k
        const int Hd = Hs >> k, Wd = Ws >> k;
        if ((int)threadIdx.x < n * n) {
            const int oy = threadIdx.x / n, ox = threadIdx.x - oy * n;
            const int Y = ty * n + oy, X = tx * n + ox;
            if (Y < Hd && X < Wd) {
                const float* b = buf[cur] + ((2 * oy) * (2 * n) + 2 * ox) * C;
                float* o = rest + d.off[s + k] + ((size_t)Y * Wd + X) * C;
#pragma unroll
                for (int c = 0; c < C; c++) { const float v = 0.25f * (b[c] + b[C + c] + b[2 * n * C + c] + b[2 * n * C + C + c]); o[c] = v; buf[cur ^ 1][(oy * n + ox) * C + c] = v; }
            }
        }
        cur ^= 1;
    }
}

template <int C>
__global__ __launch_bounds__(256) void mip_pyr_down_kernel(const float* __restrict__ src, float* __restrict__ rest, MipDesc d, int s, int n_out)
{
    __shared__ float buf[2][16 * 16 * C];
    mip_pyr_down_body<C>(src, rest, d, s, n_out, blockIdx.x, buf);
}

// The folds of a whole gradient stack in one launch: level f (the finest one to fold INTO) += 0.25 * level f+1 += 0.25 * level f+2 ...
// A block owns a 32x32 tile of level f.  It first walks its chain of ancestors down from the top level (one texel per level: the
// same fused multiply-adds the level-by-level kernels perform on those texels), then folds its own sub-pyramid (levels f+5 .. f+1)
// through LDS and finally read-modify-writes its tile of level f.  Only level f is written back: nothing reads the coarser
// gradient levels afterwards.  Bit-identical to mip_fold_tail_kernel + one mip_fold_kernel launch per level.
// `mask` (nullable): one bit per texel of the `rest` stack (bit t & 31 of word t >> 5, t = d.off[l] / C + y * W_l + x).  With a mask the stack is NEVER
// cleared between steps: only the texels a view's tap lists wrote this step (their bit is set) hold values, every other texel counts as zero --
// reads of levels f+1 .. top go through the mask, and level f itself is WRITTEN (own masked value + a quarter of the parent) instead of
// read-modify-written.  Same floats as folding a zero-filled stack.
template <int C>
__device__ __forceinline__ float rest_read(const float* __restrict__ rest, const uint32_t* __restrict__ mask, int64_t off, int64_t texel, int c)
{
    // (the value is loaded whether its bit is set or not and SELECTED afterwards: a branch on the mask word would put every value load behind its own
    // mask load -- a block of the fold issues ~20 of these per thread, and as dependent round trips they were the whole launch, 23 us.  What an unset
    // texel holds is never used in arithmetic.)
    const float val = rest[off + texel * C + c];
    if (mask) {
        const int64_t t = off / C + texel;
        return ((mask[t >> 5] >> (t & 31)) & 1u) ? val : 0.f;
    }
    return val;
}

// The tile of a stack with at least five levels above f whose level f is made of whole 32 x 32 tiles with 16-byte aligned rows (every texture that matters): the
// same fused multiply-adds as the general body below, but
//   * EVERY global load of the tile -- the ancestors' texels, the tile's own texels of levels f+4 .. f+1 and of level f, through the mask where there is one --
//     is issued before the first barrier (the general form walks the levels with a load + barrier each);
//   * level f, where the bytes are, moves as float4: C loads and C stores per thread instead of 4C.  On gfx9 a store holds its address / data registers until it
//     completes, and the compiler reuses them for the next one: the twelve scalar stores of the 3-channel tile ran as twelve dependent round trips (an
//     `s_waitcnt vmcnt(0)` in front of each, see the ISA) -- 17 us for 12 MB against 8 us for the 1-channel texture's 4 MB.
template <int C>
__device__ __forceinline__ void mip_pyr_fold_tile5(float* __restrict__ fine_base, const float* __restrict__ rest, const MipDesc& d, int f,
                                                   const uint32_t* __restrict__ mask, int ty, int tx, float (*buf)[16 * 16 * C])
{
    const int tid = threadIdx.x, top = d.levels - 1;
    const int Wf = d.W >> f;
    const bool masked_f = mask != nullptr && f >= 1;
    // element i of the tile's n x n x C block of level f+k (n = 32 >> k, k = 1 .. 4): channel, texel inside the level, in bounds?
    auto locate = [&](int k, int i, int& c, int& ox, int& oy, int64_t& texel) -> bool {
        const int n = 32 >> k, Hl = d.H >> (f + k), Wl = d.W >> (f + k);
        c = i % C; const int t = i / C; ox = t % n; oy = t / n;
        const int Y = ty * n + oy, X = tx * n + ox;
        texel = (int64_t)Y * Wl + X;
        return i < n * n * C && Y < Hl && X < Wl;
    };
    auto load = [&](int k, int i) -> float {
        int c, ox, oy; int64_t texel;
        return locate(k, i, c, ox, oy, texel) ? rest_read<C>(rest, mask, d.off[f + k], texel, c) : 0.f;
    };
    // ---- all loads ----
    // (ancestors: thread j * C + c holds the tile's texel of level top - j, channel c; they meet in LDS below)
    float anc = 0.f;
    if (tid < 16 * C) {
        const int j = tid / C, l = top - j;
        if (l >= f + 5) {
            const int sh = l - f - 5;                           // tile coordinate -> texel of level l (tile = 32 texels of level f = 1 texel of level f+5)
            anc = rest_read<C>(rest, mask, d.off[l], (int64_t)(ty >> sh) * (d.W >> l) + (tx >> sh), tid - j * C);
        }
    }
    const float l4 = load(4, tid), l3 = load(3, tid), l2 = load(2, tid);
    float l1[C];
#pragma unroll
    for (int j = 0; j < C; j++) l1[j] = load(1, tid + 256 * j);
    // level f: the tile is 32 rows of 32 * C floats = 8 * C float4; thread tid owns float4 number tid + 256 * j (j < C) of the tile
    float4 own[C];
    uint32_t ownbits = 0xffffu;                                  // bit 4 * j + m: element m of float4 j holds a value of this step (mask)
    const int64_t f_off = d.off[f] / C;
#pragma unroll
    for (int j = 0; j < C; j++) {
        const int q = tid + 256 * j, row = q / (8 * C), e0 = (q - row * (8 * C)) * 4;
        const int64_t texel0 = (int64_t)(ty * 32 + row) * Wf + tx * 32;
        own[j] = *reinterpret_cast<const float4*>(fine_base + texel0 * C + e0);
        if (masked_f) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int64_t tt = f_off + texel0 + (e0 + m) / C;
                if (!((mask[tt >> 5] >> (tt & 31)) & 1u)) ownbits &= ~(1u << (4 * j + m));
            }
        }
    }
    // ---- the chain, top down ----
    if (tid < 16 * C) buf[1][tid] = anc;
    __syncthreads();
    if (tid < C) {
        float acc = buf[1][tid];
        for (int j = 1; j < 16; j++) if (top - j >= f + 5) acc = __builtin_fmaf(0.25f, acc, buf[1][j * C + tid]);
        buf[0][tid] = acc;                                       // folded level f+5: the tile's one texel there
    }
    int cur = 0;
    auto fold_level = [&](int k, int i, float val) {
        int c, ox, oy; int64_t texel;
        const int n = 32 >> k;
        if (i < n * n * C) {
            const bool in = locate(k, i, c, ox, oy, texel);
            buf[cur ^ 1][i] = in ? __builtin_fmaf(0.25f, buf[cur][((oy >> 1) * (n >> 1) + (ox >> 1)) * C + c], val) : 0.f;
        }
    };
    __syncthreads(); fold_level(4, tid, l4); cur ^= 1;
    __syncthreads(); fold_level(3, tid, l3); cur ^= 1;
    __syncthreads(); fold_level(2, tid, l2); cur ^= 1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C; j++) fold_level(1, tid + 256 * j, l1[j]);
    cur ^= 1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C; j++) {
        const int q = tid + 256 * j, row = q / (8 * C), e0 = (q - row * (8 * C)) * 4;
        const int64_t texel0 = (int64_t)(ty * 32 + row) * Wf + tx * 32;
        const float o4[4] = {own[j].x, own[j].y, own[j].z, own[j].w};
        float r4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int ox = (e0 + m) / C, c = (e0 + m) - ox * C;
            const float mine = ((ownbits >> (4 * j + m)) & 1u) ? o4[m] : 0.f;
            r4[m] = __builtin_fmaf(0.25f, buf[cur][((row >> 1) * 16 + (ox >> 1)) * C + c], mine);
        }
        *reinterpret_cast<float4*>(fine_base + texel0 * C + e0) = make_float4(r4[0], r4[1], r4[2], r4[3]);
    }
}

template <int C>
__device__ __forceinline__ void mip_pyr_fold_body(float* __restrict__ fine_base /* level f */, const float* __restrict__ rest, const MipDesc& d, int f,
                                                  const uint32_t* __restrict__ mask, int bid, float (*buf)[16 * 16 * C])
{
    const int Hf = d.H >> f, Wf = d.W >> f;
    const int tiles_x = (Wf + 31) / 32;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int top = d.levels - 1;
    const int K = min(5, top - f);                              // levels f+1 .. f+K live inside the tile
    // (whole tiles, 16-byte aligned rows and level base: the float4 form)
    if (K == 5 && (Hf & 31) == 0 && (Wf & 31) == 0 && ((uintptr_t)fine_base & 15) == 0) { mip_pyr_fold_tile5<C>(fine_base, rest, d, f, mask, ty, tx, buf); return; }
    // ancestors: levels top .. f+K (one texel each for this tile), folded from the top down by the first C threads
    if (K == 5 && (int)threadIdx.x < C) {
        const int c = threadIdx.x;
        // (all the loads first -- they are independent -- then the dependent chain of fused multiply-adds)
        float gl[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int l = top - j;
            gl[j] = 0.f;
            if (l >= f + 5) {
                const int sh = l - f - 5;                       // tile coordinate -> texel of level l (tile = 32 texels of level f = 1 texel of level f+5)
                gl[j] = rest_read<C>(rest, mask, d.off[l], (int64_t)(ty >> sh) * (d.W >> l) + (tx >> sh), c);
            }
        }
        float acc = gl[0];
#pragma unroll
        for (int j = 1; j < 16; j++) if (top - j >= f + 5) acc = __builtin_fmaf(0.25f, acc, gl[j]);
        buf[0][c] = acc;                                         // folded level f+5: the tile's one texel there
    }
    // (K < 5 only for stacks with fewer than six levels above f: then level f+K is the top level and the tile covers all of it)
    int cur = 0;
    if (K < 5) {
        __syncthreads();
        const int n = 32 >> K, Hl = d.H >> (f + K), Wl = d.W >> (f + K);
        for (int i = threadIdx.x; i < n * n * C; i += 256) {
            const int c = i % C, t = i / C, ox = t % n, oy = t / n;
            const int Y = ty * n + oy, X = tx * n + ox;
            buf[1][i] = (Y < Hl && X < Wl) ? rest_read<C>(rest, mask, d.off[f + K], (int64_t)Y * Wl + X, c) : 0.f;
        }
        cur = 1;
    }
    for (int k = K - 1; k >= 1; k--) {
        __syncthreads();
        const int n = 32 >> k, Hl = d.H >> (f + k), Wl = d.W >> (f + k);
        for (int i = threadIdx.x; i < n * n * C; i += 256) {
            const int c = i % C, t = i / C, ox = t % n, oy = t / n;
            const int Y = ty * n + oy, X = tx * n + ox;
            float v = 0.f;
            if (Y < Hl && X < Wl) v = __builtin_fmaf(0.25f, buf[cur][((oy >> 1) * (n >> 1) + (ox >> 1)) * C + c], rest_read<C>(rest, mask, d.off[f + k], (int64_t)Y * Wl + X, c));
            buf[cur ^ 1][i] = v;
        }
        cur ^= 1;
    }
    __syncthreads();
    // level f: 32x32 texels, read-modify-write -- or, under a mask, a plain write (when K == 0 there is nothing above f: the launcher does not call us)
    const bool masked_f = mask != nullptr && f >= 1;                    // (level 0 is not part of the `rest` stack: never masked)
    for (int i = threadIdx.x; i < 32 * 32 * C; i += 256) {
        const int c = i % C, t = i / C, ox = t & 31, oy = t >> 5;
        const int Y = ty * 32 + oy, X = tx * 32 + ox;
        if (Y < Hf && X < Wf) {
            float* o = fine_base + ((size_t)Y * Wf + X) * C + c;
            float own = *o;
            if (masked_f) {
                const int64_t tt = d.off[f] / C + (int64_t)Y * Wf + X;
                own = ((mask[tt >> 5] >> (tt & 31)) & 1u) ? own : 0.f;
            }
            *o = __builtin_fmaf(0.25f, buf[cur][((oy >> 1) * 16 + (ox >> 1)) * C + c], own);
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void mip_pyr_fold_kernel(float* __restrict__ fine_base /* level f */, const float* __restrict__ rest, MipDesc d, int f)
{
    __shared__ float buf[2][16 * 16 * C];
    mip_pyr_fold_body<C>(fine_base, rest, d, f, nullptr, blockIdx.x, buf);
}

__device__ __forceinline__ float mip_level_from_da(float4 da, int W, int H, int maxl)
{
    float dsdx = da.x * (float)W, dsdy = da.y * (float)W, dtdx = da.z * (float)H, dtdy = da.w * (float)H;
    float A = dsdx * dsdx + dtdx * dtdx, B = dsdy * dsdy + dtdy * dtdy, Cc = dsdx * dsdy + dtdx * dtdy;
    float l2b = 0.5f * (A + B), l2n = 0.25f * (A - B) * (A - B) + Cc * Cc;
    float major = l2b + sqrtf(l2n);
    float lv = 0.5f * log2f(major);
    return fminf(fmaxf(lv, 0.f), (float)maxl);       // log2(0) = -inf clamps to 0
}

struct Tap { int64_t i00, i10, i01, i11; float w00, w10, w01, w11; };

__device__ __forceinline__ Tap bilinear_wrap(float u, float v, int W, int H)
{
    u = u - floorf(u); v = v - floorf(v);                      // boundary_mode='wrap'
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 += W;
    if (y0 < 0) y0 += H;
    if (x1 >= W) x1 -= W;
    if (y1 >= H) y1 -= H;
    Tap t;
    t.i00 = (int64_t)y0 * W + x0; t.i10 = (int64_t)y0 * W + x1; t.i01 = (int64_t)y1 * W + x0; t.i11 = (int64_t)y1 * W + x1;
    t.w00 = (1.f - fx) * (1.f - fy); t.w10 = fx * (1.f - fy); t.w01 = (1.f - fx) * fy; t.w11 = fx * fy;
    return t;
}

template <bool BWD>
__device__ __forceinline__ void tex_fetch_body(float* __restrict__ lvl0, float* __restrict__ rest, const MipDesc& d, const float* __restrict__ uv,
                                               const float* __restrict__ uvda, int trilinear, int64_t P, float* __restrict__ io, int bid, int nb)
{
    const int C = d.C;
    for (int64_t p = (int64_t)bid * 256 + threadIdx.x; p < P; p += (int64_t)nb * 256) {
        const float u = uv[2 * p], v = uv[2 * p + 1];
        int l0 = 0, l1 = 0; float f = 0.f;
        if (trilinear && d.levels > 1) {
            float4 da = *reinterpret_cast<const float4*>(uvda + 4 * p);
            float lv = mip_level_from_da(da, d.W, d.H, d.levels - 1);
            l0 = (int)floorf(lv); l1 = min(l0 + 1, d.levels - 1); f = lv - (float)l0;
        }
        float out[4] = {0.f, 0.f, 0.f, 0.f};
        for (int pass = 0; pass < 2; pass++) {
            const int l = pass ? l1 : l0;
            const float wl = pass ? f : 1.f - f;
            if (wl == 0.f) continue;
            Tap t = bilinear_wrap(u, v, d.W >> l, d.H >> l);
            float* base = l == 0 ? lvl0 : rest + d.off[l];
            for (int ch = 0; ch < C; ch++) {
                if (BWD) {
                    float g = io[(int64_t)C * p + ch] * wl;
                    if (g != 0.f) {
                        atomicAdd(base + t.i00 * C + ch, g * t.w00); atomicAdd(base + t.i10 * C + ch, g * t.w10);
                        atomicAdd(base + t.i01 * C + ch, g * t.w01); atomicAdd(base + t.i11 * C + ch, g * t.w11);
                    }
                } else {
                    out[ch] += wl * (base[t.i00 * C + ch] * t.w00 + base[t.i10 * C + ch] * t.w10 + base[t.i01 * C + ch] * t.w01 + base[t.i11 * C + ch] * t.w11);
                }
            }
        }
        if (!BWD) for (int ch = 0; ch < C; ch++) io[(int64_t)C * p + ch] = out[ch];
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void tex_fetch_kernel(float* __restrict__ lvl0, float* __restrict__ rest, MipDesc d, const float* __restrict__ uv,
                                                        const float* __restrict__ uvda, int trilinear, int64_t P, float* __restrict__ io)
{
    tex_fetch_body<BWD>(lvl0, rest, d, uv, uvda, trilinear, P, io, blockIdx.x, gridDim.x);
}

static int grid1d(int64_t n, int bs) { int64_t nb = (n + bs - 1) / bs; return (int)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb)); }

constexpr int kTailElems = 16 * 16 * 4;     // levels with at most this many elements are handled by the single-block tail kernels (measured: 64^2 1.36, 32^2 1.31, 16^2 1.29, 8^2 1.29, none 1.31 ms per material step)

static int tail_begin(const MipDesc& d)
{
    int l = 1;
    while (l < d.levels && (int64_t)(d.H >> l) * (d.W >> l) * d.C > kTailElems) l++;
    return l;        // first level produced by the tail kernel (>= 1)
}

static void launch_down(const float* src, float* dst, int Hd, int Wd, int C, hipStream_t st)
{
    dim3 grid((Wd * C + 255) / 256, Hd > 4096 ? 4096 : Hd);
    if (C == 1) hipLaunchKernelGGL(mip_down_kernel<1>, grid, dim3(256), 0, st, src, dst, Hd, Wd);
    else if (C == 2) hipLaunchKernelGGL(mip_down_kernel<2>, grid, dim3(256), 0, st, src, dst, Hd, Wd);
    else if (C == 3) hipLaunchKernelGGL(mip_down_kernel<3>, grid, dim3(256), 0, st, src, dst, Hd, Wd);
    else hipLaunchKernelGGL(mip_down_kernel<4>, grid, dim3(256), 0, st, src, dst, Hd, Wd);
}

static void launch_fold(float* fine, const float* coarse, int Hf, int Wf, int C, hipStream_t st)
{
    dim3 grid((Wf * C + 255) / 256, Hf > 4096 ? 4096 : Hf);
    if (C == 1) hipLaunchKernelGGL(mip_fold_kernel<1>, grid, dim3(256), 0, st, fine, coarse, Hf, Wf);
    else if (C == 2) hipLaunchKernelGGL(mip_fold_kernel<2>, grid, dim3(256), 0, st, fine, coarse, Hf, Wf);
    else if (C == 3) hipLaunchKernelGGL(mip_fold_kernel<3>, grid, dim3(256), 0, st, fine, coarse, Hf, Wf);
    else hipLaunchKernelGGL(mip_fold_kernel<4>, grid, dim3(256), 0, st, fine, coarse, Hf, Wf);
}

// ---- atomics-free texture backward for FIXED (uv, uv_da): the taps of a view never change, so they are enumerated once, sorted by
// texel on the host side (torch.sort) and replayed as a gather: one thread per touched texel sums its (pixel, weight) list in a fixed
// order -- deterministic, and ~10x faster than the float-atomic scatter of tex_fetch_kernel<true>.
// key = unified texel index over [level 0 | levels 1.. in `rest` order]; skipped taps (blend weight 0) get key -1.
__global__ __launch_bounds__(256) void tex_taps_kernel(MipDesc d, const float* __restrict__ uv, const float* __restrict__ uvda, int trilinear,
                                                       int64_t P, long long* __restrict__ keys, float* __restrict__ weights)
{
    const int64_t n0 = (int64_t)d.H * d.W;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (int64_t)gridDim.x * 256) {
        const float u = uv[2 * p], v = uv[2 * p + 1];
        int l0 = 0, l1 = 0; float f = 0.f;
        if (trilinear && d.levels > 1) {
            float4 da = *reinterpret_cast<const float4*>(uvda + 4 * p);
            float lv = mip_level_from_da(da, d.W, d.H, d.levels - 1);
            l0 = (int)floorf(lv); l1 = min(l0 + 1, d.levels - 1); f = lv - (float)l0;
        }
        for (int pass = 0; pass < 2; pass++) {
            const int l = pass ? l1 : l0;
            const float wl = pass ? f : 1.f - f;
            long long* k = keys + 8 * p + 4 * pass;
            float* w = weights + 8 * p + 4 * pass;
            if (wl == 0.f) { k[0] = k[1] = k[2] = k[3] = -1; w[0] = w[1] = w[2] = w[3] = 0.f; continue; }
            Tap t = bilinear_wrap(u, v, d.W >> l, d.H >> l);
            const int64_t base = l == 0 ? 0 : n0 + d.off[l] / d.C;
            k[0] = base + t.i00; k[1] = base + t.i10; k[2] = base + t.i01; k[3] = base + t.i11;
            w[0] = wl * t.w00; w[1] = wl * t.w10; w[2] = wl * t.w01; w[3] = wl * t.w11;
        }
    }
}

template <int C>
__device__ __forceinline__ void tex_gather_body(float* __restrict__ lvl0, float* __restrict__ rest, int64_t n0, const long long* __restrict__ seg_key,
                                                const int* __restrict__ seg_start, const int* __restrict__ seg_count, int n_seg,
                                                const int* __restrict__ pix, const float* __restrict__ w, const float* __restrict__ d_out, int bid, int nb,
                                                const float* __restrict__ d_out2 = nullptr /* a second gradient of the same fetch output, added on the fly */)
{
    constexpr int R = 8;                                         // taps per round
    for (int s = bid * 256 + threadIdx.x; s < n_seg; s += nb * 256) {
        const long long key = seg_key[s];
        const int b = seg_start[s], e = b + seg_count[s];    // (all three loads before the first branch: one round trip, not three)
        if (key < n0 && !lvl0) continue;                     // (caller passed no level-0 buffer: it promised that no list samples level 0)
        float acc[C];
        for (int c = 0; c < C; c++) acc[c] = 0.f;
        // The launch lasts as long as its LONGEST list (one thread per touched texel; median 2 taps, 99.9 % below 40, a few coarse-level texels near 100 --
        // profiles/r04/tap_level_probe.txt), and a tap is two dependent loads (index + weight, then the gathered gradient).  Eight taps' loads are in flight per
        // round (a short tail re-reads the last tap), and the next round's indices and weights are requested BEFORE this round's gradient values are consumed:
        // one round trip per eight taps instead of two per four.  The fused multiply-adds stay in list order, one per real tap: the sum keeps its bits.
        int pp[R]; float ww[R];
#pragma unroll
        for (int k = 0; k < R; k++) { const int ik = min(b + k, e - 1); pp[k] = pix[ik]; ww[k] = w[ik]; }
        for (int i = b; i < e; i += R) {
            const int n = e - i;                                   // taps of this round: R, or fewer in the last one
            float v[R][C];
#pragma unroll
            for (int k = 0; k < R; k++)
#pragma unroll
                for (int c = 0; c < C; c++) v[k][c] = d_out[(int64_t)pp[k] * C + c];
            // (two consumers of one fetch output -- the specular term and the loss both read the roughness -- send two gradients: their sum, the float
            // autograd's add kernel would have written, is formed here instead)
            if (d_out2) {
#pragma unroll
                for (int k = 0; k < R; k++)
#pragma unroll
                    for (int c = 0; c < C; c++) v[k][c] += d_out2[(int64_t)pp[k] * C + c];
            }
            float wc[R];
#pragma unroll
            for (int k = 0; k < R; k++) wc[k] = ww[k];
            if (i + R < e) {
#pragma unroll
                for (int k = 0; k < R; k++) { const int ik = min(i + R + k, e - 1); pp[k] = pix[ik]; ww[k] = w[ik]; }
            }
#pragma unroll
            for (int c = 0; c < C; c++) {
                float a = acc[c];
#pragma unroll
                for (int k = 0; k < R; k++) if (k < n) a = __builtin_fmaf(v[k][c], wc[k], a);
                acc[c] = a;
            }
        }
        float* o = key < n0 ? lvl0 + key * C : rest + (key - n0) * C;
        for (int c = 0; c < C; c++) o[c] = acc[c];
    }
}

template <int C>
__global__ __launch_bounds__(256) void tex_gather_kernel(float* __restrict__ lvl0, float* __restrict__ rest, int64_t n0, const long long* __restrict__ seg_key,
                                                         const int* __restrict__ seg_start, const int* __restrict__ seg_count, int n_seg,
                                                         const int* __restrict__ pix, const float* __restrict__ w, const float* __restrict__ d_out)
{
    tex_gather_body<C>(lvl0, rest, n0, seg_key, seg_start, seg_count, n_seg, pix, w, d_out, blockIdx.x, gridDim.x);
}

template <int C>
static void launch_pyr_down_c(const float* src, float* rest, const MipDesc& d, int s_lvl, int n_out, hipStream_t st)
{
    const int Hs = d.H >> s_lvl, Ws = d.W >> s_lvl;
    const int blocks = ((Hs + 31) / 32) * ((Ws + 31) / 32);
    hipLaunchKernelGGL(mip_pyr_down_kernel<C>, dim3(blocks), dim3(256), 0, st, src, rest, d, s_lvl, n_out);
}

// builds levels from_level+1 .. levels-1 into `rest`; from_level = 0: from the caller's level-0 texture, = 1: level 1 is already
// in `rest` (the fused optimiser wrote it while it updated the texture, texir_adam_step_tex)
// TEXIR_MIP_PER_LEVEL=1 keeps the first implementation (one launch per level) alive: the parity tests run both and demand identical bits
static bool mip_per_level() { return env().mip_per_level != 0; }

hipError_t launch_mip_build(const float* tex, float* rest, int H, int W, int C, int levels, int from_level, hipStream_t st)
{
    if (levels <= 1 + from_level) return hipSuccess;
    MipDesc d = make_desc(H, W, C, levels);
    if (mip_per_level()) {
        int lt = tail_begin(d);
        if (lt < 2) lt = 2;                       // level 1 always comes from the separate level-0 pointer
        for (int l = 1 + from_level; l < lt && l < levels; l++) {
            const float* src_l = l == 1 ? tex : rest + d.off[l - 1];
            launch_down(src_l, rest + d.off[l], H >> l, W >> l, C, st);
        }
        if (lt < levels) hipLaunchKernelGGL(mip_down_tail_kernel, dim3(1), dim3(1024), 0, st, rest, d, lt);
        return hipGetLastError();
    }
    const int s_lvl = from_level;
    const int n_out = (levels - 1 - s_lvl) < 5 ? (levels - 1 - s_lvl) : 5;
    const float* src = s_lvl == 0 ? tex : rest + d.off[s_lvl];
    if (C == 1) launch_pyr_down_c<1>(src, rest, d, s_lvl, n_out, st);
    else if (C == 2) launch_pyr_down_c<2>(src, rest, d, s_lvl, n_out, st);
    else if (C == 3) launch_pyr_down_c<3>(src, rest, d, s_lvl, n_out, st);
    else launch_pyr_down_c<4>(src, rest, d, s_lvl, n_out, st);
    // (the levels above the pyramid: a last-block-finishes fusion was measured -- 16 384 same-address atomics cost 0.5 ms against this 5 us launch)
    const int lt = s_lvl + n_out + 1;
    if (lt < levels) hipLaunchKernelGGL(mip_down_tail_kernel, dim3(1), dim3(1024), 0, st, rest, d, lt);
    return hipGetLastError();
}

hipError_t launch_tex_taps(int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear, int64_t P, long long* keys,
                           float* weights, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    MipDesc d = make_desc(H, W, C, levels);
    hipLaunchKernelGGL(tex_taps_kernel, dim3(grid1d(P, 256)), dim3(256), 0, st, d, uv, uvda, trilinear, P, keys, weights);
    return hipGetLastError();
}

static void launch_folds(float* d_tex, float* grad_rest, const MipDesc& d, int H, int W, int C, int levels, int fold_to_level, hipStream_t st);

// d_tex and grad_rest zero on entry; gather of the sorted tap lists, then the same folds as launch_tex_fetch_bwd
hipError_t launch_tex_gather_bwd(float* d_tex, float* grad_rest, int H, int W, int C, int levels, const long long* seg_key, const int* seg_start,
                                 const int* seg_count, int n_seg, const int* pix, const float* w, const float* d_out, int trilinear,
                                 int fold_to_level, hipStream_t st)
{
    MipDesc d = make_desc(H, W, C, levels);
    const int64_t n0 = (int64_t)H * W;
    if (n_seg > 0) {
        dim3 grid(grid1d(n_seg, 256));
        if (C == 1) hipLaunchKernelGGL(tex_gather_kernel<1>, grid, dim3(256), 0, st, d_tex, grad_rest, n0, seg_key, seg_start, seg_count, n_seg, pix, w, d_out);
        else if (C == 2) hipLaunchKernelGGL(tex_gather_kernel<2>, grid, dim3(256), 0, st, d_tex, grad_rest, n0, seg_key, seg_start, seg_count, n_seg, pix, w, d_out);
        else if (C == 3) hipLaunchKernelGGL(tex_gather_kernel<3>, grid, dim3(256), 0, st, d_tex, grad_rest, n0, seg_key, seg_start, seg_count, n_seg, pix, w, d_out);
        else hipLaunchKernelGGL(tex_gather_kernel<4>, grid, dim3(256), 0, st, d_tex, grad_rest, n0, seg_key, seg_start, seg_count, n_seg, pix, w, d_out);
    }
    if (trilinear && levels > 1) launch_folds(d_tex, grad_rest, d, H, W, C, levels, fold_to_level, st);
    return hipGetLastError();
}

hipError_t launch_tex_fetch(const float* tex, const float* rest, int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear,
                            int64_t P, float* out, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    MipDesc d = make_desc(H, W, C, levels);
    hipLaunchKernelGGL(tex_fetch_kernel<false>, dim3(grid1d(P, 256)), dim3(256), 0, st, const_cast<float*>(tex), const_cast<float*>(rest), d, uv, uvda,
                       trilinear, P, out);
    return hipGetLastError();
}

// d_tex [H,W,C] and grad_rest [mip_total_elems] must be zero on entry; on return d_tex holds d loss / d texture
hipError_t launch_tex_fetch_bwd(float* d_tex, float* grad_rest, int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear,
                                int64_t P, const float* d_out, int fold_to_level, hipStream_t st)
{
    MipDesc d = make_desc(H, W, C, levels);
    if (P > 0)
        hipLaunchKernelGGL(tex_fetch_kernel<true>, dim3(grid1d(P, 256)), dim3(256), 0, st, d_tex, grad_rest, d, uv, uvda, trilinear, P, const_cast<float*>(d_out));
    if (trilinear && levels > 1) launch_folds(d_tex, grad_rest, d, H, W, C, levels, fold_to_level, st);
    return hipGetLastError();
}

static void launch_folds(float* d_tex, float* grad_rest, const MipDesc& d, int H, int W, int C, int levels, int fold_to_level, hipStream_t st)
{
    // one launch folds the whole stack into level `fold_to_level` (1: level 1 stays un-folded into level 0 -- texir_adam_step_tex adds
    // 0.25 * level 1 while it reads the gradient; 2: level 2 stays un-folded into level 1 as well, the optimiser step takes both folds over)
    const int f = fold_to_level;
    if (levels - 1 - f < 1) return;
    if (mip_per_level()) {
        int lt = tail_begin(d);
        if (lt < 2) lt = 2;
        // small levels: levels-1 .. lt folded down to level lt-1 inside one block
        if (lt < levels) hipLaunchKernelGGL(mip_fold_tail_kernel, dim3(1), dim3(1024), 0, st, grad_rest, d, lt - 1 > f ? lt - 1 : f);
        for (int l = (lt < levels ? lt : levels) - 1; l >= 1 + fold_to_level; l--) {
            float* fine_l = l == 1 ? d_tex : grad_rest + d.off[l - 1];
            launch_fold(fine_l, grad_rest + d.off[l], H >> (l - 1), W >> (l - 1), C, st);
        }
        return;
    }
    float* fine = f == 0 ? d_tex : grad_rest + d.off[f];
    const int Hf = H >> f, Wf = W >> f;
    const int blocks = ((Hf + 31) / 32) * ((Wf + 31) / 32);
    if (C == 1) hipLaunchKernelGGL(mip_pyr_fold_kernel<1>, dim3(blocks), dim3(256), 0, st, fine, grad_rest, d, f);
    else if (C == 2) hipLaunchKernelGGL(mip_pyr_fold_kernel<2>, dim3(blocks), dim3(256), 0, st, fine, grad_rest, d, f);
    else if (C == 3) hipLaunchKernelGGL(mip_pyr_fold_kernel<3>, dim3(blocks), dim3(256), 0, st, fine, grad_rest, d, f);
    else hipLaunchKernelGGL(mip_pyr_fold_kernel<4>, dim3(blocks), dim3(256), 0, st, fine, grad_rest, d, f);
}

// ------------------------------------------------------------------------------------------------------------------
// torch.optim.Adam (no amsgrad / weight decay / maximize), same operation order as torch's single-tensor path, fused
// with the trainer's clamp (materials_r in [1e-2, 0.8], materials_a >= 0; train_material.py:458,592-593)
// ------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// one Adam update with explicitly placed roundings (shared by both optimiser kernels so that they agree bit for bit):
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2); param.addcdiv_(exp_avg, denom, value = -step_size)
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float beta1, float beta2, float eps, float step_size,
                                            float bc2_sqrt, float lo, float hi)
{
    const float mi = __builtin_fmaf(g - m, 1.f - beta1, m);
    const float vi = __builtin_fmaf(g * g, 1.f - beta2, v * beta2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = __builtin_fmaf(mi / denom, -step_size, p);
    m = mi; v = vi; p = fminf(fmaxf(pi, lo), hi);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float beta1, float beta2, float eps, float step_size, float bc2_sqrt,
                                                   float lo, float hi, const float* __restrict__ hyp)
{
    if (hyp) { step_size = hyp[0]; bc2_sqrt = hyp[1]; }          // device-resident step size / bias correction (adam_tick_kernel): hipGraph replays
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        adam_update(p[i], g[i], m[i], v[i], beta1, beta2, eps, step_size, bc2_sqrt, lo, hi);
    }
}
#pragma clang fp contract(fast)

// Adam over a texture [H,W,C] whose gradient is level 0 + 0.25 * (level-1 gradient of the 2x2 block): the last fold of the mip
// backward happens here, while the gradient is read anyway (saves one read-modify-write of the finest level per step).
// Same arithmetic as mip_fold_kernel followed by adam_kernel, bit for bit.
//   * l0_mask (one bit per texel, nullable): g holds valid values only at the texels whose bit is set (the texels a cached view's
//     tap lists write -- the buffer is never zero-filled); every other texel's level-0 gradient is zero;
//   * g == nullptr: the level-0 part of the gradient is identically zero (no pixel of the step sampled mip level 0 -- the usual
//     case for 4k textures seen through 128^2 cube faces), so it is neither zero-filled nor read (fma(0.25, g1, 0) is the same float);
//   * mip1 != nullptr: the thread also writes the 2x2 average of the UPDATED texels = level 1 of the next forward's mip stack
//     (same expression as the mip build), which removes the build's pass over the whole level-0 texture.
template <int C>
__global__ __launch_bounds__(256) void adam_tex_kernel(float* __restrict__ p, const float* __restrict__ g, const uint32_t* __restrict__ l0_mask,
                                                       const float* __restrict__ g1, const float* __restrict__ g2, float* __restrict__ m, float* __restrict__ v, float* __restrict__ mip1,
                                                       int H, int W, float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float lo, float hi, const float* __restrict__ hyp,
                                                       const uint32_t* __restrict__ g1_mask /* nullable: see adam_tex_vec_body */)
{
    if (hyp) { step_size = hyp[0]; bc2_sqrt = hyp[1]; }
    // one thread per (2x2 block column, channel): t = bx * C + ch.  Its level-1 element and its g1 element sit at index t of the
    // half-resolution row (perfectly coalesced); its four texture elements are rows 2by / 2by+1 at bx * 2C + ch and + C (the two
    // loads of a row together cover the row densely, every 128-byte line is fetched once).
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int Wh = W >> 1, Hh = H >> 1;
    if (t >= Wh * C) return;
    const int bx = t / C, ch = t - bx * C;
    const int e0 = bx * 2 * C + ch;
    for (int by = blockIdx.y; by < Hh; by += gridDim.y) {
        float g1c = 0.f;                                             // (g1 == nullptr: no tap of the view touches level 1, its direct gradient is identically zero)
        if (g1) {
            bool has = true;
            if (g1_mask) { const size_t tt = (size_t)by * Wh + bx; has = (g1_mask[tt >> 5] >> (tt & 31)) & 1u; }
            if (has) g1c = g1[(size_t)by * Wh * C + t];
        }
        if (g2) g1c = __builtin_fmaf(0.25f, g2[((size_t)(by >> 1) * (Wh >> 1) + (bx >> 1)) * C + ch], g1c);      // the fold level 2 -> level 1, taken over as well
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const size_t i = (size_t)(2 * by + r) * W * C + e0 + q * C;
                float g0 = 0.f;
                if (g) {
                    // level-0 gradient: dense (no mask), or only where this view's tap lists wrote one (bit of the texel set)
                    const size_t texel = (size_t)(2 * by + r) * W + 2 * bx + q;
                    if (!l0_mask || ((l0_mask[texel >> 5] >> (texel & 31)) & 1u)) g0 = g[i];
                }
                const float gi = __builtin_fmaf(0.25f, g1c, g0);
                float pi = p[i], mi = m[i], vi = v[i];
                adam_update(pi, gi, mi, vi, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi);
                p[i] = pi; m[i] = mi; v[i] = vi;
                acc = (r == 0 && q == 0) ? pi : acc + pi;           // ((a + b) + c) + d, the mip build's order
            }
        }
        if (mip1) mip1[(size_t)by * Wh * C + t] = 0.25f * acc;
    }
}

// floats of a row pair one block of adam_tex_vec_kernel owns: the largest multiple of BOTH 2C (no 2x2 texel block straddles two blocks) and 32
// (a block's segment starts and ends on a 128-byte line: with 1020 floats per block -- the largest multiple of 6 -- every segment boundary of
// the 3-channel texture split a line between two blocks, i.e. two CUs / XCDs each wrote part of it) that fits 256 threads x one float4
constexpr int adam_vec_epb(int C) { return C == 3 ? 960 : (1024 / (2 * C)) * (2 * C); }

// The same step with 16-byte accesses: a block owns EPB consecutive floats of a row pair (adam_vec_epb, so
// that no 2x2 texel block straddles two blocks), every thread one float4 of each row; the level-1 gradient segment and the updated
// texels go through LDS so that the level-1 texels come out in the mip build's own summation order ((p00 + p01) + p10) + p11.
// Needs W*C % 4 == 0 (16-byte aligned rows); launch_adam_tex falls back to adam_tex_kernel otherwise.  Identical bits.
// A/B switch of the build (make EXTRA=-DTEXIR_ADAM_NT=n): non-temporal hint on the step's loads (1), stores (2) or both (3) of texels and moments
#ifndef TEXIR_ADAM_NT
#define TEXIR_ADAM_NT 3          // measured (profiles/r04/adam_batch_probe_s15.txt): 310 -> 283 us per launch over both 4k textures, 5.5 -> 6.0 TB/s; loads or stores alone: a third of it each
#endif
#ifndef TEXIR_ADAM_WAVES
#define TEXIR_ADAM_WAVES 5       // 94 VGPRs, no scratch (unbounded: 116 = 4 waves; 6: 24 bytes of scratch).  Alone the launch takes the same time at 4 / 5 / 6; inside the step 5 was 10 us ahead of 4
#endif
#if TEXIR_ADAM_WAVES
#define TEXIR_ADAM_BOUNDS __launch_bounds__(256, TEXIR_ADAM_WAVES)
#else
#define TEXIR_ADAM_BOUNDS __launch_bounds__(256)
#endif
typedef float texir_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 adam_ld4(const float* __restrict__ q)
{
#if TEXIR_ADAM_NT & 1
    const texir_vf4 t = __builtin_nontemporal_load(reinterpret_cast<const texir_vf4*>(q));
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<const float4*>(q);
#endif
}
__device__ __forceinline__ void adam_st4(float* __restrict__ q, float a, float b, float c, float d)
{
#if TEXIR_ADAM_NT & 2
    texir_vf4 t; t.x = a; t.y = b; t.z = c; t.w = d;
    __builtin_nontemporal_store(t, reinterpret_cast<texir_vf4*>(q));
#else
    *reinterpret_cast<float4*>(q) = make_float4(a, b, c, d);
#endif
}

//   * g1_mask (nullable, only together with g2): the level-1 stack is a never-cleared buffer; its texels carry this step's values only where the view's tap
//     lists wrote them (bit by * W/2 + x of the mask -- level 1 leads the `rest` stack, so this is the stack's own mask), all others count as zero and
//     are not read.
template <int C>
__device__ __forceinline__ void adam_tex_vec_body(float* __restrict__ p, const float* __restrict__ g, const uint32_t* __restrict__ l0_mask,
                                                  const float* __restrict__ g1, const uint32_t* __restrict__ g1_mask, const float* __restrict__ g2,
                                                  float* __restrict__ m, float* __restrict__ v, float* __restrict__ mip1,
                                                  int H, int W, float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float lo, float hi,
                                                  int bx, int by0, int gy, float* __restrict__ g1s /* LDS [EPB / 2] */, float (*ps)[adam_vec_epb(C)] /* LDS [2][EPB] */)
{
    constexpr int EPB = adam_vec_epb(C);
    const int row_elems = W * C, Wh = W >> 1, Hh = H >> 1;
    const int e_base = bx * EPB;
    const int n_here = min(EPB, row_elems - e_base);                 // multiple of 2C and of 4
    const int j4 = threadIdx.x * 4;
    const bool act = j4 < n_here;
    const int h_base = e_base / 2, n_half = n_here / 2;              // this block's segment of the half-resolution row
    for (int by = by0; by < Hh; by += gy) {
        // The thread's own texels, moments and level-0 gradient (or the mask bits that say where it is valid) FIRST: these loads depend on nothing, and
        // issued here they are in flight while the level-1 segment is staged (mask word -> value -> level-2 value: up to three dependent round trips
        // that the barrier below would otherwise put in front of them).
        float4 pv[2], mv[2], vv[2], gv[2];
        uint32_t l0bits = 0;                                        // bit r * 4 + q: element q of row r has a level-0 gradient to read
        if (act) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const size_t i = (size_t)(2 * by + r) * row_elems + e_base + j4;
                pv[r] = adam_ld4(p + i); mv[r] = adam_ld4(m + i); vv[r] = adam_ld4(v + i);
                if (g && !l0_mask) gv[r] = *reinterpret_cast<const float4*>(g + i);
                if (g && l0_mask) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const size_t texel = (size_t)(2 * by + r) * W + (e_base / C) + (j4 + q) / C;
                        l0bits |= ((l0_mask[texel >> 5] >> (texel & 31)) & 1u) << (r * 4 + q);
                    }
                }
            }
        }
        for (int k = threadIdx.x; k < n_half; k += 256) {
            const int txh = (h_base + k) / C, ch = (h_base + k) - txh * C;
            // (the level-2 value is loaded before the mask word is tested: it does not depend on it)
            const float x2 = g2 ? g2[((size_t)(by >> 1) * (Wh >> 1) + (txh >> 1)) * C + ch] : 0.f;
            float x = 0.f;                                          // (g1 == nullptr: level-1 direct gradient identically zero -- not read)
            if (g1) {
                bool has = true;
                if (g1_mask) { const size_t t = (size_t)by * Wh + txh; has = (g1_mask[t >> 5] >> (t & 31)) & 1u; }
                if (has) x = g1[(size_t)by * Wh * C + h_base + k];
            }
            if (g2) x = __builtin_fmaf(0.25f, x2, x);               // the fold level 2 -> level 1, taken over as well (same fma as mip_pyr_fold_kernel's)
            g1s[k] = x;
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const size_t i = (size_t)(2 * by + r) * row_elems + e_base + j4;
                float pe[4] = {pv[r].x, pv[r].y, pv[r].z, pv[r].w}, me[4] = {mv[r].x, mv[r].y, mv[r].z, mv[r].w}, ve[4] = {vv[r].x, vv[r].y, vv[r].z, vv[r].w};
                const float ge[4] = {gv[r].x, gv[r].y, gv[r].z, gv[r].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int el = j4 + q;                           // element inside the block's segment
                    const int tx = el / C, ch = el - tx * C;
                    float g0 = 0.f;
                    if (g) {
                        // dense level-0 gradient (read above), or only where this view's tap lists wrote one (bit of the texel set: rare)
                        if (!l0_mask) g0 = ge[q];
                        else if ((l0bits >> (r * 4 + q)) & 1u) g0 = g[i + q];
                    }
                    const float gi = __builtin_fmaf(0.25f, g1s[(tx >> 1) * C + ch], g0);
                    adam_update(pe[q], gi, me[q], ve[q], beta1, beta2, eps, step_size, bc2_sqrt, lo, hi);
                    ps[r][el] = pe[q];
                }
                adam_st4(p + i, pe[0], pe[1], pe[2], pe[3]);
                adam_st4(m + i, me[0], me[1], me[2], me[3]);
                adam_st4(v + i, ve[0], ve[1], ve[2], ve[3]);
            }
        }
        __syncthreads();
        if (mip1) {
            for (int k = threadIdx.x; k < n_half; k += 256) {
                const int txh = k / C, ch = k - txh * C;
                const int a = (2 * txh) * C + ch;
                mip1[(size_t)by * Wh * C + h_base + k] = 0.25f * (ps[0][a] + ps[0][a + C] + ps[1][a] + ps[1][a + C]);
            }
        }
        // (the next iteration's first barrier orders these LDS reads before the rewrite of ps; g1s is rewritten before it, and nobody
        // reads g1s after the barrier above)
    }
}

template <int C>
__global__ TEXIR_ADAM_BOUNDS void adam_tex_vec_kernel(float* __restrict__ p, const float* __restrict__ g, const uint32_t* __restrict__ l0_mask,
                                                           const float* __restrict__ g1, const float* __restrict__ g2, float* __restrict__ m, float* __restrict__ v, float* __restrict__ mip1,
                                                           int H, int W, float beta1, float beta2, float eps, float step_size, float bc2_sqrt, float lo, float hi, const float* __restrict__ hyp)
{
    if (hyp) { step_size = hyp[0]; bc2_sqrt = hyp[1]; }
    constexpr int EPB = adam_vec_epb(C);
    __shared__ float g1s[EPB / 2];
    __shared__ float ps[2][EPB];
    adam_tex_vec_body<C>(p, g, l0_mask, g1, nullptr, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, blockIdx.x, blockIdx.y, gridDim.y, g1s, ps);
}

hipError_t launch_adam_tex(float* p, const float* g, const uint32_t* l0_mask, const float* g1, const float* g2, float* m, float* v, float* mip1, int H, int W, int C,
                           float lr, float beta1, float beta2, float eps, int step, float lo, float hi, const float* hyp, hipStream_t st)
{
    // hyp == nullptr: step size and bias correction from (lr, step) here on the host; else the kernels read them from device memory
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (!hyp) {
        double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    if ((W * C) % 4 == 0 && !env().adam_scalar) {
        const int epb = adam_vec_epb(C);
        dim3 gridv((W * C + epb - 1) / epb, (H >> 1) > 2048 ? 2048 : (H >> 1));
        if (const int v = env().adam_grid_y; v >= 1 && v < (int)gridv.y) gridv.y = v;      // (probe: fewer, longer blocks leave wave slots to a concurrent kernel)
        if (C == 1) hipLaunchKernelGGL(adam_tex_vec_kernel<1>, gridv, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp);
        else if (C == 2) hipLaunchKernelGGL(adam_tex_vec_kernel<2>, gridv, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp);
        else if (C == 3) hipLaunchKernelGGL(adam_tex_vec_kernel<3>, gridv, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp);
        else hipLaunchKernelGGL(adam_tex_vec_kernel<4>, gridv, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp);
        return hipGetLastError();
    }
    dim3 grid(((W >> 1) * C + 255) / 256, (H >> 1) > 4096 ? 4096 : (H >> 1));
    if (C == 1) hipLaunchKernelGGL(adam_tex_kernel<1>, grid, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp, (const uint32_t*)nullptr);
    else if (C == 2) hipLaunchKernelGGL(adam_tex_kernel<2>, grid, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp, (const uint32_t*)nullptr);
    else if (C == 3) hipLaunchKernelGGL(adam_tex_kernel<3>, grid, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp, (const uint32_t*)nullptr);
    else hipLaunchKernelGGL(adam_tex_kernel<4>, grid, dim3(256), 0, st, p, g, l0_mask, g1, g2, m, v, mip1, H, W, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp, (const uint32_t*)nullptr);
    return hipGetLastError();
}

// One thread per parameter record: advances the step count and derives the step size / bias correction the Adam kernels read, in
// double precision with the same expressions the host path uses (torch.optim.Adam's single-tensor path): a captured hipGraph that
// contains this launch followed by the Adam kernels needs no host argument per replay.
//   state [n][4] doubles: step count, lr, beta1, beta2      hyper [n][2] floats: lr / (1 - beta1^step), sqrt(1 - beta2^step)
__global__ void adam_tick_kernel(double* __restrict__ state, float* __restrict__ hyper, int n, unsigned long long mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !((mask >> i) & 1ull)) return;
    double* s = state + 4 * i;
    const double step = s[0] + 1.0;
    s[0] = step;
    const double bc1 = 1.0 - pow(s[2], step), bc2 = 1.0 - pow(s[3], step);
    hyper[2 * i] = (float)(s[1] / bc1);
    hyper[2 * i + 1] = (float)sqrt(bc2);
}

hipError_t launch_adam_tick(double* state, float* hyper, int n, unsigned long long mask, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, state, hyper, n, mask);
    return hipGetLastError();
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                       float lo, float hi, const float* hyp, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (!hyp) {
        double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n, 256 * 4)), dim3(256), 0, st, p, g, m, v, n, beta1, beta2, eps, step_size, bc2_sqrt, lo, hi, hyp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Batched launches (include/texir_hip.h, "batched forms"): one launch per KIND of kernel for up to TEXIR_MAX_BATCH textures.  A step over the
// albedo and the roughness texture is latency-bound in these kernels (5 ... 20 us each, the launch-to-launch floor is ~4.7 us): dealing the
// blocks of ONE grid to the jobs removes a launch per kind and lets the short jobs hide in the long ones' tails.  The job bodies are the
// single-texture kernels' own (…_body above), so every job's result keeps its bits.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMaxBatch = TEXIR_MAX_BATCH;

struct PyrDownJob { const float* src; float* rest; MipDesc d; int s, n_out, first; };
struct PyrDownBatch { int n; PyrDownJob j[kMaxBatch]; };
struct TailJob { float* rest; MipDesc d; int l_begin; };
struct TailBatch { int n; TailJob j[kMaxBatch]; };
struct FetchJob { float* lvl0; float* rest; MipDesc d; const float* uv; const float* uvda; int trilinear; int64_t P; float* out; int first, nb; };
struct FetchBatch { int n; FetchJob j[kMaxBatch]; };
struct GatherJob { float* lvl0; float* rest; int64_t n0; const long long* seg_key; const int* seg_start; const int* seg_count; int n_seg; const int* pix; const float* w;
                   const float* d_out; const float* d_out2; int C, first, nb; };
struct GatherBatch { int n; GatherJob j[kMaxBatch]; };
struct FoldJob { float* fine; const float* rest; MipDesc d; int f; const uint32_t* mask; int first; };
struct FoldBatch { int n; FoldJob j[kMaxBatch]; };
struct AdamTexJob { float* p; const float* g; const uint32_t* l0_mask; const float* g1; const uint32_t* g1_mask; const float* g2; float* m; float* v; float* mip1;
                    int H, W, C; float beta1, beta2, eps, lo, hi; const float* hyp; int first, gy; };
struct AdamTexBatch { int n; AdamTexJob j[kMaxBatch]; };

// job of a block: jobs own consecutive block ranges [first, next job's first)
template <class B>
__device__ __forceinline__ int job_of_block(const B& b, int bid)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < kMaxBatch; i++) if (i < b.n && bid >= b.j[i].first) k = i;
    return k;
}

__global__ __launch_bounds__(256) void mip_pyr_down_batch_kernel(PyrDownBatch b)
{
    __shared__ float smem[2 * 16 * 16 * 4];
    const PyrDownJob& J = b.j[job_of_block(b, blockIdx.x)];
    const int bid = blockIdx.x - J.first;
    switch (J.d.C) {
        case 1: mip_pyr_down_body<1>(J.src, J.rest, J.d, J.s, J.n_out, bid, reinterpret_cast<float (*)[16 * 16 * 1]>(smem)); break;
        case 2: mip_pyr_down_body<2>(J.src, J.rest, J.d, J.s, J.n_out, bid, reinterpret_cast<float (*)[16 * 16 * 2]>(smem)); break;
        case 3: mip_pyr_down_body<3>(J.src, J.rest, J.d, J.s, J.n_out, bid, reinterpret_cast<float (*)[16 * 16 * 3]>(smem)); break;
        default: mip_pyr_down_body<4>(J.src, J.rest, J.d, J.s, J.n_out, bid, reinterpret_cast<float (*)[16 * 16 * 4]>(smem)); break;
    }
}

__global__ __launch_bounds__(1024) void mip_down_tail_batch_kernel(TailBatch b)
{
    const TailJob& J = b.j[blockIdx.x];
    mip_down_tail_body(J.rest, J.d, J.l_begin);
}

__global__ __launch_bounds__(256) void tex_fetch_batch_kernel(FetchBatch b)
{
    const FetchJob& J = b.j[job_of_block(b, blockIdx.x)];
    tex_fetch_body<false>(J.lvl0, J.rest, J.d, J.uv, J.uvda, J.trilinear, J.P, J.out, blockIdx.x - J.first, J.nb);
}

__global__ __launch_bounds__(256) void tex_gather_batch_kernel(GatherBatch b)
{
    const GatherJob& J = b.j[job_of_block(b, blockIdx.x)];
    const int bid = blockIdx.x - J.first;
    switch (J.C) {
        case 1: tex_gather_body<1>(J.lvl0, J.rest, J.n0, J.seg_key, J.seg_start, J.seg_count, J.n_seg, J.pix, J.w, J.d_out, bid, J.nb, J.d_out2); break;
        case 2: tex_gather_body<2>(J.lvl0, J.rest, J.n0, J.seg_key, J.seg_start, J.seg_count, J.n_seg, J.pix, J.w, J.d_out, bid, J.nb, J.d_out2); break;
        case 3: tex_gather_body<3>(J.lvl0, J.rest, J.n0, J.seg_key, J.seg_start, J.seg_count, J.n_seg, J.pix, J.w, J.d_out, bid, J.nb, J.d_out2); break;
        default: tex_gather_body<4>(J.lvl0, J.rest, J.n0, J.seg_key, J.seg_start, J.seg_count, J.n_seg, J.pix, J.w, J.d_out, bid, J.nb, J.d_out2); break;
    }
}

__global__ __launch_bounds__(256) void mip_pyr_fold_batch_kernel(FoldBatch b)
{
    __shared__ float smem[2 * 16 * 16 * 4];
    const FoldJob& J = b.j[job_of_block(b, blockIdx.x)];
    const int bid = blockIdx.x - J.first;
    switch (J.d.C) {
        case 1: mip_pyr_fold_body<1>(J.fine, J.rest, J.d, J.f, J.mask, bid, reinterpret_cast<float (*)[16 * 16 * 1]>(smem)); break;
        case 2: mip_pyr_fold_body<2>(J.fine, J.rest, J.d, J.f, J.mask, bid, reinterpret_cast<float (*)[16 * 16 * 2]>(smem)); break;
        case 3: mip_pyr_fold_body<3>(J.fine, J.rest, J.d, J.f, J.mask, bid, reinterpret_cast<float (*)[16 * 16 * 3]>(smem)); break;
        default: mip_pyr_fold_body<4>(J.fine, J.rest, J.d, J.f, J.mask, bid, reinterpret_cast<float (*)[16 * 16 * 4]>(smem)); break;
    }
}

// grid.x = the jobs' row segments side by side, grid.y = the largest row-pair stride of the batch (a job with fewer rows leaves the surplus rows idle)
// CSET: bit C-1 set for every channel count that occurs in the batch -- the other bodies are not compiled in (the kernel's register count is the largest of
// its bodies': 92 VGPRs with all four, 64 with the albedo + roughness pair)
template <int CSET>
__global__ TEXIR_ADAM_BOUNDS void adam_tex_vec_batch_kernel(AdamTexBatch b)
{
    __shared__ float g1s[512];
    __shared__ float ps[2 * 1024];
    const AdamTexJob& J = b.j[job_of_block(b, blockIdx.x)];
    if ((int)blockIdx.y >= J.gy) return;
    const float step_size = J.hyp[0], bc2_sqrt = J.hyp[1];
    const int bx = blockIdx.x - J.first;
#define TEXIR_ADAM_BODY(CC)                                                                                                                                   \
    if constexpr ((CSET >> (CC - 1)) & 1)                                                                                                                     \
        if (J.C == CC) {                                                                                                                                      \
            adam_tex_vec_body<CC>(J.p, J.g, J.l0_mask, J.g1, J.g1_mask, J.g2, J.m, J.v, J.mip1, J.H, J.W, J.beta1, J.beta2, J.eps, step_size, bc2_sqrt, J.lo, \
                                  J.hi, bx, blockIdx.y, J.gy, g1s, reinterpret_cast<float (*)[adam_vec_epb(CC)]>(ps));                                        \
            return;                                                                                                                                           \
        }
    TEXIR_ADAM_BODY(1) TEXIR_ADAM_BODY(2) TEXIR_ADAM_BODY(3) TEXIR_ADAM_BODY(4)
#undef TEXIR_ADAM_BODY
}

// g[t][c] += (bit t of mask) ? g0[t][c] : 0 -- the sparse level-0 gradient (valid under the view's tap mask only) folded into a DENSE level-0 gradient that another
// fetch of the same texture produced in the same backward pass (stage 1: the un-mipmapped roughness fetch is dense, the trilinear one sparse).  One pass; the
// torch form it replaces (shift, and, bool, where, add_ over the whole texture: five launches, 130 us at 4096^2) computed the same sums in the same order.
__global__ __launch_bounds__(256) void grad_add_masked_kernel(float* __restrict__ g, const float* __restrict__ g0, const uint32_t* __restrict__ mask, int64_t n_texels, int C)
{
    const int64_t n = n_texels * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / C;
        const bool on = (mask[t >> 5] >> (t & 31)) & 1u;
        g[i] = g[i] + (on ? g0[i] : 0.f);
    }
}

static thread_local char g_batch_err[384];

static int bfail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_batch_err, sizeof(g_batch_err), fmt, ap); va_end(ap);
    return code;
}

static int bcheck_tex(const char* fn, int k, int H, int W, int C, int levels)
{
    if (H <= 0 || W <= 0 || C <= 0 || C > 4 || levels < 1 || levels > 16) return bfail(TEXIR_ERR_INVALID, "%s: job %d: bad texture H=%d W=%d C=%d levels=%d", fn, k, H, W, C, levels);
    if (levels > mip_levels(H, W, 15)) return bfail(TEXIR_ERR_INVALID, "%s: job %d: %d mip levels not available for %dx%d", fn, k, levels, H, W);
    return 0;
}

#define BATCH_HIP_TRY(fn, expr)                                                                              \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return bfail(TEXIR_ERR_HIP, "%s: %s failed: %s", fn, #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace texir

using namespace texir;

extern "C" {

const char* texir_batch_last_error(void) { return g_batch_err; }

int texir_grad_add_masked(float* g, const float* g0, const uint32_t* mask, int64_t n_texels, int32_t C, void* stream)
{
    const char* fn = "texir_grad_add_masked";
    if (n_texels < 0 || C < 1 || C > 4) return bfail(TEXIR_ERR_INVALID, "%s: bad size n_texels=%lld C=%d", fn, (long long)n_texels, (int)C);
    if (n_texels == 0) return TEXIR_OK;
    if (!g || !g0 || !mask) return bfail(TEXIR_ERR_INVALID, "%s: null argument", fn);
    hipLaunchKernelGGL(grad_add_masked_kernel, dim3(grid1d(n_texels * C, 256 * 4)), dim3(256), 0, (hipStream_t)stream, g, g0, mask, (int64_t)n_texels, (int)C);
    BATCH_HIP_TRY(fn, hipGetLastError());
    return TEXIR_OK;
}

int texir_tex_fetch_forward_batch(const texir_tex_fetch_job* jobs, int32_t n, void* stream)
{
    const char* fn = "texir_tex_fetch_forward_batch";
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || n > kMaxBatch || (n > 0 && !jobs)) return bfail(TEXIR_ERR_INVALID, "%s: 0..%d jobs (got %d)", fn, kMaxBatch, n);
    for (int k = 0; k < n; k++) {
        const texir_tex_fetch_job& q = jobs[k];
        if (!q.tex || (q.P > 0 && (!q.uv || !q.out || (q.filter_mode == 1 && !q.uv_da))) || (q.filter_mode == 1 && q.levels > 1 && !q.mips_rest))
            return bfail(TEXIR_ERR_INVALID, "%s: job %d: null argument", fn, k);
        if (q.filter_mode < 0 || q.filter_mode > 1 || q.P < 0 || q.build_from < -1 || q.build_from > 1) return bfail(TEXIR_ERR_INVALID, "%s: job %d: bad filter_mode/P/build_from", fn, k);
        if (q.build_from >= 0 && q.levels > 1 && !q.mips_rest) return bfail(TEXIR_ERR_INVALID, "%s: job %d: a build needs mips_rest", fn, k);
        if (int rc = bcheck_tex(fn, k, q.H, q.W, q.C, q.levels)) return rc;
    }
    if (n == 0) return TEXIR_OK;
    // the reference form (one launch per level, TEXIR_MIP_PER_LEVEL=1) stays a sequence of single launches
    if (env().mip_per_level || n == 1) {
        for (int k = 0; k < n; k++) {
            const texir_tex_fetch_job& q = jobs[k];
            if (q.build_from >= 0) BATCH_HIP_TRY(fn, launch_mip_build(q.tex, q.mips_rest, q.H, q.W, q.C, q.levels, q.build_from, st));
        }
        for (int k = 0; k < n; k++) {
            const texir_tex_fetch_job& q = jobs[k];
            if (q.P > 0) BATCH_HIP_TRY(fn, launch_tex_fetch(q.tex, q.mips_rest, q.H, q.W, q.C, q.levels, q.uv, q.uv_da, q.filter_mode, q.P, q.out, st));
        }
        return TEXIR_OK;
    }
    PyrDownBatch pb; pb.n = 0;
    TailBatch tb; tb.n = 0;
    int blocks = 0;
    for (int k = 0; k < n; k++) {
        const texir_tex_fetch_job& q = jobs[k];
        if (q.build_from < 0 || q.levels <= 1 + q.build_from) continue;
        MipDesc d = make_desc(q.H, q.W, q.C, q.levels);
        const int s_lvl = q.build_from;
        const int n_out = (q.levels - 1 - s_lvl) < 5 ? (q.levels - 1 - s_lvl) : 5;
        PyrDownJob& J = pb.j[pb.n++];
        J.src = s_lvl == 0 ? q.tex : q.mips_rest + d.off[s_lvl]; J.rest = q.mips_rest; J.d = d; J.s = s_lvl; J.n_out = n_out; J.first = blocks;
        blocks += (((q.H >> s_lvl) + 31) / 32) * (((q.W >> s_lvl) + 31) / 32);
        const int lt = s_lvl + n_out + 1;
        if (lt < q.levels) { TailJob& T = tb.j[tb.n++]; T.rest = q.mips_rest; T.d = d; T.l_begin = lt; }
    }
    if (pb.n > 0) hipLaunchKernelGGL(mip_pyr_down_batch_kernel, dim3(blocks), dim3(256), 0, st, pb);
    if (tb.n > 0) hipLaunchKernelGGL(mip_down_tail_batch_kernel, dim3(tb.n), dim3(1024), 0, st, tb);
    FetchBatch fb; fb.n = 0;
    blocks = 0;
    for (int k = 0; k < n; k++) {
        const texir_tex_fetch_job& q = jobs[k];
        if (q.P <= 0) continue;
        FetchJob& J = fb.j[fb.n++];
        J.lvl0 = const_cast<float*>(q.tex); J.rest = q.mips_rest; J.d = make_desc(q.H, q.W, q.C, q.levels); J.uv = q.uv; J.uvda = q.uv_da; J.trilinear = q.filter_mode; J.P = q.P; J.out = q.out;
        J.first = blocks; J.nb = grid1d(q.P, 256);
        blocks += J.nb;
    }
    if (fb.n > 0) hipLaunchKernelGGL(tex_fetch_batch_kernel, dim3(blocks), dim3(256), 0, st, fb);
    BATCH_HIP_TRY(fn, hipGetLastError());
    return TEXIR_OK;
}

int texir_tex_gather_backward_batch(const texir_tex_gather_job* jobs, int32_t n, void* stream)
{
    const char* fn = "texir_tex_gather_backward_batch";
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || n > kMaxBatch || (n > 0 && !jobs)) return bfail(TEXIR_ERR_INVALID, "%s: 0..%d jobs (got %d)", fn, kMaxBatch, n);
    for (int k = 0; k < n; k++) {
        const texir_tex_gather_job& q = jobs[k];
        if ((!q.d_tex && !q.defer_last_fold) || !q.d_out || (q.n_seg > 0 && (!q.seg_key || !q.seg_start || !q.seg_count || !q.pix || !q.weights)) || (q.filter_mode == 1 && q.levels > 1 && !q.grad_rest))
            return bfail(TEXIR_ERR_INVALID, "%s: job %d: null argument", fn, k);
        if (q.filter_mode < 0 || q.filter_mode > 1 || q.n_seg < 0 || q.defer_last_fold < 0 || q.defer_last_fold > 2 || (q.defer_last_fold && (q.filter_mode != 1 || q.levels < 2))
            || (q.defer_last_fold == 2 && (q.levels < 4 || (q.H & 3) || (q.W & 3))))
            return bfail(TEXIR_ERR_INVALID, "%s: job %d: bad filter_mode/n_seg/defer_last_fold", fn, k);
        if (q.d_out2 && env().mip_per_level) return bfail(TEXIR_ERR_INVALID, "%s: job %d: d_out2 is not available with TEXIR_MIP_PER_LEVEL=1", fn, k);
        if (q.rest_mask && q.defer_last_fold != 2) return bfail(TEXIR_ERR_INVALID, "%s: job %d: rest_mask needs defer_last_fold = 2", fn, k);
        if (q.rest_mask && env().mip_per_level) return bfail(TEXIR_ERR_INVALID, "%s: job %d: rest_mask is not available with TEXIR_MIP_PER_LEVEL=1 (the per-level folds read a cleared stack)", fn, k);
        if (int rc = bcheck_tex(fn, k, q.H, q.W, q.C, q.levels)) return rc;
    }
    if (n == 0) return TEXIR_OK;
    if (env().mip_per_level) {
        for (int k = 0; k < n; k++) {
            const texir_tex_gather_job& q = jobs[k];
            BATCH_HIP_TRY(fn, launch_tex_gather_bwd(q.d_tex, q.grad_rest, q.H, q.W, q.C, q.levels, (const long long*)q.seg_key, q.seg_start, q.seg_count, q.n_seg, q.pix, q.weights, q.d_out,
                                                    q.filter_mode, q.defer_last_fold, st));
        }
        return TEXIR_OK;
    }
    GatherBatch gb; gb.n = 0;
    int blocks = 0;
    for (int k = 0; k < n; k++) {
        const texir_tex_gather_job& q = jobs[k];
        if (q.n_seg <= 0) continue;
        GatherJob& J = gb.j[gb.n++];
        J.lvl0 = q.d_tex; J.rest = q.grad_rest; J.n0 = (int64_t)q.H * q.W; J.seg_key = (const long long*)q.seg_key; J.seg_start = q.seg_start; J.seg_count = q.seg_count; J.n_seg = q.n_seg;
        J.pix = q.pix; J.w = q.weights; J.d_out = q.d_out; J.d_out2 = q.d_out2; J.C = q.C; J.first = blocks; J.nb = grid1d(q.n_seg, 256);
        blocks += J.nb;
    }
    if (gb.n > 0) hipLaunchKernelGGL(tex_gather_batch_kernel, dim3(blocks), dim3(256), 0, st, gb);
    FoldBatch fb; fb.n = 0;
    blocks = 0;
    for (int k = 0; k < n; k++) {
        const texir_tex_gather_job& q = jobs[k];
        const int f = q.defer_last_fold;
        if (!(q.filter_mode == 1 && q.levels > 1) || q.levels - 1 - f < 1) continue;
        FoldJob& J = fb.j[fb.n++];
        J.d = make_desc(q.H, q.W, q.C, q.levels);
        J.fine = f == 0 ? q.d_tex : q.grad_rest + J.d.off[f]; J.rest = q.grad_rest; J.f = f; J.mask = q.rest_mask; J.first = blocks;
        blocks += (((q.H >> f) + 31) / 32) * (((q.W >> f) + 31) / 32);
    }
    if (fb.n > 0) hipLaunchKernelGGL(mip_pyr_fold_batch_kernel, dim3(blocks), dim3(256), 0, st, fb);
    BATCH_HIP_TRY(fn, hipGetLastError());
    return TEXIR_OK;
}

int texir_adam_step_tex_dev_batch(const texir_adam_tex_job* jobs, int32_t n, void* stream)
{
    const char* fn = "texir_adam_step_tex_dev_batch";
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || n > kMaxBatch || (n > 0 && !jobs)) return bfail(TEXIR_ERR_INVALID, "%s: 0..%d jobs (got %d)", fn, kMaxBatch, n);
    bool vec = !env().adam_scalar;
    for (int k = 0; k < n; k++) {
        const texir_adam_tex_job& q = jobs[k];
        if (!q.param || (!q.grad_level1 && !q.grad_level2) || !q.exp_avg || !q.exp_avg_sq || !q.hyper) return bfail(TEXIR_ERR_INVALID, "%s: job %d: null argument", fn, k);
        if (q.H < 2 || q.W < 2 || (q.H & 1) || (q.W & 1) || q.C < 1 || q.C > 4) return bfail(TEXIR_ERR_INVALID, "%s: job %d: bad H/W/C", fn, k);
        if (q.grad_level2 && ((q.H & 3) || (q.W & 3))) return bfail(TEXIR_ERR_INVALID, "%s: job %d: a level-2 gradient needs H and W divisible by 4", fn, k);
        if (q.level1_mask && !q.grad_level2) return bfail(TEXIR_ERR_INVALID, "%s: job %d: level1_mask needs grad_level2", fn, k);
        if ((q.W * q.C) % 4 != 0) vec = false;
    }
    if (n == 0) return TEXIR_OK;
    if (!vec) {
        // scalar form (reference kernel of the parity tests, or rows that are not 16-byte aligned): one launch per job
        for (int k = 0; k < n; k++) {
            const texir_adam_tex_job& q = jobs[k];
            dim3 grid(((q.W >> 1) * q.C + 255) / 256, (q.H >> 1) > 4096 ? 4096 : (q.H >> 1));
#define TEXIR_ADAM_SCALAR_LAUNCH(CC) hipLaunchKernelGGL(adam_tex_kernel<CC>, grid, dim3(256), 0, st, q.param, q.grad, q.grad_mask, q.grad_level1, q.grad_level2, q.exp_avg, q.exp_avg_sq, \
                                                        q.mip_level1, q.H, q.W, q.beta1, q.beta2, q.eps, 0.f, 1.f, q.clamp_lo, q.clamp_hi, q.hyper, q.level1_mask)
            if (q.C == 1) TEXIR_ADAM_SCALAR_LAUNCH(1); else if (q.C == 2) TEXIR_ADAM_SCALAR_LAUNCH(2); else if (q.C == 3) TEXIR_ADAM_SCALAR_LAUNCH(3); else TEXIR_ADAM_SCALAR_LAUNCH(4);
#undef TEXIR_ADAM_SCALAR_LAUNCH
        }
        BATCH_HIP_TRY(fn, hipGetLastError());
        return TEXIR_OK;
    }
    AdamTexBatch ab; ab.n = n;
    int gx = 0, gy = 1;
    for (int k = 0; k < n; k++) {
        const texir_adam_tex_job& q = jobs[k];
        AdamTexJob& J = ab.j[k];
        J.p = q.param; J.g = q.grad; J.l0_mask = q.grad_mask; J.g1 = q.grad_level1; J.g1_mask = q.level1_mask; J.g2 = q.grad_level2; J.m = q.exp_avg; J.v = q.exp_avg_sq; J.mip1 = q.mip_level1;
        J.H = q.H; J.W = q.W; J.C = q.C; J.beta1 = q.beta1; J.beta2 = q.beta2; J.eps = q.eps; J.lo = q.clamp_lo; J.hi = q.clamp_hi; J.hyp = q.hyper;
        const int epb = adam_vec_epb(q.C);
        J.first = gx; J.gy = (q.H >> 1) > 2048 ? 2048 : (q.H >> 1);
        if (const int v = env().adam_grid_y; v >= 1 && v < J.gy) J.gy = v;
        gx += (q.W * q.C + epb - 1) / epb;
        if (J.gy > gy) gy = J.gy;
    }
    int cset = 0;
    for (int k = 0; k < n; k++) cset |= 1 << (jobs[k].C - 1);
    if (cset == 0b0001) hipLaunchKernelGGL(adam_tex_vec_batch_kernel<0b0001>, dim3(gx, gy), dim3(256), 0, st, ab);
    else if (cset == 0b0100) hipLaunchKernelGGL(adam_tex_vec_batch_kernel<0b0100>, dim3(gx, gy), dim3(256), 0, st, ab);
    else if (cset == 0b0101) hipLaunchKernelGGL(adam_tex_vec_batch_kernel<0b0101>, dim3(gx, gy), dim3(256), 0, st, ab);
    else hipLaunchKernelGGL(adam_tex_vec_batch_kernel<0b1111>, dim3(gx, gy), dim3(256), 0, st, ab);
    BATCH_HIP_TRY(fn, hipGetLastError());
    return TEXIR_OK;
}

}  // extern "C"
