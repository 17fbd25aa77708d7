// gfx950 kernels of libtexir_hip.so + their launchers.
//   irt_kernel          fused sample + trace + shade + ndl-weighted reduce   (models/tracer_o3d_irt.py:156-178)
//   trace_shade_kernel  query_irf                                            (models/tracer_o3d_irt.py:240-269)
//   gen_dir_kernel      generate_dir                                         (utils/sample_util.py:63-146)
//   spec_fwd/bwd_kernel render + specular_reflectance and its analytic grad  (models/mat_nvdiffrast.py:201-279)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_common.h"
#include "kernels.h"

namespace texir {

// Sample order inside a texel.  The estimator is a plain sum, so the order is free; choose it so that the 64
// samples a wave traces together fall into one (phi-bin, cos-theta-bin) cell of the Hammersley lattice
// (i's low bits are phi's high bits after the radical inverse; i's high bits are theta).  Coherent rays share
// the upper BVH levels => the wave's node fetches coalesce.
__device__ __forceinline__ uint32_t sample_index(uint32_t pass, uint32_t lane, uint32_t N, int log2N)
{
    if (log2N < 7) return pass * 64u + lane;             // N not a power of two or N <= 64: natural order
    int cells = log2N - 6;                               // log2(#passes)
    int bphi = (cells + 1) >> 1, bth = cells - bphi;     // split the cell bits between phi and theta
    uint32_t low = pass & ((1u << bphi) - 1u);
    uint32_t th = pass >> bphi;
    return (th << (log2N - bth)) | (lane << bphi) | low;
}

// Cell-major schedule.  With N = 2^m >= 128 the samples of a texel fall into 2^(m-6) lattice cells of 64 samples; the
// kernel is launched once per ABSOLUTE direction cell J and every texel contributes the lattice pass whose cell lies
// nearest to J after its own Cranley-Patterson shift (a bijection J -> pass for a fixed shift, so every sample is traced
// exactly once over the launches).  All rays in flight on the chip then leave neighbouring texels towards the same
// ~1/32 of the hemisphere: the BVH/triangle/texel working set of a launch shrinks by the number of cells and fits the
// per-XCD L2 instead of streaming from the Infinity Cache.
__device__ __forceinline__ uint32_t cell_to_pass(uint32_t J, float sh0, float sh1, int log2N)
{
    int cells = log2N - 6;
    int bphi = (cells + 1) >> 1, bth = cells - bphi;
    uint32_t nphi = 1u << bphi, nth = 1u << bth;
    uint32_t Jphi = J & (nphi - 1u), Jth = J >> bphi;
    uint32_t dphi = (uint32_t)(sh1 * (float)nphi + 0.5f), dth = (uint32_t)(sh0 * (float)nth + 0.5f);
    uint32_t phibin = (Jphi + nphi - (dphi & (nphi - 1u))) & (nphi - 1u);
    uint32_t th = (Jth + nth - (dth & (nth - 1u))) & (nth - 1u);
    uint32_t low = __brev(phibin) >> (32 - bphi);          // phi's high bits are i's low bits reversed
    if (bphi == 0) low = 0;
    return (th << bphi) | low;
}

// MODE 0: all passes of a texel in one launch (any N).  MODE 1: one direction cell per launch (cell-major schedule);
// irr accumulates the raw sum over launches and the last launch applies the 2*pi/N scale.
template <bool STATS, int MODE, int LSTK, int MINW, int WIDTH, bool TOPLDS>
__global__ __launch_bounds__(kBlock, MINW) void irt_kernel(SceneDev sc, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                     const float* __restrict__ shift, const int32_t* __restrict__ ids, int64_t n_ids,
                                                     int N, int log2N, int mode, int cell, int last_cell, float* __restrict__ irr,
                                                     unsigned long long* __restrict__ stats)
{
    __shared__ float4 lds_top[TOPLDS ? 4 * kTopMax : 1];
    if (TOPLDS) stage_top_levels(sc, lds_top);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + wave, nw = (int64_t)gridDim.x * (kBlock / 64);
    uint32_t c_nodes = 0, c_tris = 0, c_rays = 0, c_hits = 0;
    const int passes = (N + 63) >> 6;
    for (int64_t k = gw; k < n_ids; k += nw) {
        const int64_t t = ids ? (int64_t)ids[k] : k;
        const float px = pos[3 * t], py = pos[3 * t + 1], pz = pos[3 * t + 2];
        const float nx = nrm[3 * t], ny = nrm[3 * t + 1], nz = nrm[3 * t + 2];
        const float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
        const Frame f = make_frame(nx, ny, nz);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        const int p_begin = MODE ? (int)cell_to_pass((uint32_t)cell, sh0, sh1, log2N) : 0;
        const int p_end = MODE ? p_begin + 1 : passes;
        for (int p = p_begin; p < p_end; p++) {
            uint32_t i = sample_index((uint32_t)p, (uint32_t)lane, (uint32_t)N, log2N);
            if (i < (uint32_t)N) {
                float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), sh0);
                float s1 = shift_wrap_clamp(ham1(i), sh1);
                float d[3];
                sample_dir(mode, s0, s1, 0.f, f, d);
                Hit h = trace_closest<STATS, LSTK, WIDTH, TOPLDS>(sc, px, py, pz, d[0], d[1], d[2], c_nodes, c_tris, lds_top);
                if (STATS) c_rays++;
                if (h.slot >= 0 && h.t > 1e-4f) {          // tracer_o3d_irt.py:248
                    float L[3];
                    shade_hit(sc, h.slot, h.u, h.v, L);
                    // :170 clamp(n . l, 0, 1) with the RAW normal
                    float ndl = fminf(fmaxf(nx * d[0] + ny * d[1] + nz * d[2], 0.f), 1.f);
                    acc0 += L[0] * ndl; acc1 += L[1] * ndl; acc2 += L[2] * ndl;
                    if (STATS) c_hits++;
                }
            }
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
        if (lane == 0) {
            if (MODE) {
                if (cell != 0) { acc0 += irr[3 * t]; acc1 += irr[3 * t + 1]; acc2 += irr[3 * t + 2]; }
            }
            if (!MODE || last_cell) {
                // :171  sum * 2 * np.pi / N
                const float pi = 3.141592653589793f;
                acc0 = ((acc0 * 2.f) * pi) / (float)N; acc1 = ((acc1 * 2.f) * pi) / (float)N; acc2 = ((acc2 * 2.f) * pi) / (float)N;
            }
            irr[3 * t] = acc0; irr[3 * t + 1] = acc1; irr[3 * t + 2] = acc2;
        }
    }
    if (STATS) {
        // integer wave reductions
        uint32_t v[4] = {c_rays, c_nodes, c_tris, c_hits};
        for (int q = 0; q < 4; q++) {
            unsigned long long x = v[q];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if (lane == 0 && x) atomicAdd(&stats[q], x);
        }
    }
}

// Persistent-wave variant with lane refill (wavefront ballot / prefix-sum ray compaction): a wave owns one texel at a time
// and keeps its 64 lanes busy -- whenever >= refill_min lanes have finished their ray, the finished rays are shaded in one
// batch and the idle lanes take the texel's next samples (rank among idle lanes = mbcnt of the ballot mask).
template <int WIDTH>
__global__ __launch_bounds__(kBlock) void irt_refill_kernel(SceneDev sc, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                            const float* __restrict__ shift, const int32_t* __restrict__ ids, int64_t n_ids,
                                                            int N, int log2N, int mode, int refill_min, float* __restrict__ irr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + wave, nw = (int64_t)gridDim.x * (kBlock / 64);
    int ovf[kStackCap - kLdsStack];
    for (int64_t k = gw; k < n_ids; k += nw) {
        const int64_t t = ids ? (int64_t)ids[k] : k;
        const float px = pos[3 * t], py = pos[3 * t + 1], pz = pos[3 * t + 2];
        const float nx = nrm[3 * t], ny = nrm[3 * t + 1], nz = nrm[3 * t + 2];
        const float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
        const Frame f = make_frame(nx, ny, nz);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        RayState r;
        r.node = kSentinel; r.sp = 0; r.h.slot = -1; r.h.t = 0.f; r.h.u = r.h.v = 0.f;
        r.dx = r.dy = r.dz = 0.f; r.idx = r.idy = r.idz = 0.f; r.oodx = r.oody = r.oodz = 0.f;
        bool have_ray = false;
        int q_next = 0;                                        // wave-uniform: samples handed out so far
        for (;;) {
            const bool idle = r.node == kSentinel;
            // finished rays: hit shader + accumulate (batched over all lanes that finished since the last refill)
            if (idle && have_ray) {
                if (r.h.slot >= 0 && r.h.t > 1e-4f) {              // tracer_o3d_irt.py:248
                    float L[3];
                    shade_hit(sc, r.h.slot, r.h.u, r.h.v, L);
                    float ndl = fminf(fmaxf(nx * r.dx + ny * r.dy + nz * r.dz, 0.f), 1.f);     // :170, RAW normal
                    acc0 += L[0] * ndl; acc1 += L[1] * ndl; acc2 += L[2] * ndl;
                }
                have_ray = false;
            }
            // compaction: idle lanes take the next samples, rank = number of idle lanes below this one
            const unsigned long long idle_mask = __ballot(idle);
            if (q_next < N) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const int q = q_next + rank;
                if (idle && q < N) {
                    uint32_t i = sample_index((uint32_t)(q >> 6), (uint32_t)(q & 63), (uint32_t)N, log2N);
                    if (i >= (uint32_t)N) i = (uint32_t)q;          // (N not a multiple of 64: natural order already)
                    float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), sh0);
                    float s1 = shift_wrap_clamp(ham1(i), sh1);
                    float d[3];
                    sample_dir(mode, s0, s1, 0.f, f, d);
                    ray_begin(r, px, py, pz, d[0], d[1], d[2]);
                    have_ray = true;
                }
                q_next += __popcll(idle_mask);
            }
            if (__ballot(r.node != kSentinel) == 0ull) break;
            trace_resume<kLdsStack, WIDTH>(sc, r, ovf, px, py, pz, q_next < N, refill_min);
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
        if (lane == 0) {
            const float pi = 3.141592653589793f;
            irr[3 * t] = ((acc0 * 2.f) * pi) / (float)N;
            irr[3 * t + 1] = ((acc1 * 2.f) * pi) / (float)N;
            irr[3 * t + 2] = ((acc2 * 2.f) * pi) / (float)N;
        }
    }
}

template <int WIDTH>
__global__ __launch_bounds__(kBlock) void trace_shade_kernel(SceneDev sc, const float* __restrict__ org, const float* __restrict__ dir,
                                                             int64_t R, float t_min, float* __restrict__ rad, float* __restrict__ t_hit,
                                                             uint32_t* __restrict__ prim, float* __restrict__ puv)
{
    uint32_t cn = 0, ct = 0;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < R; r += (int64_t)gridDim.x * kBlock) {
        float ox = org[3 * r], oy = org[3 * r + 1], oz = org[3 * r + 2];
        float dx = dir[3 * r], dy = dir[3 * r + 1], dz = dir[3 * r + 2];
        Hit h = trace_closest<false, kLdsStack, WIDTH>(sc, ox, oy, oz, dx, dy, dz, cn, ct);
        float L[3] = {0.f, 0.f, 0.f};
        bool hit = h.slot >= 0 && h.t > t_min;
        if (hit) shade_hit(sc, h.slot, h.u, h.v, L);
        rad[3 * r] = L[0]; rad[3 * r + 1] = L[1]; rad[3 * r + 2] = L[2];
        if (t_hit) t_hit[r] = h.t;
        if (prim) prim[r] = h.slot >= 0 ? __float_as_uint(sc.tris[3 * (size_t)h.slot].w) : 0xFFFFFFFFu;
        if (puv) { puv[2 * r] = h.u; puv[2 * r + 1] = h.v; }
    }
}

__global__ __launch_bounds__(256) void gen_dir_kernel(const float* __restrict__ normals, const float* __restrict__ rough,
                                                      const float* __restrict__ shift, int64_t b, int N, int mode, float* __restrict__ L)
{
    int64_t total = b * (int64_t)N;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int64_t p = g / N; uint32_t i = (uint32_t)(g - p * N);
        Frame f = make_frame(normals[3 * p], normals[3 * p + 1], normals[3 * p + 2]);
        float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), shift[2 * p]);
        float s1 = shift_wrap_clamp(ham1(i), shift[2 * p + 1]);
        float d[3];
        sample_dir(mode, s0, s1, rough ? rough[p] : 0.f, f, d);
        L[3 * g] = d[0]; L[3 * g + 1] = d[1]; L[3 * g + 2] = d[2];
    }
}

// ------------------------------------------------------------------------------------------------
// material pixel: forward-mode dual numbers carry d/d(roughness) through the whole sample chain
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
struct Dual { float v, d; };
__device__ __forceinline__ Dual dmul(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual dadd(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual dsub(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual ddiv(Dual a, Dual b) { float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
__device__ __forceinline__ Dual dconst(float c) { return {c, 0.f}; }
__device__ __forceinline__ Dual dscale(Dual a, float s) { return {a.v * s, a.d * s}; }
__device__ __forceinline__ Dual dsqrt(Dual a) { float s = sqrtf(a.v); return {s, a.d / (2.f * s)}; }
// torch.clamp backward: gradient passes where min <= x <= max (inclusive)
__device__ __forceinline__ Dual dclamp(Dual a, float lo, float hi) { return {fminf(fmaxf(a.v, lo), hi), (a.v >= lo && a.v <= hi) ? a.d : 0.f}; }
__device__ __forceinline__ Dual dclamp_min(Dual a, float lo) { return {fmaxf(a.v, lo), a.v >= lo ? a.d : 0.f}; }

struct SpecSample { Dual w; float l[3]; };

// one GGX sample of pixel (n, v, r): returns the reflected direction l and the estimator weight w (+dw/dr)
__device__ __forceinline__ SpecSample spec_sample(const Frame& f, float nx, float ny, float nz, float vx, float vy, float vz,
                                                  float r, float s0, float s1)
{
    // generate_dir importance branch (sample_util.py:133-143)
    Dual rr = {r, 1.f};
    Dual a = dmul(rr, rr);
    Dual den = dadd(dconst(1.0f), dscale(dsub(dmul(a, a), dconst(1.f)), s0));
    Dual ct = dsqrt(ddiv(dconst(1.0f - s0), den));
    ct = dclamp(ct, -1.0f + 1e-6f, 1.0f - 1e-6f);
    Dual st = dsqrt(dsub(dconst(1.0f), dmul(ct, ct)));
    st = dclamp(st, -1.0f + 1e-6f, 1.0f - 1e-6f);
    float phi = 6.283185307179586f * s1 - 3.141592653589793f;
    float sphi, cphi;
    sincosf(phi, &sphi, &cphi);
    Dual sp = dscale(st, sphi);
    Dual cp = dscale(st, cphi); cp.v = -cp.v; cp.d = -cp.d;
    Dual h[3];
    for (int k = 0; k < 3; k++) h[k] = dadd(dadd(dscale(sp, f.V[k]), dscale(ct, f.n[k])), dscale(cp, f.U[k]));
    // render (mat_nvdiffrast.py:235-236) and specular_reflectance (:262-279); dots use the RAW normal
    Dual vdh = dclamp(dadd(dadd(dscale(h[0], vx), dscale(h[1], vy)), dscale(h[2], vz)), 0.f, 1.f);
    Dual l[3];
    const float vv[3] = {vx, vy, vz};
    for (int k = 0; k < 3; k++) l[k] = dsub(dscale(dmul(vdh, h[k]), 2.f), dconst(vv[k]));
    Dual ndl = dclamp(dadd(dadd(dscale(l[0], nx), dscale(l[1], ny)), dscale(l[2], nz)), 0.f, 1.f);
    Dual ndh = dclamp(dadd(dadd(dscale(h[0], nx), dscale(h[1], ny)), dscale(h[2], nz)), 0.f, 1.f);
    float ndv = fminf(fmaxf(nx * vx + ny * vy + nz * vz, 0.f), 1.f);
    // f = 0.04 + 0.96 * 2^((-5.55472*vdh - 6.98316)*vdh)
    Dual e = dmul(dsub(dscale(vdh, -5.55472f), dconst(6.98316f)), vdh);
    float p2 = exp2f(e.v);
    Dual fr = {0.04f + 0.96f * p2, 0.96f * p2 * 0.6931471805599453f * e.d};
    Dual kk = dscale(dmul(dadd(rr, dconst(1.f)), dadd(rr, dconst(1.f))), 0.125f);
    Dual omk = dsub(dconst(1.f), kk);
    Dual g1v = ddiv(dconst(ndv), dclamp_min(dadd(dscale(omk, ndv), kk), 1e-14f));
    Dual g1l = ddiv(ndl, dclamp_min(dadd(dmul(ndl, omk), kk), 1e-14f));
    Dual g = dmul(g1l, g1v);
    Dual brdf = ddiv(dmul(fr, g), dclamp_min(dscale(ndl, 4.f * ndv), 1e-14f));
    Dual w = ddiv(dmul(dscale(dmul(brdf, ndl), 4.f), vdh), dclamp_min(ndh, 1e-14f));
    SpecSample o;
    o.w = w;
    for (int k = 0; k < 3; k++) o.l[k] = l[k].v;
    return o;
}
#pragma clang fp contract(fast)

// lanes-per-pixel = S when S is a power of two <= 64 (several pixels per wave), else 64 with ceil(S/64) passes
template <bool BWD, int WIDTH>
__global__ __launch_bounds__(kBlock) void spec_kernel(SceneDev sc, const float* __restrict__ normal, const float* __restrict__ albedo,
                                                      const float* __restrict__ rough, const float* __restrict__ points,
                                                      const float* __restrict__ irr, const float* __restrict__ cam,
                                                      const float* __restrict__ shift, int64_t P, int S, int lpp,
                                                      float* __restrict__ rgb, float* __restrict__ Ls_ws,
                                                      const float* __restrict__ d_rgb, float* __restrict__ d_albedo, float* __restrict__ d_rough)
{
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / lpp;                              // pixels per wave
    const int sub = lane / lpp, sl = lane % lpp;
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const float cx = cam[0], cy = cam[1], cz = cam[2];
    const int passes = (S + lpp - 1) / lpp;
    uint32_t cn = 0, ct = 0;
    for (int64_t base = gw * ppw; base < P; base += nw * ppw) {
        const int64_t p = base + sub;
        const bool live = p < P;
        float acc[3] = {0.f, 0.f, 0.f};
        float dacc = 0.f;
        float nx = 0, ny = 0, nz = 0, r = 0.1f, ox = 0, oy = 0, oz = 0, vx = 0, vy = 0, vz = 0, sh0 = 0, sh1 = 0;
        float g0 = 0, g1 = 0, g2 = 0;
        if (live) {
            nx = normal[3 * p]; ny = normal[3 * p + 1]; nz = normal[3 * p + 2];
            r = rough[p];
            ox = points[3 * p]; oy = points[3 * p + 1]; oz = points[3 * p + 2];
            sh0 = shift[2 * p]; sh1 = shift[2 * p + 1];
            // F.normalize(cam - p, eps=1e-4)  (mat_nvdiffrast.py:218)
            vx = cx - ox; vy = cy - oy; vz = cz - oz;
            float lv = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-4f);
            vx /= lv; vy /= lv; vz /= lv;
            if (BWD) { g0 = d_rgb[3 * p]; g1 = d_rgb[3 * p + 1]; g2 = d_rgb[3 * p + 2]; }
        }
        const Frame f = make_frame(nx, ny, nz);
        for (int q = 0; q < passes; q++) {
            const int i = q * lpp + sl;
            if (live && i < S) {
                float s0 = shift_wrap_clamp(ham0((uint32_t)i, (uint32_t)S), sh0);
                float s1 = shift_wrap_clamp(ham1((uint32_t)i), sh1);
                SpecSample ss = spec_sample(f, nx, ny, nz, vx, vy, vz, r, s0, s1);
                float L[3] = {0.f, 0.f, 0.f};
                if (BWD) {
                    const float* lp = Ls_ws + 3 * ((size_t)p * S + i);
                    L[0] = lp[0]; L[1] = lp[1]; L[2] = lp[2];
                    dacc += (L[0] * g0 + L[1] * g1 + L[2] * g2) * ss.w.d;
                } else {
                    Hit h = trace_closest<false, kLdsStack, WIDTH>(sc, ox, oy, oz, ss.l[0], ss.l[1], ss.l[2], cn, ct);
                    if (h.slot >= 0 && h.t > 1e-4f) shade_hit(sc, h.slot, h.u, h.v, L);
                    if (Ls_ws) { float* lp = Ls_ws + 3 * ((size_t)p * S + i); lp[0] = L[0]; lp[1] = L[1]; lp[2] = L[2]; }
                    acc[0] += L[0] * ss.w.v; acc[1] += L[1] * ss.w.v; acc[2] += L[2] * ss.w.v;
                }
            }
        }
        // reduce over the lpp lanes of this pixel
        for (int o = lpp >> 1; o > 0; o >>= 1) {
            if (BWD) dacc += __shfl_xor(dacc, o, 64);
            else { acc[0] += __shfl_xor(acc[0], o, 64); acc[1] += __shfl_xor(acc[1], o, 64); acc[2] += __shfl_xor(acc[2], o, 64); }
        }
        if (live && sl == 0) {
            const float pi = 3.141592653589793f;
            if (BWD) {
                if (d_rough) d_rough[p] = dacc / (float)S;
                if (d_albedo) {
                    d_albedo[3 * p] = g0 * irr[3 * p] / pi; d_albedo[3 * p + 1] = g1 * irr[3 * p + 1] / pi; d_albedo[3 * p + 2] = g2 * irr[3 * p + 2] / pi;
                }
            } else {
                for (int c = 0; c < 3; c++) rgb[3 * p + c] = irr[3 * p + c] * albedo[3 * p + c] / pi + acc[c] / (float)S;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int grid_for(int64_t work_items_per_block_unit, int64_t n)
{
    // >> 256 workgroups to fill 256 CUs x several blocks/CU; persistent grid-stride above that
    int64_t want = (n + work_items_per_block_unit - 1) / work_items_per_block_unit;
    int64_t cap = 256 * 8;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

static int ilog2_exact(int N) { if (N <= 0 || (N & (N - 1))) return 0; int l = 0; while ((1 << l) < N) l++; return l; }

// workgroups that are co-resident on the whole chip for a kernel (so a grid-stride loop has no second, partial round)
template <typename K>
static int resident_grid(K kernel, int block)
{
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    return cus * per_cu;
}

static int irt_variant()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("TEXIR_IRT_VARIANT"); v = e ? atoi(e) : 1; }
    return v;
}

int irt_launch_count(int N) { return (irt_variant() == 2 && ilog2_exact(N) >= 7) ? (N >> 6) : 1; }

template <bool STATS, int MODE, int LSTK, int MINW, int WIDTH, bool TOPLDS>
static void irt_launch_w(bool resident, const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t n_ids,
                         int N, int l2, int mode, int cell, int last, float* irr, unsigned long long* stats, hipStream_t st)
{
    int64_t want = (n_ids + (kBlock / 64) - 1) / (kBlock / 64);
    int grid = resident ? resident_grid(irt_kernel<STATS, MODE, LSTK, MINW, WIDTH, TOPLDS>, kBlock) : 2048;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL((irt_kernel<STATS, MODE, LSTK, MINW, WIDTH, TOPLDS>), dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, cell, last, irr, stats);
}

template <bool STATS, int MODE, int LSTK, int MINW>
static void irt_launch_one(bool resident, const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t n_ids,
                           int N, int l2, int mode, int cell, int last, float* irr, unsigned long long* stats, hipStream_t st)
{
    static const bool top_lds = getenv("TEXIR_TOP_LDS") ? atoi(getenv("TEXIR_TOP_LDS")) != 0 : false;
    if (sc.nodes4 && top_lds) irt_launch_w<STATS, MODE, LSTK, MINW, 4, true>(resident, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, cell, last, irr, stats, st);
    else if (sc.nodes4) irt_launch_w<STATS, MODE, LSTK, MINW, 4, false>(resident, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, cell, last, irr, stats, st);
    else irt_launch_w<STATS, MODE, LSTK, MINW, 2, false>(resident, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, cell, last, irr, stats, st);
}

hipError_t launch_irt(const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t n_ids,
                      int N, int mode, float* irr, unsigned long long* stats, hipStream_t st)
{
    if (n_ids <= 0) return hipSuccess;
    // variants (TEXIR_IRT_VARIANT, default 1): 0 capped grid; 1 resident grid; 2 resident + cell-major launches
    const int variant = irt_variant();
    int l2 = ilog2_exact(N);
    if (stats) { irt_launch_one<true, 0, 24, 1>(variant >= 1, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, 0, 1, irr, stats, st); return hipGetLastError(); }
    if (variant == 2 && l2 >= 7) {
        const int cells = N >> 6;
        for (int j = 0; j < cells; j++) irt_launch_one<false, 1, 24, 1>(true, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, j, j == cells - 1, irr, stats, st);
    } else if (variant >= 5) {
        // 5: persistent waves with lane refill; refill threshold from TEXIR_REFILL_MIN (default 16 idle lanes)
        static const int refill_min = getenv("TEXIR_REFILL_MIN") ? atoi(getenv("TEXIR_REFILL_MIN")) : 16;
        int64_t want = (n_ids + (kBlock / 64) - 1) / (kBlock / 64);
        if (sc.nodes4) {
            int grid = resident_grid(irt_refill_kernel<4>, kBlock); if (want < grid) grid = (int)want;
            hipLaunchKernelGGL(irt_refill_kernel<4>, dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, refill_min, irr);
        } else {
            int grid = resident_grid(irt_refill_kernel<2>, kBlock); if (want < grid) grid = (int)want;
            hipLaunchKernelGGL(irt_refill_kernel<2>, dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, refill_min, irr);
        }
    } else irt_launch_one<false, 0, 24, 1>(variant >= 1, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, 0, 1, irr, stats, st);
    return hipGetLastError();
}

hipError_t launch_trace_shade(const SceneDev& sc, const float* org, const float* dir, int64_t R, float t_min, float* rad, float* t_hit,
                              uint32_t* prim, float* puv, hipStream_t st)
{
    if (R <= 0) return hipSuccess;
    if (sc.nodes4) hipLaunchKernelGGL(trace_shade_kernel<4>, dim3(grid_for(kBlock, R)), dim3(kBlock), 0, st, sc, org, dir, R, t_min, rad, t_hit, prim, puv);
    else hipLaunchKernelGGL(trace_shade_kernel<2>, dim3(grid_for(kBlock, R)), dim3(kBlock), 0, st, sc, org, dir, R, t_min, rad, t_hit, prim, puv);
    return hipGetLastError();
}

hipError_t launch_gen_dir(const float* normals, const float* rough, const float* shift, int64_t b, int N, int mode, float* L, hipStream_t st)
{
    if (b <= 0 || N <= 0) return hipSuccess;
    hipLaunchKernelGGL(gen_dir_kernel, dim3(grid_for(256, b * (int64_t)N)), dim3(256), 0, st, normals, rough, shift, b, N, mode, L);
    return hipGetLastError();
}

static int lanes_per_pixel(int S) { return (S <= 64 && (S & (S - 1)) == 0) ? S : 64; }

hipError_t launch_spec_fwd(const SceneDev& sc, const float* normal, const float* albedo, const float* rough, const float* points,
                           const float* irr, const float* cam, const float* shift, int64_t P, int S, float* rgb, float* Ls_ws, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    int lpp = lanes_per_pixel(S);
    int64_t pix_per_block = (int64_t)(kBlock / 64) * (64 / lpp);
    if (sc.nodes4)
        hipLaunchKernelGGL((spec_kernel<false, 4>), dim3(grid_for(pix_per_block, P)), dim3(kBlock), 0, st, sc, normal, albedo, rough, points, irr, cam,
                           shift, P, S, lpp, rgb, Ls_ws, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    else
        hipLaunchKernelGGL((spec_kernel<false, 2>), dim3(grid_for(pix_per_block, P)), dim3(kBlock), 0, st, sc, normal, albedo, rough, points, irr, cam,
                           shift, P, S, lpp, rgb, Ls_ws, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    return hipGetLastError();
}

hipError_t launch_spec_bwd(const float* normal, const float* rough, const float* points, const float* irr, const float* cam,
                           const float* shift, const float* Ls_ws, const float* d_rgb, int64_t P, int S, float* d_albedo, float* d_rough,
                           hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    int lpp = lanes_per_pixel(S);
    int64_t pix_per_block = (int64_t)(kBlock / 64) * (64 / lpp);
    SceneDev none{};
    hipLaunchKernelGGL((spec_kernel<true, 2>), dim3(grid_for(pix_per_block, P)), dim3(kBlock), 0, st, none, normal, (const float*)nullptr, rough,
                       points, irr, cam, shift, P, S, lpp, (float*)nullptr, const_cast<float*>(Ls_ws), d_rgb, d_albedo, d_rough);
    return hipGetLastError();
}

}  // namespace texir
