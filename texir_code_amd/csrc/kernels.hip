// gfx950 kernels of libtexir_hip.so + their launchers.
//   irt_kernel          fused sample + trace + shade + ndl-weighted reduce   (models/tracer_o3d_irt.py:156-178)
//   trace_shade_kernel  query_irf                                            (models/tracer_o3d_irt.py:240-269)
//   gen_dir_kernel      generate_dir                                         (utils/sample_util.py:63-146)
//   spec_fwd/bwd_kernel render + specular_reflectance and its analytic grad  (models/mat_nvdiffrast.py:201-279)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "device_common.h"
#include "env.h"
#include "kernels.h"

namespace texir {

static int grid_for_static(int64_t per_block, int64_t n) { int64_t w = (n + per_block - 1) / per_block; return (int)(w < 1 ? 1 : (w > 2048 ? 2048 : w)); }

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

// stats [8] (include/texir_hip.h): rays, node fetches, triangle tests, hits, wave-level node steps, wave-level triangle steps
__device__ __forceinline__ void irt_stats_flush(unsigned long long* stats, int lane, uint32_t rays, uint32_t nodes, uint32_t tris,
                                                uint32_t hits, uint32_t wnodes, uint32_t wtris)
{
    uint32_t v[6] = {rays, nodes, tris, hits, wnodes, wtris};
    for (int q = 0; q < 6; q++) {
        unsigned long long x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0 && x) atomicAdd(&stats[q], x);
    }
}

#if TEXIR_CHAIN_PROBE
// stats[8 + q] of a probe build (device_common.h TEXIR_CHAIN_PROBE), added once per chunk by lane 0; q =
//   0-2 steps / timed steps / cycles of the timed steps of per-lane (vector) node steps, 3-5 of wave-uniform (scalar-cache) node steps, 6-8 of leaf steps,
//   9-10 timed empty regions / their cycles   [trace_core];   11 passes, 12 timed passes, 13 cycles inside trace_closest, 14 cycles in the hit shader,
//   15 cycles of whole passes (sampling + trace + shade) -- of the timed passes (every 8th);  16 cycles of whole chunks, 17 chunks
constexpr int kProbeStats = 18;
__device__ __forceinline__ void irt_probe_flush(unsigned long long* stats, int lane, const uint32_t* v)
{
    if (lane == 0 && stats) for (int q = 0; q < kProbeStats; q++) if (v[q]) atomicAdd(&stats[8 + q], (unsigned long long)v[q]);
}
#endif

// Occupancy.  Round 1 (4-byte stack entries): 5 waves / 24 entries 13.85, 6 / 24 14.79, 7 / 16 15.11, 8 / 16 15.06 Grays/s (c4).  Round 2, after the
// scalar node path took the L1 off the critical path (8-byte entries): 6 waves / 12 entries 15.01, 7 / 11 15.82, 8 / 10 15.87 (c2: 16.30, 17.11, 17.40;
// c4_scan: 4.98, 5.31, 5.54): 8 waves per SIMD = 64 VGPRs (the compiler parks the per-texel frame and the ray's shear rows in scratch across the
// traversal loop) and a 10-entry LDS stack.
#ifndef TEXIR_CULL
#define TEXIR_CULL 1
#endif
constexpr bool kCull = TEXIR_CULL != 0;
constexpr int kEstimatorCosine = 4;      // or-ed into the IrT kernels' `mode`: the cosine branch of diffuse_reflectance (mat_nvdiffrast.py:256-257)
#ifndef TEXIR_GROUP_LSTK
#define TEXIR_GROUP_LSTK (TEXIR_CULL ? 10 : 16)         // 8-byte entries with culling: 10 x 2 KiB = 20 KiB per block, 8 blocks = all 160 KiB of a CU
#endif
constexpr int kGroupLstk = TEXIR_GROUP_LSTK;
#ifndef TEXIR_GROUP_WAVES
#define TEXIR_GROUP_WAVES 8
#endif
constexpr int kGroupWaves = TEXIR_GROUP_WAVES;
constexpr int kLstk = kCull ? kLdsStack / 2 : kLdsStack;   // the other tracing kernels: 24 KiB of stack per block either way
// chunk hand-out of irt_group_kernel (see there): 0 = one counter, parts = elevation rings (round 2)
#ifndef TEXIR_XCD_SCHED
#define TEXIR_XCD_SCHED 1
#endif
#ifndef TEXIR_PART_WEDGE
#define TEXIR_PART_WEDGE 1
#endif
// azimuth sine / cosine of the fused IrT sampling from v_sin_f32 / v_cos_f32 (device_common.h sample_dir<FAST>)
#ifndef TEXIR_IRT_FAST_SINCOS
#define TEXIR_IRT_FAST_SINCOS 0
#endif

// One texel per wave: the 64 lanes trace 64 samples of the texel per pass (any N, binary or 4-wide tree).  Kept as the
// form for short texel lists (a 1024-point NIrF batch), for binary-tree scenes, and TEXIR_IRT_TEXELS_PER_WAVE=1.
template <bool STATS, int WIDTH>
__global__ __launch_bounds__(kBlock, kGroupWaves) void irt_kernel(SceneDev sc, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                     const float* __restrict__ shift, const int32_t* __restrict__ ids, int64_t n_ids,
                                                     int N, int log2N, int mode, float* __restrict__ irr,
                                                     unsigned long long* __restrict__ stats, unsigned long long* /*work*/, float* /*partial*/, int /*log2parts*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + wave, nw = (int64_t)gridDim.x * (kBlock / 64);
    uint32_t c_nodes = 0, c_tris = 0, c_rays = 0, c_hits = 0, wi[2] = {0, 0};
    const bool cosw = (mode & kEstimatorCosine) != 0;        // diffuse_reflectance's cosine branch: sum L * pi / N, no n.l factor
    mode &= 3;
    const int passes = (N + 63) >> 6;
    for (int64_t k = gw; k < n_ids; k += nw) {
        const int64_t t = ids ? (int64_t)ids[k] : k;
        const float px = pos[3 * t], py = pos[3 * t + 1], pz = pos[3 * t + 2];
        const float nx = nrm[3 * t], ny = nrm[3 * t + 1], nz = nrm[3 * t + 2];
        const float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
        const Frame f = make_frame(nx, ny, nz);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        for (int p = 0; p < passes; p++) {
            uint32_t i = sample_index((uint32_t)p, (uint32_t)lane, (uint32_t)N, log2N);
            if (i < (uint32_t)N) {
                float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), sh0);
                float s1 = shift_wrap_clamp(ham1(i), sh1);
                float d[3];
                sample_dir(mode, s0, s1, 0.f, f, d);
                Hit h = trace_closest<STATS, kGroupLstk, WIDTH, kCull>(sc, px, py, pz, d[0], d[1], d[2], c_nodes, c_tris, STATS ? wi : nullptr);
                if (STATS) c_rays++;
                if (h.slot >= 0 && h.t > 1e-4f) {          // tracer_o3d_irt.py:248
                    float L[3];
                    shade_hit(sc, h.slot, h.u, h.v, L);
                    // :170 clamp(n . l, 0, 1) with the RAW normal
                    float ndl = cosw ? 1.f : fminf(fmaxf(nx * d[0] + ny * d[1] + nz * d[2], 0.f), 1.f);
                    acc0 += L[0] * ndl; acc1 += L[1] * ndl; acc2 += L[2] * ndl;
                    if (STATS) c_hits++;
                }
            }
        }
        acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
        if (lane == 0) {
            // :171  sum * 2 * np.pi / N      (cosine estimator, mat_nvdiffrast.py:256-257: sum * np.pi / N)
            const float pi = 3.141592653589793f, two = cosw ? 1.f : 2.f;
            irr[3 * t] = ((acc0 * two) * pi) / (float)N;
            irr[3 * t + 1] = ((acc1 * two) * pi) / (float)N;
            irr[3 * t + 2] = ((acc2 * two) * pi) / (float)N;
        }
    }
    if (STATS) irt_stats_flush(stats, lane, c_rays, c_nodes, c_tris, c_hits, wi[0], wi[1]);
}

// Multi-texel passes.  A wave traces GRP neighbouring texels at once: lane group g (64/GRP lanes) belongs to texel g and all groups
// take, in the same pass, the lattice cell of THEIR texel that lies nearest to one absolute direction cell J (cell_to_pass_m).
// The 64 rays of a pass then span 1/(32*GRP) of the hemisphere instead of 1/32 (for N = 2048) and start within a few
// millimetres of each other: their traversals stay together much longer, which is what an issue-bound SIMT traversal needs.
// Every sample of every texel is still traced exactly once (J -> cell is a bijection for a fixed shift).
__device__ __forceinline__ uint32_t sample_index_m(uint32_t cell, uint32_t sub, int log2N, int log2m)
{
    int cells = log2N - log2m;
    int bphi = (cells + 1) >> 1, bth = cells - bphi;
    uint32_t low = cell & ((1u << bphi) - 1u);
    uint32_t th = bth ? (cell >> bphi) : 0u;
    return (th << (log2N - bth)) | (sub << bphi) | low;
}

__device__ __forceinline__ uint32_t cell_to_pass_m(uint32_t J, float sh0, float sh1, int log2N, int log2m)
{
    int cells = log2N - log2m;
    int bphi = (cells + 1) >> 1, bth = cells - bphi;
    uint32_t nphi = 1u << bphi, nth = 1u << bth;
    uint32_t Jphi = J & (nphi - 1u), Jth = J >> bphi;
    uint32_t dphi = (uint32_t)(sh1 * (float)nphi + 0.5f), dth = (uint32_t)(sh0 * (float)nth + 0.5f);
    uint32_t phibin = (Jphi + nphi - (dphi & (nphi - 1u))) & (nphi - 1u);
    uint32_t th = (Jth + nth - (dth & (nth - 1u))) & (nth - 1u);
    uint32_t low = bphi ? (__brev(phibin) >> (32 - bphi)) : 0u;
    return (th << bphi) | low;
}

template <bool STATS, int WIDTH, int LOG2GRP>
__global__ __launch_bounds__(kBlock, kGroupWaves) void irt_group_kernel(SceneDev sc, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                           const float* __restrict__ shift, const int32_t* __restrict__ ids, int64_t n_ids,
                                                           int N, int log2N, int mode, float* __restrict__ irr,
                                                           unsigned long long* __restrict__ stats, unsigned long long* __restrict__ work,
                                                           float* __restrict__ partial, int log2parts)
{
    constexpr int GRP = 1 << LOG2GRP, LOG2M = 6 - LOG2GRP, M = 64 >> LOG2GRP;       // texels per wave, samples per texel per pass
    const int lane = threadIdx.x & 63;
    const int grp = lane >> LOG2M, sub = lane & (M - 1);
    const int n_cells = N >> LOG2M;
    uint32_t cn = 0, ct = 0, c_rays = 0, c_hits = 0, wi[2 + kProbeSlots] = {0};
    const bool cosw = (mode & kEstimatorCosine) != 0;
    mode &= 3;
    // A chunk = GRP texels x (all passes / 2^log2parts).  With parts > 1 the raw partial sums go to partial[part][k][3] and
    // irt_combine_kernel adds them in part order: the result depends on N only, never on how the texel list is cut or scheduled.
    const int part_cells = n_cells >> log2parts;
    // Which cells a part holds (TEXIR_PART_WEDGE = 1): the cells are walked azimuth-major (all elevations of one azimuth bin, then the next
    // bin), so a part is an azimuthal WEDGE of the hemisphere (N = 2048: 2 of the 64 azimuth bins = 5.6 degrees, all 32 elevations) instead of a
    // full ring of one elevation.  The rays of a wedge leave a surface patch towards one side of the room: what they touch deep in the tree,
    // their triangles and their radiance-texture lines are shared by the chunks of neighbouring wedges -- which the hand-out below keeps on one XCD.
    const int cell_bits = log2N < 0 ? 0 : log2N - LOG2M, bphi = (cell_bits + 1) >> 1, bth = cell_bits - bphi;
    // XCD-aware hand-out (TEXIR_XCD_SCHED = 1): the 8 XCDs have private 4 MiB L2s; with ONE chunk counter consecutive chunks -- the 32 parts of the
    // same 64 texels -- go to whichever waves ask next, so every L2 sees every direction of every region.  Here each XCD owns parts / 8
    // neighbouring wedges (a 45-degree sector at N = 2048) of ALL texel groups and pulls them from its own counter (its own 128-byte line: 8
    // heads also dequeue faster than one, MI355X_MICROARCH.md "dequeue"); an XCD whose sector has run dry steals from the next one's.  HW_REG_XCC_ID
    // is a speed hint only: any wave may execute any chunk, the partial sums are indexed by (part, texel).
    const int n_own = (TEXIR_XCD_SCHED && log2parts >= 3) ? 8 : 1;
    const int log2ppo = log2parts - (n_own == 8 ? 3 : 0);                        // parts per owner
    const int64_t n_groups = (n_ids + GRP - 1) / GRP;
    int owner = n_own == 8 ? (__builtin_amdgcn_s_getreg(6164 /* hwreg(HW_REG_XCC_ID, 0, 4) */) & 7) : 0;
    int dry = 0;
    for (;;) {
        // persistent waves pull chunks from a counter: a chunk is milliseconds of work, so a static round-robin would
        // leave the slowest wave's surplus (the sum of its chunk-time deviations) as an idle tail
        unsigned long long chunk = 0;
        if (lane == 0) chunk = atomicAdd(work + owner * kWorkStride, 1ull);
        chunk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(chunk >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)chunk);
        if ((int64_t)(chunk >> log2ppo) >= n_groups) {
            if (++dry >= n_own) break;                                           // every owner's queue is empty
            owner = (owner + 1) & (n_own - 1);
            continue;
        }
        dry = 0;
#if TEXIR_CHAIN_PROBE
        uint32_t pv[kProbeStats] = {0};
        const uint32_t chunk_c0 = probe_clock();
        for (int q = 2; q < 2 + kProbeSlots; q++) wi[q] = 0;
#endif
        const int part = (owner << log2ppo) | (int)(chunk & ((1ull << log2ppo) - 1ull));
        const int64_t k0 = (int64_t)(chunk >> log2ppo) * GRP;
        const int64_t k = k0 + grp;
        const bool live = k < n_ids;
        const int64_t t = live ? (ids ? (int64_t)ids[k] : k) : 0;
        const float px = pos[3 * t], py = pos[3 * t + 1], pz = pos[3 * t + 2];
        const float nx = nrm[3 * t], ny = nrm[3 * t + 1], nz = nrm[3 * t + 2];
        const float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
        const Frame f = make_frame(nx, ny, nz);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        for (int Lc = part * part_cells; Lc < (part + 1) * part_cells; Lc++) {
            const int J = (TEXIR_PART_WEDGE && log2N >= 0) ? (((Lc & ((1 << bth) - 1)) << bphi) | (Lc >> bth)) : Lc;
#if TEXIR_CHAIN_PROBE
            const bool pass_timed = (Lc & 7) == 0;
            uint32_t pass_c0 = 0;
            if (pass_timed) pass_c0 = probe_clock();
            probe_add(pv[11], 1u);
#endif
            if (live) {
                // (N not a power of two: only the one-sample-per-pass form is launched, in natural sample order)
                const uint32_t i = log2N < 0 ? (uint32_t)J : sample_index_m(cell_to_pass_m((uint32_t)J, sh0, sh1, log2N, LOG2M), (uint32_t)sub, log2N, LOG2M);
                float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), sh0);
                float s1 = shift_wrap_clamp(ham1(i), sh1);
                float d[3];
                sample_dir<TEXIR_IRT_FAST_SINCOS != 0>(mode, s0, s1, 0.f, f, d);
                const float ndl = cosw ? 1.f : fminf(fmaxf(nx * d[0] + ny * d[1] + nz * d[2], 0.f), 1.f);       // :170, RAW normal (before the trace: one live register instead of three)
#if TEXIR_CHAIN_PROBE
                uint32_t tr_c0 = 0, tr_c1 = 0;
                if (pass_timed) tr_c0 = probe_clock();
                Hit h = trace_closest<STATS, kGroupLstk, WIDTH, kCull>(sc, px, py, pz, d[0], d[1], d[2], cn, ct, wi);
                if (pass_timed) { tr_c1 = probe_clock(); probe_add(pv[13], tr_c1 - tr_c0); probe_add(pv[12], 1u); }
#else
                Hit h = trace_closest<STATS, kGroupLstk, WIDTH, kCull>(sc, px, py, pz, d[0], d[1], d[2], cn, ct, STATS ? wi : nullptr);
#endif
                if (STATS) c_rays++;
                if (h.slot >= 0 && h.t > 1e-4f) {          // tracer_o3d_irt.py:248
                    float L[3];
                    shade_hit(sc, h.slot, h.u, h.v, L);
                    acc0 += L[0] * ndl; acc1 += L[1] * ndl; acc2 += L[2] * ndl;
                    if (STATS) c_hits++;
                }
#if TEXIR_CHAIN_PROBE
                // (the shader's loads are consumed by the three multiply-adds above: the clock read below is ordered behind a use of acc0 so that it cannot be
                // hoisted over the wait)
                if (pass_timed) {
                    const uint32_t sh_c1 = probe_clock() + (uint32_t)(__builtin_amdgcn_readfirstlane(__float_as_int(acc0)) & 0);
                    probe_add(pv[14], sh_c1 - tr_c1);
                }
#endif
            }
#if TEXIR_CHAIN_PROBE
            if (pass_timed) probe_add(pv[15], probe_clock() - pass_c0);
#endif
        }
        // reduce over the M lanes of each texel
        for (int o = M >> 1; o > 0; o >>= 1) { acc0 += __shfl_xor(acc0, o, 64); acc1 += __shfl_xor(acc1, o, 64); acc2 += __shfl_xor(acc2, o, 64); }
        if (live && sub == 0) {
            if (log2parts) {
                float* o = partial + ((int64_t)part * n_ids + k) * 3;
                o[0] = acc0; o[1] = acc1; o[2] = acc2;
            } else {
                const float pi = 3.141592653589793f, two = cosw ? 1.f : 2.f;
                irr[3 * t] = ((acc0 * two) * pi) / (float)N;
                irr[3 * t + 1] = ((acc1 * two) * pi) / (float)N;
                irr[3 * t + 2] = ((acc2 * two) * pi) / (float)N;
            }
        }
#if TEXIR_CHAIN_PROBE
        if constexpr (!STATS) {
            for (int q = 0; q < kProbeSlots; q++) pv[q] = wi[2 + q];
            pv[16] = probe_clock() - chunk_c0; pv[17] = 1u;
            irt_probe_flush(stats, lane, pv);
        }
#endif
    }
    if (STATS) irt_stats_flush(stats, lane, c_rays, cn, ct, c_hits, wi[0], wi[1]);
}

// irt_group_kernel with compaction by refill (device_common.h trace_core<STREAM>): same chunks, same hand-out, same per-lane sample order and
// partial sums; a lane whose ray has ended takes its texel's next direction cell as soon as `refill_at` lanes of the wave are idle instead of
// waiting for the slowest ray of the pass.  Launched when TEXIR_IRT_REFILL = 1..63 (A/B switch); power-of-two N only.
#ifndef TEXIR_STREAM_WAVES
#define TEXIR_STREAM_WAVES TEXIR_GROUP_WAVES          // waves per SIMD the stream kernel is compiled for (A/B: fewer waves = more registers, fewer spills)
#endif
template <bool STATS, int WIDTH>
__global__ __launch_bounds__(kBlock, TEXIR_STREAM_WAVES) void irt_stream_kernel(SceneDev sc, const float* __restrict__ pos, const float* __restrict__ nrm,
                                                            const float* __restrict__ shift, const int32_t* __restrict__ ids, int64_t n_ids,
                                                            int N, int log2N, int mode, float* __restrict__ irr,
                                                            unsigned long long* __restrict__ stats, unsigned long long* __restrict__ work,
                                                            float* __restrict__ partial, int log2parts, int refill_at)
{
    constexpr int GRP = 64;
    const int lane = threadIdx.x & 63;
    const int n_cells = N;
    uint32_t cn = 0, ct = 0, c_rays = 0, c_hits = 0, wi[2] = {0, 0};
    const bool cosw = (mode & kEstimatorCosine) != 0;
    mode &= 3;
    const int part_cells = n_cells >> log2parts;
    const int bphi = (log2N + 1) >> 1, bth = log2N - bphi;
    const int n_own = (TEXIR_XCD_SCHED && log2parts >= 3) ? 8 : 1;
    const int log2ppo = log2parts - (n_own == 8 ? 3 : 0);
    const int64_t n_groups = (n_ids + GRP - 1) / GRP;
    int owner = n_own == 8 ? (__builtin_amdgcn_s_getreg(6164 /* hwreg(HW_REG_XCC_ID, 0, 4) */) & 7) : 0;
    int dry = 0;
    for (;;) {
        unsigned long long chunk = 0;
        if (lane == 0) chunk = atomicAdd(work + owner * kWorkStride, 1ull);
        chunk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(chunk >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)chunk);
        if ((int64_t)(chunk >> log2ppo) >= n_groups) {
            if (++dry >= n_own) break;
            owner = (owner + 1) & (n_own - 1);
            continue;
        }
        dry = 0;
        const int part = (owner << log2ppo) | (int)(chunk & ((1ull << log2ppo) - 1ull));
        const int64_t k = (int64_t)(chunk >> log2ppo) * GRP + lane;
        const bool live = k < n_ids;
        const int64_t t = live ? (ids ? (int64_t)ids[k] : k) : 0;
        const float px = pos[3 * t], py = pos[3 * t + 1], pz = pos[3 * t + 2];
        const float nx = nrm[3 * t], ny = nrm[3 * t + 1], nz = nrm[3 * t + 2];
        const float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (live) {
            // What a refill needs of the texel (frame, raw normal, shifts: 14 floats) is PARKED in private memory for the whole chunk and read back only
            // inside `next`: held in registers across the traversal loop it pushes the loop's own temporaries into scratch (first build of this kernel:
            // spills in every node step, 4x slower than the lock-step kernel).  volatile = the compiler keeps the array in memory.
            volatile float keep[14];
            {
                const Frame f = make_frame(nx, ny, nz);
                for (int a = 0; a < 3; a++) { keep[a] = f.n[a]; keep[3 + a] = f.U[a]; keep[6 + a] = f.V[a]; }
                keep[9] = nx; keep[10] = ny; keep[11] = nz; keep[12] = sh0; keep[13] = sh1;
            }
            int Lc = part * part_cells;
            const int Lend = Lc + part_cells;
            float ndl = 0.f;
            auto next = [&](bool finished, const Hit& h, float& dx, float& dy, float& dz) -> bool {
                if (finished && h.slot >= 0 && h.t > 1e-4f) {          // tracer_o3d_irt.py:248
                    float L[3];
                    shade_hit(sc, h.slot, h.u, h.v, L);
                    acc0 += L[0] * ndl; acc1 += L[1] * ndl; acc2 += L[2] * ndl;
                    if (STATS) c_hits++;
                }
                if (Lc >= Lend) return false;
                const int J = TEXIR_PART_WEDGE ? (((Lc & ((1 << bth) - 1)) << bphi) | (Lc >> bth)) : Lc;
                Lc++;
                Frame f;
                for (int a = 0; a < 3; a++) { f.n[a] = keep[a]; f.U[a] = keep[3 + a]; f.V[a] = keep[6 + a]; }
                const float rnx = keep[9], rny = keep[10], rnz = keep[11], rs0 = keep[12], rs1 = keep[13];
                const uint32_t i = sample_index_m(cell_to_pass_m((uint32_t)J, rs0, rs1, log2N, 0), 0u, log2N, 0);
                const float s0 = shift_wrap_clamp(ham0(i, (uint32_t)N), rs0);
                const float s1 = shift_wrap_clamp(ham1(i), rs1);
                float d[3];
                sample_dir<TEXIR_IRT_FAST_SINCOS != 0>(mode, s0, s1, 0.f, f, d);
                ndl = cosw ? 1.f : fminf(fmaxf(rnx * d[0] + rny * d[1] + rnz * d[2], 0.f), 1.f);       // :170, RAW normal
                dx = d[0]; dy = d[1]; dz = d[2];
                if (STATS) c_rays++;
                return true;
            };
            trace_stream<STATS, kGroupLstk, WIDTH, kCull>(sc, px, py, pz, cn, ct, STATS ? wi : nullptr, refill_at, next);
            float* o = partial + ((int64_t)part * n_ids + k) * 3;
            o[0] = acc0; o[1] = acc1; o[2] = acc2;
        }
    }
    if (STATS) irt_stats_flush(stats, lane, c_rays, cn, ct, c_hits, wi[0], wi[1]);
}

__global__ __launch_bounds__(128) void clear_u64_kernel(unsigned long long* __restrict__ p, int n)
{
    for (int i = threadIdx.x; i < n; i += 128) p[i] = 0ull;
}

// irr[t] = (2 pi / N) * (partial sums of the texel's pass ranges, added in part order)          (tracer_o3d_irt.py:171)
__global__ __launch_bounds__(256) void irt_combine_kernel(const float* __restrict__ partial, const int32_t* __restrict__ ids, int64_t n_ids,
                                                          int parts, int N, float two, float* __restrict__ irr)
{
    const float pi = 3.141592653589793f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 3 * n_ids; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e / 3;
        const int c = (int)(e - 3 * k);
        float a = partial[e];
        for (int p = 1; p < parts; p++) a += partial[(int64_t)p * n_ids * 3 + e];
        const int64_t t = ids ? (int64_t)ids[k] : k;
        irr[3 * t + c] = ((a * two) * pi) / (float)N;
    }
}

// Retiled copies of the radiance texture for the hit shader (device_common.h shade_hit): same floats, other addresses.
//   layout 1: 8x8-texel tiles, 12-byte texels (768 B per tile), tile-row major
//   layout 2: one 128-byte line per 3x3 block of texels at stride 2 (27 floats + 5 pad), so that every 2x2 bilinear footprint
//             lies inside ONE line; texels past the right/bottom edge repeat the edge texel (their bilinear weight is 0)
__global__ __launch_bounds__(256) void tex_retile_kernel(const float* __restrict__ src, float* __restrict__ dst, int Ht, int Wt, int layout, int tiles_x, int tiles_y)
{
    if (layout == 2) {
        const int64_t n = (int64_t)tiles_x * tiles_y * 32;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
            const int64_t tile = e >> 5; const int f = (int)(e & 31);
            float v = 0.f;
            if (f < 27) {
                const int ty = (int)(tile / tiles_x), tx = (int)(tile - (int64_t)ty * tiles_x);
                const int r = f / 9, c = (f - 9 * r) / 3, ch = f - 9 * r - 3 * c;
                const int y = min(2 * ty + r, Ht - 1), x = min(2 * tx + c, Wt - 1);
                v = src[((size_t)y * Wt + x) * 3 + ch];
            }
            dst[e] = v;
        }
    } else {
        const int64_t n = (int64_t)tiles_x * tiles_y * 192;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
            const int64_t tile = e / 192; const int f = (int)(e - tile * 192);
            const int ty = (int)(tile / tiles_x), tx = (int)(tile - (int64_t)ty * tiles_x);
            const int t = f / 3, ch = f - 3 * t;
            const int y = min(8 * ty + (t >> 3), Ht - 1), x = min(8 * tx + (t & 7), Wt - 1);
            dst[e] = src[((size_t)y * Wt + x) * 3 + ch];
        }
    }
}

// Layouts 3, 4: 4-byte shared-exponent texels (device_common.h pack_texel) in overlapping tiles of one 128-byte line each -- 5x5 texels at stride 4
// (25 of 32 words used) or 8x4 texels at stride 7x3 -- so that every 2x2 bilinear footprint lies inside ONE line.  `bad` counts the texels that are
// NOT three 8-bit integers times one power of two (the caller then keeps the float32 layout 2): the packed copy is only ever used when it decodes
// to the identical floats.
__global__ __launch_bounds__(256) void tex_pack_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, int Ht, int Wt, int layout, int tiles_x, int tiles_y,
                                                       unsigned int* __restrict__ bad)
{
    const int tw = layout == 3 ? 5 : 8, th = layout == 3 ? 5 : 4, sx = tw - 1, sy = th - 1;
    const int64_t n = (int64_t)tiles_x * tiles_y * 32;
    unsigned int nbad = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t tile = e >> 5; const int f = (int)(e & 31);
        uint32_t w = 0u;
        if (f < tw * th) {
            const int ty = (int)(tile / tiles_x), tx = (int)(tile - (int64_t)ty * tiles_x);
            const int r = f / tw, c = f - r * tw;
            const int y = min(sy * ty + r, Ht - 1), x = min(sx * tx + c, Wt - 1);
            const float* p = src + ((size_t)y * Wt + x) * 3;
            if (!pack_texel(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), w)) { nbad++; w = 0u; }
        }
        dst[e] = w;
    }
    if (nbad) atomicAdd(bad, nbad);
}

size_t tex_pack_bytes(int Ht, int Wt, int layout, int* tiles_x, int* tiles_y)
{
    if (layout == 3) { *tiles_x = (Wt - 1) / 4 + 1; *tiles_y = (Ht - 1) / 4 + 1; }
    else if (layout == 4) { *tiles_x = (Wt - 1) / 7 + 1; *tiles_y = (Ht - 1) / 3 + 1; }
    else { *tiles_x = *tiles_y = 0; return 0; }
    return (size_t)*tiles_x * *tiles_y * 128;
}

hipError_t launch_tex_pack(const float* src, uint32_t* dst, int Ht, int Wt, int layout, unsigned int* bad, hipStream_t st)
{
    int tx, ty;
    const size_t bytes = tex_pack_bytes(Ht, Wt, layout, &tx, &ty);
    if (!bytes) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tex_pack_kernel, dim3(grid_for_static(256, (int64_t)(bytes / 4))), dim3(256), 0, st, src, dst, Ht, Wt, layout, tx, ty, bad);
    return hipGetLastError();
}

// Streams `n16` 16-byte words through the memory hierarchy and keeps nothing: after a kernel that has flushed the caches (the 1.8 GB stream of the
// fused Adam), this brings the traversal data back into the memory-side Infinity Cache before the latency-bound specular trace starts.
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9E3779B9u && sink) *sink = acc;          // (practically never: keeps the loads alive)
}

hipError_t launch_prefetch(const void* p, size_t bytes, int blocks, uint32_t* sink, hipStream_t st)
{
    if (!p || bytes < 16) return hipSuccess;
    hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, st, (const uint4*)p, bytes / 16, sink);
    return hipGetLastError();
}

size_t tex_retile_bytes(int Ht, int Wt, int layout, int* tiles_x, int* tiles_y)
{
    if (layout == 2) { *tiles_x = (Wt + 1) / 2; *tiles_y = (Ht + 1) / 2; return (size_t)*tiles_x * *tiles_y * 128; }
    if (layout == 1) { *tiles_x = (Wt + 7) / 8; *tiles_y = (Ht + 7) / 8; return (size_t)*tiles_x * *tiles_y * 768; }
    *tiles_x = *tiles_y = 0;
    return 0;
}

hipError_t launch_tex_retile(const float* src, float* dst, int Ht, int Wt, int layout, hipStream_t st)
{
    int tx, ty;
    const size_t bytes = tex_retile_bytes(Ht, Wt, layout, &tx, &ty);
    if (!bytes) return hipSuccess;
    hipLaunchKernelGGL(tex_retile_kernel, dim3(grid_for_static(256, (int64_t)(bytes / 4))), dim3(256), 0, st, src, dst, Ht, Wt, layout, tx, ty);
    return hipGetLastError();
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
        Hit h = trace_closest<false, kLstk, WIDTH, kCull>(sc, ox, oy, oz, dx, dy, dz, cn, ct);
        float L[3] = {0.f, 0.f, 0.f};
        bool hit = h.slot >= 0 && h.t > t_min;
        if (hit) shade_hit(sc, h.slot, h.u, h.v, L);
        rad[3 * r] = L[0]; rad[3 * r + 1] = L[1]; rad[3 * r + 2] = L[2];
        if (t_hit) t_hit[r] = h.t;
        if (prim) prim[r] = h.slot >= 0 ? tri_prim(sc, h.slot) : 0xFFFFFFFFu;
        if (puv) {
            // the caller's barycentrics (weights of ITS corners 1 and 2): the slot stores its corners rotated (bvh_build.h), stored corner k = corner (rot + k) % 3
            float u = h.u, v = h.v;
            if (h.slot >= 0) {
                const uint32_t rot = __float_as_uint(sc.tris[3 * (size_t)h.slot + 1].w);
                const float w = 1.f - u - v;
                if (rot == 1u) { const float u1 = w, v1 = u; u = u1; v = v1; }           // stored (c1, c2, c0): weights (w, u, v) of (c1, c2, c0)
                else if (rot == 2u) { const float u1 = v, v1 = w; u = u1; v = v1; }      // stored (c2, c0, c1)
            }
            puv[2 * r] = u; puv[2 * r + 1] = v;
        }
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
                                                  float r, float s0, float s1, float ceps)
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
    Dual g1v = ddiv(dconst(ndv), dclamp_min(dadd(dscale(omk, ndv), kk), ceps));
    Dual g1l = ddiv(ndl, dclamp_min(dadd(dmul(ndl, omk), kk), ceps));
    Dual g = dmul(g1l, g1v);
    Dual brdf = ddiv(dmul(fr, g), dclamp_min(dscale(ndl, 4.f * ndv), ceps));
    Dual w = ddiv(dmul(dscale(dmul(brdf, ndl), 4.f), vdh), dclamp_min(ndh, ceps));
    SpecSample o;
    o.w = w;
    for (int k = 0; k < 3; k++) o.l[k] = l[k].v;
    return o;
}
#pragma clang fp contract(fast)

// lanes-per-pixel = S when S is a power of two <= 64 (several pixels per wave), else 64 with ceil(S/64) passes
#ifndef TEXIR_SPEC_WAVES
#define TEXIR_SPEC_WAVES 0
#endif
#ifndef TEXIR_SPEC_LSTK
#define TEXIR_SPEC_LSTK (TEXIR_SPEC_WAVES >= 7 && TEXIR_CULL ? (TEXIR_SPEC_WAVES >= 8 ? 10 : 11) : kLstk)
#endif
constexpr int kSpecLstk = TEXIR_SPEC_LSTK;
// DW (forward only): the sample weights' derivatives d w_i / d roughness -- which the dual-number chain yields next to the weights at no extra fetch, in a
// kernel that waits on memory with two thirds of its issue slots free -- are written to dw_ws [P,S]; the backward is then spec_bwd_ws_kernel, a stream over
// (Ls, dw, d rgb), instead of this kernel's BWD form recomputing the whole sample chain (38 us of the material step: round 4).
template <bool BWD, int WIDTH, bool DW = false>
__global__
#if TEXIR_SPEC_WAVES
__launch_bounds__(kBlock, TEXIR_SPEC_WAVES)
#else
__launch_bounds__(kBlock)
#endif
void spec_kernel(SceneDev sc, const float* __restrict__ normal, const float* __restrict__ albedo,
                                                      const float* __restrict__ rough, const float* __restrict__ points,
                                                      const float* __restrict__ irr, const float* __restrict__ cam,
                                                      const float* __restrict__ shift, int64_t P, int S, int lpp,
                                                      float* __restrict__ rgb, float* __restrict__ Ls_ws,
                                                      const float* __restrict__ d_rgb, float* __restrict__ d_albedo, float* __restrict__ d_rough, float ceps,
                                                      int ls_given, float* __restrict__ dw_ws = nullptr)
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
                SpecSample ss = spec_sample(f, nx, ny, nz, vx, vy, vz, r, s0, s1, ceps);
                if constexpr (DW) dw_ws[(size_t)p * S + i] = ss.w.d;          // (before the trace: the derivative does not stay live across the traversal)
                float L[3] = {0.f, 0.f, 0.f};
                if (BWD) {
                    const float* lp = Ls_ws + 3 * ((size_t)p * S + i);
                    L[0] = lp[0]; L[1] = lp[1]; L[2] = lp[2];
                    dacc += (L[0] * g0 + L[1] * g1 + L[2] * g2) * ss.w.d;
                } else {
                    if (ls_given) {
                        // specular_reflectance on the caller's lighting (mat_nvdiffrast.py:260-279 as a function seam): no tracing
                        const float* lp = Ls_ws + 3 * ((size_t)p * S + i);
                        L[0] = lp[0]; L[1] = lp[1]; L[2] = lp[2];
                    } else {
                        Hit h = trace_closest<false, kSpecLstk, WIDTH, kCull>(sc, ox, oy, oz, ss.l[0], ss.l[1], ss.l[2], cn, ct);
                        if (h.slot >= 0 && h.t > 1e-4f) shade_hit(sc, h.slot, h.u, h.v, L);
                        if (Ls_ws) { float* lp = Ls_ws + 3 * ((size_t)p * S + i); lp[0] = L[0]; lp[1] = L[1]; lp[2] = L[2]; }
                    }
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

// Backward of the specular term on what the forward kept: d roughness[p] = (1/S) sum_i (Ls_i . d rgb[p]) * dw_i, d albedo[p] = d rgb[p] * irr[p] / pi.  Same lane
// assignment, same expression and same reduction order as spec_kernel<BWD>.
__global__ __launch_bounds__(kBlock) void spec_bwd_ws_kernel(const float* __restrict__ irr, const float* __restrict__ Ls_ws, const float* __restrict__ dw_ws,
                                                            const float* __restrict__ d_rgb, int64_t P, int S, int lpp,
                                                            float* __restrict__ d_albedo, float* __restrict__ d_rough)
{
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / lpp;
    const int sub = lane / lpp, sl = lane % lpp;
    const int64_t gw = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * kBlock) >> 6;
    const int passes = (S + lpp - 1) / lpp;
    for (int64_t base = gw * ppw; base < P; base += nw * ppw) {
        const int64_t p = base + sub;
        const bool live = p < P;
        float dacc = 0.f, g0 = 0, g1 = 0, g2 = 0;
        if (live) { g0 = d_rgb[3 * p]; g1 = d_rgb[3 * p + 1]; g2 = d_rgb[3 * p + 2]; }
        for (int q = 0; q < passes; q++) {
            const int i = q * lpp + sl;
            if (live && i < S) {
                const float* lp = Ls_ws + 3 * ((size_t)p * S + i);
                dacc += (lp[0] * g0 + lp[1] * g1 + lp[2] * g2) * dw_ws[(size_t)p * S + i];
            }
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o, 64);
        if (live && sl == 0) {
            const float pi = 3.141592653589793f;
            if (d_rough) d_rough[p] = dacc / (float)S;
            if (d_albedo) { d_albedo[3 * p] = g0 * irr[3 * p] / pi; d_albedo[3 * p + 1] = g1 * irr[3 * p + 1] / pi; d_albedo[3 * p + 2] = g2 * irr[3 * p + 2] / pi; }
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

// TEXIR_IRT_TEXELS_PER_WAVE = 1 | 64 forces the kernel form (A/B measurements, parity tests of each form); unset or 0 = automatic (env.h).
static int irt_forced_texels_per_wave() { return env().irt_texels_per_wave; }

// Which kernel form a call launches (shared by launch_irt and texir_irt_kernel_name, so that a bench line names what really ran):
// texels per wave: 64 from 32 768 listed texels up, else 1 (a short list does not fill the chip with 64-texel groups; measured on the
// c2 scene, Grays/s for 1 / 64 per wave: 16 k texels 10.7 / 10.1, 65 k 11.3 / 14.3, 131 k 11.5 / 14.8, 524 k 11.3 / 16.0; a 16-texel
// form was slower than both at every length and is gone).
// 64 texels per wave: the passes of a texel are cut into 2^log2parts ranges of >= 64 passes (N = 2048: 32 ranges), each range its
// own chunk.  Fine chunks matter for lists up to ~1 M texels (262 k texels: 8 ranges 14.1, 32 ranges 15.7 Grays/s), i.e. for every
// rank's share of a multi-GPU run; the number of ranges depends on N alone, so results do not depend on the sharding.
IrtPlan irt_plan(const SceneDev& sc, int64_t n_ids, int N)
{
    IrtPlan p{};
    const int forced = irt_forced_texels_per_wave();
    const bool pow2 = (N & (N - 1)) == 0;
    p.width = sc.nodes4 ? 4 : 2;
    p.per_wave = !sc.nodes4 ? 1 : (forced ? forced : (n_ids >= 32768 ? 64 : 1));
    p.log2parts = 0;
    // parts per texel: up to 32, down to 8 passes per part (N = 2048: 32 parts of 64 passes; N = 64 -- the reference's own configuration -- 8 parts of 8:
    // with 64-pass parts its 3 053 chunks left two thirds of the 8 192 resident waves without work).  A function of N alone.
    const int min_cells = env().irt_min_part_cells;                                                                      // A/B switch (default 8; round 2: 64)
    if (sc.nodes4 && p.per_wave == 64 && pow2) { while (p.log2parts < 5 && (N >> (p.log2parts + 1)) >= min_cells) p.log2parts++; }
    if (const int cap = env().irt_log2parts_cap; cap >= 0 && p.log2parts > cap) p.log2parts = cap;                        // A/B switch
    const bool stream = p.per_wave == 64 && env().irt_refill && pow2 && p.log2parts > 0;
    if (stream) snprintf(p.name, sizeof(p.name), "irt_stream_kernel<false, %d>", p.width);
    else snprintf(p.name, sizeof(p.name), p.per_wave == 64 ? "irt_group_kernel<false, %d, 6>" : "irt_kernel<false, %d>", p.width);
    return p;
}

template <typename K>
static void irt_launch(K kernel, int64_t waves_wanted, const SceneDev& sc, const float* pos, const float* nrm, const float* shift,
                       const int32_t* ids, int64_t n_ids, int N, int l2, int mode, float* irr, unsigned long long* stats,
                       unsigned long long* work, float* partial, int log2parts, hipStream_t st)
{
    // persistent grid: exactly the workgroups that are co-resident, each wave strides over the texel list
    const int64_t want = (waves_wanted + (kBlock / 64) - 1) / (kBlock / 64);
    int grid = resident_grid(kernel, kBlock);
    if (want < grid) grid = (int)want;
    if (const int cap = env().irt_grid_cap; cap > 0 && grid > cap) grid = cap;            // occupancy sweeps only (any grid gives the same result: the waves pull chunks)
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, irr, stats, work, partial, log2parts);
}

size_t irt_scratch_bytes(const SceneDev& sc, int64_t n_ids, int N)
{
    if (n_ids <= 0) return 0;
    const IrtPlan plan = irt_plan(sc, n_ids, N);
    return plan.log2parts ? (sizeof(float) * 3 * (size_t)n_ids << plan.log2parts) : 0;
}

hipError_t launch_irt(const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t n_ids,
                      int N, int mode, float* irr, unsigned long long* stats, unsigned long long* work, hipStream_t st, float* scratch)
{
    if (n_ids <= 0) return hipSuccess;
    // the chunk counters of this launch (one per XCD), cleared by a KERNEL: recorded into a hipGraph, a hipMemsetAsync node was observed to stop taking
    // effect from the second replay on (tools/graph_memset_probe.py; every later replay of texir_irt_generate then found its queues exhausted)
    hipLaunchKernelGGL(clear_u64_kernel, dim3(1), dim3(128), 0, st, work, 8 * kWorkStride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const bool pow2 = (N & (N - 1)) == 0;
    const int l2 = ilog2_exact(N);
    const IrtPlan plan = irt_plan(sc, n_ids, N);
#if TEXIR_CHAIN_PROBE
    const bool probe_group = stats && sc.nodes4 && plan.per_wave == 64 && !(env().irt_refill && pow2 && plan.log2parts > 0);
#else
    constexpr bool probe_group = false;
#endif
#define TEXIR_IRT(WAVES, L2, NAME, ...) { if (stats && !probe_group) irt_launch(NAME<true, __VA_ARGS__>, WAVES, sc, pos, nrm, shift, ids, n_ids, N, L2, mode, irr, stats, work, partial, log2parts, st); \
                                        else irt_launch(NAME<false, __VA_ARGS__>, WAVES, sc, pos, nrm, shift, ids, n_ids, N, L2, mode, irr, stats, work, partial, log2parts, st); }
    const int per_wave = plan.per_wave;
    float* partial = nullptr;
    const int log2parts = plan.log2parts;
    if (log2parts) {
        const size_t bytes = sizeof(float) * 3 * (size_t)n_ids << log2parts;
        // stream-ordered scratch (384 B per listed texel at N >= 2048: 4.8 GB at 4k^2 texels); keep it cached in the device's pool between calls instead of
        // returning it to the OS at every synchronisation
        static bool pool_set = false;
        if (!pool_set) {
            int dev = 0; hipMemPool_t pool;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                uint64_t keep = ~0ull;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            }
            pool_set = true;
        }
        if (scratch) partial = scratch;                  // (caller-owned: the form a recorded hipGraph uses -- texir_scene_reserve_scratch)
        else if ((e = hipMallocAsync((void**)&partial, bytes, st)) != hipSuccess) return e;
    }
    if (!sc.nodes4) TEXIR_IRT(n_ids, l2, irt_kernel, 2)                                        // deep binary tree (capi.hip fallback)
    else if (per_wave == 1) TEXIR_IRT(n_ids, l2, irt_kernel, 4)
    else if (env().irt_refill && pow2 && log2parts > 0) {
        // compaction by refill (A/B switch TEXIR_IRT_REFILL = lanes that must be idle before a refill)
        const int64_t waves = ((n_ids + 63) / 64) << log2parts;
        const int64_t want = (waves + (kBlock / 64) - 1) / (kBlock / 64);
        if (stats) {
            int grid = resident_grid(irt_stream_kernel<true, 4>, kBlock); if (want < grid) grid = (int)want;
            hipLaunchKernelGGL((irt_stream_kernel<true, 4>), dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, irr, stats, work, partial, log2parts, env().irt_refill);
        } else {
            int grid = resident_grid(irt_stream_kernel<false, 4>, kBlock); if (want < grid) grid = (int)want;
            hipLaunchKernelGGL((irt_stream_kernel<false, 4>), dim3(grid), dim3(kBlock), 0, st, sc, pos, nrm, shift, ids, n_ids, N, l2, mode, irr, stats, work, partial, log2parts, env().irt_refill);
        }
    }
    else TEXIR_IRT(((n_ids + 63) / 64) << log2parts, pow2 ? l2 : -1, irt_group_kernel, 4, 6)   // any N (natural sample order if not 2^k)
#undef TEXIR_IRT
    if (log2parts) {
        hipLaunchKernelGGL(irt_combine_kernel, dim3(grid_for(256, 3 * n_ids)), dim3(256), 0, st, partial, ids, n_ids, 1 << log2parts, N, (mode & kEstimatorCosine) ? 1.f : 2.f, irr);
        if (!scratch && (e = hipFreeAsync(partial, st)) != hipSuccess) return e;
    }
    return hipGetLastError();
}

// How full do this scene's node steps run?  A counting launch of the 64-texel form over `count` listed texels starting at `first` (min(N, 256) samples,
// weight 2, partial sums into scratch: the caller's irradiance buffer is not touched) -> lanes taking part per wave-level node step / 64.
// Blocks until the counters are back (once per scene: texir_scene_tune -- never from a *_generate / *_forward entry point).
hipError_t irt_probe_node_utilisation(const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t first, int64_t count,
                                      int N, int mode, unsigned long long* work, hipStream_t st, double* util)
{
    *util = -1.0;
    if (!sc.nodes4 || !ids || count < 64 || (N & (N - 1))) return hipSuccess;
    const int Nc = N > 256 ? 256 : N;
    const int l2 = ilog2_exact(Nc);
    int log2parts = 0;
    while (log2parts < 5 && (Nc >> (log2parts + 1)) >= 8) log2parts++;
    if (!log2parts) return hipSuccess;
    hipError_t e;
    float* partial = nullptr; unsigned long long* stats = nullptr;
    hipLaunchKernelGGL(clear_u64_kernel, dim3(1), dim3(128), 0, st, work, 8 * kWorkStride);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = hipMallocAsync((void**)&partial, (sizeof(float) * 3 * (size_t)count << log2parts) + 8 * sizeof(unsigned long long), st)) != hipSuccess) return e;
    stats = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(partial) + (sizeof(float) * 3 * (size_t)count << log2parts));
    auto drop = [&](hipError_t err) { (void)hipFreeAsync(partial, st); return err; };           // (no error path keeps the scratch)
    hipLaunchKernelGGL(clear_u64_kernel, dim3(1), dim3(128), 0, st, stats, 8);
    if ((e = hipGetLastError()) != hipSuccess) return drop(e);
    SceneDev probe = sc;
    probe.sched_weight = kSchedNodeWeight;
    irt_launch(irt_group_kernel<true, 4, 6>, ((count + 63) / 64) << log2parts, probe, pos, nrm, shift, ids + first, count, Nc, l2, mode, (float*)nullptr, stats, work, partial, log2parts, st);
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((e = hipGetLastError()) != hipSuccess) return drop(e);
    if ((e = hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, st)) != hipSuccess) return drop(e);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return drop(e);
    if ((e = hipFreeAsync(partial, st)) != hipSuccess) return e;
    if (h[4]) *util = (double)h[1] / (64.0 * (double)h[4]);          // node fetches / (64 x wave-level node steps)
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

// lanes per pixel: S when S is a power of two <= 64 (one pass), else 64.  TEXIR_SPEC_LPP = 1..64 (a power of two) forces fewer lanes and
// more passes per pixel: the lanes of a wave then belong to more, neighbouring pixels and take the SAME sample indices in a pass
// (more coherent rays, fewer and longer waves) -- A/B switch.
// the specular kernels' grid: one block per 4 waves of pixels up to this many blocks, grid-stride beyond (TEXIR_SPEC_GRID_CAP: A/B switch).
// A pixel group is ~15 node steps of work with a long tail, so letting the hardware hand out many short blocks balances better than a
// static stride over a few long ones.
static int spec_grid(int64_t pix_per_block, int64_t P)
{
    const int64_t cap = env().spec_grid_cap;
    const int64_t want = (P + pix_per_block - 1) / pix_per_block;
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

static int lanes_per_pixel(int S)
{
    int lpp = (S <= 64 && (S & (S - 1)) == 0) ? S : 64;
    if (const int v = env().spec_lpp; v >= 1 && v <= lpp) lpp = v;
    return lpp;
}

hipError_t launch_spec_fwd(const SceneDev& sc, const float* normal, const float* albedo, const float* rough, const float* points,
                           const float* irr, const float* cam, const float* shift, int64_t P, int S, float clamp_eps, int ls_given, float* rgb, float* Ls_ws,
                           hipStream_t st, float* dw_ws)
{
    if (P <= 0) return hipSuccess;
    int lpp = lanes_per_pixel(S);
    int64_t pix_per_block = (int64_t)(kBlock / 64) * (64 / lpp);
    const dim3 grid(spec_grid(pix_per_block, P));
#define TEXIR_SPEC_FWD(W, DW) hipLaunchKernelGGL((spec_kernel<false, W, DW>), grid, dim3(kBlock), 0, st, sc, normal, albedo, rough, points, irr, cam, shift, P, S, lpp, rgb, Ls_ws, \
                                                 (const float*)nullptr, (float*)nullptr, (float*)nullptr, clamp_eps, ls_given, dw_ws)
    if (sc.nodes4 || ls_given) { if (dw_ws) TEXIR_SPEC_FWD(4, true); else TEXIR_SPEC_FWD(4, false); }
    else { if (dw_ws) TEXIR_SPEC_FWD(2, true); else TEXIR_SPEC_FWD(2, false); }
#undef TEXIR_SPEC_FWD
    return hipGetLastError();
}

hipError_t launch_spec_bwd_ws(const float* irr, const float* Ls_ws, const float* dw_ws, const float* d_rgb, int64_t P, int S, float* d_albedo, float* d_rough,
                              hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    int lpp = lanes_per_pixel(S);
    int64_t pix_per_block = (int64_t)(kBlock / 64) * (64 / lpp);
    hipLaunchKernelGGL(spec_bwd_ws_kernel, dim3(spec_grid(pix_per_block, P)), dim3(kBlock), 0, st, irr, Ls_ws, dw_ws, d_rgb, P, S, lpp, d_albedo, d_rough);
    return hipGetLastError();
}

hipError_t launch_spec_bwd(const float* normal, const float* rough, const float* points, const float* irr, const float* cam,
                           const float* shift, const float* Ls_ws, const float* d_rgb, int64_t P, int S, float clamp_eps, float* d_albedo, float* d_rough,
                           hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    int lpp = lanes_per_pixel(S);
    int64_t pix_per_block = (int64_t)(kBlock / 64) * (64 / lpp);
    SceneDev none{};
    hipLaunchKernelGGL((spec_kernel<true, 2>), dim3(spec_grid(pix_per_block, P)), dim3(kBlock), 0, st, none, normal, (const float*)nullptr, rough,
                       points, irr, cam, shift, P, S, lpp, (float*)nullptr, const_cast<float*>(Ls_ws), d_rgb, d_albedo, d_rough, clamp_eps, 0);
    return hipGetLastError();
}

}  // namespace texir
