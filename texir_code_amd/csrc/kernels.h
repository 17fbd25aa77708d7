// launchers of the gfx950 kernels (kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"

namespace texir {
hipError_t launch_irt(const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t n_ids,
                      int N, int mode, float* irr, unsigned long long* stats, unsigned long long* work /*dev: 8 * kWorkStride chunk counters of this launch*/,
                      hipStream_t st, float* scratch = nullptr /*dev: caller-owned partial-sum scratch of >= irt_scratch_bytes(), else stream-ordered*/);
size_t irt_scratch_bytes(const SceneDev& sc, int64_t n_ids, int N);      // bytes of partial-sum scratch one launch_irt call over n_ids listed texels needs (0: none)
constexpr int kWorkStride = 16;              // unsigned long longs between the per-XCD chunk counters of one launch (a 128-byte line each)
hipError_t irt_probe_node_utilisation(const SceneDev& sc, const float* pos, const float* nrm, const float* shift, const int32_t* ids, int64_t first, int64_t count,
                                      int N, int mode, unsigned long long* work, hipStream_t st, double* util);
struct IrtPlan { int per_wave, log2parts, width; char name[64]; };
IrtPlan irt_plan(const SceneDev& sc, int64_t n_ids, int N);      // the kernel form launch_irt picks for this call
hipError_t launch_prefetch(const void* p, size_t bytes, int blocks, uint32_t* sink, hipStream_t st);
size_t tex_retile_bytes(int Ht, int Wt, int layout, int* tiles_x, int* tiles_y);
hipError_t launch_tex_retile(const float* src_row_major, float* dst, int Ht, int Wt, int layout, hipStream_t st);
size_t tex_pack_bytes(int Ht, int Wt, int layout, int* tiles_x, int* tiles_y);          // layouts 3, 4 (4-byte shared-exponent texels)
hipError_t launch_tex_pack(const float* src_row_major, uint32_t* dst, int Ht, int Wt, int layout, unsigned int* bad /*dev: texels that do not pack exactly*/, hipStream_t st);
hipError_t launch_trace_shade(const SceneDev& sc, const float* org, const float* dir, int64_t R, float t_min, float* rad, float* t_hit,
                              uint32_t* prim, float* puv, hipStream_t st);
hipError_t launch_gen_dir(const float* normals, const float* rough, const float* shift, int64_t b, int N, int mode, float* L, hipStream_t st);
hipError_t launch_spec_fwd(const SceneDev& sc, const float* normal, const float* albedo, const float* rough, const float* points,
                           const float* irr, const float* cam, const float* shift, int64_t P, int S, float clamp_eps, int ls_given, float* rgb, float* Ls_ws,
                           hipStream_t st, float* dw_ws = nullptr /*dev [P,S], nullable: d w_i / d roughness for launch_spec_bwd_ws*/);
hipError_t launch_spec_bwd_ws(const float* irr, const float* Ls_ws, const float* dw_ws, const float* d_rgb, int64_t P, int S, float* d_albedo, float* d_rough,
                              hipStream_t st);
hipError_t launch_spec_bwd(const float* normal, const float* rough, const float* points, const float* irr, const float* cam,
                           const float* shift, const float* Ls_ws, const float* d_rgb, int64_t P, int S, float clamp_eps, float* d_albedo, float* d_rough,
                           hipStream_t st);
size_t loss_workspace_bytes(int64_t P, int C, int R);
hipError_t launch_loss(int stage, int l2, const float* gt, const float* rgb, const float* albedo, const float* rough, const float* rough_womip,
                       const float* empty, const float* gtm, const uint8_t* seg, const uint8_t* hl, const uint8_t* room, int64_t P, int C, int R,
                       int hw, void* workspace, float* out, float* d_rgb, float* d_albedo, float* d_rough, hipStream_t st);
hipError_t launch_gbuffer(const SceneDev& sc, const float* mvp_host, const float4* cnrm, int c, int flip_v, float* pos, float* nrm, float* mask,
                          float* uv, float* uvda, int32_t* tri, hipStream_t st);
int mip_levels(int H, int W, int max_mip_level);
int64_t mip_total_elems(int H, int W, int C, int levels);
hipError_t launch_mip_build(const float* tex, float* rest, int H, int W, int C, int levels, int from_level, hipStream_t st);
hipError_t launch_tex_fetch(const float* tex, const float* rest, int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear,
                            int64_t P, float* out, hipStream_t st);
hipError_t launch_tex_fetch_bwd(float* d_tex, float* grad_rest, int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear,
                                int64_t P, const float* d_out, int fold_to_level, hipStream_t st);
hipError_t launch_tex_taps(int H, int W, int C, int levels, const float* uv, const float* uvda, int trilinear, int64_t P, long long* keys,
                           float* weights, hipStream_t st);
hipError_t launch_tex_gather_bwd(float* d_tex, float* grad_rest, int H, int W, int C, int levels, const long long* seg_key, const int* seg_start,
                                 const int* seg_count, int n_seg, const int* pix, const float* w, const float* d_out, int trilinear,
                                 int fold_to_level, hipStream_t st);
hipError_t launch_adam_tex(float* p, const float* g /*nullable*/, const uint32_t* l0_mask /*nullable*/, const float* g1, const float* g2 /*nullable*/, float* m, float* v, float* mip1 /*nullable*/,
                           int H, int W, int C, float lr,
                           float beta1, float beta2, float eps, int step, float lo, float hi, const float* hyper /*dev, nullable: step size + bias correction*/, hipStream_t st);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                       float lo, float hi, const float* hyper /*dev, nullable*/, hipStream_t st);
hipError_t launch_adam_tick(double* state, float* hyper, int n, unsigned long long mask, hipStream_t st);
}  // namespace texir
