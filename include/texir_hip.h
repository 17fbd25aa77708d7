/*
 * texir_hip.h -- C-ABI of libtexir_hip.so: the MI355X (gfx950) implementation of TexIR's
 * irradiance-texture + material-estimation hot path.
 *
 * The reference (LZleejean/TexIR_code) is pure Python and has no FFI of its own; its seams
 * on this path are Python-level.  Each entry point below replaces one of those
 * seams and cites it.  A reference maintainer binds them with ctypes (INTEGRATION.md shows the
 * stubs).  Conventions:
 *   - every function returns 0 on success, <0 on error; texir_last_error() gives the
 *     thread-local message.  Nothing falls back to a CPU path.
 *   - `const float* x /+dev+/` pointers are caller-owned, contiguous DEVICE pointers
 *     (tensor.data_ptr()); host pointers are marked /+host+/.
 *   - launches are asynchronous on the hipStream_t passed as `stream` (void*; 0 = null stream).
 *   - a texir_scene is immutable after creation (except texir_scene_set_texture) and may be
 *     shared by host threads; one handle per device.
 */
#ifndef TEXIR_HIP_H
#define TEXIR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEXIR_OK 0
#define TEXIR_ERR_INVALID (-1)
#define TEXIR_ERR_HIP (-2)
#define TEXIR_ERR_NOMEM (-3)

/* sampling modes of utils/sample_util.py:115-143 */
#define TEXIR_MODE_UNIFORM 0
#define TEXIR_MODE_COSINE 1
#define TEXIR_MODE_IMPORTANCE 2

#if defined(__GNUC__)
#define TEXIR_API __attribute__((visibility("default")))
#else
#define TEXIR_API
#endif

typedef struct texir_scene texir_scene;

TEXIR_API const char* texir_last_error(void);
TEXIR_API int texir_version(void);
/* The library's run-time switches (TEXIR_* environment variables, csrc/env.h) are parsed once when the library is loaded; this re-reads them
 * (for test suites that flip a switch between two launches; not to be called while launches are being issued from other threads). */
TEXIR_API int texir_reload_env(void);
/* the value the library's snapshot holds for one switch, by variable name ("TEXIR_MIP_PER_LEVEL" ...): host code that must agree with the library
 * on a switch asks the library instead of parsing the environment a second time with its own rule. */
TEXIR_API int texir_env_switch(const char* name, int32_t* value);

/* Replaces TracerO3d.__init__ scene part (models/tracer_o3d_irt.py:75-89) and MaterialModel.__init__
 * (models/mat_nvdiffrast.py:87-101): o3d.t.geometry.RaycastingScene().add_triangles(mesh) + the CPU-resident
 * radiance texture.  Builds the BVH on the host and uploads it.
 *   verts   [V,3]  f32 host      tris [T,3] i32 host (primitive id = row index, as in Open3D)
 *   tri_uvs [3T,2] f32 host      per-corner uvs = np.asarray(trianglemesh.triangle_uvs)
 *   hdr_tex [Ht,Wt,3] f32 host   ALREADY BGR->RGB, vertically flipped and scaled by 2^hdr_exposure exactly as
 *                                tracer_o3d_irt.py:77-81 does. */
TEXIR_API int texir_scene_create(const float* verts /*host*/, int32_t V, const int32_t* tris /*host*/, int32_t T,
                       const float* tri_uvs /*host*/, const float* hdr_tex /*host*/, int32_t Ht, int32_t Wt,
                       int32_t device, texir_scene** out);
TEXIR_API int texir_scene_destroy(texir_scene* scene);

/* Replaces the temporary `self.texture = torch.where(intensity>=0.5, ...)` swap of stage -1
 * (models/mat_nvdiffrast.py:141-150).  tex [Ht,Wt,3] f32; is_device selects pointer kind. */
TEXIR_API int texir_scene_set_texture(texir_scene* scene, const float* tex, int32_t Ht, int32_t Wt, int32_t is_device, void* stream);
/* The hit shader's copy of the radiance texture (`self.texture`, models/tracer_o3d_irt.py:77-81: an RGBE file times 2^hdr_exposure).  When EVERY
 * texel is three 8-bit integers times one power of two -- always true of such a texture -- it is kept as 4-byte shared-exponent texels that decode
 * to the identical float32 values (layout 3: 5x5-texel lines at stride 4; 4: 8x4 at stride 7x3); any other texture keeps float32 tiles (2; 1 / 0 are the
 * older A/B layouts).  Decided at texir_scene_create and again at every texir_scene_set_texture (which then synchronises `stream` once to read the
 * pack kernel's verdict; on a capturing stream the float32 layout is taken).  Results are bit-identical in every layout. */
TEXIR_API int texir_scene_texture_layout(const texir_scene* scene, int32_t* layout);
/* Host statement of the 4-byte texel: word = m_r | m_g << 8 | m_b << 16 | E << 24, value_c = m_c * 2^(E - 127).  rgb /+host+/ [n,3] f32 ->
 * words /+host+/ [n], exact /+host+/ [n] (nullable; 1 where the triple has this form: non-negative, finite, normal, one shared power of two with
 * 8-bit integers -- e.g. any cv2-decoded RGBE pixel times 2^k; 0 -> the word is 0 and the device keeps float32 texels).  unpack is the decode. */
TEXIR_API int texir_texel_pack(const float* rgb, int64_t n, uint32_t* words, uint8_t* exact);
TEXIR_API int texir_texel_unpack(const uint32_t* words, int64_t n, float* rgb);

/* out[0]=inner nodes of the traversal tree (4-wide quantised by default), [1]=triangles, [2]=max depth, [3]=node bytes,
 * [4]=triangle bytes (leaf-order slots + the quad records the 4-wide leaves name), [5]=uv bytes, [6]=texture bytes, [7]=device */
TEXIR_API int texir_scene_info(const texir_scene* scene, int64_t out[8]);
/* The traversal's phase scheduler weighs the lanes at inner nodes against the lanes at leaves (csrc/device_common.h); the weight is a property of the
 * scene.  texir_scene_tune decides it ONCE per scene from the measured fullness of the scene's node steps on a sample of the caller's own texel
 * list (a counting launch over 16 384 listed texels, ~2 ms, BLOCKING: it synchronises `stream` and must not be called while the stream is being
 * captured); lists shorter than 65 536 texels or N < 256 decide nothing.  No other entry point measures or synchronises: texir_irt_generate,
 * texir_spec_forward, ... launch with the weight in force (2 until tuned) and can be recorded into a hipGraph from their first call on.  The
 * Python host layer calls it before the first long irt_generate of a scene (scene.Scene.irt_generate); the reference has no counterpart (Embree
 * picks its traversal internally, models/tracer_o3d_irt.py:243-244).  Speed only: results never depend on the weight.
 * texir_scene_scheduler: out[0] = the weight in use (0 = not decided -> 2), out[1] = measured fullness (-1 = not measured). */
TEXIR_API int texir_scene_tune(const texir_scene* scene, const float* pos /*dev*/, const float* nrm /*dev*/, const float* shift /*dev*/,
                               const int32_t* texel_ids /*dev*/, int64_t n_ids, int32_t N, int32_t mode, void* stream);
TEXIR_API int texir_scene_scheduler(const texir_scene* scene, double out[2]);
/* Streams the scene's traversal data through the memory hierarchy (what: bit 0 quantised nodes, 1 float nodes, 2 triangles, 3 corner uvs;
 * blocks = grid size, 0 -> 512): a cache warm-up to launch beside / before a latency-bound tracing kernel that follows a cache-flushing
 * stream (the material step's fused Adam).  No reference counterpart (a speed hint: results never depend on it). */
TEXIR_API int texir_scene_prefetch(const texir_scene* scene, int32_t what, int32_t blocks, void* stream);

/* Replaces query_irf (models/tracer_o3d_irt.py:240-269, models/mat_nvdiffrast.py:292-320):
 * closest hit (Embree semantics: t>0, t in units of |dir|), hit mask t>t_min (reference: 1e-4) & finite,
 * barycentric clip, corner-uv interpolation, bilinear/border/align_corners=False fetch of the radiance
 * texture, misses -> 0.   org,dir [R,3] dev -> radiance [R,3] dev.
 * Optional raw intersection outputs (the cast_rays dict, tracer_o3d_irt.py:245-251): t_hit [R] (inf on miss),
 * prim_id [R] (0xFFFFFFFF on miss), prim_uv [R,2] (weights of the CALLER's corners 1 and 2 of triangle prim_id, as Open3D's primitive_uvs: the library
 * stores a triangle's corners rotated where its leaf record wants the shared edge, csrc/bvh_build.h, and turns the barycentrics back here); pass NULL to skip. */
TEXIR_API int texir_trace_shade(const texir_scene* scene, const float* org /*dev*/, const float* dir /*dev*/, int64_t R,
                      float t_min, float* radiance /*dev*/, float* t_hit /*dev, nullable*/,
                      uint32_t* prim_id /*dev, nullable*/, float* prim_uv /*dev, nullable*/, void* stream);

/* Replaces generate_dir (utils/sample_util.py:63-146) with pre_mode='Hammersley'.  The per-point random
 * shift (torch.rand(b,1,2) on the CPU generator, :102) is an INPUT so that parity is exact.
 *   normals [b,3] dev, roughness [b] dev (importance only, else NULL), shift [b,2] dev -> L [b,N,3] dev */
TEXIR_API int texir_generate_dir(const float* normals /*dev*/, const float* roughness /*dev, nullable*/,
                       const float* shift /*dev*/, int64_t b, int32_t N, int32_t mode, float* L /*dev*/, void* stream);

/* Replaces the hot loop of TracerO3d.forward (models/tracer_o3d_irt.py:156-178): for every listed texel
 *   E = (2*pi/N) * sum_i L(pos, d_i) * clamp(nrm . d_i, 0, 1),  d_i = generate_dir(nrm, N, mode)[i]
 * fused in one kernel (sample + trace + shade + reduce).
 *   pos,nrm [Nt,3] dev (pos already offset by +1e-2*n, :110), shift [Nt,2] dev
 *   texel_ids [n_ids] i32 dev: the texels to compute (NULL => all Nt, n_ids ignored).  Seam texels
 *     (index texture all-zero, :137-139,176-178) are simply not listed; irr must be zero-initialised by the caller.
 *     texel_ids == NULL means ALL Nt texels whatever n_ids says: a caller whose list can be empty (a rank's shard of a short list) must skip the call
 *     -- an empty device array has no address to pass (the Python wrapper does, scene.Scene.irt_generate).
 *   irr [Nt,3] dev: only listed texels are written.  A texel's value has a fixed summation order per kernel form, so for lists of
 *     >= 32768 texels (the 64-texels-per-wave form; shorter lists use the one-texel-per-wave form, which differs in the last bits) the texture does
 *     not depend on the order or sharding of texel_ids nor on the launch configuration.
 *   Long lists (>= 32768 texels) use a stream-ordered scratch allocation (hipMallocAsync/hipFreeAsync on `stream`,
 *     384 bytes per listed texel at N >= 2048) for the per-pass-range partial sums.
 *   stats [8] u64 dev, nullable: += rays, 64-byte node fetches, triangle tests, hits, wave-level node steps, wave-level
 *   triangle steps (how often a wavefront executed each loop body: lane utilisation = lane count / (64 * wave count)), 2 reserved. */
TEXIR_API int texir_irt_generate(const texir_scene* scene, const float* pos /*dev*/, const float* nrm /*dev*/,
                       const float* shift /*dev*/, const int32_t* texel_ids /*dev, nullable*/, int64_t n_ids,
                       int64_t Nt, int32_t N, int32_t mode, float* irr /*dev*/, uint64_t* stats /*dev, nullable*/,
                       void* stream);

/* Recording texir_irt_generate into a hipGraph: a long list's partial-sum scratch cannot be allocated stream-ordered inside a recorded graph (ROCm 7.2:
 * some replays then read wrong partial sums), so a launch on a CAPTURING stream uses scratch reserved on the scene beforehand and fails with a clear
 * message if there is not enough: call this once, outside any capture, with the largest (n_ids, N) that will be recorded (384 bytes per listed texel at
 * N >= 2048; nothing for lists < 32 768 texels).  One recorded launch per scene at a time may be in flight (they share the scratch).  Eager launches are
 * unaffected.  No reference counterpart. */
TEXIR_API int texir_scene_reserve_scratch(texir_scene* scene, int64_t n_ids, int32_t N);

/* name of the kernel form ONE texir_irt_generate call over n_ids listed texels at N samples launches on this scene
 * ("irt_group_kernel<false, 4, 6>": 64 texels per wave; "irt_kernel<false, 4|2>": one texel per wave) -- the launcher's own
 * decision, so that bench.py's roofline names the kernel that really ran.  buf receives a NUL-terminated string. */
TEXIR_API int texir_irt_kernel_name(const texir_scene* scene, int64_t n_ids, int32_t N, char* buf, int32_t cap);

/* Replaces MaterialModel.render + specular_reflectance (models/mat_nvdiffrast.py:201-249, 260-279), forward:
 *   rgb = irr*albedo/pi + (1/S) sum_i Ls_i * w_i(roughness)        
 * normal,albedo,points,irr [P,3] dev; rough [P] dev; cam [3] dev; shift [P,2] dev (GGX sample shift, as above)
 * Ls_ws [P,S,3] dev, nullable: traced radiance saved for texir_spec_backward.
 * clamp_eps: the floor of the BRDF denominators -- 1e-14 (TINY_TINY_NUMBER) in models/mat_nvdiffrast.py:270-279, 1e-6 (TINY_NUMBER) in
 * the evaluation model models/test_nvdiffrast.py:320-333.
 * ls_given = 1: Ls_ws is an INPUT (the `lighting` argument of specular_reflectance, function seam 3 of SURVEY 8b) and nothing is
 * traced -- specular_reflectance(lighting, h, n, v, l, roughness)/S + irr*albedo/pi on the caller's lighting. */
TEXIR_API int texir_spec_forward(const texir_scene* scene /*nullable when ls_given*/, const float* normal, const float* albedo, const float* rough,
                       const float* points, const float* irr, const float* cam, const float* shift, int64_t P,
                       int32_t S, float clamp_eps, int32_t ls_given, float* rgb /*dev [P,3]*/, float* Ls_ws /*dev, nullable*/, void* stream);

/* The training form of the pair (round 4): the forward also writes dw_ws [P,S] = d w_i / d roughness (its dual-number sample chain yields them next to the
 * weights), and the backward is a stream over what the forward kept -- d_rough[p] = (1/S) sum_i (Ls_i . d_rgb[p]) dw_i, d_albedo = d_rgb*irr/pi -- instead of
 * texir_spec_backward's recomputation of the whole sample chain.  Same values as the pair above (tests/test_gpu_parity.py). */
TEXIR_API int texir_spec_forward_train(const texir_scene* scene /*nullable when ls_given*/, const float* normal, const float* albedo, const float* rough,
                       const float* points, const float* irr, const float* cam, const float* shift, int64_t P,
                       int32_t S, float clamp_eps, int32_t ls_given, float* rgb /*dev [P,3]*/, float* Ls_ws /*dev [P,S,3]*/, float* dw_ws /*dev [P,S]*/, void* stream);
TEXIR_API int texir_spec_backward_ws(const float* irr, const float* Ls_ws, const float* dw_ws, const float* d_rgb, int64_t P, int32_t S,
                       float* d_albedo /*dev [P,3], nullable*/, float* d_rough /*dev [P], nullable*/, void* stream);

/* Analytic backward of the above (what autograd computes in the reference): given d_rgb [P,3],
 *   d_albedo [P,3] = d_rgb*irr/pi ;  d_rough [P] = sum_c d_rgb_c * (1/S) sum_i Ls_ic * dw_i/dr
 * (gradient through a=r^2 -> cos/sin theta -> h -> vdh -> l -> ndl, ndh and through k=(r+1)^2/8; Ls constant,
 * torch clamp sub-gradients).  Either output may be NULL. */
TEXIR_API int texir_spec_backward(const float* normal, const float* rough, const float* points, const float* irr,
                        const float* cam, const float* shift, const float* Ls_ws, const float* d_rgb, int64_t P,
                        int32_t S, float clamp_eps, float* d_albedo /*dev, nullable*/, float* d_rough /*dev, nullable*/, void* stream);

/* Replaces the lighting integral of diffuse_reflectance (models/mat_nvdiffrast.py:252-258; live in the evaluation model's
 * relighting branch, models/test_nvdiffrast.py:268-274): per point  E = (2*pi/N) sum_i L_i * clamp(n.l_i, 0, 1)  over uniform
 * directions (sample_type 0: the IrT estimator) or  E = (pi/N) sum_i L_i  over cosine-distributed directions (sample_type 1), so that
 * diffuse_reflectance(...)/N == E * albedo / pi.  pos (already offset), nrm [P,3], shift [P,2] dev -> irr [P,3] dev. */
TEXIR_API int texir_diffuse_irradiance(const texir_scene* scene, const float* pos, const float* nrm, const float* shift, int64_t P,
                        int32_t N, int32_t sample_type, float* irr /*dev [P,3]*/, void* stream);

/* Replaces RenderLoss.forward + SegLoss.forward + hdr_scale (models/loss.py:81-115, 214-295; utils/general.py:61-66),
 * value AND gradient in one call.  The reference's one-hot mask tensors (seg_mask/floor_max_mask [C,6,h,w,1],
 * room_seg_mask [R,6,h,w,1], built at trainer/train_material.py:255-296) are passed in their compact form:
 *   seg_id [P] u8: class of the pixel (255 = none); hl [P] u8: floor_max_mask of the pixel's own class;
 *   room_id [P] u8 (255 = none; stage 2 only, else NULL).
 * stage 0/1/2 as in RenderLoss.forward; loss_type 0 = 'L1', 1 = 'L2' (applies to the rendered-radiance term only), 2 = no
 * rendered-radiance term (SegLoss alone: the psnr / ssim / msssim variants of loss.py:65-73 add their own image term).
 * gt,rgb,albedo [P,3]; rough,rough_womip,empty_mask,gt_mask [P] (all dev; inputs a stage does not read may be NULL).
 * out [2] dev: (total loss, seg term) = the reference's (loss, seg_loss.item()).
 * d_rgb [P,3], d_albedo [P,3] (stage 0), d_rough [P] (stages 1,2): d loss / d input for an upstream gradient of 1
 *   (the means of stages 0 and 2 are differentiated through, the stage-1 quantile target is detached, as in the reference).
 * workspace: texir_loss_workspace_bytes(P, C, R) bytes of device scratch. hw = h*w (the stage-1 scale, loss.py:101). */
TEXIR_API int64_t texir_loss_workspace_bytes(int64_t P, int32_t C, int32_t R);
TEXIR_API int texir_loss_forward(int32_t stage, int32_t loss_type, const float* gt, const float* rgb, const float* albedo,
                       const float* rough, const float* rough_womip, const float* empty_mask, const float* gt_mask,
                       const uint8_t* seg_id, const uint8_t* hl, const uint8_t* room_id, int64_t P, int32_t C, int32_t R,
                       int32_t hw, void* workspace, float* out /*dev [2]*/, float* d_rgb, float* d_albedo, float* d_rough,
                       void* stream);

/* ---- G-buffer production: replaces nvdiffrast rasterize + interpolate (models/mat_nvdiffrast.py:119-128,
 * models/tracer_o3d_irt.py:102-108) by casting one primary ray per cube-map pixel through the scene's BVH.
 * Geometry and cameras never change (optim_cam=False, configs/syn.conf:20), so hosts cache the result per view. */

/* per-corner shading normals [3T,3] host (normals[indices] of pyredner.load_obj, tracer_o3d_irt.py:61,105);
 * without them the G-buffer carries geometric normals. */
TEXIR_API int texir_scene_set_corner_normals(texir_scene* scene, const float* corner_normals /*host*/);

/* mvp [6,4,4] f32 HOST: the reference's per-face mvp as passed to MaterialModel.forward (row-vector convention
 * clip = [x,y,z,1] @ mvp, datasets/dataset.py:464-465); inverted in double inside the library.  Pixel (row i, col j) is at ndc ((j+.5)/c*2-1, (i+.5)/c*2-1), i.e. nvdiffrast's layout.
 * Outputs, P = 6*c*c, all dev: pos [P,3] (interpolated vertex position; empty -> (1,0,0)), nrm [P,3] (interpolated
 * corner normals; empty -> (1,0,0)), mask [P] (1 where rast[...,3] > 0), uv [P,2] (= texc), uv_da [P,4]
 * (= texd: du/dX, du/dY, dv/dX, dv/dY per pixel), tri_id [P] (primitive id + 1, 0 = empty; nvdiffrast rast[...,3]).
 * flip_v: 1 -> uv.v := 1 - v (pyredner's OBJ convention for the nvdiffrast-side textures). */
TEXIR_API int texir_gbuffer_cast(const texir_scene* scene, const float* mvp /*host*/, int32_t cube_res, int32_t flip_v,
                       float* pos, float* nrm, float* mask, float* uv, float* uv_da, int32_t* tri_id, void* stream);

/* ---- nvdiffrast `texture` restated (models/mat_nvdiffrast.py:131-139).  Level 0 of the mip stack IS the caller's texture
 * [H,W,C] (C <= 4, no copy); levels 1.. live in a caller-provided "rest" buffer of texir_mip_elems() floats that
 * texir_mip_build fills by 2x2 box filtering.
 * filter_mode 0 = 'linear' (bilinear, level 0; mips_rest may be NULL), 1 = 'linear-mipmap-linear' (trilinear, LOD from
 * uv_da); boundary_mode 'wrap'. */
TEXIR_API int32_t texir_mip_levels(int32_t H, int32_t W, int32_t max_mip_level);
TEXIR_API int64_t texir_mip_elems(int32_t H, int32_t W, int32_t C, int32_t levels);
/* from_level 0: levels 1.. from the texture; 1: level 1 of mips_rest is already current (texir_adam_step_tex wrote it with the
 * update), build levels 2.. from it.  (TEXIR_MIP_PER_LEVEL=1 selects the one-launch-per-level reference implementation; the
 * default builds five levels per launch through LDS -- identical bits.) */
TEXIR_API int texir_mip_build(const float* tex /*dev*/, float* mips_rest /*dev*/, int32_t H, int32_t W, int32_t C, int32_t levels,
                       int32_t from_level, void* stream);
TEXIR_API int texir_tex_fetch_forward(const float* tex /*dev*/, const float* mips_rest /*dev, nullable*/, int32_t H, int32_t W,
                       int32_t C, int32_t levels, const float* uv /*dev [P,2]*/, const float* uv_da /*dev [P,4], nullable for mode 0*/,
                       int32_t filter_mode, int64_t P, float* out /*dev [P,C]*/, void* stream);
/* d_tex [H,W,C] and grad_rest [texir_mip_elems] (dev) must be zero on entry; on return d_tex holds d loss / d texture
 * (the gradient scattered into the mip levels is folded down to level 0). */
TEXIR_API int texir_tex_fetch_backward(float* d_tex /*dev*/, float* grad_rest /*dev, nullable for mode 0*/, int32_t H, int32_t W,
                       int32_t C, int32_t levels, const float* uv, const float* uv_da, int32_t filter_mode, int64_t P,
                       const float* d_out /*dev [P,C]*/, void* stream);

/* As texir_tex_fetch_backward (trilinear; the autograd backward of dr.texture, models/mat_nvdiffrast.py:131,134), but the last fold is left out: on return d_tex holds the level-0 scatter only and the first
 * (H/2)*(W/2)*C floats of grad_rest hold the level-1 gradient with all coarser levels folded in.  texir_adam_step_tex consumes the pair. */
TEXIR_API int texir_tex_fetch_backward_deferred(float* d_tex /*dev*/, float* grad_rest /*dev*/, int32_t H, int32_t W, int32_t C,
                       int32_t levels, const float* uv, const float* uv_da, int64_t P, const float* d_out /*dev [P,C]*/, void* stream);

/* Atomics-free backward of the dr.texture fetches (models/mat_nvdiffrast.py:131-139) whose (uv, uv_da) never change (a cached view:
 * geometry and cameras are constant).
 * texir_tex_taps lists the taps of every pixel: keys/weights [P*8] (4 bilinear taps x 2 mip levels; bilinear mode uses the first 4),
 * key = texel index in the unified order [level 0 | levels 1.. as in mips_rest], -1 for unused slots, weight = bilinear x level blend.
 * The caller sorts them by key once (stable), forms segments (key, start, count) + the sorted (pixel, weight) lists, and then every
 * backward is texir_tex_gather_backward: one thread per touched texel adds its list in order (deterministic), followed by the
 * same folds as texir_tex_fetch_backward (defer_last_fold = 1: as texir_tex_fetch_backward_deferred).  d_tex / grad_rest zero on entry.
 * With defer_last_fold = 1, d_tex may be NULL when no listed tap samples level 0 (no key < H*W): the level-0 gradient is then
 * identically zero and texir_adam_step_tex(grad = NULL) never reads it.
 * defer_last_fold = 2 (levels >= 4, H and W divisible by 4): the folds stop at level 2 -- grad_rest then holds the raw level-1 gradient and the
 * level-2 gradient with everything coarser folded in; texir_adam_step_tex(grad_level2 = ...) takes both remaining folds over and the
 * read-modify-write of the level-1 stack (half the traffic of the folds) disappears. */
TEXIR_API int texir_tex_taps(int32_t H, int32_t W, int32_t C, int32_t levels, const float* uv, const float* uv_da, int32_t filter_mode,
                       int64_t P, int64_t* keys /*dev [P*8]*/, float* weights /*dev [P*8]*/, void* stream);
TEXIR_API int texir_tex_gather_backward(float* d_tex, float* grad_rest, int32_t H, int32_t W, int32_t C, int32_t levels,
                       const int64_t* seg_key /*dev [n_seg]*/, const int32_t* seg_start, const int32_t* seg_count, int32_t n_seg,
                       const int32_t* pix /*dev, sorted*/, const float* weights /*dev, sorted*/, const float* d_out /*dev [P,C]*/,
                       int32_t filter_mode, int32_t defer_last_fold, void* stream);

/* ---- optimiser step of the material textures: torch.optim.Adam(lr, betas, eps) (trainer/train_material.py:122-123,448-450)
 * fused with the clamp the trainer applies right after it (:458, :592-593).  step >= 1; lo/hi = clamp range (+-inf = none). */
TEXIR_API int texir_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, int32_t step, float clamp_lo, float clamp_hi, void* stream);

/* The same step (trainer/train_material.py:448-458) for a texture [H,W,C] whose gradient is grad + 0.25 * grad_level1[y/2][x/2] (texir_tex_fetch_backward_deferred): the
 * last mip fold is fused into the optimiser's read of the gradient; results equal fold + texir_adam_step bit for bit.
 * grad == NULL: no pixel of the step sampled mip level 0, the level-0 gradient is neither materialised nor read.
 * grad_mask != NULL: grad is a buffer that is never cleared; only the texels a view's tap lists touch (their bit is set) carry this
 * step's values, all others count as zero -- the usual case of a handful of level-0 taps costs 1 bit per texel instead of a fill + a read.
 * mip_level1 != NULL: level 1 of the NEXT forward's mip stack (models/mat_nvdiffrast.py:131-134 rebuild it from the updated
 * texture every step) is written on the way: texir_mip_build(..., from_level = 1) then skips the pass over the full texture. */
TEXIR_API int texir_adam_step_tex(float* param, const float* grad /*nullable: level-0 gradient identically zero*/,
                       const uint32_t* grad_mask /*nullable: 1 bit per texel (bit t&31 of word t>>5): grad is valid -- and read -- only where set*/,
                       const float* grad_level1,
                       const float* grad_level2 /*nullable: [H/4,W/4,C] level-2 gradient NOT yet folded into grad_level1 (texir_tex_gather_backward with
                                                  defer_last_fold = 2): the step also performs grad_level1 += 0.25 * grad_level2 on the fly, same fma*/,
                       float* exp_avg, float* exp_avg_sq, float* mip_level1 /*nullable: [H/2,W/2,C] <- 2x2 average of the updated texels*/,
                       int32_t H, int32_t W, int32_t C, float lr, float beta1, float beta2, float eps, int32_t step, float clamp_lo,
                       float clamp_hi, void* stream);

/* The same two steps for launches that must not depend on host arguments that change every step (a captured hipGraph of the whole
 * optimisation step, trainer/train_material.py:408-458 -- forward, loss, backward AND optimizer.step()): the step count, learning rate
 * and betas of up to 64 parameters live in device memory (`state` [n][4] doubles: step count, lr, beta1, beta2).  texir_adam_tick
 * advances the step count of the records selected by `mask` (bit i = record i) and writes hyper[i] = (lr / (1 - beta1^step),
 * sqrt(1 - beta2^step)) in double precision, the expressions of torch.optim.Adam's single-tensor path; the *_dev steps read their
 * record's pair instead of taking (lr, step).  A learning-rate scheduler writes state[i][1] between steps.
 * texir_adam_step_tex_dev: grad_level1 may be NULL when grad_level2 is given and no tap of the view's lists touches mip level 1 (its direct
 * gradient is identically zero: the level-1 stack -- a quarter of the texture -- is then not read at all). */
TEXIR_API int texir_adam_tick(double* state /*dev [n][4]*/, float* hyper /*dev [n][2]*/, int32_t n_records, uint64_t mask, void* stream);
TEXIR_API int texir_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper /*dev [2]*/,
                       float beta1, float beta2, float eps, float clamp_lo, float clamp_hi, void* stream);
TEXIR_API int texir_adam_step_tex_dev(float* param, const float* grad, const uint32_t* grad_mask, const float* grad_level1, const float* grad_level2,
                       float* exp_avg, float* exp_avg_sq, float* mip_level1, int32_t H, int32_t W, int32_t C, const float* hyper /*dev [2]*/,
                       float beta1, float beta2, float eps, float clamp_lo, float clamp_hi, void* stream);

/* ---- batched forms of the texture-side launches of one material step (trainer/train_material.py:408-458: the reference fetches its albedo and roughness
 * textures with separate dr.texture calls, models/mat_nvdiffrast.py:131-139, and steps them with one torch.optim.Adam) --------------------------------------
 * A step over k material textures launches k x (mip build + tail, fetch, gather, fold, Adam); these entry points take up to TEXIR_MAX_BATCH jobs and issue ONE
 * launch per kind (blocks are dealt to the jobs by block index; every job keeps its own sizes and channel count).  Each job's result is bit-identical to the
 * corresponding single-texture entry point above.  Errors of these three functions are reported through texir_batch_last_error() (thread-local). */
#define TEXIR_MAX_BATCH 4
TEXIR_API const char* texir_batch_last_error(void);

/* texir_mip_build (when build_from >= 0) followed by texir_tex_fetch_forward, for every job: one pyramid launch, one tail launch, one fetch launch */
typedef struct texir_tex_fetch_job {
    const float* tex;          /* dev [H,W,C] */
    float* mips_rest;          /* dev: levels 1.. (texir_mip_elems floats); nullable when levels == 1 */
    int32_t H, W, C, levels;
    int32_t build_from;        /* -1: the stack is valid as it is; 0: build levels 1.. from tex; 1: level 1 is valid (texir_adam_step_tex wrote it), build levels 2.. */
    int32_t filter_mode;       /* 0 bilinear | 1 trilinear */
    const float* uv;           /* dev [P,2] */
    const float* uv_da;        /* dev [P,4]; nullable for mode 0 */
    int64_t P;
    float* out;                /* dev [P,C] */
} texir_tex_fetch_job;
TEXIR_API int texir_tex_fetch_forward_batch(const texir_tex_fetch_job* jobs /*host*/, int32_t n_jobs, void* stream);

/* texir_tex_gather_backward for every job: one gather launch, one fold launch.
 * rest_mask (nullable; needs defer_last_fold = 2): one bit per texel of grad_rest (bit t & 31 of word t >> 5; t = the key of texir_tex_taps minus H*W), set for
 * exactly the texels the job's tap lists name.  With it grad_rest need NOT be zero on entry and is never cleared: the folds read the levels above 2 through the
 * mask (an unset texel counts as zero) and WRITE level 2; level 1 is left as the gather wrote it, valid where the mask says so --
 * texir_adam_step_tex_dev_batch(level1_mask = the same mask) reads it accordingly.  The per-step fill of the gradient stacks (a third of the texture) and the
 * optimiser's dense read of the level-1 stack (a quarter) disappear; same floats as with a zero-filled stack. */
typedef struct texir_tex_gather_job {
    float* d_tex;              /* dev [H,W,C]; nullable as in texir_tex_gather_backward */
    float* grad_rest;          /* dev */
    int32_t H, W, C, levels;
    const int64_t* seg_key; const int32_t* seg_start; const int32_t* seg_count; int32_t n_seg;
    const int32_t* pix; const float* weights;
    const float* d_out;        /* dev [P,C] */
    int32_t filter_mode, defer_last_fold;
    const uint32_t* rest_mask; /* dev, nullable */
    const float* d_out2;       /* dev [P,C], nullable: a second gradient of the same fetch output (two consumers of one dr.texture result, models/mat_nvdiffrast.py:134 ->
                                  :179 render and models/loss.py:108): the gather adds the two on the fly -- the sum autograd's add launch would have written */
} texir_tex_gather_job;
TEXIR_API int texir_tex_gather_backward_batch(const texir_tex_gather_job* jobs /*host*/, int32_t n_jobs, void* stream);

/* texir_adam_step_tex_dev for every job in one launch */
typedef struct texir_adam_tex_job {
    float* param; const float* grad; const uint32_t* grad_mask;
    const float* grad_level1;
    const uint32_t* level1_mask;   /* dev, nullable; only with grad_level2: grad_level1 is valid -- and read -- only where the bit of its texel is set (see above) */
    const float* grad_level2;
    float* exp_avg; float* exp_avg_sq; float* mip_level1;
    int32_t H, W, C;
    const float* hyper;            /* dev [2] (texir_adam_tick) */
    float beta1, beta2, eps, clamp_lo, clamp_hi;
} texir_adam_tex_job;
/* g [n_texels][C] += g0 [n_texels][C] where bit t of mask (one bit per texel) is set: the sparse level-0 gradient of a trilinear fetch folded into the dense level-0
 * gradient an un-mipmapped fetch of the SAME texture produced in the same backward pass (stage 1, models/mat_nvdiffrast.py:131-139: both fetches read materials_r).
 * Errors: texir_batch_last_error(). */
TEXIR_API int texir_grad_add_masked(float* g /*dev*/, const float* g0 /*dev*/, const uint32_t* mask /*dev*/, int64_t n_texels, int32_t C, void* stream);
TEXIR_API int texir_adam_step_tex_dev_batch(const texir_adam_tex_job* jobs /*host*/, int32_t n_jobs, void* stream);

/* ---- host-side codec loops of the file formats around the path (both take HOST pointers) ----------------------------
 * PNG scanline un-filtering (filters 0-4, PNG spec 9.2) of zlib-inflated IDAT data: raw [H][stride+1] -> out [H][stride]; replaces the
 * decode half of cv2.imread("0.png", -1) (models/tracer_o3d_irt.py:91, datasets/dataset.py:489-492). */
TEXIR_API int texir_png_unfilter(const uint8_t* raw /*host*/, int32_t H, int32_t stride, int32_t bytes_per_pixel, uint8_t* out /*host*/);
/* Radiance .hdr scanlines (flat or new-style RLE, per scanline) after the resolution line -> RGBE bytes [H][W][4]; returns the bytes
 * consumed (< 0: error).  Replaces the decode half of cv2.imread(".hdr", -1) (models/tracer_o3d_irt.py:77, datasets/dataset.py:480). */
TEXIR_API int64_t texir_hdr_decode_scanlines(const uint8_t* data /*host*/, int64_t n, int32_t W, int32_t H, uint8_t* rgbe /*host*/);
/* Radiance RGBE pixel codec, float RGB [npix][3] <-> RGBE [npix][4] (host pointers, multi-threaded): the pixel arithmetic of
 * cv2.imwrite(".hdr") / cv2.imread(".hdr", -1) (trainer/generate_ir_texture.py:82, trainer/train_material.py:350-353,
 * models/mat_nvdiffrast.py:73).  Byte-identical to io_formats.rgbe_encode / rgbe_decode (numpy, kept as the test reference). */
TEXIR_API int texir_rgbe_encode(const float* rgb /*host*/, int64_t npix, uint8_t* rgbe /*host*/);
/* RGBE bytes [H][W][4] -> new-style RLE scanlines laid out as cv2.imwrite(".hdr") writes them (its default IMWRITE_HDR_COMPRESSION_RLE);
 * returns the bytes written (< 0: error / cap too small; 4 + 4 * (W + W / 64 + 4) bytes per scanline always suffice). */
TEXIR_API int64_t texir_hdr_encode_rle(const uint8_t* rgbe /*host*/, int32_t W, int32_t H, uint8_t* out /*host*/, int64_t cap);
TEXIR_API int texir_rgbe_decode(const uint8_t* rgbe /*host*/, int64_t npix, float* rgb /*host*/);
/* Wavefront OBJ text -> arrays (host pointers, multi-threaded): the parse half of o3d.io.read_triangle_mesh / pyredner.load_obj
 * (models/tracer_o3d_irt.py:75,85,183-189; models/mat_nvdiffrast.py:87,193-199).  texir_obj_parse classifies lines by their first token
 * (v / vt / vn / f; LF, CRLF, CR), fan-triangulates polygons, resolves negative indices and fills counts = {n_v, n_vt, n_vn, n_tri};
 * texir_obj_take copies into caller-owned buffers sized from the counts (v [n_v][3], vt [n_vt][2], vn [n_vn][3] float32 -- a correctly
 * rounded double rounded once more, as float() + numpy do; fi / ft / fn [n_tri][3] int32, 0-based, -1 = absent) and frees the handle
 * (all-null outputs: only frees).  Array-identical to io_formats.load_obj_py. */
TEXIR_API int texir_obj_parse(const char* text /*host*/, int64_t n, void** handle, int64_t counts[4]);
TEXIR_API int texir_obj_take(void* handle, float* v, float* vt, float* vn, int32_t* fi, int32_t* ft, int32_t* fn);

#ifdef __cplusplus
}
#endif
#endif
