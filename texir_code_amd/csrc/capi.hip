// C-ABI of libtexir_hip.so (include/texir_hip.h).  Thin: argument checks, handle ownership, launches.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/texir_hip.h"
#include "bvh_build.h"
#include "env.h"
#include "kernels.h"

using namespace texir;

struct texir_scene {
    int device = 0;
    SceneDev dev{};
    void* d_nodes4 = nullptr; void* d_nodes4f = nullptr;
    void* d_nodes = nullptr; void* d_tris = nullptr; void* d_quads = nullptr; void* d_uvs = nullptr; float* d_tex = nullptr;
    float* d_tex_tiled = nullptr;        // retiled copy read by the hit shader (texture layouts 1, 2); d_tex stays the row-major master
    size_t tiled_bytes = 0;
    int tiled_layout = 0;                // layout d_tex_tiled was sized for
    uint32_t* d_tex_packed = nullptr;    // 4-byte shared-exponent texels (layouts 3, 4), in force only while the texture packs EXACTLY (install_texture)
    size_t packed_bytes = 0;
    int packed_layout = 0;
    unsigned int* d_tex_flag = nullptr;  // device word: texels of the last pack that were not representable
    int64_t n_nodes = 0, n_nodes4 = 0, n_tris = 0, n_slots = 0 /* leaf-order slots behind d_tris / d_uvs / d_cnrm: = n_tris, or 2 per quad record (bvh_build.h) */, n_quads = 0, n_uv_recs = 0 /* 32-byte uv records behind d_uvs: one per quad record (TEXIR_UV_QUAD) or per slot */, max_depth = 0;
    int width = 2;
    size_t tex_bytes = 0;
    std::vector<uint32_t> slot_prim;     // leaf slot -> primitive id (host copy, for per-corner attribute uploads; 0xFFFFFFFF: an empty slot)
    std::vector<uint8_t> slot_rot;       // leaf slot -> rotation of the stored corners (stored corner k = the caller's corner (rot + k) % 3)
    void* d_cnrm = nullptr;              // leaf-ordered corner normals, 3 x float4 per triangle
    float* d_scratch = nullptr;          // texir_scene_reserve_scratch: the IrT partial-sum scratch of RECORDED launches (eager launches allocate stream-ordered)
    size_t scratch_bytes = 0;
    std::vector<float*> retired_scratch;  // smaller blocks a grown reservation replaced: recorded graphs may still name them, so they live until destroy
    // chunk counters of the persistent IrT kernel: one slot per launch, handed out round-robin so that launches of one scene that
    // overlap on different streams do not share a counter (kWorkSlots launches would have to be in flight at once)
    static constexpr int kWorkSlots = 64;
    unsigned long long* d_work = nullptr;
    mutable std::atomic<unsigned> work_next{0};      // (launching on an immutable scene still advances the slot)
    // phase-scheduler weight of this scene (device_common.h TEXIR_SCHED): decided once by texir_scene_tune, read by every tracing launch
    mutable std::atomic<int> sched_state{0};         // 0 undecided, 1 being decided, 2 decided
    mutable std::atomic<int> sched_weight{0};        // 0 = the compile-time default (2)
    mutable std::atomic<double> node_utilisation{-1.0};   // what the decision measured (texir_scene_scheduler)
};

// the kernel-argument view of a scene for one launch: the immutable part + the scheduler weight in force now
// (`coherent`: the launch traces direction cells -- texir_irt_generate, where quad leaves made weight 3 the better one for scenes whose node steps run full,
// tools/r04_session33.sh; the other kernels' rays share no direction and keep what they were measured with: at most 2)
static SceneDev dev_of(const texir_scene* s, bool coherent = false)
{
    SceneDev d = s->dev;
    const int forced = env().sched_weight;
    int w = forced ? forced : s->sched_weight.load(std::memory_order_relaxed);
    if (!forced && !coherent && w > 2) w = 2;
    d.sched_weight = w;
    return d;
}

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(TEXIR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Chooses and fills the hit shader's copy of the radiance texture; d_tex (row-major float32 master) has been written on `st` already.
// TEXIR_TEX_LAYOUT 4 (default) / 3: 4-byte shared-exponent texels when EVERY texel is three 8-bit integers times one power of two -- what an RGBE file
// times 2^hdr_exposure always is (tracer_o3d_irt.py:77-81) -- decoding to the identical floats; any other texture (a float-valued synthetic one,
// 2.5 x an RGBE one, negative values) keeps the float32 tiles of layout 2.  The decision needs the pack kernel's verdict, i.e. one stream
// synchronisation: on a capturing stream the float32 layout is taken without asking.  0 / 1 / 2 force the float32 layouts (A/B runs).
static int install_texture(texir_scene* s, hipStream_t st)
{
    const int want = env().tex_layout, Ht = s->dev.Ht, Wt = s->dev.Wt;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if ((want == 3 || want == 4) && !capturing && Ht < (1 << 16) && Wt < (1 << 16)) {
        int tx, ty;
        const size_t bytes = tex_pack_bytes(Ht, Wt, want, &tx, &ty);
        if (s->d_tex_packed && (s->packed_layout != want || s->packed_bytes != bytes)) { HIP_TRY(hipFree(s->d_tex_packed)); s->d_tex_packed = nullptr; }
        if (!s->d_tex_packed) { HIP_TRY(hipMalloc((void**)&s->d_tex_packed, bytes)); s->packed_bytes = bytes; s->packed_layout = want; }
        if (!s->d_tex_flag) HIP_TRY(hipMalloc((void**)&s->d_tex_flag, sizeof(unsigned int)));
        HIP_TRY(launch_tex_pack(s->d_tex, s->d_tex_packed, Ht, Wt, want, s->d_tex_flag, st));
        unsigned int bad = 1;
        HIP_TRY(hipMemcpyAsync(&bad, s->d_tex_flag, sizeof(bad), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (bad == 0) {
            s->dev.tex = reinterpret_cast<const float*>(s->d_tex_packed); s->dev.tex_layout = want; s->dev.tiles_x = tx;
            return TEXIR_OK;
        }
    }
    const int layout = want >= 3 ? 2 : want;
    if (layout == 1 || layout == 2) {
        int tx, ty;
        const size_t bytes = tex_retile_bytes(Ht, Wt, layout, &tx, &ty);
        if (s->d_tex_tiled && (s->tiled_layout != layout || s->tiled_bytes != bytes)) { HIP_TRY(hipFree(s->d_tex_tiled)); s->d_tex_tiled = nullptr; }
        if (!s->d_tex_tiled) {
            if (capturing) return fail(TEXIR_ERR_INVALID, "texir_scene_set_texture: the stream is being captured and the float32 tiles are not allocated yet");
            HIP_TRY(hipMalloc((void**)&s->d_tex_tiled, bytes)); s->tiled_bytes = bytes; s->tiled_layout = layout;
        }
        HIP_TRY(launch_tex_retile(s->d_tex, s->d_tex_tiled, Ht, Wt, layout, st));
        s->dev.tex = s->d_tex_tiled; s->dev.tex_layout = layout; s->dev.tiles_x = tx;
    } else {
        s->dev.tex = s->d_tex; s->dev.tex_layout = 0; s->dev.tiles_x = 0;
    }
    return TEXIR_OK;
}

extern "C" {

const char* texir_last_error(void) { return g_err.c_str(); }
int texir_version(void) { return 100; }

int texir_scene_create(const float* verts, int32_t V, const int32_t* tris, int32_t T, const float* tri_uvs, const float* hdr_tex,
                       int32_t Ht, int32_t Wt, int32_t device, texir_scene** out)
{
    if (!verts || !tris || !tri_uvs || !hdr_tex || !out) return fail(TEXIR_ERR_INVALID, "texir_scene_create: null argument");
    if (V <= 0 || T <= 0 || Ht <= 0 || Wt <= 0) return fail(TEXIR_ERR_INVALID, "texir_scene_create: empty mesh or texture");
    // the traversal addresses nodes and triangles with 32-bit byte offsets (64-byte nodes, 48-byte triangle records)
    if ((uint64_t)(2 * (uint64_t)T + 1) * sizeof(GpuTri) >= (1ull << 32)) return fail(TEXIR_ERR_INVALID, "texir_scene_create: too many triangles (%d; limit 44 M)", (int)T);
    for (int64_t i = 0; i < 3 * (int64_t)T; i++)
        if (tris[i] < 0 || tris[i] >= V) return fail(TEXIR_ERR_INVALID, "texir_scene_create: triangle index %d out of range", (int)tris[i]);
    HIP_TRY(hipSetDevice(device));
    BvhHost h;
    try { build_bvh(verts, V, tris, T, tri_uvs, h); }
    catch (const std::bad_alloc&) { return fail(TEXIR_ERR_NOMEM, "BVH build: out of host memory"); }
    catch (const std::exception& ex) { return fail(TEXIR_ERR_INVALID, "BVH build: %s", ex.what()); }
    texir_scene* s = new (std::nothrow) texir_scene;
    if (!s) return fail(TEXIR_ERR_NOMEM, "out of host memory");
    s->n_slots = h.n_slots; s->n_quads = (int64_t)h.quads.size(); s->n_uv_recs = (int64_t)h.uvs.size();
    s->slot_prim.resize((size_t)h.n_slots); s->slot_rot.resize((size_t)h.n_slots);
    for (int64_t i = 0; i < h.n_slots; i++) { s->slot_prim[i] = h.tris[i].prim; uint32_t r; std::memcpy(&r, &h.tris[i].pad1, 4); s->slot_rot[i] = (uint8_t)(r % 3u); }
    s->device = device; s->n_nodes = (int64_t)h.nodes.size(); s->n_tris = T; s->max_depth = h.max_depth;
    s->tex_bytes = sizeof(float) * 3 * (size_t)Ht * Wt;
    auto bail = [&](hipError_t e, const char* what) { texir_scene_destroy(s); return fail(TEXIR_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
    hipError_t e;
    // traversal tree: 4-wide quantised by default (TEXIR_BVH_WIDTH=2 keeps the binary tree); falls back to binary when the wide
    // tree's worst-case stack (3 pushes per level) would not fit the traversal stack
    const int want_w = env().bvh_width;
    s->width = (want_w == 4 && 3 * h.max_depth4 + 2 <= kStackCap) ? 4 : 2;
    if (s->width == 4) {
        s->n_nodes4 = (int64_t)h.nodes4.size(); s->max_depth = h.max_depth4;
        if ((e = hipMalloc(&s->d_nodes4, h.nodes4.size() * sizeof(GpuNode4))) != hipSuccess) return bail(e, "hipMalloc nodes4");
        if ((e = hipMemcpy(s->d_nodes4, h.nodes4.data(), h.nodes4.size() * sizeof(GpuNode4), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload nodes4");
        // the float form of the same nodes, read by wave-uniform node steps through the scalar cache (TEXIR_UNIFORM_FLOAT=0: A/B switch)
        if (env().uniform_float) {
            if ((e = hipMalloc(&s->d_nodes4f, h.nodes4f.size() * sizeof(GpuNode4F))) != hipSuccess) return bail(e, "hipMalloc nodes4f");
            if ((e = hipMemcpy(s->d_nodes4f, h.nodes4f.data(), h.nodes4f.size() * sizeof(GpuNode4F), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload nodes4f");
        }
    }
    if ((e = hipMalloc(&s->d_nodes, h.nodes.size() * sizeof(GpuNode))) != hipSuccess) return bail(e, "hipMalloc nodes");
    if ((e = hipMalloc(&s->d_tris, h.tris.size() * sizeof(GpuTri))) != hipSuccess) return bail(e, "hipMalloc tris");
    if (!h.quads.empty()) {
        if ((e = hipMalloc(&s->d_quads, h.quads.size() * sizeof(GpuQuad))) != hipSuccess) return bail(e, "hipMalloc quads");
        if ((e = hipMemcpy(s->d_quads, h.quads.data(), h.quads.size() * sizeof(GpuQuad), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload quads");
    }
    if (!h.uvs.empty() && (e = hipMalloc(&s->d_uvs, h.uvs.size() * sizeof(GpuTriUV))) != hipSuccess) return bail(e, "hipMalloc uvs");
    if ((e = hipMalloc((void**)&s->d_tex, s->tex_bytes)) != hipSuccess) return bail(e, "hipMalloc texture");
    if ((e = hipMalloc((void**)&s->d_work, texir_scene::kWorkSlots * 8 * kWorkStride * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc work counters");
    if ((e = hipMemcpy(s->d_nodes, h.nodes.data(), h.nodes.size() * sizeof(GpuNode), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload nodes");
    if ((e = hipMemcpy(s->d_tris, h.tris.data(), h.tris.size() * sizeof(GpuTri), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload tris");
    if (!h.uvs.empty() && (e = hipMemcpy(s->d_uvs, h.uvs.data(), h.uvs.size() * sizeof(GpuTriUV), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload uvs");
    if ((e = hipMemcpy(s->d_tex, hdr_tex, s->tex_bytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "upload texture");
    s->dev.nodes4 = (const float4*)s->d_nodes4; s->dev.nodes4f = (const float4*)s->d_nodes4f;
    s->dev.nodes = (const float4*)s->d_nodes; s->dev.tris = (const float4*)s->d_tris; s->dev.quads = (const float4*)s->d_quads; s->dev.uvs = (const float4*)s->d_uvs;
    s->dev.tex = s->d_tex; s->dev.Ht = Ht; s->dev.Wt = Wt; s->dev.tex_layout = 0; s->dev.tiles_x = 0; s->dev.sched_weight = 0;
    // hit-shader copy of the texture (install_texture): 4-byte texels when they decode to the identical floats, else float32 tiles
    if (const int rc = install_texture(s, 0)) { texir_scene_destroy(s); return rc; }
    if ((e = hipStreamSynchronize(0)) != hipSuccess) return bail(e, "texture layout");
    *out = s;
    return TEXIR_OK;
}

int texir_scene_destroy(texir_scene* s)
{
    if (!s) return TEXIR_OK;
    (void)hipSetDevice(s->device);
    if (s->d_nodes4) (void)hipFree(s->d_nodes4);
    if (s->d_nodes4f) (void)hipFree(s->d_nodes4f);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_tris) (void)hipFree(s->d_tris);
    if (s->d_quads) (void)hipFree(s->d_quads);
    if (s->d_uvs) (void)hipFree(s->d_uvs);
    if (s->d_tex) (void)hipFree(s->d_tex);
    if (s->d_tex_tiled) (void)hipFree(s->d_tex_tiled);
    if (s->d_tex_packed) (void)hipFree(s->d_tex_packed);
    if (s->d_tex_flag) (void)hipFree(s->d_tex_flag);
    if (s->d_cnrm) (void)hipFree(s->d_cnrm);
    if (s->d_scratch) (void)hipFree(s->d_scratch);
    for (float* p : s->retired_scratch) (void)hipFree(p);
    if (s->d_work) (void)hipFree(s->d_work);
    delete s;
    return TEXIR_OK;
}

int texir_scene_set_texture(texir_scene* s, const float* tex, int32_t Ht, int32_t Wt, int32_t is_device, void* stream)
{
    if (!s || !tex) return fail(TEXIR_ERR_INVALID, "texir_scene_set_texture: null argument");
    if (Ht != s->dev.Ht || Wt != s->dev.Wt) return fail(TEXIR_ERR_INVALID, "texir_scene_set_texture: size %dx%d != scene texture %dx%d", Ht, Wt, s->dev.Ht, s->dev.Wt);
    HIP_TRY(hipMemcpyAsync(s->d_tex, tex, s->tex_bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, (hipStream_t)stream));
    return install_texture(s, (hipStream_t)stream);
}

int texir_scene_info(const texir_scene* s, int64_t out[8])
{
    if (!s || !out) return fail(TEXIR_ERR_INVALID, "texir_scene_info: null argument");
    out[0] = s->width == 4 ? s->n_nodes4 : s->n_nodes; out[1] = s->n_tris; out[2] = s->max_depth;
    out[3] = s->width == 4 ? s->n_nodes4 * (int64_t)(sizeof(GpuNode4) + (s->d_nodes4f ? sizeof(GpuNode4F) : 0)) : s->n_nodes * (int64_t)sizeof(GpuNode);
    out[4] = s->n_slots * (int64_t)sizeof(GpuTri) + s->n_quads * (int64_t)sizeof(GpuQuad); out[5] = s->d_uvs ? s->n_uv_recs * (int64_t)sizeof(GpuTriUV) : 0; out[6] = (int64_t)(s->dev.tex_layout >= 3 ? s->packed_bytes : s->dev.tex_layout >= 1 ? s->tiled_bytes : s->tex_bytes); out[7] = s->device;
    return TEXIR_OK;
}

int texir_scene_texture_layout(const texir_scene* s, int32_t* layout)
{
    if (!s || !layout) return fail(TEXIR_ERR_INVALID, "texir_scene_texture_layout: null argument");
    *layout = s->dev.tex_layout;
    return TEXIR_OK;
}

int texir_texel_pack(const float* rgb, int64_t n, uint32_t* words, uint8_t* exact)
{
    if ((!rgb || !words) && n > 0) return fail(TEXIR_ERR_INVALID, "texir_texel_pack: null argument");
    for (int64_t i = 0; i < n; i++) {
        uint32_t b[3], w = 0u;
        std::memcpy(b, rgb + 3 * i, 12);
        const bool ok = pack_texel(b[0], b[1], b[2], w);
        words[i] = ok ? w : 0u;
        if (exact) exact[i] = ok ? 1 : 0;
    }
    return TEXIR_OK;
}

int texir_texel_unpack(const uint32_t* words, int64_t n, float* rgb)
{
    if ((!rgb || !words) && n > 0) return fail(TEXIR_ERR_INVALID, "texir_texel_unpack: null argument");
    for (int64_t i = 0; i < n; i++) {
        const uint32_t q = words[i], sb = (q >> 1) & 0x7F800000u;
        float scale; std::memcpy(&scale, &sb, 4);
        for (int c = 0; c < 3; c++) rgb[3 * i + c] = (float)((q >> (8 * c)) & 0xFFu) * scale;
    }
    return TEXIR_OK;
}

int texir_scene_prefetch(const texir_scene* s, int32_t what, int32_t blocks, void* stream)
{
    if (!s) return fail(TEXIR_ERR_INVALID, "texir_scene_prefetch: null argument");
    if (blocks < 1) blocks = 512;
    uint32_t* sink = reinterpret_cast<uint32_t*>(s->d_work);         // (never written in practice; any device word will do)
    if ((what & 1) && s->d_nodes4) HIP_TRY(launch_prefetch(s->d_nodes4, (size_t)s->n_nodes4 * sizeof(GpuNode4), blocks, sink, (hipStream_t)stream));
    if ((what & 2) && s->d_nodes4f) HIP_TRY(launch_prefetch(s->d_nodes4f, (size_t)s->n_nodes4 * sizeof(GpuNode4F), blocks, sink, (hipStream_t)stream));
    // (what the traversal reads of the triangles: the quad records where the library has them, else the leaf-ordered triangles)
    if ((what & 4) && s->d_quads && s->width == 4) HIP_TRY(launch_prefetch(s->d_quads, (size_t)s->n_quads * sizeof(GpuQuad), blocks, sink, (hipStream_t)stream));
    else if ((what & 4) && s->d_tris) HIP_TRY(launch_prefetch(s->d_tris, (size_t)s->n_slots * sizeof(GpuTri), blocks, sink, (hipStream_t)stream));
    if ((what & 8) && s->d_uvs) HIP_TRY(launch_prefetch(s->d_uvs, (size_t)s->n_uv_recs * sizeof(GpuTriUV), blocks, sink, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_scene_scheduler(const texir_scene* s, double out[2])
{
    if (!s || !out) return fail(TEXIR_ERR_INVALID, "texir_scene_scheduler: null argument");
    out[0] = (double)s->sched_weight.load(); out[1] = s->node_utilisation.load();
    return TEXIR_OK;
}

int texir_scene_tune(const texir_scene* s, const float* pos, const float* nrm, const float* shift, const int32_t* texel_ids, int64_t n_ids,
                     int32_t N, int32_t mode, void* stream)
{
    if (!s || !pos || !nrm || !shift) return fail(TEXIR_ERR_INVALID, "texir_scene_tune: null argument");
    if (mode < 0 || mode > 1) return fail(TEXIR_ERR_INVALID, "texir_scene_tune: mode must be uniform(0) or cosine(1), got %d", mode);
    // one host thread decides (compare-exchange: 0 undecided -> 1 deciding -> 2 decided); the others return at once and launch with the weight in force
    int expect = 0;
    if (!s->sched_state.compare_exchange_strong(expect, 1)) return TEXIR_OK;
    int w = env().sched_weight;
    double util = -1.0;
    if (w == 0 && texel_ids && n_ids >= 65536 && N >= 256 && s->width == 4) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            s->sched_state.store(0);                 // a blocking measurement cannot be recorded: stay undecided, keep the default weight
            return fail(TEXIR_ERR_INVALID, "texir_scene_tune: the stream is being captured (tune the scene before recording)");
        }
        unsigned long long* work = s->d_work + (size_t)(s->work_next.fetch_add(1) % texir_scene::kWorkSlots) * 8 * kWorkStride;
        const int64_t count = 16384, first = ((n_ids / 3) / 64) * 64;
        const hipError_t e = irt_probe_node_utilisation(s->dev, pos, nrm, shift, texel_ids, first, count, N, mode, work, (hipStream_t)stream, &util);
        if (e != hipSuccess) { s->sched_state.store(0); return fail(TEXIR_ERR_HIP, "texir_scene_tune: %s", hipGetErrorString(e)); }
        w = (util >= 0.0 && util < 0.60) ? 1 : 3;          // (3 since the quad leaves: a leaf visit is one record step, the node lanes are worth waiting for a little longer)
    }
    if (w == 0) { s->sched_state.store(0); return TEXIR_OK; }       // a short list decides nothing: a later, longer call may
    s->sched_weight.store(w);
    s->node_utilisation.store(util);
    s->sched_state.store(2);
    return TEXIR_OK;
}

int texir_reload_env(void)
{
    env_reload();
    return TEXIR_OK;
}

int texir_env_switch(const char* name, int32_t* value)
{
    int v = 0;
    if (!name || !value || env_switch(name, &v) != 0) return fail(TEXIR_ERR_INVALID, "texir_env_switch: unknown switch %s", name ? name : "(null)");
    *value = v;
    return TEXIR_OK;
}

int texir_trace_shade(const texir_scene* s, const float* org, const float* dir, int64_t R, float t_min, float* radiance, float* t_hit,
                      uint32_t* prim_id, float* prim_uv, void* stream)
{
    if (!s || !org || !dir || !radiance) return fail(TEXIR_ERR_INVALID, "texir_trace_shade: null argument");
    if (R < 0) return fail(TEXIR_ERR_INVALID, "texir_trace_shade: negative ray count");
    HIP_TRY(launch_trace_shade(dev_of(s), org, dir, R, t_min, radiance, t_hit, prim_id, prim_uv, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_generate_dir(const float* normals, const float* roughness, const float* shift, int64_t b, int32_t N, int32_t mode, float* L, void* stream)
{
    if (!normals || !shift || !L) return fail(TEXIR_ERR_INVALID, "texir_generate_dir: null argument");
    if (mode < 0 || mode > 2) return fail(TEXIR_ERR_INVALID, "texir_generate_dir: unknown mode %d", mode);
    if (mode == TEXIR_MODE_IMPORTANCE && !roughness) return fail(TEXIR_ERR_INVALID, "texir_generate_dir: importance mode needs roughness");
    if (b < 0 || N < 0) return fail(TEXIR_ERR_INVALID, "texir_generate_dir: negative size");
    HIP_TRY(launch_gen_dir(normals, roughness, shift, b, N, mode, L, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_irt_generate(const texir_scene* s, const float* pos, const float* nrm, const float* shift, const int32_t* texel_ids, int64_t n_ids,
                       int64_t Nt, int32_t N, int32_t mode, float* irr, uint64_t* stats, void* stream)
{
    if (!s || !pos || !nrm || !shift || !irr) return fail(TEXIR_ERR_INVALID, "texir_irt_generate: null argument");
    if (mode < 0 || mode > 1) return fail(TEXIR_ERR_INVALID, "texir_irt_generate: mode must be uniform(0) or cosine(1), got %d", mode);
    if (N <= 0 || Nt < 0 || n_ids < 0) return fail(TEXIR_ERR_INVALID, "texir_irt_generate: bad sizes N=%d Nt=%lld n_ids=%lld", N, (long long)Nt, (long long)n_ids);
    if (Nt >= (1ll << 31)) return fail(TEXIR_ERR_INVALID, "texir_irt_generate: Nt too large");
    int64_t n = texel_ids ? n_ids : Nt;
    unsigned long long* work = s->d_work + (size_t)(s->work_next.fetch_add(1) % texir_scene::kWorkSlots) * 8 * kWorkStride;
    // (No measurement and no synchronisation in here: the phase scheduler's weight is whatever texir_scene_tune / TEXIR_SCHED_WEIGHT has decided for
    // the scene, 2 until then.  The texture is the same bits either way.)
    // A launch that is being RECORDED into a hipGraph must not allocate: stream-ordered allocations inside a recorded graph gave wrong partial sums in
    // some replays on this stack (ROCm 7.2: profiles/r04, graph replay probe).  It uses the scratch the caller reserved on the scene beforehand.
    float* scratch = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        const size_t need = irt_scratch_bytes(s->dev, n, N);
        if (need > s->scratch_bytes)
            return fail(TEXIR_ERR_INVALID, "texir_irt_generate: the stream is being captured and the launch needs %zu bytes of scratch, %zu reserved -- call "
                                           "texir_scene_reserve_scratch(scene, n_ids, N) before recording", need, s->scratch_bytes);
        scratch = need ? s->d_scratch : nullptr;
    }
    HIP_TRY(launch_irt(dev_of(s, true), pos, nrm, shift, texel_ids, n, N, mode, irr, (unsigned long long*)stats, work, (hipStream_t)stream, scratch));
    return TEXIR_OK;
}

int texir_scene_reserve_scratch(texir_scene* s, int64_t n_ids, int32_t N)
{
    if (!s || n_ids < 0 || N <= 0) return fail(TEXIR_ERR_INVALID, "texir_scene_reserve_scratch: bad argument");
    const size_t need = irt_scratch_bytes(s->dev, n_ids, N);
    if (need <= s->scratch_bytes) return TEXIR_OK;
    HIP_TRY(hipSetDevice(s->device));
    // Growing never frees: hipGraphs recorded against the smaller block have its address baked into their kernel nodes and may still be replayed.
    // The old block is retired (kept until texir_scene_destroy); launches recorded from now on use the new one.
    float* grown = nullptr;
    HIP_TRY(hipMalloc((void**)&grown, need));
    if (s->d_scratch) s->retired_scratch.push_back(s->d_scratch);
    s->d_scratch = grown;
    s->scratch_bytes = need;
    return TEXIR_OK;
}

int texir_irt_kernel_name(const texir_scene* s, int64_t n_ids, int32_t N, char* buf, int32_t cap)
{
    if (!s || !buf || cap < 1) return fail(TEXIR_ERR_INVALID, "texir_irt_kernel_name: null argument");
    const IrtPlan p = irt_plan(s->dev, n_ids, N);
    snprintf(buf, (size_t)cap, "%s", p.name);
    return TEXIR_OK;
}

int texir_spec_forward(const texir_scene* s, const float* normal, const float* albedo, const float* rough, const float* points, const float* irr,
                       const float* cam, const float* shift, int64_t P, int32_t S, float clamp_eps, int32_t ls_given, float* rgb, float* Ls_ws, void* stream)
{
    if ((!s && !ls_given) || !normal || !albedo || !rough || !points || !irr || !cam || !shift || !rgb) return fail(TEXIR_ERR_INVALID, "texir_spec_forward: null argument");
    if (ls_given && !Ls_ws) return fail(TEXIR_ERR_INVALID, "texir_spec_forward: ls_given needs the lighting in Ls_ws");
    if (P < 0 || S <= 0 || !(clamp_eps > 0.f)) return fail(TEXIR_ERR_INVALID, "texir_spec_forward: bad sizes P=%lld S=%d clamp_eps=%g", (long long)P, S, (double)clamp_eps);
    SceneDev none{};
    HIP_TRY(launch_spec_fwd(s ? dev_of(s) : none, normal, albedo, rough, points, irr, cam, shift, P, S, clamp_eps, ls_given ? 1 : 0, rgb, Ls_ws, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_spec_forward_train(const texir_scene* s, const float* normal, const float* albedo, const float* rough, const float* points, const float* irr,
                             const float* cam, const float* shift, int64_t P, int32_t S, float clamp_eps, int32_t ls_given, float* rgb, float* Ls_ws, float* dw_ws,
                             void* stream)
{
    if ((!s && !ls_given) || !normal || !albedo || !rough || !points || !irr || !cam || !shift || !rgb || !Ls_ws || !dw_ws)
        return fail(TEXIR_ERR_INVALID, "texir_spec_forward_train: null argument");
    if (P < 0 || S <= 0 || !(clamp_eps > 0.f)) return fail(TEXIR_ERR_INVALID, "texir_spec_forward_train: bad sizes P=%lld S=%d clamp_eps=%g", (long long)P, S, (double)clamp_eps);
    SceneDev none{};
    HIP_TRY(launch_spec_fwd(s ? dev_of(s) : none, normal, albedo, rough, points, irr, cam, shift, P, S, clamp_eps, ls_given ? 1 : 0, rgb, Ls_ws, (hipStream_t)stream, dw_ws));
    return TEXIR_OK;
}

int texir_spec_backward_ws(const float* irr, const float* Ls_ws, const float* dw_ws, const float* d_rgb, int64_t P, int32_t S, float* d_albedo, float* d_rough, void* stream)
{
    if (!irr || !Ls_ws || !dw_ws || !d_rgb) return fail(TEXIR_ERR_INVALID, "texir_spec_backward_ws: null argument");
    if (P < 0 || S <= 0) return fail(TEXIR_ERR_INVALID, "texir_spec_backward_ws: bad sizes P=%lld S=%d", (long long)P, S);
    HIP_TRY(launch_spec_bwd_ws(irr, Ls_ws, dw_ws, d_rgb, P, S, d_albedo, d_rough, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_spec_backward(const float* normal, const float* rough, const float* points, const float* irr, const float* cam, const float* shift,
                        const float* Ls_ws, const float* d_rgb, int64_t P, int32_t S, float clamp_eps, float* d_albedo, float* d_rough, void* stream)
{
    if (!normal || !rough || !points || !irr || !cam || !shift || !Ls_ws || !d_rgb) return fail(TEXIR_ERR_INVALID, "texir_spec_backward: null argument");
    if (P < 0 || S <= 0 || !(clamp_eps > 0.f)) return fail(TEXIR_ERR_INVALID, "texir_spec_backward: bad sizes P=%lld S=%d", (long long)P, S);
    HIP_TRY(launch_spec_bwd(normal, rough, points, irr, cam, shift, Ls_ws, d_rgb, P, S, clamp_eps, d_albedo, d_rough, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_diffuse_irradiance(const texir_scene* s, const float* pos, const float* nrm, const float* shift, int64_t P, int32_t N, int32_t sample_type,
                             float* irr, void* stream)
{
    if (!s || !pos || !nrm || !shift || !irr) return fail(TEXIR_ERR_INVALID, "texir_diffuse_irradiance: null argument");
    if (sample_type < 0 || sample_type > 1) return fail(TEXIR_ERR_INVALID, "texir_diffuse_irradiance: sample_type must be uniform(0) or cosine(1)");
    if (N <= 0 || P < 0 || P >= (1ll << 31)) return fail(TEXIR_ERR_INVALID, "texir_diffuse_irradiance: bad sizes N=%d P=%lld", N, (long long)P);
    unsigned long long* work = s->d_work + (size_t)(s->work_next.fetch_add(1) % texir_scene::kWorkSlots) * 8 * kWorkStride;
    // uniform: (2 pi / N) sum L n.l -- the IrT estimator; cosine: (pi / N) sum L over cosine-distributed directions
    const int mode = sample_type == 1 ? (1 | 4) : 0;
    HIP_TRY(launch_irt(dev_of(s), pos, nrm, shift, nullptr, P, N, mode, irr, nullptr, work, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_scene_set_corner_normals(texir_scene* s, const float* cn)
{
    if (!s || !cn) return fail(TEXIR_ERR_INVALID, "texir_scene_set_corner_normals: null argument");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<float> buf((size_t)s->n_slots * 12, 0.f);
    for (int64_t i = 0; i < s->n_slots; i++) {
        if (s->slot_prim[i] == 0xFFFFFFFFu) continue;                            // an empty slot (bvh_build.h): never hit
        const float* src = cn + 9 * (size_t)s->slot_prim[i];
        for (int k = 0; k < 3; k++) {                                            // stored corner k = the caller's corner (rot + k) % 3
            const float* c = src + 3 * ((s->slot_rot[i] + k) % 3);
            buf[12 * i + 4 * k] = c[0]; buf[12 * i + 4 * k + 1] = c[1]; buf[12 * i + 4 * k + 2] = c[2]; buf[12 * i + 4 * k + 3] = 0.f;
        }
    }
    if (!s->d_cnrm) HIP_TRY(hipMalloc(&s->d_cnrm, buf.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(s->d_cnrm, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice));
    return TEXIR_OK;
}

int texir_gbuffer_cast(const texir_scene* s, const float* mvp, int32_t c, int32_t flip_v, float* pos, float* nrm, float* mask, float* uv,
                       float* uv_da, int32_t* tri_id, void* stream)
{
    if (!s || !mvp || !pos || !nrm || !mask || !uv || !uv_da || !tri_id) return fail(TEXIR_ERR_INVALID, "texir_gbuffer_cast: null argument");
    if (c <= 0 || c > 16384) return fail(TEXIR_ERR_INVALID, "texir_gbuffer_cast: bad cube_res %d", c);
    HIP_TRY(launch_gbuffer(dev_of(s), mvp, (const float4*)s->d_cnrm, c, flip_v, pos, nrm, mask, uv, uv_da, tri_id, (hipStream_t)stream));
    return TEXIR_OK;
}

int32_t texir_mip_levels(int32_t H, int32_t W, int32_t max_mip_level) { return mip_levels(H, W, max_mip_level); }
int64_t texir_mip_elems(int32_t H, int32_t W, int32_t C, int32_t levels) { return mip_total_elems(H, W, C, levels); }

static int check_tex(const char* fn, int H, int W, int C, int levels)
{
    if (H <= 0 || W <= 0 || C <= 0 || C > 4 || levels < 1 || levels > 16) return fail(TEXIR_ERR_INVALID, "%s: bad texture H=%d W=%d C=%d levels=%d", fn, H, W, C, levels);
    if (levels > mip_levels(H, W, 15)) return fail(TEXIR_ERR_INVALID, "%s: %d mip levels not available for %dx%d", fn, levels, H, W);
    return 0;
}

int texir_mip_build(const float* tex, float* mips_rest, int32_t H, int32_t W, int32_t C, int32_t levels, int32_t from_level, void* stream)
{
    if (!tex || !mips_rest) return fail(TEXIR_ERR_INVALID, "texir_mip_build: null argument");
    if (from_level < 0 || from_level > 1) return fail(TEXIR_ERR_INVALID, "texir_mip_build: from_level must be 0 or 1");
    if (int rc = check_tex("texir_mip_build", H, W, C, levels)) return rc;
    HIP_TRY(launch_mip_build(tex, mips_rest, H, W, C, levels, from_level, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_tex_fetch_forward(const float* tex, const float* mips_rest, int32_t H, int32_t W, int32_t C, int32_t levels, const float* uv,
                            const float* uv_da, int32_t filter_mode, int64_t P, float* out, void* stream)
{
    if (!tex || !uv || !out || (filter_mode == 1 && (!uv_da || (levels > 1 && !mips_rest)))) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_forward: null argument");
    if (filter_mode < 0 || filter_mode > 1 || P < 0) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_forward: bad filter_mode/P");
    if (int rc = check_tex("texir_tex_fetch_forward", H, W, C, levels)) return rc;
    HIP_TRY(launch_tex_fetch(tex, mips_rest, H, W, C, levels, uv, uv_da, filter_mode, P, out, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_tex_fetch_backward(float* d_tex, float* grad_rest, int32_t H, int32_t W, int32_t C, int32_t levels, const float* uv, const float* uv_da,
                             int32_t filter_mode, int64_t P, const float* d_out, void* stream)
{
    if (!d_tex || !uv || !d_out || (filter_mode == 1 && (!uv_da || (levels > 1 && !grad_rest)))) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_backward: null argument");
    if (filter_mode < 0 || filter_mode > 1 || P < 0) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_backward: bad filter_mode/P");
    if (int rc = check_tex("texir_tex_fetch_backward", H, W, C, levels)) return rc;
    HIP_TRY(launch_tex_fetch_bwd(d_tex, grad_rest, H, W, C, levels, uv, uv_da, filter_mode, P, d_out, 0, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_tex_fetch_backward_deferred(float* d_tex, float* grad_rest, int32_t H, int32_t W, int32_t C, int32_t levels, const float* uv,
                                      const float* uv_da, int64_t P, const float* d_out, void* stream)
{
    if (!d_tex || !uv || !d_out || !uv_da || !grad_rest) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_backward_deferred: null argument");
    if (P < 0 || levels < 2) return fail(TEXIR_ERR_INVALID, "texir_tex_fetch_backward_deferred: needs P >= 0 and at least two mip levels");
    if (int rc = check_tex("texir_tex_fetch_backward_deferred", H, W, C, levels)) return rc;
    HIP_TRY(launch_tex_fetch_bwd(d_tex, grad_rest, H, W, C, levels, uv, uv_da, 1, P, d_out, 1, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_tex_taps(int32_t H, int32_t W, int32_t C, int32_t levels, const float* uv, const float* uv_da, int32_t filter_mode, int64_t P,
                   int64_t* keys, float* weights, void* stream)
{
    if (!uv || !keys || !weights || (filter_mode == 1 && !uv_da)) return fail(TEXIR_ERR_INVALID, "texir_tex_taps: null argument");
    if (filter_mode < 0 || filter_mode > 1 || P < 0) return fail(TEXIR_ERR_INVALID, "texir_tex_taps: bad filter_mode/P");
    if (int rc = check_tex("texir_tex_taps", H, W, C, levels)) return rc;
    HIP_TRY(launch_tex_taps(H, W, C, levels, uv, uv_da, filter_mode, P, (long long*)keys, weights, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_tex_gather_backward(float* d_tex, float* grad_rest, int32_t H, int32_t W, int32_t C, int32_t levels, const int64_t* seg_key,
                              const int32_t* seg_start, const int32_t* seg_count, int32_t n_seg, const int32_t* pix, const float* weights,
                              const float* d_out, int32_t filter_mode, int32_t defer_last_fold, void* stream)
{
    // d_tex may be NULL with defer_last_fold when no tap of the lists samples level 0 (then nothing is written there and the caller's
    // optimiser step treats the level-0 gradient as identically zero)
    if ((!d_tex && !defer_last_fold) || !d_out || (n_seg > 0 && (!seg_key || !seg_start || !seg_count || !pix || !weights)) || (filter_mode == 1 && levels > 1 && !grad_rest))
        return fail(TEXIR_ERR_INVALID, "texir_tex_gather_backward: null argument");
    if (filter_mode < 0 || filter_mode > 1 || n_seg < 0 || defer_last_fold < 0 || defer_last_fold > 2 || (defer_last_fold && (filter_mode != 1 || levels < 2))
        || (defer_last_fold == 2 && (levels < 4 || (H & 3) || (W & 3))))
        return fail(TEXIR_ERR_INVALID, "texir_tex_gather_backward: bad filter_mode/n_seg/defer_last_fold");
    if (int rc = check_tex("texir_tex_gather_backward", H, W, C, levels)) return rc;
    HIP_TRY(launch_tex_gather_bwd(d_tex, grad_rest, H, W, C, levels, (const long long*)seg_key, seg_start, seg_count, n_seg, pix, weights, d_out,
                                  filter_mode, defer_last_fold, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_adam_step_tex(float* param, const float* grad, const uint32_t* grad_mask, const float* grad_level1, const float* grad_level2, float* exp_avg,
                        float* exp_avg_sq, float* mip_level1, int32_t H, int32_t W, int32_t C, float lr, float beta1, float beta2, float eps, int32_t step,
                        float clamp_lo, float clamp_hi, void* stream)
{
    if (!param || !grad_level1 || !exp_avg || !exp_avg_sq) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex: null argument");
    if (H < 2 || W < 2 || (H & 1) || (W & 1) || C < 1 || C > 4 || step < 1) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex: bad H/W/C/step");
    if (grad_level2 && ((H & 3) || (W & 3))) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex: a level-2 gradient needs H and W divisible by 4");
    HIP_TRY(launch_adam_tex(param, grad, grad_mask, grad_level1, grad_level2, exp_avg, exp_avg_sq, mip_level1, H, W, C, lr, beta1, beta2, eps, step, clamp_lo, clamp_hi, nullptr, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    int32_t step, float clamp_lo, float clamp_hi, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(TEXIR_ERR_INVALID, "texir_adam_step: null argument");
    if (n < 0 || step < 1) return fail(TEXIR_ERR_INVALID, "texir_adam_step: bad n/step");
    HIP_TRY(launch_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, clamp_lo, clamp_hi, nullptr, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_adam_tick(double* state, float* hyper, int32_t n_records, uint64_t mask, void* stream)
{
    if (!state || !hyper) return fail(TEXIR_ERR_INVALID, "texir_adam_tick: null argument");
    if (n_records < 0 || n_records > 64) return fail(TEXIR_ERR_INVALID, "texir_adam_tick: 0..64 records per call (got %d)", n_records);
    HIP_TRY(launch_adam_tick(state, hyper, n_records, (unsigned long long)mask, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper, float beta1, float beta2,
                        float eps, float clamp_lo, float clamp_hi, void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper) return fail(TEXIR_ERR_INVALID, "texir_adam_step_dev: null argument");
    if (n < 0) return fail(TEXIR_ERR_INVALID, "texir_adam_step_dev: bad n");
    HIP_TRY(launch_adam(param, grad, exp_avg, exp_avg_sq, n, 0.f, beta1, beta2, eps, 1, clamp_lo, clamp_hi, hyper, (hipStream_t)stream));
    return TEXIR_OK;
}

int texir_adam_step_tex_dev(float* param, const float* grad, const uint32_t* grad_mask, const float* grad_level1, const float* grad_level2, float* exp_avg,
                            float* exp_avg_sq, float* mip_level1, int32_t H, int32_t W, int32_t C, const float* hyper, float beta1, float beta2, float eps,
                            float clamp_lo, float clamp_hi, void* stream)
{
    if (!param || (!grad_level1 && !grad_level2) || !exp_avg || !exp_avg_sq || !hyper) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex_dev: null argument");
    if (H < 2 || W < 2 || (H & 1) || (W & 1) || C < 1 || C > 4) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex_dev: bad H/W/C");
    if (grad_level2 && ((H & 3) || (W & 3))) return fail(TEXIR_ERR_INVALID, "texir_adam_step_tex_dev: a level-2 gradient needs H and W divisible by 4");
    HIP_TRY(launch_adam_tex(param, grad, grad_mask, grad_level1, grad_level2, exp_avg, exp_avg_sq, mip_level1, H, W, C, 0.f, beta1, beta2, eps, 1, clamp_lo, clamp_hi, hyper,
                            (hipStream_t)stream));
    return TEXIR_OK;
}

int64_t texir_loss_workspace_bytes(int64_t P, int32_t C, int32_t R) { return (int64_t)loss_workspace_bytes(P, C, R); }

int texir_loss_forward(int32_t stage, int32_t loss_type, const float* gt, const float* rgb, const float* albedo, const float* rough,
                       const float* rough_womip, const float* empty_mask, const float* gt_mask, const uint8_t* seg_id, const uint8_t* hl,
                       const uint8_t* room_id, int64_t P, int32_t C, int32_t R, int32_t hw, void* workspace, float* out, float* d_rgb,
                       float* d_albedo, float* d_rough, void* stream)
{
    if (stage < 0 || stage > 2) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: stage must be 0, 1 or 2 (got %d)", stage);
    if (loss_type < 0 || loss_type > 2) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: loss_type must be 0 (L1), 1 (L2) or 2 (segmentation term only)");
    if (!gt || !rgb || !empty_mask || !seg_id || !workspace || !out || !d_rgb) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: null argument");
    if (P <= 0 || C <= 0 || C > 255) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: bad sizes P=%lld C=%d", (long long)P, C);
    if (stage == 0 && (!albedo || !gt_mask || !d_albedo)) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: stage 0 needs albedo, gt_mask, d_albedo");
    if (stage == 1 && (!rough || !rough_womip || !hl || !d_rough)) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: stage 1 needs rough, rough_womip, hl, d_rough");
    if (stage == 2 && (!rough || !room_id || !d_rough || R <= 0 || R > 255)) return fail(TEXIR_ERR_INVALID, "texir_loss_forward: stage 2 needs rough, room_id, d_rough, 0<R<256");
    HIP_TRY(launch_loss(stage, loss_type, gt, rgb, albedo, rough, rough_womip, empty_mask, gt_mask, seg_id, hl, room_id, P, C, R, hw, workspace, out,
                        d_rgb, d_albedo, d_rough, (hipStream_t)stream));
    return TEXIR_OK;
}

}  // extern "C"
