/*
 * texir_oracle.c -- CPU ORACLE for the TexIR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm.  It exists so that the
 * HIP product path (texir_code_amd/csrc) can be CHECKED against it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built
 * from this file.  Nothing under texir_code_amd/ imports, links or executes it.
 *
 * Parity pin: every function below is checked (tests/test_oracle_golden.py) against
 * golden vectors captured by importing the reference's own pure-torch code in the build
 * container (oracle/make_golden.py -> tests/golden NPZ files).  The ray/triangle boundary
 * (Open3D RaycastingScene = Embree, un-vendored, unpinned in requirements.txt:4) cannot be
 * executed here: its published semantics are restated (closest hit, t in units of |dir|,
 * miss => t=+inf, barycentrics (u,v) with hit=(1-u-v)v0+u*v1+v*v2) and are "parity
 * unpinned" at that one boundary.
 *
 * Reference lines followed (relative to /root/reference):
 *   utils/sample_util.py:28-41      RadicalInverse / Hammersley        -> txo_hammersley
 *   utils/sample_util.py:63-146     generate_dir (uniform/cosine/GGX)  -> txo_generate_dir
 *   models/tracer_o3d_irt.py:240-269 query_irf (cast + uv interp + bilinear border fetch)
 *                                                                      -> txo_cast_rays_*, txo_shade_hits
 *   models/tracer_o3d_irt.py:156-178 IrT estimator                     -> txo_irt_generate
 *   models/mat_nvdiffrast.py:201-249,260-279 render + specular_reflectance -> txo_spec_forward
 *
 * The "canonical BVH2" (binary, binned SAH, <=4 tris/leaf, 32-byte node, 36-byte triangle)
 * defined in SURVEY.md 8(d) lives here; its traversal counters define the ALGORITHMIC
 * bytes per ray that bench.py's roofline figure is computed from.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TXO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* scene                                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float bmin[3]; int32_t left_first; /* inner: index of left child (right = left+1); leaf: first tri slot */
    float bmax[3]; int32_t count;      /* 0 => inner node, >0 => leaf with `count` triangles               */
} TxoNode;                             /* 32 bytes: the canonical node of SURVEY.md 8(d)                    */

typedef struct {
    int V, T, Ht, Wt;
    float *verts;      /* [V,3]  */
    int32_t *tris;     /* [T,3]  */
    float *tri_uvs;    /* [3T,2] per-corner uvs, Open3D triangle_uvs order */
    float *hdr;        /* [Ht,Wt,3] already flipped + exposure-scaled (tracer_o3d_irt.py:77-81) */
    /* canonical BVH2 */
    TxoNode *nodes; int n_nodes;
    int32_t *tri_order; /* leaf order -> original primitive id */
    float *tri_v;       /* [T,9] leaf-ordered vertex positions: the 36-byte canonical triangle */
    int max_depth;
} TxoScene;

static void tri_bounds(const TxoScene *s, int prim, float *mn, float *mx, float *cen)
{
    for (int a = 0; a < 3; a++) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
    for (int k = 0; k < 3; k++) {
        const float *p = s->verts + 3 * (size_t)s->tris[3 * (size_t)prim + k];
        for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
    }
    for (int a = 0; a < 3; a++) cen[a] = 0.5f * (mn[a] + mx[a]);
}

typedef struct { float *mn, *mx, *cen; } TriInfo;

static float half_area(const float *mn, const float *mx)
{
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}

#define TXO_BINS 16
#define TXO_LEAF 4

static void build_rec(TxoScene *s, const TriInfo *ti, int node_idx, int first, int count, int depth)
{
    TxoNode *n = &s->nodes[node_idx];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = first; i < first + count; i++) {
        int p = s->tri_order[i];
        for (int a = 0; a < 3; a++) {
            if (ti->mn[3 * (size_t)p + a] < mn[a]) mn[a] = ti->mn[3 * (size_t)p + a];
            if (ti->mx[3 * (size_t)p + a] > mx[a]) mx[a] = ti->mx[3 * (size_t)p + a];
            if (ti->cen[3 * (size_t)p + a] < cmn[a]) cmn[a] = ti->cen[3 * (size_t)p + a];
            if (ti->cen[3 * (size_t)p + a] > cmx[a]) cmx[a] = ti->cen[3 * (size_t)p + a];
        }
    }
    for (int a = 0; a < 3; a++) { n->bmin[a] = mn[a]; n->bmax[a] = mx[a]; }
    if (depth > s->max_depth) s->max_depth = depth;
    if (count <= TXO_LEAF) { n->left_first = first; n->count = count; return; }

    /* binned SAH over the three axes */
    int best_axis = -1, best_split = -1; float best_cost = FLT_MAX;
    for (int a = 0; a < 3; a++) {
        float ext = cmx[a] - cmn[a];
        if (!(ext > 0.f)) continue;
        float scale = (float)TXO_BINS / ext;
        int bcnt[TXO_BINS]; float bmn[TXO_BINS][3], bmx[TXO_BINS][3];
        for (int b = 0; b < TXO_BINS; b++) {
            bcnt[b] = 0;
            for (int k = 0; k < 3; k++) { bmn[b][k] = FLT_MAX; bmx[b][k] = -FLT_MAX; }
        }
        for (int i = first; i < first + count; i++) {
            int p = s->tri_order[i];
            int b = (int)((ti->cen[3 * (size_t)p + a] - cmn[a]) * scale);
            if (b >= TXO_BINS) b = TXO_BINS - 1;
            if (b < 0) b = 0;
            bcnt[b]++;
            for (int k = 0; k < 3; k++) {
                if (ti->mn[3 * (size_t)p + k] < bmn[b][k]) bmn[b][k] = ti->mn[3 * (size_t)p + k];
                if (ti->mx[3 * (size_t)p + k] > bmx[b][k]) bmx[b][k] = ti->mx[3 * (size_t)p + k];
            }
        }
        float la[TXO_BINS], ra[TXO_BINS]; int lc[TXO_BINS], rc[TXO_BINS];
        float amn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, amx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; int c = 0;
        for (int b = 0; b < TXO_BINS - 1; b++) {
            c += bcnt[b];
            for (int k = 0; k < 3; k++) { if (bmn[b][k] < amn[k]) amn[k] = bmn[b][k]; if (bmx[b][k] > amx[k]) amx[k] = bmx[b][k]; }
            lc[b] = c; la[b] = c ? half_area(amn, amx) : 0.f;
        }
        for (int k = 0; k < 3; k++) { amn[k] = FLT_MAX; amx[k] = -FLT_MAX; }
        c = 0;
        for (int b = TXO_BINS - 1; b > 0; b--) {
            c += bcnt[b];
            for (int k = 0; k < 3; k++) { if (bmn[b][k] < amn[k]) amn[k] = bmn[b][k]; if (bmx[b][k] > amx[k]) amx[k] = bmx[b][k]; }
            rc[b - 1] = c; ra[b - 1] = c ? half_area(amn, amx) : 0.f;
        }
        for (int b = 0; b < TXO_BINS - 1; b++) {
            if (lc[b] == 0 || rc[b] == 0) continue;
            float cost = la[b] * (float)lc[b] + ra[b] * (float)rc[b];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
        }
    }
    int mid;
    if (best_axis < 0) {
        mid = first + count / 2; /* all centroids coincide: median split in current order */
    } else {
        float ext = cmx[best_axis] - cmn[best_axis];
        float scale = (float)TXO_BINS / ext;
        int i = first, j = first + count - 1;
        while (i <= j) {
            int p = s->tri_order[i];
            int b = (int)((ti->cen[3 * (size_t)p + best_axis] - cmn[best_axis]) * scale);
            if (b >= TXO_BINS) b = TXO_BINS - 1;
            if (b < 0) b = 0;
            if (b <= best_split) i++;
            else { int t = s->tri_order[i]; s->tri_order[i] = s->tri_order[j]; s->tri_order[j] = t; j--; }
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int left = s->n_nodes; s->n_nodes += 2;
    n->left_first = left; n->count = 0;
    build_rec(s, ti, left, first, mid - first, depth + 1);
    build_rec(s, ti, left + 1, mid, first + count - mid, depth + 1);
}

TXO_API TxoScene *txo_scene_create(const float *verts, int V, const int32_t *tris, int T,
                                   const float *tri_uvs, const float *hdr, int Ht, int Wt)
{
    TxoScene *s = (TxoScene *)calloc(1, sizeof(TxoScene));
    s->V = V; s->T = T; s->Ht = Ht; s->Wt = Wt;
    s->verts = (float *)malloc(sizeof(float) * 3 * (size_t)V); memcpy(s->verts, verts, sizeof(float) * 3 * (size_t)V);
    s->tris = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)T); memcpy(s->tris, tris, sizeof(int32_t) * 3 * (size_t)T);
    s->tri_uvs = (float *)malloc(sizeof(float) * 6 * (size_t)T); memcpy(s->tri_uvs, tri_uvs, sizeof(float) * 6 * (size_t)T);
    size_t nh = (size_t)Ht * Wt * 3;
    s->hdr = (float *)malloc(sizeof(float) * nh); memcpy(s->hdr, hdr, sizeof(float) * nh);

    TriInfo ti;
    ti.mn = (float *)malloc(sizeof(float) * 3 * (size_t)T);
    ti.mx = (float *)malloc(sizeof(float) * 3 * (size_t)T);
    ti.cen = (float *)malloc(sizeof(float) * 3 * (size_t)T);
    s->tri_order = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    for (int p = 0; p < T; p++) { tri_bounds(s, p, ti.mn + 3 * (size_t)p, ti.mx + 3 * (size_t)p, ti.cen + 3 * (size_t)p); s->tri_order[p] = p; }
    s->nodes = (TxoNode *)malloc(sizeof(TxoNode) * (2 * (size_t)T + 2));
    s->n_nodes = 1; s->max_depth = 0;
    build_rec(s, &ti, 0, 0, T, 0);
    free(ti.mn); free(ti.mx); free(ti.cen);
    s->tri_v = (float *)malloc(sizeof(float) * 9 * (size_t)T);
    for (int i = 0; i < T; i++) {
        int p = s->tri_order[i];
        for (int k = 0; k < 3; k++) memcpy(s->tri_v + 9 * (size_t)i + 3 * k, s->verts + 3 * (size_t)s->tris[3 * (size_t)p + k], sizeof(float) * 3);
    }
    return s;
}

TXO_API void txo_scene_destroy(TxoScene *s)
{
    if (!s) return;
    free(s->verts); free(s->tris); free(s->tri_uvs); free(s->hdr); free(s->nodes); free(s->tri_order); free(s->tri_v); free(s);
}

TXO_API void txo_scene_info(const TxoScene *s, int64_t *out /* n_nodes, max_depth, T */)
{
    out[0] = s->n_nodes; out[1] = s->max_depth; out[2] = s->T;
}

/* ------------------------------------------------------------------------------------------ */
/* closest-hit ray casting (Open3D RaycastingScene.cast_rays semantics, restated)             */
/* ------------------------------------------------------------------------------------------ */

/* double-precision brute force Moeller-Trumbore: the ground truth for small scenes */
TXO_API void txo_cast_rays_bruteforce(const TxoScene *s, const float *org, const float *dir, int64_t R,
                                      float *t_hit, uint32_t *prim_id, float *prim_uv)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < R; r++) {
        double o[3] = {org[3 * r], org[3 * r + 1], org[3 * r + 2]};
        double d[3] = {dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
        double best = INFINITY, bu = 0, bv = 0; uint32_t bp = 0xFFFFFFFFu;
        for (int p = 0; p < s->T; p++) {
            const float *a = s->verts + 3 * (size_t)s->tris[3 * (size_t)p];
            const float *b = s->verts + 3 * (size_t)s->tris[3 * (size_t)p + 1];
            const float *c = s->verts + 3 * (size_t)s->tris[3 * (size_t)p + 2];
            double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]};
            double e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
            double pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
            double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
            if (det == 0.0) continue;
            double inv = 1.0 / det;
            double tv[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
            double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
            if (u < 0.0 || u > 1.0) continue;
            double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
            double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
            if (v < 0.0 || u + v > 1.0) continue;
            double t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
            if (t > 0.0 && t < best) { best = t; bu = u; bv = v; bp = (uint32_t)p; }
        }
        t_hit[r] = (float)best; prim_id[r] = bp; prim_uv[2 * r] = (float)bu; prim_uv[2 * r + 1] = (float)bv;
    }
}

static inline int slab(const TxoNode *n, const float *o, const float *inv, float tbest, float *tnear)
{
    float t0 = 0.f, t1 = tbest;
    for (int a = 0; a < 3; a++) {
        float ta = (n->bmin[a] - o[a]) * inv[a], tb = (n->bmax[a] - o[a]) * inv[a];
        float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
        /* NaN (0*inf) never narrows the interval */
        if (lo > t0) t0 = lo;
        if (hi < t1) t1 = hi;
    }
    *tnear = t0;
    return t0 <= t1;
}

/* float32 canonical-BVH2 traversal, ordered, closest hit, with the visit counters that define
 * algorithmic bytes/ray: counters[0] += node records fetched (32 B each),
 * counters[1] += triangles tested (36 B each), counters[2] += rays, counters[3] += hits (t>1e-4). */
static void cast_one_bvh(const TxoScene *s, const float *o, const float *d, float *t_out, uint32_t *p_out,
                         float *u_out, float *v_out, uint64_t *cn, uint64_t *ct)
{
    float inv[3];
    for (int a = 0; a < 3; a++) inv[a] = 1.0f / d[a];
    float best = INFINITY, bu = 0, bv = 0; uint32_t bp = 0xFFFFFFFFu;
    int stack[128]; int sp = 0; float tn;
    uint64_t nn = 1, nt = 0;
    if (slab(&s->nodes[0], o, inv, best, &tn)) stack[sp++] = 0;
    while (sp) {
        const TxoNode *n = &s->nodes[stack[--sp]];
        if (n->count) {
            for (int i = n->left_first; i < n->left_first + n->count; i++) {
                const float *a = s->tri_v + 9 * (size_t)i, *b = a + 3, *c = a + 6;
                nt++;
                float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                if (det == 0.f) continue;
                float idet = 1.0f / det;
                float tv[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
                float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * idet;
                if (u < 0.f || u > 1.f) continue;
                float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * idet;
                if (v < 0.f || u + v > 1.f) continue;
                float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * idet;
                if (t > 0.f && t < best) { best = t; bu = u; bv = v; bp = (uint32_t)s->tri_order[i]; }
            }
            continue;
        }
        int l = n->left_first; float tl, tr;
        nn += 2;
        int hl = slab(&s->nodes[l], o, inv, best, &tl), hr = slab(&s->nodes[l + 1], o, inv, best, &tr);
        if (hl && hr) {
            if (tl <= tr) { stack[sp++] = l + 1; stack[sp++] = l; } else { stack[sp++] = l; stack[sp++] = l + 1; }
        } else if (hl) stack[sp++] = l;
        else if (hr) stack[sp++] = l + 1;
    }
    *t_out = best; *p_out = bp; *u_out = bu; *v_out = bv; *cn += nn; *ct += nt;
}

TXO_API void txo_cast_rays_bvh(const TxoScene *s, const float *org, const float *dir, int64_t R,
                               float *t_hit, uint32_t *prim_id, float *prim_uv, uint64_t *counters)
{
    uint64_t tn = 0, tt = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tn, tt)
    for (int64_t r = 0; r < R; r++) {
        uint64_t cn = 0, ct = 0;
        cast_one_bvh(s, org + 3 * r, dir + 3 * r, &t_hit[r], &prim_id[r], &prim_uv[2 * r], &prim_uv[2 * r + 1], &cn, &ct);
        tn += cn; tt += ct;
    }
    if (counters) { counters[0] += tn; counters[1] += tt; counters[2] += (uint64_t)R; }
}

/* query_irf post-intersection math (tracer_o3d_irt.py:248-267): hit mask t>1e-4 & finite; clip bary to
 * [0,1]; corner-uv interpolation in DOUBLE (numpy f64 triangle_uvs * f32 bary) then cast to f32;
 * grid_sample(bilinear, border, align_corners=False) on the flipped texture; misses -> 0. */
static void shade_one(const TxoScene *s, float t, uint32_t prim, float bu, float bv, float *rgb)
{
    int hit = isfinite(t) && t > 1e-4f;
    if (!hit) { rgb[0] = rgb[1] = rgb[2] = 0.f; return; }
    float u = bu < 0.f ? 0.f : (bu > 1.f ? 1.f : bu), v = bv < 0.f ? 0.f : (bv > 1.f ? 1.f : bv);
    const float *tu = s->tri_uvs + 6 * (size_t)prim;
    /* numpy: (1 - u - v) evaluated in float32 (python int 1 is weak), products promote to float64 */
    float w = 1.0f - u - v;
    double gx = (double)tu[0] * (double)w + (double)tu[2] * (double)u + (double)tu[4] * (double)v;
    double gy = (double)tu[1] * (double)w + (double)tu[3] * (double)u + (double)tu[5] * (double)v;
    float fx = (float)gx, fy = (float)gy;
    fx = fx * 2.f - 1.f;
    fy = -(1.f - fy * 2.f);
    /* grid_sample unnormalize, align_corners=False: ((g+1)*size-1)/2, then clip to [0,size-1] */
    float x = ((fx + 1.f) * (float)s->Wt - 1.f) * 0.5f, y = ((fy + 1.f) * (float)s->Ht - 1.f) * 0.5f;
    x = fminf(fmaxf(x, 0.f), (float)(s->Wt - 1)); y = fminf(fmaxf(y, 0.f), (float)(s->Ht - 1));
    float x0f = floorf(x), y0f = floorf(y);
    int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    float wx1 = x - x0f, wx0 = 1.f - wx1, wy1 = y - y0f, wy0 = 1.f - wy1; /* tw = x_e - x, etc. */
    for (int c = 0; c < 3; c++) {
        float acc = 0.f;
        const float *h = s->hdr;
        size_t W = (size_t)s->Wt;
        acc += h[((size_t)y0 * W + x0) * 3 + c] * (wx0 * wy0);
        if (x1 < s->Wt) acc += h[((size_t)y0 * W + x1) * 3 + c] * (wx1 * wy0);
        if (y1 < s->Ht) acc += h[((size_t)y1 * W + x0) * 3 + c] * (wx0 * wy1);
        if (x1 < s->Wt && y1 < s->Ht) acc += h[((size_t)y1 * W + x1) * 3 + c] * (wx1 * wy1);
        rgb[c] = acc;
    }
}

TXO_API void txo_shade_hits(const TxoScene *s, const float *t_hit, const uint32_t *prim_id, const float *prim_uv,
                            int64_t R, float *radiance)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; r++) {
        uint32_t p = prim_id[r];
        int hit = isfinite(t_hit[r]) && t_hit[r] > 1e-4f;
        if (!hit) p = 0;
        if (hit && p >= (uint32_t)s->T) { radiance[3 * r] = radiance[3 * r + 1] = radiance[3 * r + 2] = 0.f; continue; }
        shade_one(s, t_hit[r], p, prim_uv[2 * r], prim_uv[2 * r + 1], radiance + 3 * r);
    }
}

/* query_irf: cast + shade.  tracer: 0 = brute force f64, 1 = canonical BVH2 f32 */
TXO_API void txo_trace_shade(const TxoScene *s, const float *org, const float *dir, int64_t R, int tracer,
                             float *radiance, uint64_t *counters)
{
    if (tracer == 0) {
        float *t = (float *)malloc(sizeof(float) * R); uint32_t *p = (uint32_t *)malloc(sizeof(uint32_t) * R);
        float *uv = (float *)malloc(sizeof(float) * 2 * R);
        txo_cast_rays_bruteforce(s, org, dir, R, t, p, uv);
        txo_shade_hits(s, t, p, uv, R, radiance);
        free(t); free(p); free(uv);
        return;
    }
    uint64_t tn = 0, tt = 0, th = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tn, tt, th)
    for (int64_t r = 0; r < R; r++) {
        uint64_t cn = 0, ct = 0; float t, u, v; uint32_t p;
        cast_one_bvh(s, org + 3 * r, dir + 3 * r, &t, &p, &u, &v, &cn, &ct);
        int hit = isfinite(t) && t > 1e-4f;
        shade_one(s, t, hit ? p : 0, u, v, radiance + 3 * r);
        tn += cn; tt += ct; th += hit;
    }
    if (counters) { counters[0] += tn; counters[1] += tt; counters[2] += (uint64_t)R; counters[3] += th; }
}

/* ------------------------------------------------------------------------------------------ */
/* sampling (utils/sample_util.py)                                                            */
/* ------------------------------------------------------------------------------------------ */
static uint32_t bitrev32(uint32_t b)
{
    b = (b << 16) | (b >> 16);
    b = ((b & 0x55555555u) << 1) | ((b & 0xAAAAAAAAu) >> 1);
    b = ((b & 0x33333333u) << 2) | ((b & 0xCCCCCCCCu) >> 2);
    b = ((b & 0x0F0F0F0Fu) << 4) | ((b & 0xF0F0F0F0u) >> 4);
    b = ((b & 0x00FF00FFu) << 8) | ((b & 0xFF00FF00u) >> 8);
    return b;
}

/* sample_util.py:28-41, stored into np.float32 (:94-98) */
TXO_API void txo_hammersley(int N, float *out /* [N,2] */)
{
    for (int i = 0; i < N; i++) {
        out[2 * i] = (float)((double)i / (double)N);
        out[2 * i + 1] = (float)((double)bitrev32((uint32_t)i) * 2.3283064365386963e-10);
    }
}

#define TXO_EPS6 1e-6f

static inline float shift_wrap_clamp(float s, float shift)
{
    s = s + shift;                 /* sample_util.py:103 */
    if (s > 1.f) s = s - 1.f;      /* :104-105, strict > */
    if (s < 0.f) s = s + 1.f;      /* :106-107 */
    const float lo = 0.f + TXO_EPS6, hi = (float)(1.0 - 1e-6);
    return fminf(fmaxf(s, lo), hi);/* :108 */
}

static inline void frame_from_normal(const float *n, float *nh, float *U, float *V)
{
    /* sample_util.py:84-91 : axis choice on the RAW normal, normalisation x/(|x|+1e-6) */
    float xa[3] = {1.f, 0.f, 0.f};
    if (fabsf(n[0]) > 0.99f) { xa[0] = 0.f; xa[1] = 1.f; }
    float ln = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) + TXO_EPS6;
    for (int a = 0; a < 3; a++) nh[a] = n[a] / ln;
    float c[3] = {xa[1] * nh[2] - xa[2] * nh[1], xa[2] * nh[0] - xa[0] * nh[2], xa[0] * nh[1] - xa[1] * nh[0]};
    float lc = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + TXO_EPS6;
    for (int a = 0; a < 3; a++) U[a] = c[a] / lc;
    float e[3] = {nh[1] * U[2] - nh[2] * U[1], nh[2] * U[0] - nh[0] * U[2], nh[0] * U[1] - nh[1] * U[0]};
    float le = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + TXO_EPS6;
    for (int a = 0; a < 3; a++) V[a] = e[a] / le;
}

/* mode: 0 uniform, 1 cosine, 2 importance(GGX).  sample_util.py:113-143 */
static inline void sample_dir(int mode, float s0, float s1, float rough, const float *nh, const float *U, const float *V, float *L)
{
    const float two_pi = (float)(2 * 3.141592653589793), pi = (float)3.141592653589793;
    float phi = two_pi * s1 - pi;
    float ct, st;
    if (mode == 0) { ct = 1.0f - s0; st = sqrtf(1.0f - ct * ct); }
    else if (mode == 1) { ct = sqrtf(1.0f - s0); st = sqrtf(1.0f - ct * ct); }
    else {
        float a = rough * rough;
        ct = sqrtf((1.0f - s0) / (1.0f + (a * a - 1.f) * s0));
        const float lo = -1.0f + TXO_EPS6, hi = 1.0f - TXO_EPS6;
        ct = fminf(fmaxf(ct, lo), hi);
        st = fminf(fmaxf(sqrtf(1.0f - ct * ct), lo), hi);
    }
    float sp = sinf(phi) * st, cp = -(cosf(phi) * st);
    for (int a = 0; a < 3; a++) L[a] = V[a] * sp + nh[a] * ct + U[a] * cp;
}

TXO_API void txo_generate_dir(const float *normals, int b, int N, int mode, const float *roughness,
                              const float *shift /* [b,2] */, float *L /* [b,N,3] */)
{
    float *ham = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    txo_hammersley(N, ham);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < b; p++) {
        float nh[3], U[3], V[3];
        frame_from_normal(normals + 3 * (size_t)p, nh, U, V);
        float r = roughness ? roughness[p] : 0.f;
        for (int i = 0; i < N; i++) {
            float s0 = shift_wrap_clamp(ham[2 * i], shift[2 * (size_t)p]);
            float s1 = shift_wrap_clamp(ham[2 * i + 1], shift[2 * (size_t)p + 1]);
            sample_dir(mode, s0, s1, r, nh, U, V, L + 3 * ((size_t)p * N + i));
        }
    }
    free(ham);
}

/* ------------------------------------------------------------------------------------------ */
/* IrT estimator  (models/tracer_o3d_irt.py:156-178)                                          */
/* E_p = (2*pi/N) * sum_i L(o_p, d_i) * clamp(n_p . d_i, 0, 1)   with the RAW n_p             */
/* ------------------------------------------------------------------------------------------ */
TXO_API void txo_irt_generate(const TxoScene *s, const float *pos, const float *nrm, const uint8_t *valid,
                              const float *shift, int64_t Nt, int N, int mode, int tracer,
                              float *irr /* [Nt,3] */, uint64_t *counters)
{
    float *ham = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    txo_hammersley(N, ham);
    uint64_t tn = 0, tt = 0, tr = 0, th = 0;
    const float pi = (float)3.141592653589793;
    /* mode | 4: the cosine branch of diffuse_reflectance (models/mat_nvdiffrast.py:256-257): sum L * pi / N, no n.l factor */
    const int cosw = (mode & 4) != 0;
    mode &= 3;
#pragma omp parallel
    {
        float *dirs = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *orgs = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *rad = (float *)malloc(sizeof(float) * 3 * (size_t)N);
        float *th_ = (float *)malloc(sizeof(float) * (size_t)N);
        uint32_t *pid = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)N);
        float *puv = (float *)malloc(sizeof(float) * 2 * (size_t)N);
#pragma omp for schedule(dynamic, 1) reduction(+ : tn, tt, tr, th)
        for (int64_t p = 0; p < Nt; p++) {
            if (valid && !valid[p]) { irr[3 * p] = irr[3 * p + 1] = irr[3 * p + 2] = 0.f; continue; }
            const float *n = nrm + 3 * p;
            float nh[3], U[3], V[3];
            frame_from_normal(n, nh, U, V);
            for (int i = 0; i < N; i++) {
                float s0 = shift_wrap_clamp(ham[2 * i], shift[2 * p]);
                float s1 = shift_wrap_clamp(ham[2 * i + 1], shift[2 * p + 1]);
                sample_dir(mode, s0, s1, 0.f, nh, U, V, dirs + 3 * i);
                orgs[3 * i] = pos[3 * p]; orgs[3 * i + 1] = pos[3 * p + 1]; orgs[3 * i + 2] = pos[3 * p + 2];
            }
            if (tracer == 0) {
                /* nested parallel region is inactive: runs serially inside this thread */
                txo_cast_rays_bruteforce(s, orgs, dirs, N, th_, pid, puv);
                for (int i = 0; i < N; i++) {
                    int hit = isfinite(th_[i]) && th_[i] > 1e-4f;
                    shade_one(s, th_[i], hit ? pid[i] : 0, puv[2 * i], puv[2 * i + 1], rad + 3 * i);
                }
            } else {
                for (int i = 0; i < N; i++) {
                    uint64_t cn = 0, ct = 0; float t, u, v; uint32_t q;
                    cast_one_bvh(s, orgs + 3 * i, dirs + 3 * i, &t, &q, &u, &v, &cn, &ct);
                    int hit = isfinite(t) && t > 1e-4f;
                    shade_one(s, t, hit ? q : 0, u, v, rad + 3 * i);
                    tn += cn; tt += ct; th += hit;
                }
                tr += (uint64_t)N;
            }
            double acc[3] = {0, 0, 0};
            for (int i = 0; i < N; i++) {
                float ndl = n[0] * dirs[3 * i] + n[1] * dirs[3 * i + 1] + n[2] * dirs[3 * i + 2];
                ndl = cosw ? 1.f : fminf(fmaxf(ndl, 0.f), 1.f);
                for (int c = 0; c < 3; c++) acc[c] += (double)(rad[3 * i + c] * ndl);
            }
            for (int c = 0; c < 3; c++) irr[3 * p + c] = (((float)acc[c] * (cosw ? 1.f : 2.f)) * pi) / (float)N;
        }
        free(dirs); free(orgs); free(rad); free(th_); free(pid); free(puv);
    }
    free(ham);
    if (counters) { counters[0] += tn; counters[1] += tt; counters[2] += tr; counters[3] += th; }
}

/* ------------------------------------------------------------------------------------------ */
/* Material pixel forward (models/mat_nvdiffrast.py:201-249, 260-279)                          */
/* rgb = irr*albedo/pi + (1/S) sum_i Ls_i * w_i                                                */
/* ------------------------------------------------------------------------------------------ */
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

TXO_API void txo_spec_forward(const TxoScene *s, const float *normal, const float *albedo, const float *rough,
                              const float *points, const float *irr, const float *cam, const float *shift,
                              int64_t P, int S, int tracer, float *rgb /* [P,3] */, float *Ls_out /* [P,S,3] or NULL */,
                              const float *Ls_in /* [P,S,3] or NULL: the `lighting` argument of specular_reflectance -- nothing is traced */,
                              float clamp_eps /* 1e-14 mat_nvdiffrast.py:270-279; 1e-6 test_nvdiffrast.py:320-333 */)
{
    float *ham = (float *)malloc(sizeof(float) * 2 * (size_t)S);
    txo_hammersley(S, ham);
    const float pi = (float)3.141592653589793, eps14 = clamp_eps;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t p = 0; p < P; p++) {
        const float *n = normal + 3 * p, *pt = points + 3 * p;
        float r = rough[p];
        float vv[3] = {cam[0] - pt[0], cam[1] - pt[1], cam[2] - pt[2]};
        float lv = fmaxf(sqrtf(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]), 1e-4f); /* F.normalize eps=1e-4 */
        for (int a = 0; a < 3; a++) vv[a] /= lv;
        float nh[3], U[3], V[3];
        frame_from_normal(n, nh, U, V);
        float spec[3] = {0, 0, 0};
        float ndv = clamp01(n[0] * vv[0] + n[1] * vv[1] + n[2] * vv[2]);
        float k = (r + 1.f) * (r + 1.f) / 8.f;
        for (int i = 0; i < S; i++) {
            float s0 = shift_wrap_clamp(ham[2 * i], shift[2 * p]);
            float s1 = shift_wrap_clamp(ham[2 * i + 1], shift[2 * p + 1]);
            float h[3], l[3];
            sample_dir(2, s0, s1, r, nh, U, V, h);
            float vdh = clamp01(h[0] * vv[0] + h[1] * vv[1] + h[2] * vv[2]);
            for (int a = 0; a < 3; a++) l[a] = 2.f * vdh * h[a] - vv[a];
            float Ls[3];
            if (Ls_in) memcpy(Ls, Ls_in + 3 * ((size_t)p * S + i), sizeof(float) * 3);
            else txo_trace_shade(s, pt, l, 1, tracer, Ls, NULL);
            if (Ls_out) memcpy(Ls_out + 3 * ((size_t)p * S + i), Ls, sizeof(float) * 3);
            float ndl = clamp01(n[0] * l[0] + n[1] * l[1] + n[2] * l[2]);
            float ndh = clamp01(n[0] * h[0] + n[1] * h[1] + n[2] * h[2]);
            float f = 0.04f + 0.96f * powf(2.0f, (-5.55472f * vdh - 6.98316f) * vdh);
            float g1v = ndv / fmaxf(ndv * (1.f - k) + k, eps14);
            float g1l = ndl / fmaxf(ndl * (1.f - k) + k, eps14);
            float g = g1l * g1v;
            float brdf = f * g / fmaxf(4.f * ndl * ndv, eps14);
            float w = brdf * ndl * 4.f * vdh / fmaxf(ndh, eps14);
            for (int c = 0; c < 3; c++) spec[c] += Ls[c] * w;
        }
        for (int c = 0; c < 3; c++) rgb[3 * p + c] = irr[3 * p + c] * albedo[3 * p + c] / pi + spec[c] / (float)S;
    }
    free(ham);
}

/* number of OpenMP threads the following calls use (PyTorch's torch.set_num_threads() changes the process-wide default) */
TXO_API void txo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

TXO_API int txo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
