// Binned-SAH BVH2 builder (host, multi-threaded over the top subtrees) emitting the 64-byte
// two-child-box node layout consumed by the gfx950 traversal kernels (see bvh_build.h).
#include <stdexcept>
#include <string>
#include <cstdio>
#include <cstdlib>
#include "bvh_build.h"
#include "env.h"
#include <cstdlib>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <future>
#include <memory>
#include <thread>

namespace texir {
namespace {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; a++) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; } }
    void grow(const Box& b) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Tmp {
    Box box;
    int first = 0, count = 0;           // leaf payload: a run of `order`
    int slot_first = 0, slot_count = 0; // ... as leaf-order slots (assign_slots)
    int rec_first = 0, rec_count = 0;   // ... as quad records (TEXIR_QUAD)
    std::unique_ptr<Tmp> l, r;
    int depth_below = 0;
};

struct Ctx {
    const std::vector<Box>* tb;
    const std::vector<float>* cen;      // [T,3]
    std::vector<int32_t>* order;
};

constexpr int kBins = 32;

// triangles per leaf (leaf codes hold count-1 in 3 bits); TEXIR_MAX_LEAF overrides the default for A/B measurements
static int max_leaf()
{
    const int v = env().max_leaf ? env().max_leaf : kMaxLeaf;
#if TEXIR_QUAD
    // quad records own two leaf-order slots each: a leaf of n unpaired triangles spans up to 2 n - 1 slots, and the binary tree's leaf code has 3 bits
    // for (slots - 1) -- more than 4 triangles per leaf would spill into the first-slot field (the deep-tree fallback and TEXIR_BVH_WIDTH=2 read it)
    return std::min(v, 4);
#else
    return v;
#endif
}

std::unique_ptr<Tmp> build(const Ctx& c, int first, int count, int depth, int par_levels)
{
    std::unique_ptr<Tmp> n(new Tmp);
    Box cb; cb.reset(); n->box.reset();
    auto& ord = *c.order;
    for (int i = first; i < first + count; i++) {
        int p = ord[i];
        n->box.grow((*c.tb)[p]);
        cb.grow(&(*c.cen)[3 * (size_t)p]);
    }
    if (count <= max_leaf()) { n->first = first; n->count = count; return n; }

    int mid = -1;
    // depth guard: a balanced split always terminates within ceil(log2(count)) further levels
    bool force_median = depth + (int)std::ceil(std::log2((double)count)) >= kMaxDepth - 2;
    if (!force_median) {
        int best_axis = -1, best_split = -1; float best_cost = FLT_MAX;
        for (int a = 0; a < 3; a++) {
            float ext = cb.mx[a] - cb.mn[a];
            if (!(ext > 0.f)) continue;
            float scale = (float)kBins / ext;
            int cnt[kBins]; Box bb[kBins];
            for (int b = 0; b < kBins; b++) { cnt[b] = 0; bb[b].reset(); }
            for (int i = first; i < first + count; i++) {
                int p = ord[i];
                int b = std::min(kBins - 1, std::max(0, (int)(((*c.cen)[3 * (size_t)p + a] - cb.mn[a]) * scale)));
                cnt[b]++; bb[b].grow((*c.tb)[p]);
            }
            float la[kBins]; int lc[kBins];
            Box acc; acc.reset(); int k = 0;
            for (int b = 0; b < kBins - 1; b++) { k += cnt[b]; if (cnt[b]) acc.grow(bb[b]); lc[b] = k; la[b] = k ? acc.half_area() : 0.f; }
            acc.reset(); k = 0;
            for (int b = kBins - 1; b > 0; b--) {
                k += cnt[b]; if (cnt[b]) acc.grow(bb[b]);
                int s = b - 1;
                if (lc[s] == 0 || k == 0) continue;
                float cost = la[s] * (float)lc[s] + acc.half_area() * (float)k;
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = s; }
            }
        }
        if (best_axis >= 0) {
            float scale = (float)kBins / (cb.mx[best_axis] - cb.mn[best_axis]);
            float base = cb.mn[best_axis];
            auto it = std::partition(ord.begin() + first, ord.begin() + first + count, [&](int p) {
                int b = std::min(kBins - 1, std::max(0, (int)(((*c.cen)[3 * (size_t)p + best_axis] - base) * scale)));
                return b <= best_split;
            });
            mid = (int)(it - ord.begin());
        }
    }
    if (mid <= first || mid >= first + count) {
        // median split along the widest centroid axis
        int a = 0; float e = -1.f;
        for (int k = 0; k < 3; k++) { float x = cb.mx[k] - cb.mn[k]; if (x > e) { e = x; a = k; } }
        mid = first + count / 2;
        std::nth_element(ord.begin() + first, ord.begin() + mid, ord.begin() + first + count,
                         [&](int p, int q) { return (*c.cen)[3 * (size_t)p + a] < (*c.cen)[3 * (size_t)q + a]; });
    }
    if (par_levels > 0 && count > 50000) {
        auto fut = std::async(std::launch::async, [&, first, mid, depth, par_levels]() { return build(c, first, mid - first, depth + 1, par_levels - 1); });
        n->r = build(c, mid, first + count - mid, depth + 1, par_levels - 1);
        n->l = fut.get();
    } else {
        n->l = build(c, first, mid - first, depth + 1, 0);
        n->r = build(c, mid, first + count - mid, depth + 1, 0);
    }
    n->depth_below = 1 + std::max(n->l->depth_below, n->r->depth_below);
    return n;
}

// leaf codes (bvh_build.h): the binary tree names leaf-order slots, the 4-wide tree names quad records when the library has them
inline int32_t leaf_code(const Tmp* t)
{
    // (cannot fire with TEXIR_MAX_LEAF <= 8; thrown -- texir_scene_create turns it into TEXIR_ERR_INVALID -- rather than aborting the host process)
    if (t->slot_count < 1 || t->slot_count > 8) throw std::runtime_error("texir bvh: leaf of " + std::to_string(t->slot_count) + " slots does not fit its 3-bit count");
    return ~(int32_t)(((uint32_t)t->slot_first << 3) | (uint32_t)(t->slot_count - 1));
}
inline int32_t leaf_code4(const Tmp* t)
{
#if TEXIR_QUAD
    return ~(int32_t)(((uint32_t)t->rec_first << 3) | (uint32_t)(t->rec_count - 1));
#else
    return leaf_code(t);
#endif
}

void put_box(GpuNode& g, int slot, const Box& b)
{
    float* xy = slot == 0 ? g.n0 : g.n1;
    xy[0] = b.mn[0]; xy[1] = b.mx[0]; xy[2] = b.mn[1]; xy[3] = b.mx[1];
    g.n2[2 * slot] = b.mn[2]; g.n2[2 * slot + 1] = b.mx[2];
}

// depth-first emission: the first child of every inner node directly follows it in memory
int32_t emit(const Tmp* t, std::vector<GpuNode>& out)
{
    int32_t idx = (int32_t)out.size();
    out.emplace_back();
    const Tmp* ch[2] = {t->l.get(), t->r.get()};
    int32_t code[2];
    for (int s = 0; s < 2; s++) {
        put_box(out[idx], s, ch[s]->box);
        code[s] = ch[s]->count ? leaf_code(ch[s]) : emit(ch[s], out);
    }
    out[idx].c[0] = code[0]; out[idx].c[1] = code[1]; out[idx].c[2] = out[idx].c[3] = 0;
    return idx;
}

// ---- 4-wide collapse: repeatedly open the inner child with the largest surface area until four children ----
struct Quant { uint8_t lo[3], hi[3]; };

// Child boxes are stored with a small ABSOLUTE slack on every side: the traversal's slab test runs in float32 with a folded
// origin (t = q * (cell / d) + (origin / d - o / d)); its rounding error, a few ulps of the coordinates involved (<= ~4 * 2^-24 * M
// in space, M = largest |coordinate| of the scene), would otherwise cull a box the ray only grazes -- and with it the triangle
// behind a shared edge or vertex that lies exactly on the box's face (flat, axis-aligned geometry: every ray that hits it).
// slack = 2^-19 * M is 8x that bound and ~1/50 of a leaf-level cell: it moves a quantised plane by one cell only when the true
// plane happens to sit within `slack` of a cell boundary (a few percent of the planes) -- no measurable extra node visits.
// (the slack and the output vectors travel in Emit4Ctx: texir_scene_create may run on several host threads at once)
struct Emit4Ctx { float slack; std::vector<GpuNode4>* out; std::vector<GpuNode4F>* out4f; int32_t dummy_leaf; int max_depth; };

void emit4_fill(GpuNode4& g, const Box& nb, const Tmp* const* kids, int nk, const int32_t* codes, int32_t dummy_leaf, float slack_f)
{
    const double slack = (double)slack_f;
    float scale[3];
    for (int a = 0; a < 3; a++) {
        const double ext = (double)nb.mx[a] - (double)nb.mn[a] + 2.0 * slack;
        int ex = 1;
        if (ext > 0.0) { ex = 127 + (int)std::ceil(std::log2(ext / 255.0)); if (ex < 1) ex = 1; if (ex > 254) ex = 254; }
        const float org = std::nextafter((float)((double)nb.mn[a] - slack), -FLT_MAX);
        // make sure 255 cells really cover the padded extent in float arithmetic
        for (;;) { uint32_t bits = (uint32_t)ex << 23; float sc; std::memcpy(&sc, &bits, 4); if ((double)org + 255.0 * (double)sc >= (double)nb.mx[a] + slack || ex >= 254) { scale[a] = sc; break; } ex++; }
        g.origin[a] = org;
    }
    g.cell_x = scale[0]; g.cell_y = scale[1]; g.cell_z = scale[2];
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int k = 0; k < 4; k++) {
        uint32_t ql[3] = {255, 255, 255}, qh[3] = {0, 0, 0};          // unused slot: inverted box
        if (k < nk) {
            for (int a = 0; a < 3; a++) {
                const float org = g.origin[a];
                const double mn = (double)kids[k]->box.mn[a] - slack, mx = (double)kids[k]->box.mx[a] + slack;
                double l = std::floor((mn - (double)org) / (double)scale[a]);
                double h = std::ceil((mx - (double)org) / (double)scale[a]);
                int li = (int)std::max(0.0, std::min(255.0, l)), hi_ = (int)std::max(0.0, std::min(255.0, h));
                // conservative in FLOAT decode: origin + q*scale must bracket the padded child box
                while (li > 0 && (double)(org + (float)li * scale[a]) > mn) li--;
                while (hi_ < 255 && (double)(org + (float)hi_ * scale[a]) < mx) hi_++;
                ql[a] = (uint32_t)li; qh[a] = (uint32_t)hi_;
            }
        }
        for (int a = 0; a < 3; a++) { lo[a] |= ql[a] << (8 * k); hi[a] |= qh[a] << (8 * k); }
        g.c[k] = k < nk ? codes[k] : dummy_leaf;
    }
    g.lox = lo[0]; g.loy = lo[1]; g.loz = lo[2]; g.hix = hi[0]; g.hiy = hi[1]; g.hiz = hi[2];
}

static void emit4f_fill(GpuNode4F& g, const Tmp* const* kids, int nk, const int32_t* codes, int32_t dummy_leaf, float slack)
{
    for (int k = 0; k < 4; k++) {
        for (int a = 0; a < 3; a++) {
            // unused slot: inverted box.  Used slot: the child's box widened by the absolute slack, rounded outwards
            g.plane[2 * a][k] = k < nk ? std::nextafter((float)((double)kids[k]->box.mn[a] - (double)slack), -FLT_MAX) : 1e30f;
            g.plane[2 * a + 1][k] = k < nk ? std::nextafter((float)((double)kids[k]->box.mx[a] + (double)slack), FLT_MAX) : -1e30f;
        }
        g.c[k] = k < nk ? codes[k] : dummy_leaf;
        g.pad[k] = 0;
    }
}

int32_t emit4(const Tmp* t, Emit4Ctx& cx, int depth)
{
    const int32_t idx = (int32_t)cx.out->size();
    cx.out->emplace_back();
    cx.out4f->emplace_back();
    if (depth > cx.max_depth) cx.max_depth = depth;
    const Tmp* kids[4]; int nk = 0;
    if (t->count) { kids[nk++] = t; }                      // degenerate root leaf
    else { kids[nk++] = t->l.get(); kids[nk++] = t->r.get(); }
    while (nk < 4) {
        int best = -1; float ba = -1.f;
        for (int k = 0; k < nk; k++) if (!kids[k]->count) { float a = kids[k]->box.half_area(); if (a > ba) { ba = a; best = k; } }
        if (best < 0) break;
        const Tmp* o = kids[best];
        kids[best] = o->l.get(); kids[nk++] = o->r.get();
    }
    int32_t codes[4];
    for (int k = 0; k < nk; k++) codes[k] = kids[k]->count ? leaf_code4(kids[k]) : emit4(kids[k], cx, depth + 1);
    emit4_fill((*cx.out)[idx], t->box, kids, nk, codes, cx.dummy_leaf, cx.slack);
    emit4f_fill((*cx.out4f)[idx], kids, nk, codes, cx.dummy_leaf, cx.slack);
    return idx;
}

// Memory order of the 4-wide tree (TEXIR_BVH_LAYOUT; the quantised and the float form stay index for index):
//   0  depth-first as emitted: a node's FIRST child follows it, the other children lie a whole subtree away
//   1  sibling blocks: the inner children of a node occupy consecutive slots (up to 4 x 64 B = two 128-byte lines), blocks in depth-first order
//   2 / 3  treelets of 2 / 3 levels below their root, breadth-first inside (1 + 4 + 16 [+ 64] nodes contiguous), treelets in depth-first order
// Only indices change: every ray visits the same nodes in the same order and finds the same hit.
static void relayout4(std::vector<GpuNode4>& n4, std::vector<GpuNode4F>& n4f, int mode)
{
    const size_t n = n4.size();
    if (mode <= 0 || n < 2) return;
    std::vector<int32_t> order; order.reserve(n);
    if (mode == 1) {
        std::vector<int32_t> stack{0};
        order.push_back(0);
        while (!stack.empty()) {
            const int32_t v = stack.back(); stack.pop_back();
            int32_t kids[4]; int nk = 0;
            for (int k = 0; k < 4; k++) if (n4[v].c[k] >= 0) kids[nk++] = n4[v].c[k];
            for (int k = 0; k < nk; k++) order.push_back(kids[k]);
            for (int k = nk - 1; k >= 0; k--) stack.push_back(kids[k]);
        }
    } else {
        const int levels = mode == 2 ? 2 : 3;
        std::vector<int32_t> roots{0};
        std::vector<int32_t> cur, nxt;
        while (!roots.empty()) {
            const int32_t r = roots.back(); roots.pop_back();
            cur.assign(1, r);
            order.push_back(r);
            for (int l = 0; l <= levels; l++) {
                nxt.clear();
                for (int32_t v : cur) for (int k = 0; k < 4; k++) if (n4[v].c[k] >= 0) nxt.push_back(n4[v].c[k]);
                if (l < levels) for (int32_t v : nxt) order.push_back(v);
                else for (auto it = nxt.rbegin(); it != nxt.rend(); ++it) roots.push_back(*it);        // next treelets, first child's first
                cur.swap(nxt);
            }
        }
    }
    if (order.size() != n) return;                    // (cannot happen: every inner node is reachable exactly once)
    std::vector<int32_t> where(n);
    for (size_t i = 0; i < n; i++) where[(size_t)order[i]] = (int32_t)i;
    std::vector<GpuNode4> a(n); std::vector<GpuNode4F> b(n4f.size());
    for (size_t i = 0; i < n; i++) {
        a[i] = n4[(size_t)order[i]];
        for (int k = 0; k < 4; k++) if (a[i].c[k] >= 0) a[i].c[k] = where[(size_t)a[i].c[k]];
        if (!n4f.empty()) { b[i] = n4f[(size_t)order[i]]; for (int k = 0; k < 4; k++) b[i].c[k] = a[i].c[k]; }
    }
    n4.swap(a);
    if (!n4f.empty()) n4f.swap(b);
}

}  // namespace

void build_bvh(const float* verts, int V, const int32_t* tris, int T, const float* tri_uvs, BvhHost& out)
{
    std::vector<Box> tb((size_t)T);
    std::vector<float> cen(3 * (size_t)T);
    std::vector<int32_t> order((size_t)T);
    for (int p = 0; p < T; p++) {
        tb[p].reset();
        for (int k = 0; k < 3; k++) tb[p].grow(verts + 3 * (size_t)tris[3 * (size_t)p + k]);
        for (int a = 0; a < 3; a++) cen[3 * (size_t)p + a] = 0.5f * (tb[p].mn[a] + tb[p].mx[a]);
        order[p] = p;
    }
    float slack = 0.f;
    {   // absolute box slack of this scene (see emit4_fill); TEXIR_BOX_SLACK_LOG2 overrides the exponent for A/B runs (99 = none)
        float M = 0.f;
        for (int64_t i = 0; i < 3 * (int64_t)V; i++) M = std::max(M, std::fabs(verts[i]));
        const int l2 = env().box_slack_log2;
        slack = l2 == 99 ? 0.f : std::ldexp(M, l2);
    }
    Ctx c{&tb, &cen, &order};
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int par_levels = 0;
    while ((1u << par_levels) < hw && par_levels < 5) par_levels++;
    std::unique_ptr<Tmp> root = build(c, 0, T, 0, par_levels);

    // ---- leaf-order slots and quad records (bvh_build.h) ----
    struct Rec { int p0, r0, p1, r1; };                    // triangle, rotation; p1 < 0: no partner
    std::vector<Rec> recs;
    recs.reserve((size_t)T);
    {
        // two triangles pair up when they share an edge that they walk in OPPOSITE directions (a consistently wound mesh): both can then be stored as
        // rotations of themselves, triangle 0 = (q0, q1, q2), triangle 1 = (q3, q2, q1)
        auto vtx = [&](int p, int k) { return verts + 3 * (size_t)tris[3 * (size_t)p + k]; };
        auto same = [&](const float* a, const float* b) { return std::memcmp(a, b, 3 * sizeof(float)) == 0; };
        auto pair_up = [&](int p, int q, int& r0, int& r1) {
#if TEXIR_QUAD
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
                if (same(vtx(p, i), vtx(q, (j + 1) % 3)) && same(vtx(p, (i + 1) % 3), vtx(q, j))) {
                    // degenerate partners (a repeated vertex) stay single: their "shared edge" is not one
                    if (same(vtx(p, i), vtx(p, (i + 1) % 3)) || same(vtx(p, (i + 2) % 3), vtx(q, (j + 2) % 3))) return false;
#if TEXIR_UV_QUAD
                    // one uv per record corner: a uv seam along the shared edge keeps the two triangles single
                    if (tri_uvs) {
                        auto cuv = [&](int t_, int k) { return tri_uvs + 6 * (size_t)t_ + 2 * (size_t)k; };
                        if (std::memcmp(cuv(p, i), cuv(q, (j + 1) % 3), 8) || std::memcmp(cuv(p, (i + 1) % 3), cuv(q, j), 8)) return false;
                    }
#endif
                    r0 = (i + 2) % 3; r1 = (j + 2) % 3; return true;
                }
#endif
            (void)p; (void)q; (void)r0; (void)r1;
            return false;
        };
        std::vector<Tmp*> stack{root.get()};
        while (!stack.empty()) {                            // leaves in emission order (left to right = increasing `first`)
            Tmp* t = stack.back(); stack.pop_back();
            if (!t->count) { stack.push_back(t->r.get()); stack.push_back(t->l.get()); continue; }
            t->rec_first = (int)recs.size();
            bool last_single = false;
            // (leaves of more than two triangles, TEXIR_MAX_LEAF: bring partners next to each other first -- the order inside a leaf is free)
            if (TEXIR_QUAD && t->count > 2)
                for (int i = t->first; i + 1 < t->first + t->count; i++) {
                    int r0, r1, j = i + 1;
                    while (j < t->first + t->count && !pair_up(order[i], order[j], r0, r1)) j++;
                    if (j < t->first + t->count) { std::swap(order[i + 1], order[j]); i++; }
                }
            for (int i = t->first; i < t->first + t->count;) {
                int r0 = 0, r1 = 0;
                if (TEXIR_QUAD && i + 1 < t->first + t->count && pair_up(order[i], order[i + 1], r0, r1)) { recs.push_back(Rec{order[i], r0, order[i + 1], r1}); i += 2; last_single = false; }
                else if (TEXIR_QUAD) { recs.push_back(Rec{order[i], 0, -1, 0}); i += 1; last_single = true; }
                else { recs.push_back(Rec{order[i], 0, -1, 0}); i += 1; }
            }
            t->rec_count = (int)recs.size() - t->rec_first;
#if TEXIR_QUAD
            t->slot_first = 2 * t->rec_first; t->slot_count = 2 * t->rec_count - (last_single ? 1 : 0);
#else
            (void)last_single;
            t->slot_first = t->rec_first; t->slot_count = t->rec_count;
#endif
        }
    }
    const size_t n_rec = recs.size();
    const size_t n_slots = TEXIR_QUAD ? 2 * n_rec : n_rec;
    out.n_slots = (int64_t)n_slots;

    out.nodes.clear();
    out.nodes.reserve((size_t)T);
    if (root->count) {
        // whole mesh fits one leaf: synthesise a root whose second child can never be entered
        GpuNode g; std::memset(&g, 0, sizeof(g));
        put_box(g, 0, root->box);
        // second slot: a point box far outside any scene; if a ray ever grazed it the (duplicate) leaf is harmless
        Box far; for (int a = 0; a < 3; a++) far.mn[a] = far.mx[a] = 1e30f;
        put_box(g, 1, far);
        g.c[0] = leaf_code(root.get()); g.c[1] = g.c[0];
        out.nodes.push_back(g);
        out.max_depth = 1;
    } else {
        emit(root.get(), out.nodes);
        out.max_depth = root->depth_below;
    }
    out.nodes4.clear();
    out.nodes4.reserve((size_t)T / 2 + 16);
    out.nodes4f.clear();
    out.nodes4f.reserve((size_t)T / 2 + 16);
    // the slot (quad libraries: the record) after the last one holds a degenerate all-zero triangle (record): the target of unused child slots
    Emit4Ctx cx{slack, &out.nodes4, &out.nodes4f, ~(int32_t)(((uint32_t)(TEXIR_QUAD ? n_rec : n_slots) << 3) | 0u), 0};
    emit4(root.get(), cx, 1);
    out.max_depth4 = cx.max_depth;
    relayout4(out.nodes4, out.nodes4f, env().bvh_layout);
    // slot data in STORED corner order: stored corner k = the caller's corner (rot + k) % 3
    out.tris.assign(n_slots + 1, GpuTri{});
    out.uvs.assign((TEXIR_UV_QUAD ? n_rec : n_slots) + 1, GpuTriUV{});
    for (auto& g : out.tris) g.prim = 0xFFFFFFFFu;          // holes (the odd slot of a single) and the dummy: degenerate, never hit
    auto put_slot = [&](size_t slot, int p, int rot) {
        GpuTri& g = out.tris[slot];
        const float* v[3];
        for (int k = 0; k < 3; k++) v[k] = verts + 3 * (size_t)tris[3 * (size_t)p + (rot + k) % 3];
#if TEXIR_TRI_WATERTIGHT
        for (int k = 0; k < 3; k++) { g.v0[k] = v[0][k]; g.e1[k] = v[1][k]; g.e2[k] = v[2][k]; }         // the vertices themselves, bit for bit
#else
        for (int k = 0; k < 3; k++) { g.v0[k] = v[0][k]; g.e1[k] = v[1][k] - v[0][k]; g.e2[k] = v[2][k] - v[0][k]; }
#endif
        g.prim = (uint32_t)p;
        const uint32_t r = (uint32_t)rot;
        std::memcpy(&g.pad1, &r, 4); g.pad2 = 0.f;
#if TEXIR_UV_QUAD
        // record form: the even slot writes its three stored corners (q0, q1, q2), the odd slot its first stored corner (q3; its other two are q2, q1)
        GpuTriUV& u = out.uvs[slot >> 1];
        const int n_put = (slot & 1) ? 1 : 3;
        for (int k = 0; k < n_put; k++) {
            const float* uv = tri_uvs + 6 * (size_t)p + 2 * (size_t)((rot + k) % 3);
            const int at = (slot & 1) ? 3 : k;
            u.uv[2 * at] = uv[0]; u.uv[2 * at + 1] = uv[1];
        }
#else
        GpuTriUV& u = out.uvs[slot];
        for (int k = 0; k < 3; k++) { const float* uv = tri_uvs + 6 * (size_t)p + 2 * (size_t)((rot + k) % 3); u.uv[2 * k] = uv[0]; u.uv[2 * k + 1] = uv[1]; }
        u.uv[6] = u.uv[7] = 0.f;
#endif
    };
#if TEXIR_QUAD
    out.quads.assign(n_rec + 1, GpuQuad{});
    for (size_t r = 0; r < n_rec; r++) {
        const Rec& rc = recs[r];
        put_slot(2 * r, rc.p0, rc.r0);
        if (rc.p1 >= 0) put_slot(2 * r + 1, rc.p1, rc.r1);
        GpuQuad& q = out.quads[r];
        const GpuTri& a = out.tris[2 * r];
        for (int k = 0; k < 3; k++) { q.q[k] = a.v0[k]; q.q[3 + k] = a.e1[k]; q.q[6 + k] = a.e2[k]; }
        // (q3: the partner's first stored corner -- its other two are q2, q1 by construction; a single repeats q2: a triangle without area)
        for (int k = 0; k < 3; k++) q.q[9 + k] = rc.p1 >= 0 ? out.tris[2 * r + 1].v0[k] : a.e2[k];
    }
#else
    for (size_t r = 0; r < n_rec; r++) put_slot(r, recs[r].p0, 0);
#endif
}

}  // namespace texir
