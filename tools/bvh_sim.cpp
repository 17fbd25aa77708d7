// Design-exploration tool (not product, not oracle): replays the IrT kernel's wave-level schedule on the CPU --
// 64 neighbouring texels per wave, one absolute direction cell per pass, lock-step while-while traversal of the product's
// own 4-wide quantised BVH (texir_code_amd/csrc/bvh_build.cpp is linked in) -- and counts what the GPU's issue-bound inner
// loops execute: per-lane node visits / triangle tests and WAVE-level node steps / triangle steps.  Variants of the
// traversal (pop culling with a stored entry distance, leaf sizes via TEXIR_MAX_LEAF, ...) can be compared here for free
// before they cost GPU minutes.
//
//   g++ -O2 -fopenmp -std=c++17 -I texir_code_amd/csrc tools/bvh_sim.cpp texir_code_amd/csrc/bvh_build.cpp -o /tmp/bvh_sim -lpthread
//   python tools/bvh_sim_dump.py c4 /tmp/sim_c4 && /tmp/bvh_sim /tmp/sim_c4 [groups] [passes]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.h"

using namespace texir;

template <typename T>
static std::vector<T> load(const std::string& p)
{
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(1);
    fclose(f);
    return v;
}

static inline uint32_t brev(uint32_t x)
{
    x = (x >> 16) | (x << 16); x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4); x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1); return x;
}
static float swc(float s, float sh) { s += sh; if (s > 1.f) s -= 1.f; if (s < 0.f) s += 1.f; return fminf(fmaxf(s, 1e-6f), (float)(1.0 - 1e-6)); }

struct Frame { float n[3], U[3], V[3]; };
static Frame make_frame(float nx, float ny, float nz)
{
    Frame f; float ax = 1.f, ay = 0.f; if (fabsf(nx) > 0.99f) { ax = 0.f; ay = 1.f; }
    float ln = sqrtf(nx * nx + ny * ny + nz * nz) + 1e-6f; f.n[0] = nx / ln; f.n[1] = ny / ln; f.n[2] = nz / ln;
    float c0 = ay * f.n[2], c1 = -ax * f.n[2], c2 = ax * f.n[1] - ay * f.n[0];
    float lc = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + 1e-6f; f.U[0] = c0 / lc; f.U[1] = c1 / lc; f.U[2] = c2 / lc;
    float e0 = f.n[1] * f.U[2] - f.n[2] * f.U[1], e1 = f.n[2] * f.U[0] - f.n[0] * f.U[2], e2 = f.n[0] * f.U[1] - f.n[1] * f.U[0];
    float le = sqrtf(e0 * e0 + e1 * e1 + e2 * e2) + 1e-6f; f.V[0] = e0 / le; f.V[1] = e1 / le; f.V[2] = e2 / le; return f;
}
static uint32_t cell_to_pass(uint32_t J, float sh0, float sh1, int log2N)
{
    int cells = log2N, bphi = (cells + 1) >> 1, bth = cells - bphi;
    uint32_t nphi = 1u << bphi, nth = 1u << bth, Jphi = J & (nphi - 1u), Jth = J >> bphi;
    uint32_t dphi = (uint32_t)(sh1 * (float)nphi + 0.5f), dth = (uint32_t)(sh0 * (float)nth + 0.5f);
    uint32_t phibin = (Jphi + nphi - (dphi & (nphi - 1u))) & (nphi - 1u), th = (Jth + nth - (dth & (nth - 1u))) & (nth - 1u);
    uint32_t low = bphi ? (brev(phibin) >> (32 - bphi)) : 0u;
    return (th << bphi) | low;
}
static uint32_t sample_index(uint32_t cell, int log2N)
{
    int cells = log2N, bphi = (cells + 1) >> 1, bth = cells - bphi;
    uint32_t low = cell & ((1u << bphi) - 1u), th = bth ? (cell >> bphi) : 0u;
    return (th << (log2N - bth)) | low;
}

struct Ray { float o[3], d[3], id[3], ood[3]; float t; int slot; int node; int pleaf = 0; std::vector<int> stk; std::vector<float> stk_t; };

struct Counters { double rays = 0, nodes = 0, tris = 0, wnode = 0, wtri = 0, culled = 0, hits = 0, wcull = 0, maxsp = 0, wuni = 0, wuni0 = 0, wdeep[4] = {0, 0, 0, 0}, lines = 0, wmaxn = 0, rounds = 0, wmaxt = 0, refills = 0, refill_lanes = 0, spread = 0, specpops = 0, un_nodes = 0, un_leaves = 0, dh[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, parked = 0, cpass = 0, cwnode = 0, cwtri = 0, cnodes = 0, ctris = 0, ah[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; };

static const int kSent = 0x7FFFFFFF;

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: bvh_sim <dir> [groups=64] [passes=64] [variant flags: cull cull8 nosort]\n"); return 1; }
    std::string dir = argv[1];
    int n_groups = argc > 2 ? atoi(argv[2]) : 64, n_pass = argc > 3 ? atoi(argv[3]) : 64;
    bool cull = false, cull8 = false, nosort = false, psort3 = false; int policy = 0; double alpha = 1.0, alpha2 = 1.0, tricost = 1.5; int refillK = 0; int spec = 0; int switch_after = 0; int parkK = 0, parkM = 0; int regB = 0, regKey = 0; bool consec = false; int texK = 64;
    for (int i = 4; i < argc; i++) { if (!strcmp(argv[i], "cull")) cull = true; if (!strcmp(argv[i], "cull8")) cull = cull8 = true; if (!strcmp(argv[i], "nosort")) nosort = true; if (!strcmp(argv[i], "psort3")) psort3 = true; if (!strncmp(argv[i], "after:", 6)) switch_after = atoi(argv[i] + 6); if (!strcmp(argv[i], "spec")) spec = 1; if (!strcmp(argv[i], "spec2")) spec = 2; if (!strncmp(argv[i], "refill:", 7)) refillK = atoi(argv[i] + 7); if (!strncmp(argv[i], "arr", 3)) { policy = 2; if (argv[i][3] == ':') { alpha = alpha2 = atof(argv[i] + 4); const char* c2 = strchr(argv[i] + 4, ','); if (c2) alpha2 = atof(c2 + 1); } } if (!strncmp(argv[i], "maj", 3)) { policy = 1; if (argv[i][3] == ':') { alpha = alpha2 = atof(argv[i] + 4); const char* c2 = strchr(argv[i] + 4, ','); if (c2) alpha2 = atof(c2 + 1); } } if (!strncmp(argv[i], "tricost:", 8)) tricost = atof(argv[i] + 8); if (!strcmp(argv[i], "consec")) consec = true; if (!strncmp(argv[i], "texk:", 5)) texK = atoi(argv[i] + 5); if (!strncmp(argv[i], "regroup:", 8)) { regB = atoi(argv[i] + 8); const char* c2 = strchr(argv[i] + 8, ','); regKey = c2 ? atoi(c2 + 1) : 0; } if (!strncmp(argv[i], "park:", 5)) { parkK = atoi(argv[i] + 5); const char* c2 = strchr(argv[i] + 5, ','); parkM = c2 ? atoi(c2 + 1) : 0; } }
    auto verts = load<float>(dir + "/verts.f32"); auto tris = load<int32_t>(dir + "/tris.i32"); auto uvs = load<float>(dir + "/tri_uvs.f32");
    auto pos = load<float>(dir + "/pos.f32"); auto nrm = load<float>(dir + "/nrm.f32"); auto shift = load<float>(dir + "/shift.f32");
    auto ids = load<int32_t>(dir + "/ids.i32"); auto meta = load<int32_t>(dir + "/meta.i32");
    const int N = meta[0]; int log2N = 0; while ((1 << log2N) < N) log2N++;
    const int T = (int)tris.size() / 3, V = (int)verts.size() / 3;
    BvhHost h; build_bvh(verts.data(), V, tris.data(), T, uvs.data(), h);
    printf("T=%d nodes4=%zu depth4=%d N=%d groups=%d passes=%d cull=%d cull8=%d\n", T, h.nodes4.size(), h.max_depth4, N, n_groups, n_pass, cull, cull8);
    // 8-bit entry distance (conservative lower bound): minifloat with 5 exponent + 3 mantissa bits below an exponent ceiling taken from the scene
    float diag = 0.f; { float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f}; for (int i = 0; i < V; i++) for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], verts[3 * i + a]); mx[a] = fmaxf(mx[a], verts[3 * i + a]); } for (int a = 0; a < 3; a++) diag += (mx[a] - mn[a]) * (mx[a] - mn[a]); diag = sqrtf(diag); }
    int emax; frexpf(diag, &emax);            // diag < 2^emax
    auto q8 = [&](float t) -> float {         // round DOWN to the minifloat grid
        if (!(t > 0.f)) return 0.f;
        uint32_t b; memcpy(&b, &t, 4);
        int e = (int)(b >> 23) - 127;         // t in [2^e, 2^(e+1))
        int elo = emax - 31;
        if (e < elo) return 0.f;
        if (e >= emax) return ldexpf(1.f, emax);
        b &= 0xFFF00000u;                     // keep 3 mantissa bits
        float r; memcpy(&r, &b, 4); return r;
    };
    if (argc > 2 && !strcmp(argv[2], "rays")) {
        // watertightness study: trace org.f32/dir.f32 one ray at a time with the kernel's float arithmetic (fmaf where hipcc contracts),
        // count rays that find no hit.  variants (argv[3..]): pad = far plane distances x (1 + 2^-21); sub = node origin minus ray origin first;
        // inflate = quantised child boxes grown by one cell (done here at decode time)
        bool pad = false, sub = false, inflate = false, mt = false;
        for (int i = 3; i < argc; i++) { pad |= !strcmp(argv[i], "pad"); sub |= !strcmp(argv[i], "sub"); inflate |= !strcmp(argv[i], "inflate"); mt |= !strcmp(argv[i], "mt"); }
        auto org = load<float>(dir + "/org.f32"); auto dd = load<float>(dir + "/dir.f32");
        const int64_t R = (int64_t)org.size() / 3;
        int64_t miss = 0, boxmiss = 0; double nodes = 0;
#pragma omp parallel for reduction(+ : miss, nodes) schedule(dynamic, 4096)
        for (int64_t r = 0; r < R; r++) {
            float o[3] = {org[3 * r], org[3 * r + 1], org[3 * r + 2]}, d[3] = {dd[3 * r], dd[3 * r + 1], dd[3 * r + 2]}, id[3], ood[3];
            for (int a = 0; a < 3; a++) { float x = fabsf(d[a]) > 8.271806e-25f ? d[a] : copysignf(8.271806e-25f, d[a]); id[a] = 1.f / x; ood[a] = o[a] * id[a]; }
            float ht = INFINITY; int slot = -1;
            std::vector<int> stk; int node = 0;
            float wmx[3] = {0, 0, 0}, wmy[3] = {0, 0, 0}, wmz[3] = {0, 0, 0};
            {
                int kz = 0; if (fabsf(d[1]) > fabsf(d[kz])) kz = 1; if (fabsf(d[2]) > fabsf(d[kz])) kz = 2;
                int kx = (kz + 1) % 3, ky = (kx + 1) % 3; if (d[kz] < 0.f) std::swap(kx, ky);
                float Sz = 1.f / d[kz], Sx = d[kx] * Sz, Sy = d[ky] * Sz;
                wmx[kx] = 1.f; wmx[kz] = -Sx; wmy[ky] = 1.f; wmy[kz] = -Sy; wmz[kz] = Sz;
            }
            while (node != kSent) {
                if (node >= 0) {
                    nodes++;
                    const GpuNode4& n = h.nodes4[node];
                    const float cell[3] = {n.cell_x, n.cell_y, n.cell_z};
                    const uint32_t lo[3] = {n.lox, n.loy, n.loz}, hi[3] = {n.hix, n.hiy, n.hiz};
                    float key[4]; int code[4];
                    for (int k = 0; k < 4; k++) {
                        float tn = 0.f, tf = ht;
                        for (int a = 0; a < 3; a++) {
                            float sc = cell[a] * id[a];
                            float b = sub ? (n.origin[a] - o[a]) * id[a] : fmaf(n.origin[a], id[a], -ood[a]);
                            uint32_t nq = id[a] < 0.f ? hi[a] : lo[a], fq = id[a] < 0.f ? lo[a] : hi[a];
                            int nb = (int)((nq >> (8 * k)) & 255u), fb = (int)((fq >> (8 * k)) & 255u);
                            if (inflate && !(((lo[a] >> (8 * k)) & 255u) > ((hi[a] >> (8 * k)) & 255u))) { if (id[a] < 0.f) { nb += 1; fb -= 1; } else { nb -= 1; fb += 1; } }
                            float tnn = fmaf((float)nb, sc, b), tff = fmaf((float)fb, sc, b);
                            if (pad) tff *= 1.0000005f;
                            tn = fmaxf(tn, tnn); tf = fminf(tf, tff);
                        }
                        key[k] = tn <= tf ? tn : INFINITY; code[k] = n.c[k];
                    }
                    for (int k = 0; k < 4; k++) if (key[k] < INFINITY) stk.push_back(code[k]);
                } else {
                    uint32_t code = ~(uint32_t)node; int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
#if TEXIR_QUAD
                    first *= 2; cnt *= 2;          // the 4-wide leaves name quad records: record r = slots 2 r, 2 r + 1 (an empty slot holds a degenerate triangle)
#endif
                    for (int i = first; i < first + cnt; i++) {
                        const GpuTri& tr = h.tris[i];
#if TEXIR_TRI_WATERTIGHT
                        // Woop/Benthin/Wald: shear to ray space (per-ray vectors mx, my, mz), exact-sign 2D edge functions
                        float P[3][3];
                        const float* vv[3] = {tr.v0, tr.e1, tr.e2};
                        for (int k = 0; k < 3; k++) {
                            float A0 = vv[k][0] - o[0], A1 = vv[k][1] - o[1], A2 = vv[k][2] - o[2];
                            P[k][0] = fmaf(A2, wmx[2], fmaf(A1, wmx[1], A0 * wmx[0]));
                            P[k][1] = fmaf(A2, wmy[2], fmaf(A1, wmy[1], A0 * wmy[0]));
                            P[k][2] = fmaf(A2, wmz[2], fmaf(A1, wmz[1], A0 * wmz[0]));
                        }
                        auto edge2 = [&](const float* b_, const float* c_) {        // Cx*By - Cy*Bx with an exact sign
                            float p_ = c_[0] * b_[1], q_ = c_[1] * b_[0], r_ = p_ - q_;
                            if (r_ == 0.f) r_ = fmaf(c_[0], b_[1], -p_) - fmaf(c_[1], b_[0], -q_);
                            return r_; };
                        auto edge2f = [&](const float* b_, const float* c_) { return c_[0] * b_[1] - c_[1] * b_[0]; };
                        float U = edge2(P[1], P[2]), V = edge2(P[2], P[0]), W = edge2(P[0], P[1]);
                        float Uf = edge2f(P[1], P[2]), Vf = edge2f(P[2], P[0]), Wf = edge2f(P[0], P[1]);
                        float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
                        float det = Uf + Vf + Wf;
                        float T_ = Uf * P[0][2] + Vf * P[1][2] + Wf * P[2][2], t = T_ / det;
                        bool ok = !(mn < 0.f && mx > 0.f) && det != 0.f && t > 0.f && t < ht;
                        if (ok) { ht = t; slot = i; }
#endif
                    }
                }
                if (stk.empty()) node = kSent; else { node = stk.back(); stk.pop_back(); }
            }
            if (slot < 0) miss++;
        }
        printf("rays %lld: %lld escaped (pad=%d sub=%d inflate=%d), %.1f node visits per ray (unordered, no early-out)\n", (long long)R, (long long)miss, pad, sub, inflate, nodes / R);
        (void)boxmiss; (void)mt;
        return 0;
    }
    // regroup:B,key -- B consecutive wave groups (B x 64 Morton-neighbouring texels, one workgroup) re-deal their texels to their B waves before every pass, sorted by a
    // key taken from the texel's PREVIOUS pass (the neighbouring direction cell): key 0 = no regrouping (the same blocks, for comparison), 1 = the subtree (top two levels
    // of the 4-wide tree: <= 16 treelets + miss) of the previous hit, 2 = the previous pass's per-lane node visits (long rays together), 3 = (subtree, node visits)
    std::vector<int> slot_subtree(h.tris.size(), 16);
    if (regB) {
        std::vector<std::pair<int, int>> st;          // (node code, subtree id)
        const GpuNode4& root = h.nodes4[0];
        int next_id = 0;
        for (int k = 0; k < 4; k++) {
            const int c1 = root.c[k];
            if (c1 == kSent) continue;
            if (c1 < 0) { st.push_back({c1, next_id++}); continue; }
            for (int k2 = 0; k2 < 4; k2++) { const int c2 = h.nodes4[c1].c[k2]; if (c2 != kSent) st.push_back({c2, next_id++}); }
        }
        while (!st.empty()) {
            auto [code, id] = st.back(); st.pop_back();
            if (code == kSent) continue;
            if (code >= 0) { for (int k = 0; k < 4; k++) st.push_back({h.nodes4[code].c[k], id}); continue; }
            uint32_t lc = ~(uint32_t)code; int first = (int)(lc >> 3), cnt = (int)(lc & 7u) + 1;
#if TEXIR_QUAD
            first *= 2; cnt *= 2;
#endif
            for (int i = first; i < first + cnt && i < (int)slot_subtree.size(); i++) slot_subtree[i] = id;
        }
        printf("regroup: blocks of %d waves, key %d, %d treelets\n", regB, regKey, next_id);
    }
    Counters tot;
    const int64_t n_ids = (int64_t)ids.size();
    const int64_t n_grp_total = n_ids / 64;
#pragma omp parallel
    {
        Counters c;
        std::vector<Ray> R(64);
#pragma omp for schedule(dynamic, 1)
        for (int gi = 0; gi < n_groups; gi++) {
            const int B = regB ? regB : 1;
            int64_t g = (int64_t)((double)gi / n_groups * n_grp_total);
            if (regB) g = std::min<int64_t>((g / B) * B, n_grp_total - B);
            std::vector<int> blk(B * 64), order(B * 64), lane_tex(64); std::vector<double> rkey(B * 64, 0.0);
            for (int i = 0; i < B * 64; i++) { blk[i] = ids[g * 64 + i]; order[i] = i; }
            Frame fr[64];
            int lane_pass[64];
            auto init_ray = [&](int l, uint32_t J) {
                    int t = lane_tex[l];
                    fr[l] = make_frame(nrm[3 * t], nrm[3 * t + 1], nrm[3 * t + 2]);
                    float sh0 = shift[2 * t], sh1 = shift[2 * t + 1];
                    uint32_t i = sample_index(cell_to_pass(J, sh0, sh1, log2N), log2N);
                    float s0 = swc((float)i / (float)N, sh0), s1 = swc((float)((double)brev(i) * 2.3283064365386963e-10), sh1);
                    float phi = 6.283185307179586f * s1 - 3.141592653589793f, ct = 1.f - s0, st = sqrtf(1.f - ct * ct);
                    float sp = sinf(phi) * st, cp = -(cosf(phi) * st);
                    Ray& r = R[l];
                    for (int a = 0; a < 3; a++) { r.o[a] = pos[3 * t + a]; r.d[a] = fr[l].V[a] * sp + fr[l].n[a] * ct + fr[l].U[a] * cp; }
                    for (int a = 0; a < 3; a++) { float d = fabsf(r.d[a]) > 8.271806e-25f ? r.d[a] : copysignf(8.271806e-25f, r.d[a]); r.id[a] = 1.f / d; r.ood[a] = r.o[a] * r.id[a]; }
                    r.t = INFINITY; r.slot = -1; r.node = 0; r.pleaf = 0; r.stk.clear(); r.stk_t.clear();
            };
            // refill mode: the lanes of a wave walk n_pass CONSECUTIVE cells each at its own pace; finished lanes take their next cell when at least
            // refillK lanes are idle (or nothing else is left to do)
            const uint32_t Jbase = (uint32_t)((gi * 97) % 32) * (uint32_t)(N / 32);
            std::vector<Ray> queue; bool in_compact = false;
            for (int pj = 0; pj < (refillK ? 1 : n_pass) || !queue.empty(); pj++) {
                const uint32_t J = refillK ? Jbase : consec ? (Jbase + (uint32_t)pj) % (uint32_t)N : (uint32_t)((double)pj / n_pass * N);      // consec: the passes of ONE wedge, neighbouring cells in order (what a chunk of the kernel walks)
                if (regB && regKey && pj > 0) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rkey[a] < rkey[b]; });
              const int CPW = 64 / texK;
              for (int wv = 0; wv < (regB ? B : CPW); wv++) {
                for (int l = 0; l < 64; l++) lane_tex[l] = texK < 64 ? blk[wv * texK + l % texK] : blk[order[wv * 64 + l]];
                // texk:K -- a wave = K texels x (64 / K) NEIGHBOURING cells per pass instead of 64 texels x 1 cell: 64 / K such waves share a group's texels (sub-wave `sw` of the
                // pass takes texels [sw K, sw K + K)); lane l traces texel (l % K) in cell J (64 / K) + l / K.  Same rays in total, fewer origins and more directions per wave.

                in_compact = false;
                const double w0n = c.wnode, w0t = c.wtri, n0 = c.nodes, t0 = c.tris;
                if (parkK && (queue.size() >= 64 || pj >= n_pass)) {
                    // a compact pass over (up to) 64 parked rays
                    in_compact = true; pj--;
                    const size_t take = std::min<size_t>(64, queue.size());
                    for (size_t l = 0; l < 64; l++) { if (l < take) R[l] = queue[queue.size() - take + l]; else { R[l].node = kSent; R[l].pleaf = 0; R[l].stk.clear(); R[l].stk_t.clear(); R[l].slot = -1; } }
                    queue.resize(queue.size() - take);
                    c.cpass++; c.rays -= 64;          // (not new rays)
                } else
                for (int l = 0; l < 64; l++) { init_ray(l, texK < 64 ? (J * (uint32_t)CPW + (uint32_t)(l / texK)) % (uint32_t)N : J); lane_pass[l] = 0; }
                if (refillK) { c.refills++; c.refill_lanes += 64; }
                c.rays += 64;
                int lane_nodes[64] = {0}, lane_tris[64] = {0};
                std::vector<int> pass_nodes, pass_leaves;      // every (lane, node) / (lane, leaf) visit of this pass: their distinct counts = what ONE packet traversal of the wave would visit
                bool still_uni = true;      // no divergent node step yet in this pass
                auto pop = [&](Ray& r) -> int {
                    for (;;) {
                        if (r.stk.empty()) return kSent;
                        int v = r.stk.back(); float tn = r.stk_t.back(); r.stk.pop_back(); r.stk_t.pop_back();
                        if (cull && tn >= r.t) { c.culled++; continue; }
                        return v;
                    }
                };
                auto node_step = [&]() {
                        c.wnode++;
                        {   // wave-uniform step? (all lanes that take part hold the same node); how deep are the stacks; distinct 128-B lines fetched
                            int first = -1; bool uni = true; size_t deep = 0; std::vector<int> ln;
                            for (auto& r : R) if (r.node >= 0 && r.node != kSent) { if (first < 0) first = r.node; else if (r.node != first) uni = false; if (r.stk.size() > deep) deep = r.stk.size(); ln.push_back(r.node >> 1); }
                            if (uni) c.wuni++;
                            if (uni && still_uni) c.wuni0++; else still_uni = false;
                            if (deep + 3 > 8) c.wdeep[0]++; if (deep + 3 > 10) c.wdeep[1]++; if (deep + 3 > 11) c.wdeep[2]++; if (deep + 3 > 16) c.wdeep[3]++;
                            std::sort(ln.begin(), ln.end()); c.lines += (double)(std::unique(ln.begin(), ln.end()) - ln.begin());
                            {   // histogram of DISTINCT NODES per wave-level step (and of the lanes that take part), bucket = min(distinct, 8) - 1
                                std::vector<int> nd; for (auto& r : R) if (r.node >= 0 && r.node != kSent) nd.push_back(r.node);
                                const int lanes = (int)nd.size();
                                std::sort(nd.begin(), nd.end()); const int dn = (int)(std::unique(nd.begin(), nd.end()) - nd.begin());
                                const int bk = std::min(dn, 8) - 1; c.dh[bk] += 1; c.dl[bk] += lanes;
                            }
                        }
                        for (auto& r : R) {
                            if (!(r.node >= 0 && r.node != kSent)) continue;
                            c.nodes++; lane_nodes[&r - &R[0]]++; pass_nodes.push_back(r.node);
                            const GpuNode4& n = h.nodes4[r.node];
                            float key[4]; int code[4];
                            const float cell[3] = {n.cell_x, n.cell_y, n.cell_z};
                            const uint32_t lo[3] = {n.lox, n.loy, n.loz}, hi[3] = {n.hix, n.hiy, n.hiz};
                            for (int k = 0; k < 4; k++) {
                                float tn = 0.f, tf = r.t;
                                for (int a = 0; a < 3; a++) {
                                    float s = cell[a] * r.id[a], b = n.origin[a] * r.id[a] - r.ood[a];
                                    uint32_t nq = r.id[a] < 0.f ? hi[a] : lo[a], fq = r.id[a] < 0.f ? lo[a] : hi[a];
                                    float tnn = (float)((nq >> (8 * k)) & 255u) * s + b, tff = (float)((fq >> (8 * k)) & 255u) * s + b;
                                    tn = fmaxf(tn, tnn); tf = fminf(tf, tff);
                                }
                                key[k] = tn <= tf ? tn : INFINITY; code[k] = n.c[k];
                            }
                            if (!nosort) {
#define CS(a, b) if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(code[a], code[b]); }
                                CS(0, 1) CS(2, 3) CS(0, 2) if (!psort3) { CS(1, 3) CS(1, 2) }
#undef CS
                            }
                            for (int k = 3; k >= 1; k--) if (key[k] < INFINITY) { r.stk.push_back(code[k]); r.stk_t.push_back(cull8 ? q8(key[k]) : key[k]); }
                            if ((double)r.stk.size() > c.maxsp) c.maxsp = (double)r.stk.size();
                            if (key[0] < INFINITY) r.node = code[0]; else r.node = pop(r);
                            if (spec && r.node < 0 && r.pleaf == 0) { r.pleaf = r.node; r.node = pop(r); c.specpops++; }
                        }
                };
                auto leaf_step = [&]() {
                        int mx = 0;
                        for (auto& r : R) {
                            if (r.node >= 0 && r.pleaf == 0) continue;
                            const bool from_p = r.pleaf != 0;
                            uint32_t code = ~(uint32_t)(from_p ? r.pleaf : r.node);
                            int first = (int)(code >> 3), cnt = (int)(code & 7u) + 1;
#if TEXIR_QUAD
                            first *= 2; cnt *= 2;  // (quad records, as above: the replay still counts per-triangle tests -- the kernel's leaf step covers a record's two at once)
#endif
                            if (cnt > mx) mx = cnt;
                            pass_leaves.push_back((int)code);
                            for (int i = first; i < first + cnt; i++) {
                                c.tris++; lane_tris[&r - &R[0]]++;
                                const GpuTri& tr = h.tris[i];
                                float px = r.d[1] * tr.e2[2] - r.d[2] * tr.e2[1], py = r.d[2] * tr.e2[0] - r.d[0] * tr.e2[2], pz = r.d[0] * tr.e2[1] - r.d[1] * tr.e2[0];
                                float det = tr.e1[0] * px + tr.e1[1] * py + tr.e1[2] * pz, inv = 1.f / det;
                                float tx = r.o[0] - tr.v0[0], ty = r.o[1] - tr.v0[1], tz = r.o[2] - tr.v0[2];
                                float u = (tx * px + ty * py + tz * pz) * inv;
                                float qx = ty * tr.e1[2] - tz * tr.e1[1], qy = tz * tr.e1[0] - tx * tr.e1[2], qz = tx * tr.e1[1] - ty * tr.e1[0];
                                float v = (r.d[0] * qx + r.d[1] * qy + r.d[2] * qz) * inv, t = (tr.e2[0] * qx + tr.e2[1] * qy + tr.e2[2] * qz) * inv;
                                if (det != 0.f && u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t > 0.f && t < r.t) { r.t = t; r.slot = i; }
                            }
                            if (from_p) r.pleaf = 0; else r.node = pop(r);
                            if (spec && r.node < 0 && r.pleaf == 0) { r.pleaf = r.node; r.node = pop(r); }
                        }
                        c.wtri += mx;
                };
                // scheduling policy: 0 = the kernel's while-while (node phase until no lane holds an inner node, then leaf phase until no lane holds a leaf);
                // 1 = per step, the phase with more waiting lanes (node lanes weighted by `alpha`)
                int phase = 0, prev_nl = 0, sched_iter = -1, few_steps = 0;
                for (;;) {
                    sched_iter++;
                    int nn = 0, nl = 0;
                    int nl_any = 0;
                    for (auto& r : R) { if (r.pleaf) nl_any++; if (r.node == kSent) { if (r.pleaf) nl++; continue; } if (r.node >= 0) nn++; else { nl++; if (!r.pleaf) nl_any++; } }
                    if (spec == 2) nl = nl_any;
                    if (refillK) {
                        int idle = 0; for (int l = 0; l < 64; l++) if (R[l].node == kSent && !R[l].pleaf && lane_pass[l] + 1 < n_pass) idle++;
                        if (idle && (idle >= refillK || (!nn && !nl))) {
                            c.refills++; c.refill_lanes += idle;
                            for (int l = 0; l < 64; l++) if (R[l].node == kSent && !R[l].pleaf && lane_pass[l] + 1 < n_pass) { if (R[l].slot >= 0) c.hits++; lane_pass[l]++; init_ray(l, Jbase + lane_pass[l]); c.rays++; nn++; }
                            { int lo = 1 << 30, hi = 0; for (int l = 0; l < 64; l++) { lo = std::min(lo, lane_pass[l]); hi = std::max(hi, lane_pass[l]); } c.spread += hi - lo; }
                        }
                    }
                    if (!nn && !nl) break;
                    {   // histogram of wave-level steps by lanes under way (1, 2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-63, 64)
                        const int act = nn + nl; const int bk = act <= 1 ? 0 : act == 2 ? 1 : act <= 4 ? 2 : act <= 8 ? 3 : act <= 16 ? 4 : act <= 32 ? 5 : act <= 48 ? 6 : act < 64 ? 7 : 8;
                        c.ah[bk] += 1;
                    }
                    if (parkK && !in_compact) {
                        // straggler parking: at most parkK lanes under way for more than parkM consecutive wave-level steps -> their rays (with the closest hit found
                        // so far) go to the wave's queue and the pass ends; the queue is traced 64 rays at a time, restarted from the root with t_max = that hit
                        if (nn + nl <= parkK) few_steps++; else few_steps = 0;
                        if (few_steps > parkM) {
                            for (auto& r : R) if (r.node != kSent) { Ray q = r; q.node = 0; q.stk.clear(); q.stk_t.clear(); q.pleaf = 0; queue.push_back(q); r.node = kSent; c.parked++; }
                            break;
                        }
                    }
                    int want;
                    if (policy == 0) want = phase == 0 ? (nn ? 0 : 1) : (nl ? 1 : 0);
                    else if (policy == 1) want = !nn ? 1 : (!nl_any ? 0 : ((double)nn * (switch_after > 0 ? (sched_iter < switch_after ? alpha : alpha2) : (phase == 0 ? alpha : alpha2)) >= (double)nl ? 0 : 1));
                    else {
                        // policy 2 ("arrivals"): node steps while lanes keep ARRIVING at leaves (the batch is still growing) and the node lanes are the
                        // weighted majority (alpha); once a node step brought no new leaf lane, the batch is taken if it holds at least nn / alpha2 lanes
                        const bool grew = nl > prev_nl;
                        if (!nn) want = 1; else if (!nl_any) want = 0;
                        else if ((double)nl >= (double)nn * 1.0) want = 1;                     // leaf lanes are the plain majority
                        else if (!grew && phase == 0 && (double)nl * alpha2 >= (double)nn) want = 1;
                        else want = ((double)nn * alpha >= (double)nl) ? 0 : 1;
                    }
                    prev_nl = (want == 1) ? 0 : nl;
                    if (want != phase || c.rays == 0) c.rounds += 0.5;
                    phase = want;
                    if (phase == 0) node_step(); else leaf_step();
                }
                if (in_compact) { c.cwnode += c.wnode - w0n; c.cwtri += c.wtri - w0t; c.cnodes += c.nodes - n0; c.ctris += c.tris - t0; }
                for (auto& r : R) if (r.slot >= 0) c.hits++;
                { int mn = 0, mt = 0; for (int l = 0; l < 64; l++) { mn = std::max(mn, lane_nodes[l]); mt = std::max(mt, lane_tris[l]); } c.wmaxn += mn; c.wmaxt += mt; }
                if (regB && texK == 64) for (int l = 0; l < 64; l++) {
                    const int sub = R[l].slot >= 0 ? slot_subtree[R[l].slot] : 17;
                    rkey[order[wv * 64 + l]] = regKey == 1 ? (double)sub : regKey == 2 ? (double)lane_nodes[l] : (double)sub * 4096.0 + (double)lane_nodes[l];
                }
                { std::sort(pass_nodes.begin(), pass_nodes.end()); c.un_nodes += (double)(std::unique(pass_nodes.begin(), pass_nodes.end()) - pass_nodes.begin());
                  std::sort(pass_leaves.begin(), pass_leaves.end()); c.un_leaves += (double)(std::unique(pass_leaves.begin(), pass_leaves.end()) - pass_leaves.begin()); }
              }
            }
        }
#pragma omp critical
        { tot.rays += c.rays; tot.nodes += c.nodes; tot.tris += c.tris; tot.wnode += c.wnode; tot.wtri += c.wtri; tot.culled += c.culled; tot.hits += c.hits; tot.wuni += c.wuni; tot.wuni0 += c.wuni0; tot.lines += c.lines; tot.wmaxn += c.wmaxn; tot.wmaxt += c.wmaxt; tot.rounds += c.rounds; for (int q = 0; q < 8; q++) { tot.dh[q] += c.dh[q]; tot.dl[q] += c.dl[q]; } tot.refills += c.refills; tot.refill_lanes += c.refill_lanes; tot.spread += c.spread; tot.un_nodes += c.un_nodes; tot.un_leaves += c.un_leaves; tot.parked += c.parked; tot.cpass += c.cpass; tot.cwnode += c.cwnode; tot.cwtri += c.cwtri; tot.cnodes += c.cnodes; tot.ctris += c.ctris; for (int q = 0; q < 9; q++) tot.ah[q] += c.ah[q]; for (int q = 0; q < 4; q++) tot.wdeep[q] += c.wdeep[q]; if (c.maxsp > tot.maxsp) tot.maxsp = c.maxsp; }
    }
    double wr = tot.rays / 64.0;
    printf("per ray: %.2f node visits, %.2f tri tests, %.2f culled pops, hit %.4f, max stack %.0f\n", tot.nodes / tot.rays, tot.tris / tot.rays, tot.culled / tot.rays, tot.hits / tot.rays, tot.maxsp);
    printf("per pass: %.2f wave node steps (util %.3f), %.2f wave tri steps (util %.3f)\n", tot.wnode / wr, tot.nodes / (64.0 * tot.wnode), tot.wtri / wr, tot.tris / (64.0 * tot.wtri));
    printf("wave node steps: %.3f uniform (%.3f in the initial all-uniform run), %.2f distinct node lines per step; steps whose push could pass 8/10/11/16 entries: %.4f %.4f %.4f %.4f\n", tot.wuni / tot.wnode, tot.wuni0 / tot.wnode, tot.lines / tot.wnode,
           tot.wdeep[0] / tot.wnode, tot.wdeep[1] / tot.wnode, tot.wdeep[2] / tot.wnode, tot.wdeep[3] / tot.wnode);
    printf("per pass: max-lane node visits %.2f, max-lane tri tests %.2f, while-while rounds %.2f\n", tot.wmaxn / wr, tot.wmaxt / wr, tot.rounds / wr);
    printf("union over the 64 rays of a pass (= the visits of ONE packet traversal of the wave): %.1f inner nodes, %.1f leaves\n", tot.un_nodes / wr, tot.un_leaves / wr);
    printf("distinct nodes per wave-level node step (share of steps : mean lanes taking part):"); for (int q = 0; q < 8; q++) printf(" %d%s %.3f:%.1f", q + 1, q == 7 ? "+" : "", tot.dh[q] / tot.wnode, tot.dh[q] > 0 ? tot.dl[q] / tot.dh[q] : 0.0); printf("\n");
    if (tot.refills > 0) printf("refill events per 64 rays %.3f (lanes per event %.1f, pass spread at refill %.1f); cost incl. %.1f per refill event: %.2f per pass\n", tot.refills / wr, tot.refill_lanes / tot.refills, tot.spread / tot.refills, 2.6, tot.wnode / wr + tricost * tot.wtri / wr + 2.6 * tot.refills / wr);
    { const char* nm[9] = {"1", "2", "3-4", "5-8", "9-16", "17-32", "33-48", "49-63", "64"}; double all = 0; for (int q = 0; q < 9; q++) all += tot.ah[q];
      printf("wave-level steps by lanes under way:"); for (int q = 0; q < 9; q++) printf(" %s %.3f", nm[q], tot.ah[q] / all); printf("\n"); }
    if (parkK) printf("parking (<= %d lanes for > %d steps): %.3f of the rays parked; compact passes per main pass %.3f: %.2f node steps (util %.3f) + %.2f tri steps (util %.3f) each; of all wave-level steps %.3f node / %.3f tri run in compact passes\n",
                      parkK, parkM, tot.parked / tot.rays, tot.cpass / wr, tot.cwnode / std::max(1.0, tot.cpass), tot.cnodes / (64.0 * std::max(1.0, tot.cwnode)), tot.cwtri / std::max(1.0, tot.cpass), tot.ctris / (64.0 * std::max(1.0, tot.cwtri)),
                      tot.cwnode / tot.wnode, tot.cwtri / tot.wtri);
    printf("cost model (node step 1, triangle test %.2f): %.2f per pass\n", tricost, tot.wnode / wr + tricost * tot.wtri / wr);
    return 0;
}
