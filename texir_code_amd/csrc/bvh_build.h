// Host-side BVH builder for libtexir_hip.so.  Product code (never linked with oracle/).
// Replaces Open3D/Embree's RaycastingScene.add_triangles (models/tracer_o3d_irt.py:86-89).
#pragma once
#include <cstdint>
#include <vector>

namespace texir {

// 64-byte inner node holding BOTH children's boxes (one fetch per traversal step, 4 x dwordx4):
//   n0 = (c0.min.x, c0.max.x, c0.min.y, c0.max.y)
//   n1 = (c1.min.x, c1.max.x, c1.min.y, c1.max.y)
//   n2 = (c0.min.z, c0.max.z, c1.min.z, c1.max.z)
//   n3 = (child0, child1, 0, 0)   child >= 0: inner node index; child < 0: leaf, ~child = (first_tri << 3) | (count-1)
struct alignas(16) GpuNode {
    float n0[4], n1[4], n2[4];
    int32_t c[4];
};
static_assert(sizeof(GpuNode) == 64, "node must be 64 bytes");

// Triangle intersector (compile-time, the builder and the kernels of one library agree):
//   TEXIR_TRI_WATERTIGHT = 1 (default): edge functions in ray space on the three stored VERTICES (the Pluecker form Embree's
//     robust mode uses): the value a ray gets for an edge shared by two triangles is exactly the negative of the neighbour's,
//     so no ray slips between them;  = 0: Moeller-Trumbore on (v0, e1, e2), ~30 VALU cheaper per test, leaks at shared edges.
#ifndef TEXIR_TRI_WATERTIGHT
#define TEXIR_TRI_WATERTIGHT 1
#endif

// 48-byte triangle (v0, prim), (a, 0), (b, 0) with (a, b) = (v1, v2) [watertight] or (e1, e2) = (v1-v0, v2-v0), + a separate 32-byte uv
// record fetched only for the closest hit.  (A 64-byte record carrying the uvs measured -2 %: tools/experiments/.)
struct alignas(16) GpuTri {
    float v0[3]; uint32_t prim;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
};
static_assert(sizeof(GpuTri) == 48, "triangle must be 48 bytes");
constexpr int kTriQuads = 3;          // float4s per triangle record

// Quad leaves (TEXIR_QUAD = 1, default; watertight intersector only): the two triangles of a leaf that share an edge -- in a tessellated mesh nearly every
// 2-triangle leaf -- are stored as ONE 48-byte record of their four vertices (q0, q1, q2, q3): triangle 0 = (q0, q1, q2), triangle 1 = (q3, q2, q1), the
// shared edge being (q1, q2).  The 4-wide traversal fetches three 16-byte words per leaf record instead of six 12-byte vertices per pair, shears four
// vertices instead of six and evaluates five edge functions instead of six (the shared edge's value is the neighbour's negated, exactly).
//   Slots.  A record owns the two leaf-order slots 2 r and 2 r + 1 (sc.tris, sc.uvs, corner normals; a hit carries its slot): a triangle without a
//   partner (or whose partner would have to be mirrored) is a record whose q3 repeats q2 -- its second triangle has no area and accepts no ray -- and
//   whose odd slot holds the degenerate dummy triangle.  Each triangle is stored ROTATED so that the shared edge comes to lie where the record wants it:
//   slot data (vertices, corner uvs, corner normals) are all in the stored order, GpuTri::pad1 holds the rotation (stored corner k = the caller's
//   corner (rot + k) % 3), and only a barycentric that leaves the library (texir_trace_shade's primitive uvs) is turned back.
//   Leaf codes: the binary tree's name slots (first slot << 3 | slots - 1: a dummy slot in the range is tested and never hit), the 4-wide tree's name
//   records (first record << 3 | records - 1).
#ifndef TEXIR_QUAD
#define TEXIR_QUAD TEXIR_TRI_WATERTIGHT
#endif
struct alignas(16) GpuQuad { float q[12]; };          // q0.xyz q1.xyz q2.xyz q3.xyz
static_assert(sizeof(GpuQuad) == 48, "quad record must be 48 bytes");

// Corner uvs, fetched only for the closest hit.
//   TEXIR_UV_QUAD = 1 (default with quad leaves): ONE 32-byte record per quad RECORD -- the uvs of its four vertices (q0, q1, q2, q3); triangle 0 (even slot)
//     reads (q0, q1, q2), triangle 1 (odd slot) reads (q3, q2, q1), the stored corner order of its slot.  Two triangles are only paired when their corner uvs
//     agree on the shared edge (no uv seam along it), so the four pairs are all there is.  Half the bytes of the per-slot form: a 128-byte line holds the
//     uvs of 8 neighbouring triangles instead of 4 (the hit shader's uv fetch was 1 of its 2 lines per ray; round 5).
//   TEXIR_UV_QUAD = 0: 32 bytes per leaf-order SLOT: (uv0, uv1), (uv2, 0, 0).
#ifndef TEXIR_UV_QUAD
#define TEXIR_UV_QUAD TEXIR_QUAD
#endif
struct alignas(16) GpuTriUV {
    float uv[8];
};

// 64-byte 4-wide node with 8-bit child boxes quantised relative to the node's own box (one cache line per traversal step,
// half the steps of the binary tree):
//   q0 = (origin.x, origin.y, origin.z, cell.x)                         cell size per axis = a power of two (float)
//   q1 = (lo.x[4], lo.y[4], lo.z[4], hi.x[4])                           one byte per child, child k in byte k
//   q2 = (hi.y[4], hi.z[4], cell.y, cell.z)
//   q3 = (child0..3)  child >= 0: inner node index; < 0: leaf code as above.  Unused slots carry an inverted box and the
//        leaf code of a degenerate dummy triangle appended after the mesh, so they need no test in the traversal loop.
struct alignas(16) GpuNode4 {
    float origin[3]; float cell_x;
    uint32_t lox, loy, loz, hix;
    uint32_t hiy, hiz; float cell_y, cell_z;
    int32_t c[4];
};
static_assert(sizeof(GpuNode4) == 64, "wide node must be 64 bytes");

// The same 4-wide tree with full float child boxes (padded by the same absolute slack, not quantised), 128 bytes per node, index for
// index with the quantised nodes.  Wave-uniform node steps read this form through the scalar cache (device_common.h, node_step4):
//   plane[0..5] = (lo.x[4], hi.x[4], lo.y[4], hi.y[4], lo.z[4], hi.z[4])   one float per child, child k in lane k
//   c           = (child0..3),  pad
struct alignas(16) GpuNode4F {
    float plane[6][4];
    int32_t c[4];
    int32_t pad[4];
};
static_assert(sizeof(GpuNode4F) == 128, "float wide node must be 128 bytes");

constexpr int32_t kEmptyChild = INT32_MIN;   // child slot with an inverted box, never entered
constexpr int kMaxLeaf = 2;
constexpr int kMaxDepth = 60;                // traversal stack bound (LDS part + private overflow)

struct BvhHost {
    std::vector<GpuNode> nodes;
    std::vector<GpuNode4> nodes4;
    std::vector<GpuNode4F> nodes4f;     // index-for-index with nodes4
    std::vector<GpuTri> tris;           // by leaf-order slot (+ one degenerate dummy at the end)
    std::vector<GpuTriUV> uvs;
    std::vector<GpuQuad> quads;         // TEXIR_QUAD: by record (+ one all-zero dummy at the end); record r owns slots 2 r, 2 r + 1
    int64_t n_slots = 0;                // leaf-order slots (= triangles without TEXIR_QUAD)
    int max_depth = 0, max_depth4 = 0;
};

// verts [V,3], tris [T,3], tri_uvs [3T,2]
void build_bvh(const float* verts, int V, const int32_t* tris, int T, const float* tri_uvs, BvhHost& out);

}  // namespace texir
