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

// TEXIR_TRI64 = 0 (default): 48-byte triangle (v0, prim), (a, 0), (b, 0) + a separate 32-byte uv record (uv0, uv1), (uv2, 0, 0).
// TEXIR_TRI64 = 1 (A/B variant, measured -2 % on c4 / c2: 16 more bytes per triangle TEST cost more cache than the one uv line per RAY they save):
//   64-byte triangle, one cache line holding the three positions AND the corner uvs the hit shader needs:
//   (v0, uv0.x), (a, uv0.y), (b, uv1.x), (uv1.y, uv2.x, uv2.y, prim id bits)   with (a, b) = (v1, v2) [watertight] or (e1, e2) = (v1-v0, v2-v0).
#ifndef TEXIR_TRI64
#define TEXIR_TRI64 0
#endif
#if TEXIR_TRI64
struct alignas(16) GpuTri {
    float v0[3]; float uv0x;
    float e1[3]; float uv0y;
    float e2[3]; float uv1x;
    float uv1y, uv2x, uv2y; uint32_t prim;
};
static_assert(sizeof(GpuTri) == 64, "triangle must be 64 bytes");
constexpr int kTriQuads = 4;          // float4s per triangle record
#else
struct alignas(16) GpuTri {
    float v0[3]; uint32_t prim;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
};
static_assert(sizeof(GpuTri) == 48, "triangle must be 48 bytes");
constexpr int kTriQuads = 3;
#endif

// 32-byte leaf-ordered corner uvs: (uv0, uv1), (uv2, 0, 0)   (TEXIR_TRI64 = 0 only)
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

// TEXIR_NODE_F32 = 1: the same 4-wide tree with full float child boxes, 128 bytes per node (A/B variant: no byte -> float conversion
// and no sign select in the node step -- the near / far planes are picked by per-ray load offsets -- for twice the node bytes):
//   f0..f5 = (lo.x[4], hi.x[4], lo.y[4], hi.y[4], lo.z[4], hi.z[4])   one float per child, child k in lane k
//   f6     = (child0..3),  f7 = padding
#ifndef TEXIR_NODE_F32
#define TEXIR_NODE_F32 0
#endif
// TEXIR_UNIFORM_SLOAD (see traverse in device_common.h) = 2 (default) keeps BOTH forms of the tree: the quantised nodes for per-lane
// fetches, the float nodes for wave-uniform node steps, which read them through the scalar cache.  1: wave-uniform steps read the quantised
// node through the scalar cache; 0: every step fetches per lane (round-1 / early round-2 kernel).
#ifndef TEXIR_UNIFORM_SLOAD
#define TEXIR_UNIFORM_SLOAD 2
#endif
#define TEXIR_BUILD_F32NODES (TEXIR_NODE_F32 || TEXIR_UNIFORM_SLOAD >= 2)
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
    std::vector<GpuNode4F> nodes4f;     // filled (index-for-index with nodes4) when TEXIR_BUILD_F32NODES
    std::vector<GpuTri> tris;
    std::vector<GpuTriUV> uvs;
    int max_depth = 0, max_depth4 = 0;
};

// verts [V,3], tris [T,3], tri_uvs [3T,2]
void build_bvh(const float* verts, int V, const int32_t* tris, int T, const float* tri_uvs, BvhHost& out);

}  // namespace texir
