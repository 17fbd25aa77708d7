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

// 48-byte leaf-ordered triangle: (v0, prim id bits), (e1, 0), (e2, 0)
struct alignas(16) GpuTri {
    float v0[3]; uint32_t prim;
    float e1[3]; float pad1;
    float e2[3]; float pad2;
};
static_assert(sizeof(GpuTri) == 48, "triangle must be 48 bytes");

// 32-byte leaf-ordered corner uvs: (uv0, uv1), (uv2, 0, 0)
struct alignas(16) GpuTriUV {
    float uv[8];
};

constexpr int32_t kEmptyChild = INT32_MIN;   // child slot with an inverted box, never entered
constexpr int kMaxLeaf = 4;
constexpr int kMaxDepth = 60;                // traversal stack bound (LDS part + private overflow)

struct BvhHost {
    std::vector<GpuNode> nodes;
    std::vector<GpuTri> tris;
    std::vector<GpuTriUV> uvs;
    int max_depth = 0;
};

// verts [V,3], tris [T,3], tri_uvs [3T,2]
void build_bvh(const float* verts, int V, const int32_t* tris, int T, const float* tri_uvs, BvhHost& out);

}  // namespace texir
