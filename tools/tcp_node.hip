// Micro-benchmark (tool): what a BVH node fetch costs in the CU's vector L1 (TCP) on gfx950, for the traversal loop's access shape:
// the 64 lanes of a wave read D distinct nodes (runs of 64/D neighbouring lanes share a node) of NODE bytes each, as NODE/(4*VEC)
// consecutive VEC-dword loads per lane.  Pool: 128 nodes (L1 resident).  Prints time per node fetch per CU in ns; run under
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum
// for the tag-lookup counts (one kernel launch per row, in the order printed).
//   hipcc --offload-arch=gfx950 -O2 tools/tcp_node.hip -o tools/tcp_node && tools/tcp_node
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VEC> struct V;
template <> struct V<4> { using T = float4; static __device__ float s(const T& v) { return v.x + v.w; } };
template <> struct V<2> { using T = float2; static __device__ float s(const T& v) { return v.x + v.y; } };
template <> struct V<1> { using T = float;  static __device__ float s(const T& v) { return v; } };

template <int NODE, int VEC>
__global__ __launch_bounds__(256) void k_node(const char* __restrict__ pool, float* out, int iters, int d, int align_mask)
{
    using T = typename V<VEC>::T;
    const int lane = threadIdx.x & 63;
    const uint32_t group = (uint32_t)(lane * d) >> 6;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = (group * 37u + (uint32_t)i * 11u + (uint32_t)u * 5u + (uint32_t)(threadIdx.x >> 6) * 3u) & 127u;
            const char* p = pool + ((idx * (uint32_t)NODE) & (uint32_t)align_mask);
#pragma unroll
            for (int k = 0; k < NODE / (4 * VEC); k++) acc += V<VEC>::s(*reinterpret_cast<const T*>(p + k * 4 * VEC));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    char* pool; (void)hipMalloc(&pool, 65536); (void)hipMemset(pool, 0, 65536);
    float* out; (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int it = 2000, blocks = cus * 8;      // 8 waves per SIMD
    auto run = [&](const char* name, auto kern, int d, int mask) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, pool, out, 10, d, mask); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, pool, out, it, d, mask);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double fetches_per_cu = (double)it * 4 * 4 * 8;       // iters x unroll x waves per block x blocks per CU
        printf("%-28s d=%2d  %8.2f ns per node fetch per CU\n", name, d, ms * 1e6 / fetches_per_cu);
        fflush(stdout);
    };
    const int ds[6] = {1, 4, 8, 16, 32, 64};
    for (int d : ds) run("64B node, 4 x dwordx4", k_node<64, 4>, d, 0xffff);
    for (int d : ds) run("48B node, 3 x dwordx4", k_node<48, 4>, d, 0xffff);
    for (int d : ds) run("64B node, 8 x dwordx2", k_node<64, 2>, d, 0xffff);
    for (int d : ds) run("48B node, 6 x dwordx2", k_node<48, 2>, d, 0xffff);
    for (int d : ds) run("64B node, 16 x dword", k_node<64, 1>, d, 0xffff);
    for (int d : ds) run("32B node, 2 x dwordx4", k_node<32, 4>, d, 0xffff);
    for (int d : ds) run("128B node, 8 x dwordx4", k_node<128, 4>, d, 0xffff);
    for (int d : ds) run("80B node, 5 x dwordx4", k_node<80, 4>, d, 0xffff);
    return 0;
}
