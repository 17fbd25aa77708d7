// Calibration of rocprofv3's FETCH_SIZE on gfx950 for THIS code base's access pattern (MI355X_MICROARCH.md asks for it before an
// absolute is trusted): every lane gathers one random, 64-byte aligned 64-byte record as 4 x dwordx4 -- exactly how the traversal
// kernels fetch BVH nodes -- from a buffer far larger than L2 + Infinity Cache, each record touched once.
//   known bytes = records * 64      ->   compare with FETCH_SIZE(KB) * 1024 of `gather64`
// `stream16` reads the same number of bytes as a coalesced 16 B/lane stream (the case the guide documents as tallied at half).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/calib_fetch.hip -o /tmp/calib_fetch && rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void gather64(const float4* __restrict__ buf, uint64_t n_rec, float* __restrict__ out, uint64_t mul, uint64_t add)
{
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_rec) return;
    uint64_t r = (g * mul + add) % n_rec;                  // a permutation of the records when gcd(mul, n_rec) = 1
    const float4* p = buf + 4 * r;
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    out[g] = a.x + b.y + c.z + d.w;
}

__global__ void stream16(const float4* __restrict__ buf, uint64_t n16, float* __restrict__ out)
{
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n16) return;
    float4 a = buf[g];
    if (a.x == 123.456f) out[0] = a.y;
}

int main()
{
    const uint64_t bytes = 4ull << 30;                     // 4 GiB >> 256 MiB Infinity Cache
    const uint64_t n_rec = bytes / 64, n16 = bytes / 16;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, n_rec * sizeof(float));
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(gather64, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, 0, buf, n_rec, out, 2654435761ull, 12345ull);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, buf, n16, out);
    hipDeviceSynchronize();
    printf("gather64: %llu records x 64 B = %llu bytes;  stream16: %llu bytes\n", (unsigned long long)n_rec, (unsigned long long)(n_rec * 64),
           (unsigned long long)(n16 * 16));
    return 0;
}
