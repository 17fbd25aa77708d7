// Micro-benchmark (tool): how many distinct cache lines per cycle a CU's vector L1 (TCP) serves on gfx950, for the access shapes of
// the traversal loop: a wave-wide global_load_dwordx4 whose 64 lanes touch 1 / 8 / 32 / 64 distinct 128-byte lines of an L1-resident
// 16 KiB window.  Prints time per wave-instruction per CU and lines per cycle (at the clock measured with a v_fma loop).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tcp_rate.hip -o tools/probes/tcp_rate && tools/probes/tcp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int STRIDE>     // bytes between consecutive lanes' 16-byte loads (0: all lanes the same address)
__global__ __launch_bounds__(256) void k_load(const float4* __restrict__ buf, float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    const char* base = reinterpret_cast<const char*>(buf) + (size_t)blockIdx.x % 4 * 16384;     // 16 KiB window per block (L1 resident)
    float acc = 0.f;
    uint32_t off = (uint32_t)(lane * STRIDE) & 16383u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float4 v = *reinterpret_cast<const float4*>(base + off);
            acc += v.x + v.w;
            off = (off + 4096u + 16u * (STRIDE == 16 ? 64 : 1)) & 16383u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_fma(float* out, int iters)
{
    float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + d + e + f;
}

int main()
{
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float4* buf; (void)hipMalloc(&buf, 65536); (void)hipMemset(buf, 0, 65536);
    float* out; (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](auto launch) { launch(); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); return (double)ms; };
    const int it = 4000;
    // clock: 64 fma per iteration per wave, 8 waves per SIMD, 2 cycles each
    double ms = time([&] { hipLaunchKernelGGL(k_fma, dim3(cus * 8), dim3(256), 0, 0, out, it); });
    const double ghz = (double)it * 64 * 8 * 2 / (ms * 1e6);
    printf("clock under a v_fma stream: %.2f GHz (if a wave64 fma issues in 2 cycles)\n", ghz);
    printf("%-34s %14s %14s %16s\n", "wave-wide dwordx4 load touching", "1 wave/SIMD", "8 waves/SIMD", "lines/clk/CU @8");
    struct { const char* name; int lines; double ms1, ms8; } r[4] = {{"1 line (all lanes same 16 B)", 1, 0, 0}, {"8 lines (lanes contiguous)", 8, 0, 0}, {"32 lines (lane stride 64 B)", 32, 0, 0}, {"64 lines (lane stride 128 B)", 64, 0, 0}};
    for (int occ = 0; occ < 2; occ++) {
        const int blocks = cus * (occ ? 8 : 1);
        double t0 = time([&] { hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(256), 0, 0, buf, out, it); });
        double t1 = time([&] { hipLaunchKernelGGL(k_load<16>, dim3(blocks), dim3(256), 0, 0, buf, out, it); });
        double t2 = time([&] { hipLaunchKernelGGL(k_load<64>, dim3(blocks), dim3(256), 0, 0, buf, out, it); });
        double t3 = time([&] { hipLaunchKernelGGL(k_load<128>, dim3(blocks), dim3(256), 0, 0, buf, out, it); });
        double t[4] = {t0, t1, t2, t3};
        for (int k = 0; k < 4; k++) (occ ? r[k].ms8 : r[k].ms1) = t[k];
    }
    for (int k = 0; k < 4; k++) {
        // wave-instructions per CU: iters * 8 loads * 4 waves per block * blocks per CU
        const double n1 = (double)it * 8 * 4 * 1, n8 = (double)it * 8 * 4 * 8;
        const double ns1 = r[k].ms1 * 1e6 / n1, ns8 = r[k].ms8 * 1e6 / n8;
        printf("%-34s %11.2f ns %11.2f ns %16.2f\n", r[k].name, ns1, ns8, r[k].lines / (ns8 * ghz));
    }
    return 0;
}
