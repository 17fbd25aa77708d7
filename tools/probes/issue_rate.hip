// Micro-benchmark (tool, not product): issue cost of the VALU instructions the traversal loop is made of, on gfx950.
// For every pattern: 8 independent register chains, 64 instructions per loop body, run at 1 and at 8 waves per SIMD on every
// CU.  Prints time per wave-instruction per SIMD relative to v_fma_f32 (= 1.00 by definition) -- the missing column of the
// guide's per-instruction table for integer / conversion / select ops.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)                      \
    {                                                                                                   \
        float r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        float a = seed * 0.5f, b = seed * 0.25f;                                                        \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc", "scc", "s40", "s41", "s42", "s43"); \
        }                                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                    \
    }

#define S(x) #x
#define I_FMA(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_MUL(k) "v_mul_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_ADD(k) "v_add_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_MAX(k) "v_max_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_MAX3(k) "v_max3_f32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_MIN3(k) "v_min3_f32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_CVTUB0(k) "v_cvt_f32_ubyte0 %" S(k) ", %" S(k) "\n"
#define I_CVTUB2(k) "v_cvt_f32_ubyte2 %" S(k) ", %" S(k) "\n"
#define I_CVTU32(k) "v_cvt_f32_u32 %" S(k) ", %" S(k) "\n"
#define I_CVTI32(k) "v_cvt_u32_f32 %" S(k) ", %" S(k) "\n"
#define I_CNDVCC(k) "v_cndmask_b32 %" S(k) ", %" S(k) ", %8, vcc\n"
#define I_CNDSGPR(k) "v_cndmask_b32 %" S(k) ", %" S(k) ", %8, s[40:41]\n"
#define I_CMPVCC(k) "v_cmp_lt_f32 vcc, %" S(k) ", %8\n"
#define I_CMPSGPR(k) "v_cmp_lt_f32 s[40:41], %" S(k) ", %8\n"
#define I_CMPCND(k) "v_cmp_lt_f32 vcc, %" S(k) ", %8\nv_cndmask_b32 %" S(k) ", %" S(k) ", %9, vcc\n"
#define I_CMPCND4(k) "v_cmp_lt_f32 vcc, %" S(k) ", %8\nv_cndmask_b32 %" S(k) ", %" S(k) ", %9, vcc\nv_cndmask_b32 %" S(k) ", %8, %" S(k) ", vcc\nv_cndmask_b32 %" S(k) ", %" S(k) ", %9, vcc\nv_cndmask_b32 %" S(k) ", %8, %" S(k) ", vcc\n"
#define I_ADDU(k) "v_add_u32 %" S(k) ", %" S(k) ", %8\n"
#define I_LSHL(k) "v_lshlrev_b32 %" S(k) ", 3, %" S(k) "\n"
#define I_AND(k) "v_and_b32 %" S(k) ", %" S(k) ", %8\n"
#define I_BFE(k) "v_bfe_u32 %" S(k) ", %" S(k) ", 8, 8\n"
#define I_PERM(k) "v_perm_b32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_RCP(k) "v_rcp_f32 %" S(k) ", %" S(k) "\n"
#define I_SQRT(k) "v_sqrt_f32 %" S(k) ", %" S(k) "\n"
#define I_SIN(k) "v_sin_f32 %" S(k) ", %" S(k) "\n"
#define I_EXP(k) "v_exp_f32 %" S(k) ", %" S(k) "\n"
#define I_MULLO(k) "v_mul_lo_u32 %" S(k) ", %" S(k) ", %8\n"
#define I_MAD24(k) "v_mad_u32_u24 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_MOV(k) "v_mov_b32 %" S(k) ", %8\n"
#define I_FMAMIX(k) "v_fma_mix_f32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_MED3(k) "v_med3_f32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_SALU(k) "s_add_u32 s40, s40, 1\n"
#define I_FMA_SALU(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\ns_add_u32 s42, s42, 1\n"

#define I_FMA_MAX(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_max_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_FMA_CVT(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_cvt_f32_ubyte0 %" S(k) ", %" S(k) "\n"
#define I_FMA_CND(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_cndmask_b32 %" S(k) ", %" S(k) ", %8, s[40:41]\n"
#define I_FMA_CMP(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_cmp_lt_f32 s[40:41], %" S(k) ", %8\n"
#define I_MAX_CVT(k) "v_max_f32 %" S(k) ", %" S(k) ", %8\nv_cvt_f32_ubyte0 %" S(k) ", %" S(k) "\n"
#define I_FMA2_MAX(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_mul_f32 %" S(k) ", %" S(k) ", %8\nv_max_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_FMA3_MAX(k) "v_fma_f32 %" S(k) ", %" S(k) ", %8, %9\nv_mul_f32 %" S(k) ", %" S(k) ", %8\nv_add_f32 %" S(k) ", %" S(k) ", %9\nv_max_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_AND_OR(k) "v_and_b32 %" S(k) ", %" S(k) ", %8\nv_or_b32 %" S(k) ", %" S(k) ", %9\n"
#define I_OR(k) "v_or_b32 %" S(k) ", %" S(k) ", %8\n"
#define I_XOR(k) "v_xor_b32 %" S(k) ", %" S(k) ", %8\n"
#define I_SUBF(k) "v_sub_f32 %" S(k) ", %" S(k) ", %8\n"
#define I_SUBU(k) "v_sub_u32 %" S(k) ", %" S(k) ", %8\n"
#define I_LSHR(k) "v_lshrrev_b32 %" S(k) ", 8, %" S(k) "\n"
#define I_MINU(k) "v_min_u32 %" S(k) ", %" S(k) ", %8\n"
#define I_ORSDWA(k) "v_or_b32_sdwa %" S(k) ", %8, %" S(k) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_ANDOR3(k) "v_and_or_b32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" S(k) ", %" S(k) ", 8, %9\n"
#define I_ADD3(k) "v_add3_u32 %" S(k) ", %" S(k) ", %8, %9\n"
#define I_CVTPKFP8(k) "v_cvt_f32_fp8 %" S(k) ", %" S(k) "\n"
#define I_MAXI(k) "v_max_i32 %" S(k) ", %" S(k) ", %8\n"
KERNEL(k_fma_max, BODY(I_FMA_MAX))
KERNEL(k_fma_cvt, BODY(I_FMA_CVT))
KERNEL(k_fma_cnd, BODY(I_FMA_CND))
KERNEL(k_fma_cmp, BODY(I_FMA_CMP))
KERNEL(k_max_cvt, BODY(I_MAX_CVT))
KERNEL(k_fma2_max, BODY(I_FMA2_MAX))
KERNEL(k_fma3_max, BODY(I_FMA3_MAX))
KERNEL(k_and_or, BODY(I_AND_OR))
KERNEL(k_or, BODY(I_OR))
KERNEL(k_xor, BODY(I_XOR))
KERNEL(k_subf, BODY(I_SUBF))
KERNEL(k_subu, BODY(I_SUBU))
KERNEL(k_lshr, BODY(I_LSHR))
KERNEL(k_minu, BODY(I_MINU))
KERNEL(k_orsdwa, BODY(I_ORSDWA))
KERNEL(k_andor3, BODY(I_ANDOR3))
KERNEL(k_lshlor, BODY(I_LSHLOR))
KERNEL(k_add3, BODY(I_ADD3))
KERNEL(k_cvtfp8, BODY(I_CVTPKFP8))
KERNEL(k_maxi, BODY(I_MAXI))
KERNEL(k_fma, BODY(I_FMA))
KERNEL(k_mul, BODY(I_MUL))
KERNEL(k_add, BODY(I_ADD))
KERNEL(k_max, BODY(I_MAX))
KERNEL(k_max3, BODY(I_MAX3))
KERNEL(k_min3, BODY(I_MIN3))
KERNEL(k_med3, BODY(I_MED3))
KERNEL(k_cvtub0, BODY(I_CVTUB0))
KERNEL(k_cvtub2, BODY(I_CVTUB2))
KERNEL(k_cvtu32, BODY(I_CVTU32))
KERNEL(k_cvti32, BODY(I_CVTI32))
KERNEL(k_cndvcc, BODY(I_CNDVCC))
KERNEL(k_cndsgpr, BODY(I_CNDSGPR))
KERNEL(k_cmpvcc, BODY(I_CMPVCC))
KERNEL(k_cmpsgpr, BODY(I_CMPSGPR))
KERNEL(k_cmpcnd, BODY(I_CMPCND))
KERNEL(k_cmpcnd4, BODY(I_CMPCND4))
KERNEL(k_addu, BODY(I_ADDU))
KERNEL(k_lshl, BODY(I_LSHL))
KERNEL(k_and, BODY(I_AND))
KERNEL(k_bfe, BODY(I_BFE))
KERNEL(k_perm, BODY(I_PERM))
KERNEL(k_rcp, BODY(I_RCP))
KERNEL(k_sqrt, BODY(I_SQRT))
KERNEL(k_sin, BODY(I_SIN))
KERNEL(k_exp, BODY(I_EXP))
KERNEL(k_mullo, BODY(I_MULLO))
KERNEL(k_mad24, BODY(I_MAD24))
KERNEL(k_mov, BODY(I_MOV))
KERNEL(k_fmamix, BODY(I_FMAMIX))
KERNEL(k_salu, BODY(I_SALU))
KERNEL(k_fma_salu, BODY(I_FMA_SALU))

typedef void (*kern_t)(float*, int, float);
struct Entry { const char* name; kern_t k; int per_slot; };   // per_slot: instructions per macro expansion

int main()
{
    std::vector<Entry> es = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_max_f32", k_max, 1}, {"v_max3_f32", k_max3, 1}, {"v_min3_f32", k_min3, 1},
        {"v_med3_f32", k_med3, 1}, {"v_cvt_f32_ubyte0", k_cvtub0, 1}, {"v_cvt_f32_ubyte2", k_cvtub2, 1}, {"v_cvt_f32_u32", k_cvtu32, 1}, {"v_cvt_u32_f32", k_cvti32, 1},
        {"v_cndmask_b32 (vcc)", k_cndvcc, 1}, {"v_cndmask_b32 (sgpr pair)", k_cndsgpr, 1}, {"v_cmp_lt_f32 -> vcc", k_cmpvcc, 1}, {"v_cmp_lt_f32 -> sgpr pair", k_cmpsgpr, 1},
        {"v_cmp + dependent v_cndmask (per instr)", k_cmpcnd, 2}, {"v_cmp + 4 dependent v_cndmask (per instr)", k_cmpcnd4, 5},
        {"v_add_u32", k_addu, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_and_b32", k_and, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_perm_b32", k_perm, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_sin_f32", k_sin, 1}, {"v_exp_f32", k_exp, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_mad_u32_u24", k_mad24, 1},
        {"v_mov_b32", k_mov, 1}, {"v_fma_mix_f32", k_fmamix, 1},
        {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_sub_f32", k_subf, 1}, {"v_sub_u32", k_subu, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_min_u32", k_minu, 1},
        {"v_max_i32", k_maxi, 1}, {"v_or_b32_sdwa (src1 BYTE_2)", k_orsdwa, 1}, {"v_and_or_b32", k_andor3, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_add3_u32", k_add3, 1},
        {"v_cvt_f32_fp8", k_cvtfp8, 1},
        {"PAIR v_fma + v_max (per pair)", k_fma_max, 1}, {"PAIR v_fma + v_cvt_ubyte (per pair)", k_fma_cvt, 1}, {"PAIR v_fma + v_cndmask (per pair)", k_fma_cnd, 1},
        {"PAIR v_fma + v_cmp (per pair)", k_fma_cmp, 1}, {"PAIR v_max + v_cvt_ubyte (per pair)", k_max_cvt, 1}, {"TRIPLE fma mul max (per triple)", k_fma2_max, 1},
        {"QUAD fma mul add max (per quad)", k_fma3_max, 1}, {"PAIR v_and + v_or (per pair)", k_and_or, 1}, {"s_add_u32 (SALU)", k_salu, 1}, {"v_fma_f32 + s_add_u32 interleaved (per pair)", k_fma_salu, 1},
    };
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    double ref[2] = {0, 0};
    printf("%-48s %12s %12s   (time per wave-instruction per SIMD, v_fma_f32 at the same occupancy = 1.00)\n", "instruction", "1 wave/SIMD", "8 waves/SIMD");
    for (auto& e : es) {
        double rel[2], ns[2];
        for (int occ = 0; occ < 2; occ++) {
            const int blocks = cus * (occ ? 8 : 1);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64.0 * e.per_slot * (occ ? 8 : 1);
            ns[occ] = ms * 1e6 / instr_per_simd;
            if (e.k == k_fma) ref[occ] = ns[occ];
            rel[occ] = ns[occ] / ref[occ];
        }
        printf("%-48s %7.3f (%5.2f ns) %7.3f (%5.2f ns)\n", e.name, rel[0], ns[0], rel[1], ns[1]); fflush(stdout);
    }
    return 0;
}
