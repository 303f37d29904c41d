// Micro-benchmark: issue cost of v_permlane32_swap / v_permlane16_swap / DPP add / plain add on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {   // plain adds, 8 independent chains
                a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
            } else if (MODE == 1) {   // permlane32 swap on 4 pairs + 4 adds
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
                a0 += a1; a2 += a3; a4 += a5; a6 += a7;
            } else if (MODE == 2) {
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(a1));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a2), "+v"(a3));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a4), "+v"(a5));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a6), "+v"(a7));
                a0 += a1; a2 += a3; a4 += a5; a6 += a7;
            } else if (MODE == 3) {   // 4 x (2 cndmask + dpp add) = the current exchange unit
                const bool s = threadIdx.x & 4;
                float x, y;
                x = s ? a1 : a0; y = s ? a0 : a1; a0 = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, true));
                x = s ? a3 : a2; y = s ? a2 : a3; a2 = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, true));
                x = s ? a5 : a4; y = s ? a4 : a5; a4 = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, true));
                x = s ? a7 : a6; y = s ? a6 : a7; a6 = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x141, 0xF, 0xF, true));
            } else if (MODE == 4) {   // bank-masked DPP adds: 2 per unit
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(a1) : "v"(a0));
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(a3) : "v"(a2));
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(a5) : "v"(a4));
                asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5" : "+v"(a7) : "v"(a6));
                a0 = a1 * 0.5f; a2 = a3 * 0.5f; a4 = a5 * 0.5f; a6 = a7 * 0.5f;
            } else if (MODE == 5) {   // v_exp_f32 x8
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            } else if (MODE == 7) {   // 8 x v_pk_fma_f32 (4 independent 2-vectors, 2 rounds)
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(p1), "v"(p2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(p2), "v"(p3));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(p3), "v"(p0));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(p0), "v"(p1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(p1), "v"(p2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(p2), "v"(p3));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(p3), "v"(p0));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(p0), "v"(p1));
                a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
            } else if (MODE == 8) {   // 8 x v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a4));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "v"(a6));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(a6), "v"(a7));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a0));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(a0), "v"(a1));
            } else if (MODE == 9) {   // 4 x (v_cmp_e64 -> sgpr, v_cndmask_e64)
                asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a0) : "v"(a1), "v"(a2) : "s20", "s21");
                asm volatile("v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[22:23]" : "=v"(a1) : "v"(a2), "v"(a3) : "s22", "s23");
                asm volatile("v_cmp_lt_f32_e64 s[24:25], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[24:25]" : "=v"(a2) : "v"(a3), "v"(a4) : "s24", "s25");
                asm volatile("v_cmp_lt_f32_e64 s[26:27], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[26:27]" : "=v"(a3) : "v"(a4), "v"(a5) : "s26", "s27");
            } else if (MODE == 10) {  // 8 x v_mul_f32 via asm (baseline for asm modes)
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(a1));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(a2));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(a3));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(a4));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(a5));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(a6));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(a7));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(a0));
            } else if (MODE == 11) {  // 8 x ds_read_b128 broadcast (same address for all lanes)
                __shared__ float4 lds[64];
                float4 r;
                asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((u & 7) * 16) : "memory");
                a0 += r.x; a1 += r.y; a2 += r.z; a3 += r.w;
                (void)lds;
            } else if (MODE == 6) {   // v_cndmask x8
                const bool s = (threadIdx.x + i) & 4;
                a0 = s ? a1 : a0; a1 = s ? a2 : a1; a2 = s ? a3 : a2; a3 = s ? a4 : a3; a4 = s ? a5 : a4; a5 = s ? a6 : a5; a6 = s ? a7 : a6; a7 = s ? a0 : a7;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char* name, int instr_per_u) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, REP);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x REP x 8 u; cycles at 2.4 GHz
    double cyc = ms * 1e-3 * 2.4e9 / (4.0 * REP * 8);
    printf("%-28s %8.3f ms  %6.1f cycles per unrolled body per wave (%d instr -> %.2f cyc/instr)\n", name, ms, cyc, instr_per_u, cyc / instr_per_u);
    hipFree(out);
}
int main() {
    run<0>("8 x v_add_f32", 8);
    run<1>("4 x (permlane32_swap + add)", 8);
    run<2>("4 x (permlane16_swap + add)", 8);
    run<3>("4 x (2 cndmask + add_dpp)", 12);
    run<4>("4 x (2 masked add_dpp + mul)", 12);
    run<5>("8 x v_exp_f32", 8);
    run<6>("8 x v_cndmask", 8);
    run<7>("8 x v_pk_fma_f32", 8);
    run<8>("8 x v_fma_f32", 8);
    run<9>("4 x (v_cmp_e64 + v_cndmask)", 8);
    run<10>("8 x v_mul_f32 (asm)", 8);
    run<11>("1 x ds_read_b128 bcast + 4 add", 5);
    return 0;
}
