// valu_rates.hip -- measure issue rates of the integer instructions the field arithmetic is made of.
// Each kernel runs ITER iterations of 8 independent dependency chains per lane; reports lane-ops/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../triton_vm_amd/csrc/platform.h"
#define ITER 4096
#define CHAINS 8

#define KERNEL(name, body)                                                       \
__global__ void __launch_bounds__(256) name(u64* out, u64 seed) {                 \
    u64 x[CHAINS]; u32 lo[CHAINS], hi[CHAINS];                                    \
    for (int c = 0; c < CHAINS; c++) { x[c] = seed + threadIdx.x * 977 + c * 131; lo[c] = (u32)x[c] | 1; hi[c] = (u32)(x[c] >> 7) | 3; } \
    for (int i = 0; i < ITER; i++) {                                              \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { body }              \
    }                                                                             \
    u64 acc = 0; for (int c = 0; c < CHAINS; c++) acc ^= x[c] + lo[c] + hi[c];    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                             \
}

KERNEL(k_mad_u64_u32, x[c] = (u64)lo[c] * (u32)x[c] + x[c];)
KERNEL(k_mul_lo_u32, lo[c] = lo[c] * hi[c] + 1;)
KERNEL(k_mul_hi_u32, lo[c] = __umulhi(lo[c], hi[c]) | 0x80000001u;)
KERNEL(k_mad_u32_u24, lo[c] = __umul24(lo[c], hi[c]) + 7;)
KERNEL(k_add_u32, lo[c] = lo[c] + hi[c];)
KERNEL(k_add_u64, x[c] = x[c] + (u64)hi[c];)
KERNEL(k_xor_shift, lo[c] = (lo[c] << 3) ^ hi[c];)
__global__ void __launch_bounds__(256) k_fma_f64(u64* out, u64 seed) {
    double x[CHAINS]; for (int c = 0; c < CHAINS; c++) x[c] = (double)(seed + threadIdx.x + c);
    for (int i = 0; i < ITER; i++) { _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], 1.0000001, 0.5); }
    double a = 0; for (int c = 0; c < CHAINS; c++) a += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)a;
}
static inline __device__ u64 montyred(u64 lo, u64 hi) { u64 a = lo + (lo << 32); u64 e = a < lo; u64 b = a - (a >> 32) - e; u64 r = hi - b; return hi < b ? r - 0xFFFFFFFFull : r; }
static inline __device__ u64 mulmod(u64 a, u64 b) {
    u32 a0 = a, a1 = a >> 32, b0 = b, b1 = b >> 32; u64 p00 = (u64)a0 * b0; u64 m1 = (u64)a0 * b1 + (p00 >> 32); u64 m2 = (u64)a1 * b0 + (u32)m1;
    return montyred((m2 << 32) | (u32)p00, (u64)a1 * b1 + (m1 >> 32) + (m2 >> 32)); }
KERNEL(k_mulmod, x[c] = mulmod(x[c], x[c] | 5);)
#include "../../triton_vm_amd/csrc/field.h"   // the product's arithmetic (asm Montgomery reduction)
KERNEL(k_bfe_mul, x[c] = bfe_mul(x[c], x[c] | 5);)
KERNEL(k_bfe_add, x[c] = bfe_add(x[c], (u64)hi[c] << 20);)
KERNEL(k_bfe_sub, x[c] = bfe_sub(x[c], (u64)hi[c] << 20);)

// asm-volatile forms (nothing for the compiler to fold): the issue rate of ONE instruction type, 8 independent chains
#define ASM_KERNEL(name, text)                                                        \
__global__ void __launch_bounds__(256) name(u64* out, u64 seed) {                     \
    u32 lo[CHAINS], hi[CHAINS];                                                        \
    for (int c = 0; c < CHAINS; c++) { lo[c] = (u32)(seed + threadIdx.x * 977 + c * 131) | 1; hi[c] = (u32)(seed >> 7) | 3 | c; } \
    for (int i = 0; i < ITER; i++) {                                                   \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(text : "+v"(lo[c]) : "v"(hi[c]));  \
    }                                                                                  \
    u64 acc = 0; for (int c = 0; c < CHAINS; c++) acc ^= lo[c];                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
}
ASM_KERNEL(k_asm_add_u32, "v_add_u32 %0, %0, %1")
ASM_KERNEL(k_asm_xor_b32, "v_xor_b32 %0, %0, %1")
ASM_KERNEL(k_asm_mov_b32, "v_mov_b32 %0, %1")
ASM_KERNEL(k_asm_add_co, "v_add_co_u32 %0, vcc, %0, %1")
ASM_KERNEL(k_asm_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
ASM_KERNEL(k_asm_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
ASM_KERNEL(k_asm_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
__global__ void __launch_bounds__(256) k_asm_mad_u64_u32(u64* out, u64 seed) {
    u64 x[CHAINS]; u32 lo[CHAINS];
    for (int c = 0; c < CHAINS; c++) { x[c] = seed + threadIdx.x * 977 + c * 131; lo[c] = (u32)x[c] | 1; }
    u64 sink;
    for (int i = 0; i < ITER; i++) {
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile("v_mad_u64_u32 %0, %1, %2, %2, %0" : "+v"(x[c]), "=s"(sink) : "v"(lo[c]));
    }
    u64 acc = 0; for (int c = 0; c < CHAINS; c++) acc ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class K> void run(const char* name, K k, double ops_per_iter_chain) {
    u64* out; hipMalloc(&out, 256 * 2048 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, 1ull);
    hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, 2ull); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = 2048.0 * 256 * ITER * CHAINS * ops_per_iter_chain;
    printf("%-16s %8.3f ms  %8.2f Glane-ops/s  = %6.2f lane-ops/clk/CU @2.4GHz x256CU\n", name, ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / (2.4e9 * 256));
    hipFree(out);
}
int main() {
    run("mad_u64_u32", k_mad_u64_u32, 1); run("mul_lo_u32", k_mul_lo_u32, 1); run("mul_hi_u32", k_mul_hi_u32, 1);
    run("mad_u32_u24", k_mad_u32_u24, 1); run("add_u32", k_add_u32, 1); run("add_u64", k_add_u64, 1);
    run("xor_shift", k_xor_shift, 1); run("fma_f64", k_fma_f64, 1); run("mulmod(C, r01a)", k_mulmod, 1); run("bfe_mul(field.h)", k_bfe_mul, 1); run("bfe_add(field.h)", k_bfe_add, 1); run("bfe_sub(field.h)", k_bfe_sub, 1);
    run("asm v_add_u32", k_asm_add_u32, 1); run("asm v_xor_b32", k_asm_xor_b32, 1); run("asm v_mov_b32", k_asm_mov_b32, 1);
    run("asm v_add_co_u32", k_asm_add_co, 1); run("asm v_cndmask", k_asm_cndmask, 1); run("asm v_alignbyte", k_asm_alignbyte, 1);
    run("asm v_lshl_add", k_asm_lshl_add, 1); run("asm mad_u64_u32", k_asm_mad_u64_u32, 1);
    return 0;
}
