// field_forms.hip -- issue cost of candidate forms of the field operations the NTT butterflies are made of, at the
// occupancy of the LDE kernels (4 wavefronts per SIMD: 256-thread workgroups with 40 KB of LDS each) and at 8 per SIMD.
// Each kernel runs ITER iterations over 8 independent values per lane; prints nanoseconds per lane-operation-wave, i.e.
// SIMD cycles per wave64 operation.   build: hipcc --offload-arch=gfx950 -O3 -I include -I triton_vm_amd/csrc tools/ubench/field_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ntt_shift.h"
using namespace tvm;
#define ITER 2048
#define CH 8

template <class F>
__global__ void __launch_bounds__(256) k_run(u64* out, u64 seed, F f) {
    extern __shared__ u64 lds[];
    u64 x[CH], y[CH];
    for (int c = 0; c < CH; c++) {
        x[c] = (seed * 0x9E3779B97F4A7C15ull + threadIdx.x * 977 + c * 131) % TVM_P;
        y[c] = (x[c] * 31 + 7) % TVM_P;
    }
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) f(x[c], y[c]);
    }
    u64 acc = 0;
    for (int c = 0; c < CH; c++) acc ^= x[c] ^ y[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (threadIdx.x == 9999 ? lds[0] : 0);
}

// ---- candidate forms ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 add_c(u64 a, u64 b) { u64 s = a + b; return (s < a || s >= TVM_P) ? s + TVM_EPS : s; }
__device__ __forceinline__ u64 sub_c(u64 a, u64 b) { u64 d = a - b; return (a < b) ? d - TVM_EPS : d; }
struct FAdd { __device__ void operator()(u64& x, u64& y) const { x = bfe_add(x, y); } };
struct FAddC { __device__ void operator()(u64& x, u64& y) const { x = add_c(x, y); } };
struct FSub { __device__ void operator()(u64& x, u64& y) const { x = bfe_sub(x, y); } };
struct FSubC { __device__ void operator()(u64& x, u64& y) const { x = sub_c(x, y); } };
struct FMul { __device__ void operator()(u64& x, u64& y) const { x = bfe_mul(x, y); } };
struct FBfly { __device__ void operator()(u64& x, u64& y) const { const u64 v = bfe_mul(y, 0x1234567812345678ull); const u64 u = x; x = bfe_add(u, v); y = bfe_sub(u, v); } };
struct FBflyC { __device__ void operator()(u64& x, u64& y) const { const u64 v = bfe_mul(y, 0x1234567812345678ull); const u64 u = x; x = add_c(u, v); y = sub_c(u, v); } };
template <int S> struct FPow { __device__ void operator()(u64& x, u64& y) const { x = bfe_mul_pow2<S>(x); } };
template <int S> struct FBflyPow { __device__ void operator()(u64& x, u64& y) const { bfe_butterfly_pow2<S>(x, y); } };
struct FBflyUnit { __device__ void operator()(u64& x, u64& y) const { const u64 u = x, v = y; x = bfe_add(u, v); y = bfe_sub(u, v); } };

template <class F>
void run(const char* name, F f, double ops) {
    u64* out;
    hipMalloc(&out, 256 * 4096 * 8);
    for (int lds_kb : {40, 0}) {   // 40 KB per 256-thread workgroup: 4 workgroups per CU = 4 wavefronts per SIMD; 0: up to 8
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipFuncSetAttribute((const void*)k_run<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        hipLaunchKernelGGL(k_run<F>, dim3(4096), dim3(256), lds_kb * 1024, 0, out, 1ull, f);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_run<F>, dim3(4096), dim3(256), lds_kb * 1024, 0, out, 2ull, f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double wave_ops = 4096.0 * 4 * ITER * CH * ops;             // wave64-level operations
        const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;              // 256 CUs x 4 SIMDs
        printf("%-22s %s waves/SIMD  %8.3f ms  %7.2f SIMD cycles per wave-operation\n", name, lds_kb ? "4" : "8", ms, simd_cycles / wave_ops);
    }
    hipFree(out);
}
int main() {
    run("bfe_add (asm)", FAdd(), 1); run("add (C, 64-bit ops)", FAddC(), 1);
    run("bfe_sub (asm)", FSub(), 1); run("sub (C, 64-bit ops)", FSubC(), 1);
    run("bfe_mul", FMul(), 1);
    run("butterfly mul+add+sub", FBfly(), 1); run("butterfly, C add/sub", FBflyC(), 1);
    run("butterfly unit", FBflyUnit(), 1);
    run("mul 2^12 (C)", FPow<12>(), 1); run("mul 2^24 (C)", FPow<24>(), 1); run("mul 2^36 (C)", FPow<36>(), 1);
    run("mul 2^48 (C)", FPow<48>(), 1); run("mul 2^60 (C)", FPow<60>(), 1); run("mul 2^72 (C)", FPow<72>(), 1); run("mul 2^84 (C)", FPow<84>(), 1);
    run("butterfly 2^48", FBflyPow<48>(), 1); run("butterfly 2^24", FBflyPow<24>(), 1); run("butterfly 2^72", FBflyPow<72>(), 1);
    return 0;
}
