// occupancy.hip -- throughput of the field multiplication as a function of wavefronts per SIMD and of the
// number of independent chains per lane: does a dependent chain of carry instructions issue back to back?
// Dynamic LDS caps the residency: 160 KiB -> 1 workgroup (256 items) per CU = 1 wave/SIMD, 80 KiB -> 2, 40 KiB -> 4.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../triton_vm_amd/csrc/platform.h"
#include "../../triton_vm_amd/csrc/field.h"
#define ITER 2048

template <int CHAINS, int OP>
__global__ void __launch_bounds__(256) k_chain(u64* out, u64 seed) {
    extern __shared__ u64 dummy[];
    u64 x[CHAINS];
    u32 y[CHAINS];
    for (int c = 0; c < CHAINS; c++) { x[c] = seed + threadIdx.x * 977 + c * 131; y[c] = (u32)x[c] * 3; }
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) x[c] = bfe_mul(x[c], x[c] | 5);
            if (OP == 1) x[c] = bfe_add(x[c], x[c] >> 3);
            if (OP == 2) x[c] = (u64)(u32)x[c] * (u32)(x[c] >> 32) + x[c];
            if (OP == 3) x[c] = ((x[c] << 3) ^ (x[c] >> 5)) + 1;
            if (OP == 4) {  // one multiply-add and three independent 32-bit logic operations: do they hide behind it?
                x[c] = (u64)(u32)x[c] * (u32)(x[c] >> 32) + x[c];
                y[c] = (y[c] ^ 0x9e3779b9u) + 0x7f4a7c15u;
                y[c] = (y[c] ^ 0x85ebca6bu) + 0xc2b2ae35u;
                y[c] = (y[c] ^ 0x27d4eb2fu) + 0x165667b1u;
            }
            if (OP == 5) {  // the logic operations alone
                y[c] = (y[c] ^ 0x9e3779b9u) + 0x7f4a7c15u;
                y[c] = (y[c] ^ 0x85ebca6bu) + 0xc2b2ae35u;
                y[c] = (y[c] ^ 0x27d4eb2fu) + 0x165667b1u;
            }
        }
    }
    u64 acc = 0;
    for (int c = 0; c < CHAINS; c++) acc ^= x[c] + y[c];
    if (acc == 12345) dummy[threadIdx.x] = acc;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int CHAINS, int OP> void run(const char* name, size_t lds, int waves) {
    u64* out; hipMalloc(&out, 256 * 4096 * 8);
    auto k = k_chain<CHAINS, OP>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), lds, 0, out, 1ull);
    hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(4096), dim3(256), lds, 0, out, 2ull); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lane_ops = 4096.0 * 256 * ITER * CHAINS;
    printf("%-10s chains %d  waves/SIMD %d  %8.3f ms  %6.2f lane-ops/clk/CU\n", name, CHAINS, waves, ms, lane_ops / (ms * 1e-3) / (2.4e9 * 256));
    hipFree(out);
}
template <int CHAINS, int OP> void sweep(const char* name) {
    run<CHAINS, OP>(name, 160 * 1024 - 256, 1);
    run<CHAINS, OP>(name, 80 * 1024 - 256, 2);
    run<CHAINS, OP>(name, 40 * 1024 - 256, 4);
    run<CHAINS, OP>(name, 0, 8);
}
int main() {
    sweep<1, 0>("bfe_mul"); sweep<2, 0>("bfe_mul"); sweep<4, 0>("bfe_mul");
    sweep<1, 1>("bfe_add"); sweep<4, 1>("bfe_add");
    sweep<1, 2>("mad64"); sweep<4, 2>("mad64");
    sweep<1, 3>("shift_xor"); sweep<4, 3>("shift_xor");
    sweep<4, 4>("mad+6alu"); sweep<4, 5>("6alu");
    return 0;
}
