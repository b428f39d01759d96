// platform.h -- the one place that decides how kernels are compiled.
//
// Product build (hipcc --offload-arch=gfx950): real HIP.  There is no CPU fallback in the product.
// Test build (g++ -DTVM_EMU, tests/emu/): the same kernel sources run under a fiber emulator so the
// GPU-less container can check them against the oracle.
#pragma once

#ifdef TVM_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define TVM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define TVM_DYN_SMEM(T, name)                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char tvm_dyn_smem_[]; \
    T* name = reinterpret_cast<T*>(tvm_dyn_smem_)
#endif

// tvm_lds_barrier(): a workgroup barrier that orders LDS traffic only.  __syncthreads() carries a memory fence
// and therefore drains every outstanding GLOBAL load and store of the wavefront first (s_waitcnt vmcnt(0)).
// Where a barrier only separates LDS phases of one tile (the NTT kernels), waiting for the LDS counter is
// enough: loads already in flight for the next tile and stores of the previous one proceed.
#ifdef TVM_EMU
static inline void tvm_lds_barrier() { __syncthreads(); }
#else
static __device__ __forceinline__ void tvm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// tvm_wave_sync(): orders the LDS traffic of ONE wavefront -- the lanes of a wavefront exchange data through LDS words that
// only this wavefront touches (ntt.hip: one transform row per wavefront), so no workgroup barrier is needed: the LDS unit
// serves a wavefront's instructions in order; the wait makes the writes visible before the reads that follow are issued and
// keeps the compiler from moving LDS accesses across.
#ifdef TVM_EMU
static inline void tvm_wave_sync() { emu::sync_wave(); }
#else
static __device__ __forceinline__ void tvm_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif

// tvm_uniform(v): v is the same in every lane of the wavefront (an index derived from the wavefront's number); says so to the
// compiler, so that addresses built from it are scalar and the loads behind them scalar loads.
#ifdef TVM_EMU
#define tvm_uniform(v) (v)
#else
#define tvm_uniform(v) __builtin_amdgcn_readfirstlane(v)
#endif

// tvm_opaque(v): v, through a move the compiler cannot see through -- what is computed from the result is computed where it is
// used, not hoisted out of the enclosing loop into registers that are then spilled.
#ifdef TVM_EMU
#define tvm_opaque(v) (v)
#else
static __device__ __forceinline__ int tvm_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
#endif

// Streaming accesses (non-temporal: the LDE's intermediates and table, written once and read a whole pass later, or read
// once): measured -3 % on the LDE at 2^20 rows (main table 46.8 -> 45.2 ms); nothing for the VALU-bound row hashing.
#if defined(TVM_EMU)
#define TVM_STORE_STREAM(ptr, value) (*(ptr) = (value))
#define TVM_LOAD_STREAM(ptr) (*(ptr))
#define TVM_STORE_STREAM_X2(ptr, a, b) ((ptr)[0] = (a), (ptr)[1] = (b))
#else
#define TVM_STORE_STREAM(ptr, value) __builtin_nontemporal_store((value), (ptr))
#define TVM_LOAD_STREAM(ptr) __builtin_nontemporal_load(ptr)
typedef unsigned long tvm_u64v2 __attribute__((ext_vector_type(2)));
// two adjacent words (16-byte aligned) in one streaming store
#define TVM_STORE_STREAM_X2(ptr, a, b) __builtin_nontemporal_store(tvm_u64v2{(a), (b)}, (tvm_u64v2*)(ptr))
#endif

#include <cstdint>

typedef uint64_t u64;
typedef uint32_t u32;

struct alignas(16) u64x2 {   // two words moved by one 16-byte access (global_load_dwordx4)
    u64 x, y;
};

#define TVM_HD __host__ __device__ __forceinline__
#define TVM_D __device__ __forceinline__
