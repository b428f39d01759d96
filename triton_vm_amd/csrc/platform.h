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

// tvm_pair_sync(): TWO wavefronts that share a transform row (2048-point rows: ntt.hip) meet WITHOUT stopping the other fourteen
// of the workgroup.  Each of the two owns a flag word in LDS; arriving, a wavefront -- whose own LDS writes are issued before it, and
// the LDS serves a wavefront's instructions in order -- stores the epoch into its flag and polls its partner's until that one shows
// the epoch too.  Both wavefronts are resident (one workgroup), so the poll terminates; everything the partner wrote before its flag is
// visible after it.  Round 6: s_barrier between the butterfly groups of such a row made all sixteen wavefronts of the CU's one
// workgroup wait for the slowest pair, five times per coset.  The emulation's fibers cannot spin: there it is the workgroup barrier
// (every wavefront executes the same sequence of synchronisations, so that is equivalent).
#define TVM_PAIR_FLAG_WORDS 64   // 32-bit words per wavefront's flag block (one per lane: ds_*_addtid)
#ifndef TVM_PAIR_SYNC
#define TVM_PAIR_SYNC 1   // 0: the workgroup barrier instead (A/B, profiles/r06_*)
#endif
#ifdef TVM_EMU
static inline void tvm_pair_sync(unsigned* flags, int mine, int partner, unsigned epoch, int& row_lane, int row_lane_base) {
    (void)flags, (void)mine, (void)partner, (void)epoch, (void)row_lane, (void)row_lane_base;
    __syncthreads();
}
#elif !TVM_PAIR_SYNC
static __device__ __forceinline__ void tvm_pair_sync(unsigned* flags, int mine, int partner, unsigned epoch, int& row_lane, int row_lane_base) {
    (void)flags, (void)mine, (void)partner, (void)epoch, (void)row_lane, (void)row_lane_base;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#else
static __device__ __forceinline__ void tvm_pair_sync(unsigned* flags, int mine, int partner, unsigned epoch, int& row_lane, int row_lane_base) {
    // One asm block that needs NO vector register of its own: the kernels that call it run at their register cap, and a temporary
    // cost k_lde_pass2_fused<11> 80 more bytes of scratch.  The flags are blocks of 64 words (TVM_PAIR_FLAG_WORDS per wavefront)
    // written and read with the address-from-lane forms of the LDS instructions (M0 + 4 * lane: no address register); the epoch
    // comes from a scalar register; the one register the loads need is BORROWED: `row_lane`, a value the caller holds anyway and that
    // is recomputed from the hardware lane number afterwards (row_lane == row_lane_base + lane, row_lane_base uniform); and the poll
    // loop is not part of the kernel's control flow.  `flags` must lie in the first 64 KB of LDS (M0 holds 16 address bits); `mine`,
    // `partner`, `epoch` are uniform over the wavefront.
    typedef __attribute__((address_space(3))) unsigned* lds_words;
    const unsigned base = (unsigned)(unsigned long)(lds_words)flags;   // LDS byte address of the flag blocks
    const unsigned my_block = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 4u * TVM_PAIR_FLAG_WORDS * (unsigned)mine));
    const unsigned partner_block = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 4u * TVM_PAIR_FLAG_WORDS * (unsigned)partner));
    const unsigned ep = (unsigned)__builtin_amdgcn_readfirstlane((int)epoch);
    const int rb = __builtin_amdgcn_readfirstlane(row_lane_base);
    unsigned got_s;
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"              // (own LDS traffic issued so far: complete before the flag)
                 "s_mov_b32 m0, %[mb]\n\t"
                 "v_mov_b32 %[t], %[eps]\n\t"
                 "ds_write_addtid_b32 %[t]\n\t"          // my block <- epoch (every lane its word)
                 "s_mov_b32 m0, %[pb]\n\t"
                 "s_nop 0\n\t"                            // (gfx9 hazard: one wait state between a write of M0 and an add-TID LDS instruction)
                 ".Ltvm_pair_poll%=:\n\t"
                 "ds_read_addtid_b32 %[t]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_readfirstlane_b32 %[ts], %[t]\n\t"
                 "s_cmp_lt_u32 %[ts], %[eps]\n\t"
                 "s_cbranch_scc0 .Ltvm_pair_done%=\n\t"
                 "s_sleep 1\n\t"
                 "s_branch .Ltvm_pair_poll%=\n\t"
                 ".Ltvm_pair_done%=:\n\t"
                 "v_mbcnt_lo_u32_b32 %[t], -1, 0\n\t"    // the borrowed register gets its value back: base + lane
                 "v_mbcnt_hi_u32_b32 %[t], -1, %[t]\n\t"
                 "v_add_u32_e32 %[t], %[rb], %[t]"
                 : [t] "+v"(row_lane), [ts] "=&s"(got_s)
                 : [mb] "s"(my_block), [pb] "s"(partner_block), [eps] "s"(ep), [rb] "s"(rb)
                 : "memory", "scc");
}
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
