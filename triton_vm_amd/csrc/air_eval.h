// air_eval.h -- what the generated parts of the AIR evaluator (air_gen_*.hip, written by
// tools/air/export.py) are compiled against: the kernel arguments, the row addressing and the
// accumulation of  sum_k w_k * c_k  per section.
//
// Replaces the build-time generated MasterAuxTable::evaluate_{initial,consistency,transition,terminal}
// _constraints of the reference (generator: /root/reference/triton-constraint-builder/src/codegen.rs:141-367)
// inside all_quotients_combined (/root/reference/triton-vm/src/table/master_table.rs:1302-1359).
//
// One lane evaluates one quotient-domain row; a wavefront is 64 consecutive rows, so every column it
// reads is four full 128-byte lines of the row-block-major table (context.h).  The code is straight-line
// and identical for every wavefront (~2 MB of instructions per row batch, streamed through the
// instruction cache; wavefronts that start together stay close enough to share the fetches --
// forcing lock-step with workgroup barriers measured 6 % slower).
#pragma once
#include "context.h"

namespace tvm {

#ifndef AIR_BLOCK
#define AIR_BLOCK 512
#endif
// challenges and weights are the same for every lane and never written while a part runs: reading them
// through the constant address space makes the loads scalar (s_load) and exempt from AIR_SYNC's clobber
#ifdef TVM_EMU
#define AIR_UNIFORM
#else
#define AIR_UNIFORM __attribute__((address_space(4)))
#endif

struct AirArgs {
    const u64* main_table;   // row-block-major [rows][main_w]
    const u64* aux_table;    // row-block-major [rows][aux_w]
    u64 main_w, aux_w;       // words per table row
    u64 stride;              // table rows per quotient-domain row
    u64 q_len, unit;         // |quotient domain|, |quotient domain| / |trace domain|
    const u64* challenges;   // 63 XFE
    const u64* weights;      // 604 XFE
    const u64* zinv;         // [4][q_len] zerofier inverses (k_zerofier_inverses)
    u64* out;                // q_len XFE
};

// sum_k w_k * c_k, kept as an XFE
struct AirAcc {
    xfe v;
};
TVM_D AirAcc air_acc_zero() {
    AirAcc a;
    a.v = xfe_zero();
    return a;
}
TVM_D void air_acc_b(AirAcc& a, xfe w, u64 c) { a.v = xfe_add(a.v, xfe_mul_bfe(w, c)); }
TVM_D void air_acc_x(AirAcc& a, xfe w, xfe c) { a.v = xfe_add(a.v, xfe_mul(w, c)); }
TVM_D xfe air_acc_value(const AirAcc& a) { return a.v; }
TVM_D xfe xfe_bfe_sub(u64 b, xfe x) { return xfe_make(bfe_sub(b, x.c0), bfe_neg(x.c1), bfe_neg(x.c2)); }

// Addressing: a workgroup covers AIR_BLOCK consecutive quotient-domain rows, so every table cell it reads
// is  (uniform block base + column offset)  +  (a 32-bit per-lane byte offset): the first term lives in
// SGPRs (scalar adds, free), the second in ONE VGPR per row kind -- global_load's saddr + voffset form --
// instead of a 64-bit VGPR pointer per 8 KiB window of columns (~50 VGPRs in the transition constraints).
// The "next" row of the last rows of the domain is read from the table's wrap rows (kernels.h: tvm_table).
#define AIR_PROLOGUE()                                                                                          \
    const u64 first_ = (u64)blockIdx.x * AIR_BLOCK;                                                             \
    const u64 i_raw_ = first_ + threadIdx.x;                                                                    \
    const bool active_ = i_raw_ < a.q_len;                                                                      \
    const u64 i_ = active_ ? i_raw_ : a.q_len - 1; /* keep every lane on the barrier path */                    \
    const u64 blk_row_ = first_ * a.stride; /* a multiple of TVM_RB */                                          \
    const u32 loc_cur_ = (u32)((i_ - first_) * a.stride), loc_next_ = loc_cur_ + (u32)(a.unit * a.stride);      \
    const char* mb_ = (const char*)(a.main_table + (blk_row_ >> TVM_RB_LOG) * a.main_w * TVM_RB);               \
    const char* ab_ = (const char*)(a.aux_table + (blk_row_ >> TVM_RB_LOG) * a.aux_w * TVM_RB);                 \
    const u32 mc_ = 8u * ((loc_cur_ >> TVM_RB_LOG) * (u32)a.main_w * TVM_RB + (loc_cur_ & (TVM_RB - 1)));       \
    const u32 mn_ = 8u * ((loc_next_ >> TVM_RB_LOG) * (u32)a.main_w * TVM_RB + (loc_next_ & (TVM_RB - 1)));     \
    const u32 ac_ = 8u * ((loc_cur_ >> TVM_RB_LOG) * (u32)a.aux_w * TVM_RB + (loc_cur_ & (TVM_RB - 1)));        \
    const u32 an_ = 8u * ((loc_next_ >> TVM_RB_LOG) * (u32)a.aux_w * TVM_RB + (loc_next_ & (TVM_RB - 1)));      \
    const AIR_UNIFORM u64* ch_ = (const AIR_UNIFORM u64*)a.challenges;                                          \
    const AIR_UNIFORM u64* wt_ = (const AIR_UNIFORM u64*)a.weights;                                             \
    xfe quot = xfe_zero()

#define AIR_CELL(base, lane_off, word) (*(const u64*)((base) + (size_t)(word) * (TVM_RB * 8) + (size_t)(lane_off)))
#define MC(c) AIR_CELL(mb_, mc_, c)
#define MN(c) AIR_CELL(mb_, mn_, c)
#define AC(c) xfe_make(AIR_CELL(ab_, ac_, 3 * (c)), AIR_CELL(ab_, ac_, 3 * (c) + 1), AIR_CELL(ab_, ac_, 3 * (c) + 2))
#define AN(c) xfe_make(AIR_CELL(ab_, an_, 3 * (c)), AIR_CELL(ab_, an_, 3 * (c) + 1), AIR_CELL(ab_, an_, 3 * (c) + 2))
#define CH(k) xfe_make(ch_[3 * (k)], ch_[3 * (k) + 1], ch_[3 * (k) + 2])
#define W(k) xfe_make(wt_[3 * (k)], wt_[3 * (k) + 1], wt_[3 * (k) + 2])
#define ZINV(s) (a.zinv[(u64)(s) * a.q_len + i_])
// AIR_SYNC: a compiler-level memory barrier -- loads of table cells are neither merged nor moved across
// it, which bounds the live range of every loaded value to one segment.
// AIR_PIN_*: an empty asm that "modifies" a value, so its computation cannot sink below this point.
#ifdef TVM_EMU
#define AIR_SYNC() __syncthreads()
#define AIR_CUT() (void)0
#define AIR_PIN_B(v) (void)0
#define AIR_PIN_X(v) (void)0
#else
#ifdef AIR_BARRIER  // experiment: lock-step the wavefronts of a workgroup (measured 6 % slower on MI355X)
#define AIR_SYNC()                      \
    do {                                \
        asm volatile("" ::: "memory");  \
        __builtin_amdgcn_s_barrier();   \
        asm volatile("" ::: "memory");  \
    } while (0)
#else
#define AIR_SYNC() asm volatile("" ::: "memory")
#endif
// AIR_CUT: a never-taken uniform branch the compiler cannot fold.  It ends the basic block, which keeps
// hipcc's per-block passes (DAG combiner, machine scheduler: super-linear) off a 50k-instruction block --
// minutes of compile time otherwise.  Placed every few segments only: the register allocation across
// many small blocks is measurably worse (~+50 VGPRs) than inside one block.
#define AIR_CUT()                                     \
    do {                                              \
        u32 never_ = 0;                               \
        asm volatile("" : "+s"(never_) : : "memory"); \
        if (never_) return;                           \
    } while (0)
#define AIR_PIN_B(v) asm volatile("" : "+v"(v))
#define AIR_PIN_X(v) asm volatile("" : "+v"((v).c0), "+v"((v).c1), "+v"((v).c2))
#endif
#define AIR_PIN_ACC(acc) AIR_PIN_X((acc).v)

#define AIR_EPILOGUE(accumulate)                                        \
    if (active_) {                                                      \
        u64* o_ = a.out + 3 * i_;                                       \
        if (accumulate) {                                               \
            o_[0] = bfe_add(o_[0], quot.c0);                            \
            o_[1] = bfe_add(o_[1], quot.c1);                            \
            o_[2] = bfe_add(o_[2], quot.c2);                            \
        } else {                                                        \
            o_[0] = quot.c0;                                            \
            o_[1] = quot.c1;                                            \
            o_[2] = quot.c2;                                            \
        }                                                               \
    }

}  // namespace tvm
