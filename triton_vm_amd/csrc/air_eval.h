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
#define AIR_BLOCK 256
#endif
#if AIR_BLOCK == 128
#define TVM_AIR_BLOCK_LOG 7
#elif AIR_BLOCK == 256
#define TVM_AIR_BLOCK_LOG 8
#elif AIR_BLOCK == 512
#define TVM_AIR_BLOCK_LOG 9
#else
#error "AIR_BLOCK is 128, 256 or 512"
#endif
// AIR_MIN_WAVES: wavefronts per SIMD the register allocation must leave room for.  Two (256 registers per lane):
// with the table cells of the next segment in flight some parts would otherwise take > 256 and run alone.
#ifndef AIR_MIN_WAVES
#define AIR_MIN_WAVES 2
#endif
#define AIR_LAUNCH_BOUNDS __launch_bounds__(AIR_BLOCK, AIR_MIN_WAVES)
// challenges and weights are the same for every lane and never written while a part runs: reading them
// through the constant address space makes the loads scalar (s_load) and exempt from AIR_SYNC's clobber
#ifdef TVM_EMU
#define AIR_UNIFORM
#else
#define AIR_UNIFORM __attribute__((address_space(4)))
#endif

struct AirArgs {
    const u64* main_table;   // row-block-major over storage rows, main_w words per row
    const u64* aux_table;    // likewise, aux_w words per row
    u64 main_w, aux_w;       // words per table row
    u64 q_len;               // |quotient domain|
    // Where the quotient domain's rows are (context.h: TabLayout): the quotient domain is X' = q_len / N cosets of the trace
    // domain, coset kq of it = coset kq * (X / X') of the table; row j = j1 + n2*j2 of a coset is storage row
    // coset base + j1*n1 + j2 and its successor j + 1 is n1 storage rows further (successor blocks).
    u64 n1;                  // rows per block
    int log_n1, log_n2, log_n;  // n1, n2 = blocks per coset, N = n1 * n2 = |trace domain|
    int log_xq;              // X' = 2^log_xq
    u64 coset_rows;          // storage rows from one coset of the quotient domain to the next
    int tiled;               // a workgroup is (AIR_BLOCK / 64 blocks) x (64 rows) of one coset, one block per wavefront
    const u64* challenges;   // 63 XFE
    const u64* weights;      // 604 XFE
    const u64* zinv;         // [4][q_len] zerofier inverses in WORK order (k_zerofier_inverses)
    u64* out;                // q_len XFE in WORK order (k_air_scatter puts them into domain order)
    int accumulate;          // 0: this part stores its share of the quotient, 1: it adds to what is there
};

// Work item t = workgroup * AIR_BLOCK + lane of the workgroup  ->  its row: `s_base` (uniform over the workgroup, a multiple
// of 16) + `rel` storage rows, and the row's index in the quotient domain.
// Tiled form (tables with n1 >= 64): wavefront w of a workgroup takes 64 consecutive rows of block g*WPB + w -- every
// column it reads is four full 128-byte lines, and the successor rows of wavefront w are the current rows of wavefront
// w + 1: WPB + 1 blocks of cells are fetched for WPB blocks of work (the successors of the last block come from the next
// workgroup's lines, which consecutive workgroups g, g + 1 have in flight together).
TVM_D void air_locate(const AirArgs& a, u64 block, int tid, u64& t, bool& active, u64& s_base, u32& rel, u64& index) {
    constexpr int WPB_LOG = TVM_AIR_BLOCK_LOG - 6;
    t = block * AIR_BLOCK + (u64)tid;
    active = t < a.q_len;
    u64 kq, j1, j2;
    if (a.tiled) {  // (q_len is a multiple of AIR_BLOCK: every lane is active)
        const int log_tiles = a.log_n - TVM_AIR_BLOCK_LOG;     // tiles per coset
        const int log_groups = a.log_n2 - WPB_LOG;             // groups of WPB blocks per coset
        kq = block >> log_tiles;
        const u64 tile = block & ((1ull << log_tiles) - 1);
        const u64 g = tile & ((1ull << log_groups) - 1), jb = tile >> log_groups;
        j1 = (g << WPB_LOG) + (u64)(tid >> 6);
        j2 = jb * 64 + (u64)(tid & 63);
        s_base = kq * a.coset_rows + (g << WPB_LOG) * a.n1 + jb * 64;
        rel = (u32)((u64)(tid >> 6) * a.n1) + (u32)(tid & 63);
    } else {
        const u64 tt = active ? t : a.q_len - 1;  // keep every lane on the common path
        kq = tt >> a.log_n;
        const u64 r = tt & ((1ull << a.log_n) - 1);
        j1 = r >> a.log_n1;
        j2 = r & (a.n1 - 1);
        s_base = 0;
        rel = (u32)(kq * a.coset_rows + r);
    }
    index = ((j1 + (j2 << a.log_n2)) << a.log_xq) + kq;
}

// The accumulator of  sum_k w_k * c_k  for one group of constraints.
#ifdef TVM_FIELD_ASM
// Device form: the 128-bit products are summed unreduced, one Montgomery reduction per coefficient and group
// instead of one per product.  With w = (w0, w1, w2), x = (x0, x1, x2) and X^3 = X - 1:
//   w*x = (d0 - d3) + (d1 + d3 - d4) X + (d2 + d4) X^2,   d0 = w0x0, d1 = w0x1 + w1x0, d2 = w0x2 + w1x1 + w2x0,
//   d3 = w1x2 + w2x1, d4 = w2x2,
// so five 160-bit sums D0..D4 (a group has <= 25 constraints: < 2^135).  A product-accumulate is 4 v_mad_u64_u32
// + a 5-limb carry chain (13 VALU instructions; reduce-then-add is 24), three of them interleaved at a time.
struct Acc160 {
    u32 l0, l1, l2, l3, l4;
};
// three product-accumulates into three DIFFERENT sums, carry chains interleaved (field.h: no wait states)
#define AIR_Q1(S, C) "v_add_co_u32_e64 %[l0" #S "], " C ", %[l0" #S "], %[x0" #S "]\n\t"
#define AIR_Q2(S, C) "v_addc_co_u32_e64 %[l1" #S "], " C ", %[l1" #S "], %[x1" #S "], " C "\n\t"
#define AIR_Q3(S, C) "v_addc_co_u32_e64 %[l2" #S "], " C ", %[l2" #S "], %[x2" #S "], " C "\n\t"
#define AIR_Q4(S, C) "v_addc_co_u32_e64 %[l3" #S "], " C ", %[l3" #S "], %[x3" #S "], " C "\n\t"
#define AIR_Q5(S, C) "v_addc_co_u32_e64 %[l4" #S "], " C ", 0, %[l4" #S "], " C "\n\t"
#define AIR_Q_IO(S, A) [l0##S] "+v"(A.l0), [l1##S] "+v"(A.l1), [l2##S] "+v"(A.l2), [l3##S] "+v"(A.l3), [l4##S] "+v"(A.l4)
#define AIR_Q_IN(S, t, v, w) [x0##S] "v"((u32)(t)), [x1##S] "v"((u32)(v)), [x2##S] "v"((u32)(w)), [x3##S] "v"((u32)((w) >> 32))
#define AIR_MAC_PARTIALS(a, b, t, u, v, w)                                       \
    const u64 t = (u64)(u32)(a) * (u32)(b);                                      \
    const u64 u = (u64)(u32)(a) * (u32)((b) >> 32) + (t >> 32);                  \
    const u64 v = (u64)(u32)((a) >> 32) * (u32)(b) + (u32)u;                     \
    const u64 w = (u64)(u32)((a) >> 32) * (u32)((b) >> 32) + ((u >> 32) + (v >> 32))
TVM_D void acc160_mac3(Acc160& A, u64 a0, u64 b0, Acc160& B, u64 a1, u64 b1, Acc160& C, u64 a2, u64 b2) {
#if TVM_MUL_CARRY_FORM
    // the partial products of field.h's carry-out form: x = (a1*b1 + (c : v1)) : v0 : t0 with (c : v) = a1*b0 + u
    TVM_MUL_LOW(a0, b0, ta, ua);
    TVM_MUL_LOW(a1, b1, tb, ub);
    TVM_MUL_LOW(a2, b2, tc, uc);
    u64 va, vb, vc, cb, cc;
    u32 ka, kb, kc;
    asm(TVM_3WAY(TVM_MV) TVM_3WAY(TVM_MC)
        : TVM_MID_OUT(a, va, ka), TVM_MID_OUT(b, vb, kb), TVM_MID_OUT(c, vc, kc), [cb] "=&s"(cb), [cc] "=&s"(cc)
        : TVM_MID_IN(a, a0, b0, ua), TVM_MID_IN(b, a1, b1, ub), TVM_MID_IN(c, a2, b2, uc)
        : "vcc");
    TVM_MUL_HIGH(a0, b0, va, ka, wa);
    TVM_MUL_HIGH(a1, b1, vb, kb, wb);
    TVM_MUL_HIGH(a2, b2, vc, kc, wc);
#else
    AIR_MAC_PARTIALS(a0, b0, ta, ua, va, wa);
    AIR_MAC_PARTIALS(a1, b1, tb, ub, vb, wb);
    AIR_MAC_PARTIALS(a2, b2, tc, uc, vc, wc);
    u64 cb, cc;
#endif
    asm(TVM_3WAY(AIR_Q1) TVM_3WAY(AIR_Q2) TVM_3WAY(AIR_Q3) TVM_3WAY(AIR_Q4) TVM_3WAY(AIR_Q5)
        : AIR_Q_IO(a, A), AIR_Q_IO(b, B), AIR_Q_IO(c, C), [cb] "=&s"(cb), [cc] "=&s"(cc)
        : AIR_Q_IN(a, ta, va, wa), AIR_Q_IN(b, tb, vb, wb), AIR_Q_IN(c, tc, vc, wc)
        : "vcc");
}
// (l4 : l3 : l2 : l1 : l0) * 2^-64 mod p, canonical: fold the top 96 bits modulo p, then one Montgomery reduction
TVM_D u64 acc160_value(const Acc160& A) {
    const u64 lo = ((u64)A.l1 << 32) | A.l0, mid = ((u64)A.l3 << 32) | A.l2;
    u64 t = (u64)A.l4 * TVM_EPS;  // l4 * 2^64 mod p, l4 < 2^32
    u64 r = mid + t;
    if (r < t) r += TVM_EPS;
    if (r >= TVM_P) r -= TVM_P;
    return bfe_montyred(lo, r);
}
struct AirAcc {
    Acc160 d[5];
};
TVM_D AirAcc air_acc_zero() {
    AirAcc a;
#pragma unroll
    for (int i = 0; i < 5; i++) a.d[i].l0 = a.d[i].l1 = a.d[i].l2 = a.d[i].l3 = a.d[i].l4 = 0;
    return a;
}
TVM_D void air_acc_b(AirAcc& a, xfe w, u64 c) { acc160_mac3(a.d[0], w.c0, c, a.d[1], w.c1, c, a.d[2], w.c2, c); }
TVM_D void air_acc_x(AirAcc& a, xfe w, xfe x) {
    // the nine products in three rounds of three different sums
    acc160_mac3(a.d[0], w.c0, x.c0, a.d[1], w.c0, x.c1, a.d[2], w.c0, x.c2);
    acc160_mac3(a.d[1], w.c1, x.c0, a.d[2], w.c1, x.c1, a.d[3], w.c1, x.c2);
    acc160_mac3(a.d[2], w.c2, x.c0, a.d[3], w.c2, x.c1, a.d[4], w.c2, x.c2);
}
TVM_D xfe air_acc_value(const AirAcc& a) {
    const u64 d0 = acc160_value(a.d[0]), d1 = acc160_value(a.d[1]), d2 = acc160_value(a.d[2]);
    const u64 d3 = acc160_value(a.d[3]), d4 = acc160_value(a.d[4]);
    return xfe_make(bfe_sub(d0, d3), bfe_sub(bfe_add(d1, d3), d4), bfe_add(d2, d4));
}
#define AIR_PIN_ACC(acc)                                                                                           \
    do {                                                                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; i_++)                                                           \
            asm volatile("" : "+v"((acc).d[i_].l0), "+v"((acc).d[i_].l1), "+v"((acc).d[i_].l2), "+v"((acc).d[i_].l3), \
                              "+v"((acc).d[i_].l4));                                                               \
    } while (0)
#else
// Emulator form: reduce every product (same value: the arithmetic is exact).
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
#define AIR_PIN_ACC(acc) (void)0
#endif
TVM_D xfe xfe_bfe_sub(u64 b, xfe x) { return xfe_bfe_minus(b, x); }

// Addressing: every table cell a workgroup reads is  (uniform row-block base + column offset)  +  (a 32-bit per-lane byte
// offset): the first term lives in SGPRs (scalar adds, free), the second in ONE VGPR per row kind -- global_load's saddr +
// voffset form -- instead of a 64-bit VGPR pointer per 8 KiB window of columns (~50 VGPRs in the transition constraints).
// The "next" row is n1 storage rows behind the current one (context.h: successor blocks), never a wrap-around.
#define AIR_PROLOGUE()                                                                                          \
    bool active_;                                                                                               \
    u64 t_, s_base_, dom_index_;                                                                                \
    u32 loc_cur_;                                                                                               \
    air_locate(a, (u64)blockIdx.x, (int)threadIdx.x, t_, active_, s_base_, loc_cur_, dom_index_);               \
    (void)dom_index_;                                                                                           \
    const u32 loc_next_ = loc_cur_ + (u32)a.n1;                                                                 \
    const char* mb_ = (const char*)(a.main_table + (s_base_ >> TVM_RB_LOG) * a.main_w * TVM_RB);                \
    const char* ab_ = (const char*)(a.aux_table + (s_base_ >> TVM_RB_LOG) * a.aux_w * TVM_RB);                  \
    const u32 mc_ = 8u * ((loc_cur_ >> TVM_RB_LOG) * (u32)a.main_w * TVM_RB + (loc_cur_ & (TVM_RB - 1)));       \
    const u32 mn_ = 8u * ((loc_next_ >> TVM_RB_LOG) * (u32)a.main_w * TVM_RB + (loc_next_ & (TVM_RB - 1)));     \
    const u32 ac_ = 8u * ((loc_cur_ >> TVM_RB_LOG) * (u32)a.aux_w * TVM_RB + (loc_cur_ & (TVM_RB - 1)));        \
    const u32 an_ = 8u * ((loc_next_ >> TVM_RB_LOG) * (u32)a.aux_w * TVM_RB + (loc_next_ & (TVM_RB - 1)));      \
    const u64 w_ = active_ ? t_ : a.q_len - 1; /* work-order slot of this lane's row */                         \
    const AIR_UNIFORM u64* ch_ = (const AIR_UNIFORM u64*)a.challenges;                                          \
    const AIR_UNIFORM u64* wt_ = (const AIR_UNIFORM u64*)a.weights;                                             \
    xfe quot = xfe_zero()

// (non-temporal loads here measured 11 % slower: the next-row cells are re-read from the cache as current-row cells)
#define AIR_CELL(base, lane_off, word) (*(const u64*)((base) + (size_t)(word) * (TVM_RB * 8) + (size_t)(lane_off)))
#define MC(c) AIR_CELL(mb_, mc_, c)
#define MN(c) AIR_CELL(mb_, mn_, c)
#define AC(c) xfe_make(AIR_CELL(ab_, ac_, 3 * (c)), AIR_CELL(ab_, ac_, 3 * (c) + 1), AIR_CELL(ab_, ac_, 3 * (c) + 2))
#define AN(c) xfe_make(AIR_CELL(ab_, an_, 3 * (c)), AIR_CELL(ab_, an_, 3 * (c) + 1), AIR_CELL(ab_, an_, 3 * (c) + 2))
#define CH(k) xfe_make(ch_[3 * (k)], ch_[3 * (k) + 1], ch_[3 * (k) + 2])
#define W(k) xfe_make(wt_[3 * (k)], wt_[3 * (k) + 1], wt_[3 * (k) + 2])
#define ZINV(s) (a.zinv[(u64)(s) * a.q_len + w_])
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

#define AIR_EPILOGUE(accumulate)                                        \
    if (active_) {                                                      \
        u64* o_ = a.out + 3 * w_;                                       \
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
