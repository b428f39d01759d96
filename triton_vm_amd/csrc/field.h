// field.h -- arithmetic in F_p, p = 2^64 - 2^32 + 1, and in F_p[X]/(X^3 - X + 1), for gfx950.
//
// Words are Montgomery representatives a*2^64 mod p, always canonical (< p), exactly the
// raw u64 inside the reference's `BFieldElement` (SURVEY.md 8b;
// /root/reference/triton-constraint-builder/src/codegen.rs:926-944), so no conversion happens at
// the library boundary and every kernel result can be compared bit-for-bit.
//
// gfx950 has no 64x64->128 multiply: the product is assembled from four 32x32->64 multiply-adds
// (v_mad_u64_u32), and the Montgomery reduction uses only the shape of p (2^64 = 2^32 - 1 mod p):
// three 64-bit add/sub with carries, no further multiplication.
#pragma once
#include "platform.h"

#define TVM_P 0xFFFFFFFF00000001ull
#define TVM_EPS 0xFFFFFFFFull          // 2^64 mod p = 2^32 - 1
#define TVM_ONE 0xFFFFFFFFull          // Montgomery word of 1
#define TVM_R2 0xFFFFFFFE00000001ull   // 2^128 mod p, to enter Montgomery form

// gfx950 does not interlock a VALU instruction that reads VCC (or an SGPR) written by the VALU instruction
// right before it: two wait states are required (LLVM: VALUWriteSGPRVALURead on gfx940+).  The compiler
// inserts them in its own code; the asm carry chains below do it by hand.
#ifdef TVM_EXPERIMENT_NO_WAIT  // timing experiment only: results are NOT guaranteed without the wait states
#define TVM_VCC_WAIT ""
#else
#define TVM_VCC_WAIT "s_nop 1\n\t"
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TVM_EMU)
#define TVM_FIELD_ASM 1
#endif
#ifndef TVM_MUL_CARRY_FORM
#define TVM_MUL_CARRY_FORM 1   // 0: the textbook form of the partial products (see bfe_mul), for A/B timings
#endif

TVM_HD u64 bfe_add(u64 a, u64 b) {
#ifdef TVM_FIELD_ASM
    // s = a + b (carry c1); t = s - p = s + EPS (carry c2); result = (c1 | c2) ? t : s.  32-bit full-rate
    // ops only: the 64-bit compare/add forms the compiler picks for the C below run at a fraction of the rate.
    u32 s0, s1, t0, t1;
    u64 c1;
    asm("v_add_co_u32 %[s0], vcc, %[a0], %[b0]\n\t" TVM_VCC_WAIT
        "v_addc_co_u32_e64 %[s1], %[c1], %[a1], %[b1], vcc\n\t"
        "v_add_co_u32 %[t0], vcc, -1, %[s0]\n\t" TVM_VCC_WAIT
        "v_addc_co_u32 %[t1], vcc, 0, %[s1], vcc\n\t" TVM_VCC_WAIT
        "s_or_b64 vcc, vcc, %[c1]\n\t" TVM_VCC_WAIT
        "v_cndmask_b32 %[s0], %[s0], %[t0], vcc\n\t"
        "v_cndmask_b32 %[s1], %[s1], %[t1], vcc"
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [t0] "=&v"(t0), [t1] "=&v"(t1), [c1] "=&s"(c1)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc", "scc");  // s_or_b64 writes SCC
    return ((u64)s1 << 32) | s0;
#else
    u64 s = a + b;
    // a + b < 2p: either the 64-bit add wrapped (then +EPS == -p mod 2^64) or s may be >= p.
    return (s < a || s >= TVM_P) ? s + TVM_EPS : s;
#endif
}
TVM_HD u64 bfe_sub(u64 a, u64 b) {
#ifdef TVM_FIELD_ASM
    // d = a - b; a borrow means d is short by p = 2^64 - EPS, i.e. subtract EPS once more
    u32 r0, r1, m;
    asm("v_sub_co_u32 %[r0], vcc, %[a0], %[b0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32 %[r1], vcc, %[a1], %[b1], vcc\n\t" TVM_VCC_WAIT
        "v_cndmask_b32_e64 %[m], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[m]\n\t" TVM_VCC_WAIT
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [m] "=&v"(m)
        : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
        : "vcc");
    return ((u64)r1 << 32) | r0;
#else
    u64 d = a - b;
    return (a < b) ? d - TVM_EPS : d;
#endif
}
TVM_HD u64 bfe_neg(u64 a) { return a ? TVM_P - a : 0; }
TVM_HD u64 bfe_dbl(u64 a) { return bfe_add(a, a); }

// x = hi*2^64 + lo  ->  x * 2^-64 mod p, canonical, for x < p*2^64.
TVM_HD u64 bfe_montyred(u64 lo, u64 hi) {
    u64 a = lo + (lo << 32);
    u64 e = (a < lo) ? 1 : 0;
    u64 b = a - (a >> 32) - e;
    u64 r = hi - b;
    return (hi < b) ? r - TVM_EPS : r;
}
TVM_HD void mul64wide(u64 a, u64 b, u64& lo, u64& hi) {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    // host code (the Fiat-Shamir sponge of the hosts, the fiber emulation of the CPU suite): the CPU's 64 x 64 -> 128 multiply
    const unsigned __int128 p = (unsigned __int128)a * b;
    lo = (u64)p;
    hi = (u64)(p >> 64);
    return;
#endif
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 m1 = (u64)a0 * b1 + (p00 >> 32);        // <= (2^32-1)^2 + 2^32 - 1 < 2^64
    u64 m2 = (u64)a1 * b0 + (u32)m1;            // same bound
    lo = (m2 << 32) | (u32)p00;
    hi = (u64)a1 * b1 + (m1 >> 32) + (m2 >> 32);
}
// Device multiplication.  With t = a0*b0, u = a0*b1 + t1 and the 65-bit sum (c : v) = a1*b0 + u the 128-bit product is
// x = (a1*b1 + (c : v1)) : v0 : t0 -- every addend but t1 is a FULL 64-bit register pair, so only t1 (and v1, next to the
// carry c) has to be moved into the low half of an aligned pair: v_mad_u64_u32 takes its addend as an even-aligned VGPR
// pair and the high half of a product always lands in an odd register.  (The textbook form u, v = a1*b0 + u0,
// w = a1*b1 + u1, x3:x2 = w + v1 costs three such moves and two more additions per product: 16 VALU instructions where
// this takes 14.)  The carry c is the mad's own carry-out (asm: C has no name for it); the tail is bfe_montyred on 32-bit
// limbs with the carries kept in VCC -- seven VALU instructions and one scalar one:
//   a1 = x1 + x0 (carry e);  b = (a1:x0) - a1 - e;  r = (x3:x2) - b;  if that borrowed (B), r -= 2^32 - 1.
// The last step is r0 += B (carry c), r1 -= B & ~c: the borrow goes to an SGPR pair, the AND-NOT is an s_andn2
// on the scalar unit, so the conditional correction costs two vector instructions instead of three.
TVM_HD u64 bfe_mul(u64 a, u64 b) {
#ifdef TVM_FIELD_ASM
#if TVM_MUL_CARRY_FORM
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    u64 v, sc;
    u32 c;
    asm("v_mad_u64_u32 %[v], %[sc], %[a1], %[b0], %[u]\n\t" TVM_VCC_WAIT
        "v_cndmask_b32_e64 %[c], 0, 1, %[sc]"
        : [v] "=v"(v), [sc] "=&s"(sc), [c] "=v"(c)
        : [a1] "v"(a1), [b0] "v"(b0), [u] "v"(u));
    const u64 w = (u64)a1 * b1 + (((u64)c << 32) | (v >> 32));
    u32 r0, r1, s1, s0;
    u64 bw;
    asm("v_add_co_u32 %[s1], vcc, %[x1], %[x0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32 %[s0], vcc, %[x0], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_subbrev_co_u32 %[s1], vcc, 0, %[s1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[w0], %[s0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32_e64 %[r1], %[bw], %[w1], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_addc_co_u32_e64 %[r0], vcc, %[r0], 0, %[bw]\n\t"
        "s_andn2_b64 vcc, %[bw], vcc\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [s1] "=&v"(s1), [s0] "=&v"(s0), [bw] "=&s"(bw)
        : [x0] "v"((u32)t), [x1] "v"((u32)v), [w0] "v"((u32)w), [w1] "v"((u32)(w >> 32))
        : "vcc", "scc");
    return ((u64)r1 << 32) | r0;
#else  // the textbook form
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    const u64 v = (u64)a1 * b0 + (u32)u;
    const u64 w = (u64)a1 * b1 + (u >> 32);
    u32 r0, r1, s1, s0;
    u64 bw;
    asm("v_add_co_u32 %[r0], vcc, %[w0], %[v1]\n\t" TVM_VCC_WAIT
        "v_addc_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_add_co_u32 %[s1], vcc, %[x1], %[x0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32 %[s0], vcc, %[x0], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_subbrev_co_u32 %[s1], vcc, 0, %[s1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[s0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32_e64 %[r1], %[bw], %[r1], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_addc_co_u32_e64 %[r0], vcc, %[r0], 0, %[bw]\n\t"
        "s_andn2_b64 vcc, %[bw], vcc\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [s1] "=&v"(s1), [s0] "=&v"(s0), [bw] "=&s"(bw)
        : [x0] "v"((u32)t), [x1] "v"((u32)v), [w0] "v"((u32)w), [w1] "v"((u32)(w >> 32)), [v1] "v"((u32)(v >> 32))
        : "vcc", "scc");
    return ((u64)r1 << 32) | r0;
#endif
#else
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    return bfe_montyred(lo, hi);
#endif
}
// ---------------------------------------------------------------- interleaved carry chains
// The wait states above are not free where few wavefronts share a SIMD (the AIR kernels run 2-3 per SIMD):
// without them the AIR evaluation measured 15 % faster, the LDE 6 %.  Two or three INDEPENDENT operations
// written as one instruction stream hide them instead: chain A carries in VCC, chains B and C in SGPR pairs
// (VOP3 encodings), and the instructions are issued A, B, C, A, B, C ... -- with three chains every
// carry consumer has two instructions between it and its producer, which is exactly the required distance;
// with two chains one single-cycle s_nop per step remains.  Results are identical to the single forms.
#ifdef TVM_FIELD_ASM
#define TVM_CA "vcc"
#define TVM_CB "%[cb]"
#define TVM_CC "%[cc]"
// S = operand suffix of the chain, C = its carry register
// the middle partial product with its carry-out, and the carry as a 0/1 word (see bfe_mul)
#define TVM_MV(S, C) "v_mad_u64_u32 %[v" #S "], " C ", %[a1" #S "], %[b0" #S "], %[u" #S "]\n\t"
#define TVM_MC(S, C) "v_cndmask_b32_e64 %[k" #S "], 0, 1, " C "\n\t"
#define TVM_MID_OUT(S, mid, carry) [v##S] "=&v"(mid), [k##S] "=&v"(carry)
#define TVM_MID_IN(S, fa, fb, low) [a1##S] "v"((u32)((fa) >> 32)), [b0##S] "v"((u32)(fb)), [u##S] "v"(low)
#define TVM_M3(S, C) "v_add_co_u32_e64 %[s1" #S "], " C ", %[x1" #S "], %[x0" #S "]\n\t"
#define TVM_M4(S, C) "v_subb_co_u32_e64 %[s0" #S "], " C ", %[x0" #S "], %[s1" #S "], " C "\n\t"
#define TVM_M5(S, C) "v_subbrev_co_u32_e64 %[s1" #S "], " C ", 0, %[s1" #S "], " C "\n\t"
#define TVM_M6(S, C) "v_sub_co_u32_e64 %[r0" #S "], " C ", %[w0" #S "], %[s0" #S "]\n\t"
#define TVM_M7(S, C) "v_subb_co_u32_e64 %[r1" #S "], %[bw" #S "], %[w1" #S "], %[s1" #S "], " C "\n\t"
#define TVM_M8(S, C) "v_addc_co_u32_e64 %[r0" #S "], " C ", %[r0" #S "], 0, %[bw" #S "]\n\t"
#define TVM_M9(S, C) "s_andn2_b64 " C ", %[bw" #S "], " C "\n\t"
#define TVM_M10(S, C) "v_subbrev_co_u32_e64 %[r1" #S "], " C ", 0, %[r1" #S "], " C "\n\t"
#define TVM_MUL_OUT(S, p, q, x, y, k) [r0##S] "=&v"(p), [r1##S] "=&v"(q), [s0##S] "=&v"(x), [s1##S] "=&v"(y), [bw##S] "=&s"(k)
#define TVM_MUL_IN(S, t, v, w) [x0##S] "v"((u32)(t)), [x1##S] "v"((u32)(v)), [w0##S] "v"((u32)(w)), [w1##S] "v"((u32)((w) >> 32))
#define TVM_MUL_LOW(a, b, t, u)                         \
    const u64 t = (u64)(u32)(a) * (u32)(b);             \
    const u64 u = (u64)(u32)(a) * (u32)((b) >> 32) + (t >> 32)
#define TVM_MUL_HIGH(a, b, v, c, w) const u64 w = (u64)(u32)((a) >> 32) * (u32)((b) >> 32) + (((u64)(c) << 32) | ((v) >> 32))
// the textbook form (TVM_MUL_CARRY_FORM 0): steps of its tail
#define TVM_M1_T(S, C) "v_add_co_u32_e64 %[r0" #S "], " C ", %[w0" #S "], %[v1" #S "]\n\t"
#define TVM_M2_T(S, C) "v_addc_co_u32_e64 %[r1" #S "], " C ", 0, %[w1" #S "], " C "\n\t"
#define TVM_M3_T(S, C) "v_add_co_u32_e64 %[s1" #S "], " C ", %[x1" #S "], %[x0" #S "]\n\t"
#define TVM_M4_T(S, C) "v_subb_co_u32_e64 %[s0" #S "], " C ", %[x0" #S "], %[s1" #S "], " C "\n\t"
#define TVM_M5_T(S, C) "v_subbrev_co_u32_e64 %[s1" #S "], " C ", 0, %[s1" #S "], " C "\n\t"
#define TVM_M6_T(S, C) "v_sub_co_u32_e64 %[r0" #S "], " C ", %[r0" #S "], %[s0" #S "]\n\t"
#define TVM_M7_T(S, C) "v_subb_co_u32_e64 %[r1" #S "], %[bw" #S "], %[r1" #S "], %[s1" #S "], " C "\n\t"
#define TVM_M8_T(S, C) "v_addc_co_u32_e64 %[r0" #S "], " C ", %[r0" #S "], 0, %[bw" #S "]\n\t"
#define TVM_M9_T(S, C) "s_andn2_b64 " C ", %[bw" #S "], " C "\n\t"
#define TVM_M10_T(S, C) "v_subbrev_co_u32_e64 %[r1" #S "], " C ", 0, %[r1" #S "], " C "\n\t"
#define TVM_MUL_OUT_T(S, p, q, x, y, k) [r0##S] "=&v"(p), [r1##S] "=&v"(q), [s0##S] "=&v"(x), [s1##S] "=&v"(y), [bw##S] "=&s"(k)
#define TVM_MUL_IN_T(S, t, v, w) \
    [x0##S] "v"((u32)(t)), [x1##S] "v"((u32)(v)), [w0##S] "v"((u32)(w)), [w1##S] "v"((u32)((w) >> 32)), [v1##S] "v"((u32)((v) >> 32))
#define TVM_MUL_PARTIALS_T(a, b, t, u, v, w)                                                        \
    const u64 t = (u64)(u32)(a) * (u32)(b);                                                       \
    const u64 u = (u64)(u32)(a) * (u32)((b) >> 32) + (t >> 32);                                   \
    const u64 v = (u64)(u32)((a) >> 32) * (u32)(b) + (u32)u;                                      \
    const u64 w = (u64)(u32)((a) >> 32) * (u32)((b) >> 32) + (u >> 32)
#define TVM_3WAY(STEP) STEP(a, TVM_CA) STEP(b, TVM_CB) STEP(c, TVM_CC)
#define TVM_2WAY(STEP) STEP(a, TVM_CA) STEP(b, TVM_CB) "s_nop 0\n\t"
#endif

// three independent products
TVM_HD void bfe_mul3(u64 a0, u64 b0, u64 a1, u64 b1, u64 a2, u64 b2, u64& p0, u64& p1, u64& p2) {
#ifdef TVM_FIELD_ASM
#if TVM_MUL_CARRY_FORM
    TVM_MUL_LOW(a0, b0, ta, ua);
    TVM_MUL_LOW(a1, b1, tb, ub);
    TVM_MUL_LOW(a2, b2, tc, uc);
    u64 va, vb, vc, cb, cc;
    u32 ca, cb_, cc_;
    asm(TVM_3WAY(TVM_MV) TVM_3WAY(TVM_MC)
        : TVM_MID_OUT(a, va, ca), TVM_MID_OUT(b, vb, cb_), TVM_MID_OUT(c, vc, cc_), [cb] "=&s"(cb), [cc] "=&s"(cc)
        : TVM_MID_IN(a, a0, b0, ua), TVM_MID_IN(b, a1, b1, ub), TVM_MID_IN(c, a2, b2, uc)
        : "vcc");
    TVM_MUL_HIGH(a0, b0, va, ca, wa);
    TVM_MUL_HIGH(a1, b1, vb, cb_, wb);
    TVM_MUL_HIGH(a2, b2, vc, cc_, wc);
    u32 r0a, r1a, s0a, s1a, r0b, r1b, s0b, s1b, r0c, r1c, s0c, s1c;
    u64 db, dc, ba, bb, bc;
    asm(TVM_3WAY(TVM_M3) TVM_3WAY(TVM_M4) TVM_3WAY(TVM_M5) TVM_3WAY(TVM_M6)
        TVM_3WAY(TVM_M7) TVM_3WAY(TVM_M8) TVM_3WAY(TVM_M9) TVM_3WAY(TVM_M10)
        : TVM_MUL_OUT(a, r0a, r1a, s0a, s1a, ba), TVM_MUL_OUT(b, r0b, r1b, s0b, s1b, bb), TVM_MUL_OUT(c, r0c, r1c, s0c, s1c, bc),
          [cb] "=&s"(db), [cc] "=&s"(dc)
        : TVM_MUL_IN(a, ta, va, wa), TVM_MUL_IN(b, tb, vb, wb), TVM_MUL_IN(c, tc, vc, wc)
        : "vcc", "scc");
    p0 = ((u64)r1a << 32) | r0a;
    p1 = ((u64)r1b << 32) | r0b;
    p2 = ((u64)r1c << 32) | r0c;
#else
    TVM_MUL_PARTIALS_T(a0, b0, ta, ua, va, wa);
    TVM_MUL_PARTIALS_T(a1, b1, tb, ub, vb, wb);
    TVM_MUL_PARTIALS_T(a2, b2, tc, uc, vc, wc);
    u32 r0a, r1a, s0a, s1a, r0b, r1b, s0b, s1b, r0c, r1c, s0c, s1c;
    u64 cb, cc, ba, bb, bc;
    asm(TVM_3WAY(TVM_M1_T) TVM_3WAY(TVM_M2_T) TVM_3WAY(TVM_M3_T) TVM_3WAY(TVM_M4_T) TVM_3WAY(TVM_M5_T) TVM_3WAY(TVM_M6_T)
        TVM_3WAY(TVM_M7_T) TVM_3WAY(TVM_M8_T) TVM_3WAY(TVM_M9_T) TVM_3WAY(TVM_M10_T)
        : TVM_MUL_OUT_T(a, r0a, r1a, s0a, s1a, ba), TVM_MUL_OUT_T(b, r0b, r1b, s0b, s1b, bb), TVM_MUL_OUT_T(c, r0c, r1c, s0c, s1c, bc),
          [cb] "=&s"(cb), [cc] "=&s"(cc)
        : TVM_MUL_IN_T(a, ta, va, wa), TVM_MUL_IN_T(b, tb, vb, wb), TVM_MUL_IN_T(c, tc, vc, wc)
        : "vcc", "scc");
    p0 = ((u64)r1a << 32) | r0a;
    p1 = ((u64)r1b << 32) | r0b;
    p2 = ((u64)r1c << 32) | r0c;
#endif
#else
    p0 = bfe_mul(a0, b0);
    p1 = bfe_mul(a1, b1);
    p2 = bfe_mul(a2, b2);
#endif
}
// two independent products
TVM_HD void bfe_mul2(u64 a0, u64 b0, u64 a1, u64 b1, u64& p0, u64& p1) {
#ifdef TVM_FIELD_ASM
#if TVM_MUL_CARRY_FORM
    TVM_MUL_LOW(a0, b0, ta, ua);
    TVM_MUL_LOW(a1, b1, tb, ub);
    u64 va, vb, cb;
    u32 ca, cb_;
    asm(TVM_2WAY(TVM_MV) TVM_MC(a, TVM_CA) TVM_MC(b, TVM_CB)
        : TVM_MID_OUT(a, va, ca), TVM_MID_OUT(b, vb, cb_), [cb] "=&s"(cb)
        : TVM_MID_IN(a, a0, b0, ua), TVM_MID_IN(b, a1, b1, ub)
        : "vcc");
    TVM_MUL_HIGH(a0, b0, va, ca, wa);
    TVM_MUL_HIGH(a1, b1, vb, cb_, wb);
    u32 r0a, r1a, s0a, s1a, r0b, r1b, s0b, s1b;
    u64 db, ba, bb;
    asm(TVM_2WAY(TVM_M3) TVM_2WAY(TVM_M4) TVM_2WAY(TVM_M5) TVM_2WAY(TVM_M6)
        TVM_2WAY(TVM_M7) TVM_2WAY(TVM_M8) TVM_M9(a, TVM_CA) TVM_M9(b, TVM_CB) TVM_M10(a, TVM_CA) TVM_M10(b, TVM_CB)
        : TVM_MUL_OUT(a, r0a, r1a, s0a, s1a, ba), TVM_MUL_OUT(b, r0b, r1b, s0b, s1b, bb), [cb] "=&s"(db)
        : TVM_MUL_IN(a, ta, va, wa), TVM_MUL_IN(b, tb, vb, wb)
        : "vcc", "scc");
    p0 = ((u64)r1a << 32) | r0a;
    p1 = ((u64)r1b << 32) | r0b;
#else
    TVM_MUL_PARTIALS_T(a0, b0, ta, ua, va, wa);
    TVM_MUL_PARTIALS_T(a1, b1, tb, ub, vb, wb);
    u32 r0a, r1a, s0a, s1a, r0b, r1b, s0b, s1b;
    u64 cb, ba, bb;
    asm(TVM_2WAY(TVM_M1_T) TVM_2WAY(TVM_M2_T) TVM_2WAY(TVM_M3_T) TVM_2WAY(TVM_M4_T) TVM_2WAY(TVM_M5_T) TVM_2WAY(TVM_M6_T)
        TVM_2WAY(TVM_M7_T) TVM_2WAY(TVM_M8_T) TVM_M9_T(a, TVM_CA) TVM_M9_T(b, TVM_CB) TVM_M10_T(a, TVM_CA) TVM_M10_T(b, TVM_CB)
        : TVM_MUL_OUT_T(a, r0a, r1a, s0a, s1a, ba), TVM_MUL_OUT_T(b, r0b, r1b, s0b, s1b, bb), [cb] "=&s"(cb)
        : TVM_MUL_IN_T(a, ta, va, wa), TVM_MUL_IN_T(b, tb, vb, wb)
        : "vcc", "scc");
    p0 = ((u64)r1a << 32) | r0a;
    p1 = ((u64)r1b << 32) | r0b;
#endif
#else
    p0 = bfe_mul(a0, b0);
    p1 = bfe_mul(a1, b1);
#endif
}

#ifdef TVM_FIELD_ASM
// a - b (mod p), steps of bfe_sub
#define TVM_S1(S, C) "v_sub_co_u32_e64 %[r0" #S "], " C ", %[a0" #S "], %[b0" #S "]\n\t"
#define TVM_S2(S, C) "v_subb_co_u32_e64 %[r1" #S "], " C ", %[a1" #S "], %[b1" #S "], " C "\n\t"
#define TVM_S3(S, C) "v_cndmask_b32_e64 %[m" #S "], 0, -1, " C "\n\t"
#define TVM_S4(S, C) "v_sub_co_u32_e64 %[r0" #S "], " C ", %[r0" #S "], %[m" #S "]\n\t"
#define TVM_S5(S, C) "v_subbrev_co_u32_e64 %[r1" #S "], " C ", 0, %[r1" #S "], " C "\n\t"
#define TVM_SUB_OUT(S, p, q, x) [r0##S] "=&v"(p), [r1##S] "=&v"(q), [m##S] "=&v"(x)
#define TVM_AB_IN(S, a, b) [a0##S] "v"((u32)(a)), [a1##S] "v"((u32)((a) >> 32)), [b0##S] "v"((u32)(b)), [b1##S] "v"((u32)((b) >> 32))
// a + b (mod p), steps of bfe_add; K = the chain's saved first carry
#define TVM_A1(S, C, K) "v_add_co_u32_e64 %[s0" #S "], " C ", %[a0" #S "], %[b0" #S "]\n\t"
#define TVM_A2(S, C, K) "v_addc_co_u32_e64 %[s1" #S "], " K ", %[a1" #S "], %[b1" #S "], " C "\n\t"
#define TVM_A3(S, C, K) "v_add_co_u32_e64 %[t0" #S "], " C ", -1, %[s0" #S "]\n\t"
#define TVM_A4(S, C, K) "v_addc_co_u32_e64 %[t1" #S "], " C ", 0, %[s1" #S "], " C "\n\t"
#define TVM_A5(S, C, K) "s_or_b64 " C ", " C ", " K "\n\t"
#define TVM_A6(S, C, K) "v_cndmask_b32_e64 %[s0" #S "], %[s0" #S "], %[t0" #S "], " C "\n\t"
#define TVM_A7(S, C, K) "v_cndmask_b32_e64 %[s1" #S "], %[s1" #S "], %[t1" #S "], " C "\n\t"
#define TVM_ADD_OUT(S, p, q, x, y) [s0##S] "=&v"(p), [s1##S] "=&v"(q), [t0##S] "=&v"(x), [t1##S] "=&v"(y)
#define TVM_3WAY_ADD(STEP) STEP(a, TVM_CA, "%[ka]") STEP(b, TVM_CB, "%[kb]") STEP(c, TVM_CC, "%[kc]")
#define TVM_2WAY_ADD(STEP) STEP(a, TVM_CA, "%[ka]") STEP(b, TVM_CB, "%[kb]") "s_nop 0\n\t"
#endif

TVM_HD void bfe_sub3(u64 a0, u64 b0, u64 a1, u64 b1, u64 a2, u64 b2, u64& d0, u64& d1, u64& d2) {
#ifdef TVM_FIELD_ASM
    u32 r0a, r1a, ma, r0b, r1b, mb, r0c, r1c, mc;
    u64 cb, cc;
    asm(TVM_3WAY(TVM_S1) TVM_3WAY(TVM_S2) TVM_3WAY(TVM_S3) TVM_3WAY(TVM_S4) TVM_3WAY(TVM_S5)
        : TVM_SUB_OUT(a, r0a, r1a, ma), TVM_SUB_OUT(b, r0b, r1b, mb), TVM_SUB_OUT(c, r0c, r1c, mc), [cb] "=&s"(cb),
          [cc] "=&s"(cc)
        : TVM_AB_IN(a, a0, b0), TVM_AB_IN(b, a1, b1), TVM_AB_IN(c, a2, b2)
        : "vcc");
    d0 = ((u64)r1a << 32) | r0a;
    d1 = ((u64)r1b << 32) | r0b;
    d2 = ((u64)r1c << 32) | r0c;
#else
    d0 = bfe_sub(a0, b0);
    d1 = bfe_sub(a1, b1);
    d2 = bfe_sub(a2, b2);
#endif
}
TVM_HD void bfe_sub2(u64 a0, u64 b0, u64 a1, u64 b1, u64& d0, u64& d1) {
#ifdef TVM_FIELD_ASM
    u32 r0a, r1a, ma, r0b, r1b, mb;
    u64 cb;
    asm(TVM_2WAY(TVM_S1) TVM_2WAY(TVM_S2) TVM_2WAY(TVM_S3) TVM_2WAY(TVM_S4) TVM_S5(a, TVM_CA) TVM_S5(b, TVM_CB)
        : TVM_SUB_OUT(a, r0a, r1a, ma), TVM_SUB_OUT(b, r0b, r1b, mb), [cb] "=&s"(cb)
        : TVM_AB_IN(a, a0, b0), TVM_AB_IN(b, a1, b1)
        : "vcc");
    d0 = ((u64)r1a << 32) | r0a;
    d1 = ((u64)r1b << 32) | r0b;
#else
    d0 = bfe_sub(a0, b0);
    d1 = bfe_sub(a1, b1);
#endif
}
TVM_HD void bfe_add3(u64 a0, u64 b0, u64 a1, u64 b1, u64 a2, u64 b2, u64& e0, u64& e1, u64& e2) {
#ifdef TVM_FIELD_ASM
    u32 s0a, s1a, t0a, t1a, s0b, s1b, t0b, t1b, s0c, s1c, t0c, t1c;
    u64 cb, cc, ka, kb, kc;
    asm(TVM_3WAY_ADD(TVM_A1) TVM_3WAY_ADD(TVM_A2) TVM_3WAY_ADD(TVM_A3) TVM_3WAY_ADD(TVM_A4) TVM_3WAY_ADD(TVM_A5)
        TVM_3WAY_ADD(TVM_A6) TVM_3WAY_ADD(TVM_A7)
        : TVM_ADD_OUT(a, s0a, s1a, t0a, t1a), TVM_ADD_OUT(b, s0b, s1b, t0b, t1b), TVM_ADD_OUT(c, s0c, s1c, t0c, t1c),
          [cb] "=&s"(cb), [cc] "=&s"(cc), [ka] "=&s"(ka), [kb] "=&s"(kb), [kc] "=&s"(kc)
        : TVM_AB_IN(a, a0, b0), TVM_AB_IN(b, a1, b1), TVM_AB_IN(c, a2, b2)
        : "vcc", "scc");
    e0 = ((u64)s1a << 32) | s0a;
    e1 = ((u64)s1b << 32) | s0b;
    e2 = ((u64)s1c << 32) | s0c;
#else
    e0 = bfe_add(a0, b0);
    e1 = bfe_add(a1, b1);
    e2 = bfe_add(a2, b2);
#endif
}
TVM_HD void bfe_add2(u64 a0, u64 b0, u64 a1, u64 b1, u64& e0, u64& e1) {
#ifdef TVM_FIELD_ASM
    u32 s0a, s1a, t0a, t1a, s0b, s1b, t0b, t1b;
    u64 cb, ka, kb;
    asm(TVM_2WAY_ADD(TVM_A1) TVM_2WAY_ADD(TVM_A2) TVM_2WAY_ADD(TVM_A3) TVM_2WAY_ADD(TVM_A4) TVM_2WAY_ADD(TVM_A5)
        TVM_A6(a, TVM_CA, "%[ka]") TVM_A6(b, TVM_CB, "%[kb]") TVM_A7(a, TVM_CA, "%[ka]") TVM_A7(b, TVM_CB, "%[kb]")
        : TVM_ADD_OUT(a, s0a, s1a, t0a, t1a), TVM_ADD_OUT(b, s0b, s1b, t0b, t1b), [cb] "=&s"(cb), [ka] "=&s"(ka),
          [kb] "=&s"(kb)
        : TVM_AB_IN(a, a0, b0), TVM_AB_IN(b, a1, b1)
        : "vcc", "scc");
    e0 = ((u64)s1a << 32) | s0a;
    e1 = ((u64)s1b << 32) | s0b;
#else
    e0 = bfe_add(a0, b0);
    e1 = bfe_add(a1, b1);
#endif
}

TVM_HD u64 bfe_sqr(u64 a) { return bfe_mul(a, a); }
// Montgomery word of a small canonical integer v
TVM_HD u64 bfe_from_u64(u64 v) { return bfe_mul(v >= TVM_P ? v - TVM_P : v, TVM_R2); }
TVM_HD u64 bfe_pow(u64 a, u64 e) {
    u64 r = TVM_ONE;
    while (e) {
        if (e & 1) r = bfe_mul(r, a);
        a = bfe_sqr(a);
        e >>= 1;
    }
    return r;
}
// a^(p - 2), p - 2 = 2^64 - 2^32 - 1 (thirty-one ones, a zero, thirty-two ones): 63 squarings and 9 multiplications along the
// chain 2^2-1, 2^3-1, 2^6-1, 2^12-1, 2^24-1, 2^30-1, 2^31-1, (2^31-1) 2^32 + 2^31-1, then one more squaring and a -- where
// square-and-multiply over the bits of the exponent takes 64 + 62.  0 -> 0.
TVM_HD u64 bfe_sqr_n(u64 a, int n) {
    for (int i = 0; i < n; i++) a = bfe_sqr(a);
    return a;
}
TVM_HD u64 bfe_inv(u64 a) {
    const u64 t2 = bfe_mul(bfe_sqr(a), a);
    const u64 t3 = bfe_mul(bfe_sqr(t2), a);
    const u64 t6 = bfe_mul(bfe_sqr_n(t3, 3), t3);
    const u64 t12 = bfe_mul(bfe_sqr_n(t6, 6), t6);
    const u64 t24 = bfe_mul(bfe_sqr_n(t12, 12), t12);
    const u64 t30 = bfe_mul(bfe_sqr_n(t24, 6), t6);
    const u64 t31 = bfe_mul(bfe_sqr(t30), a);
    const u64 t63 = bfe_mul(bfe_sqr_n(t31, 32), t31);
    return bfe_mul(bfe_sqr(t63), a);
}

// ---------------------------------------------------------------- F_p[X]/(X^3 - X + 1)
struct xfe {
    u64 c0, c1, c2;
};
TVM_HD xfe xfe_make(u64 a, u64 b, u64 c) { xfe r; r.c0 = a; r.c1 = b; r.c2 = c; return r; }
TVM_HD xfe xfe_zero() { return xfe_make(0, 0, 0); }
TVM_HD xfe xfe_one() { return xfe_make(TVM_ONE, 0, 0); }
TVM_HD xfe xfe_lift(u64 a) { return xfe_make(a, 0, 0); }
TVM_HD xfe xfe_add(xfe a, xfe b) {
    xfe r;
    bfe_add3(a.c0, b.c0, a.c1, b.c1, a.c2, b.c2, r.c0, r.c1, r.c2);
    return r;
}
TVM_HD xfe xfe_sub(xfe a, xfe b) {
    xfe r;
    bfe_sub3(a.c0, b.c0, a.c1, b.c1, a.c2, b.c2, r.c0, r.c1, r.c2);
    return r;
}
TVM_HD xfe xfe_neg(xfe a) { return xfe_make(bfe_neg(a.c0), bfe_neg(a.c1), bfe_neg(a.c2)); }
TVM_HD xfe xfe_add_bfe(xfe a, u64 b) { return xfe_make(bfe_add(a.c0, b), a.c1, a.c2); }
TVM_HD xfe xfe_sub_bfe(xfe a, u64 b) { return xfe_make(bfe_sub(a.c0, b), a.c1, a.c2); }
// b - x for a base-field b (the factor (x_i - p) of a zerofier at a base-field point)
TVM_HD xfe xfe_bfe_minus(u64 b, xfe x) { return xfe_make(bfe_sub(b, x.c0), bfe_neg(x.c1), bfe_neg(x.c2)); }
TVM_HD xfe xfe_mul_bfe(xfe a, u64 b) {
    xfe r;
    bfe_mul3(a.c0, b, a.c1, b, a.c2, b, r.c0, r.c1, r.c2);
    return r;
}
// schoolbook, then X^3 = X - 1, X^4 = X^2 - X
TVM_HD xfe xfe_mul(xfe a, xfe b) {
    // nine products, three independent ones at a time (interleaved carry chains)
    u64 p00, p01, p02, p10, p11, p12, p20, p21, p22;
    bfe_mul3(a.c0, b.c0, a.c0, b.c1, a.c0, b.c2, p00, p01, p02);
    bfe_mul3(a.c1, b.c0, a.c1, b.c1, a.c1, b.c2, p10, p11, p12);
    bfe_mul3(a.c2, b.c0, a.c2, b.c1, a.c2, b.c2, p20, p21, p22);
    u64 d1, d3, t2;
    bfe_add3(p01, p10, p12, p21, p02, p11, d1, d3, t2);  // d1, d3, and the first half of d2
    const u64 d2 = bfe_add(t2, p20);
    // (d0 - d3) + (d1 + d3 - d4) X + (d2 + d4) X^2 with d0 = p00, d4 = p22
    u64 e1, c0, c1, c2;
    bfe_add2(d1, d3, d2, p22, e1, c2);
    bfe_sub2(p00, d3, e1, p22, c0, c1);
    return xfe_make(c0, c1, c2);
}
TVM_HD xfe xfe_sqr(xfe a) { return xfe_mul(a, a); }
TVM_HD bool xfe_eq(xfe a, xfe b) { return a.c0 == b.c0 && a.c1 == b.c1 && a.c2 == b.c2; }
// inverse through the adjugate of the multiplication-by-a matrix; one base-field inversion
TVM_HD xfe xfe_inv(xfe a) {
    u64 s = bfe_add(a.c0, a.c2);
    u64 d12 = bfe_sub(a.c1, a.c2);
    u64 k0 = bfe_sub(bfe_sqr(s), bfe_mul(d12, a.c1));
    u64 k1 = bfe_neg(bfe_sub(bfe_mul(a.c1, s), bfe_mul(d12, a.c2)));
    u64 k2 = bfe_sub(bfe_sqr(a.c1), bfe_mul(s, a.c2));
    u64 det = bfe_sub(bfe_sub(bfe_mul(a.c0, k0), bfe_mul(a.c2, k1)), bfe_mul(a.c1, k2));
    u64 di = bfe_inv(det);
    return xfe_make(bfe_mul(k0, di), bfe_mul(k1, di), bfe_mul(k2, di));
}
TVM_HD xfe xfe_pow(xfe a, u64 e) {
    xfe r = xfe_one();
    while (e) {
        if (e & 1) r = xfe_mul(r, a);
        a = xfe_sqr(a);
        e >>= 1;
    }
    return r;
}
