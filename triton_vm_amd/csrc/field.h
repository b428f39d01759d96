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
#define TVM_VCC_WAIT "s_nop 1\n\t"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TVM_EMU)
#define TVM_FIELD_ASM 1
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
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 m1 = (u64)a0 * b1 + (p00 >> 32);        // <= (2^32-1)^2 + 2^32 - 1 < 2^64
    u64 m2 = (u64)a1 * b0 + (u32)m1;            // same bound
    lo = (m2 << 32) | (u32)p00;
    hi = (u64)a1 * b1 + (m1 >> 32) + (m2 >> 32);
}
// Device multiplication: the four 32x32 partial products are C (v_mad_u64_u32), the carry chain is asm.
// With t = a0*b0, u = a0*b1 + t1, v = a1*b0 + u0, w = a1*b1 + u1 the 128-bit product is
// x = (w + v1) : v0 : t0; the asm adds v1 into w and performs bfe_montyred on 32-bit limbs with the
// carries kept in VCC -- ten VALU instructions, where the compiler's rendering of the 64-bit C form takes
// fourteen plus re-pairing moves and a zero-extended register pair per partial sum:
//   a1 = x1 + x0 (carry e);  b = (a1:x0) - a1 - e;  r = (x3:x2) - b;  if that borrowed, r -= 2^32 - 1.
TVM_HD u64 bfe_mul(u64 a, u64 b) {
#ifdef TVM_FIELD_ASM
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 u = (u64)a0 * b1 + (t >> 32);
    const u64 v = (u64)a1 * b0 + (u32)u;
    const u64 w = (u64)a1 * b1 + (u >> 32);
    u32 r0, r1, s1, s0;
    asm("v_add_co_u32 %[r0], vcc, %[w0], %[v1]\n\t" TVM_VCC_WAIT
        "v_addc_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_add_co_u32 %[s1], vcc, %[x1], %[x0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32 %[s0], vcc, %[x0], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_subbrev_co_u32 %[s1], vcc, 0, %[s1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[s0]\n\t" TVM_VCC_WAIT
        "v_subb_co_u32 %[r1], vcc, %[r1], %[s1], vcc\n\t" TVM_VCC_WAIT
        "v_cndmask_b32_e64 %[s0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[s0]\n\t" TVM_VCC_WAIT
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [s1] "=&v"(s1), [s0] "=&v"(s0)
        : [x0] "v"((u32)t), [x1] "v"((u32)v), [w0] "v"((u32)w), [w1] "v"((u32)(w >> 32)), [v1] "v"((u32)(v >> 32))
        : "vcc");
    return ((u64)r1 << 32) | r0;
#else
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    return bfe_montyred(lo, hi);
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
TVM_HD u64 bfe_inv(u64 a) { return bfe_pow(a, TVM_P - 2); }

// ---------------------------------------------------------------- F_p[X]/(X^3 - X + 1)
struct xfe {
    u64 c0, c1, c2;
};
TVM_HD xfe xfe_make(u64 a, u64 b, u64 c) { xfe r; r.c0 = a; r.c1 = b; r.c2 = c; return r; }
TVM_HD xfe xfe_zero() { return xfe_make(0, 0, 0); }
TVM_HD xfe xfe_one() { return xfe_make(TVM_ONE, 0, 0); }
TVM_HD xfe xfe_lift(u64 a) { return xfe_make(a, 0, 0); }
TVM_HD xfe xfe_add(xfe a, xfe b) { return xfe_make(bfe_add(a.c0, b.c0), bfe_add(a.c1, b.c1), bfe_add(a.c2, b.c2)); }
TVM_HD xfe xfe_sub(xfe a, xfe b) { return xfe_make(bfe_sub(a.c0, b.c0), bfe_sub(a.c1, b.c1), bfe_sub(a.c2, b.c2)); }
TVM_HD xfe xfe_neg(xfe a) { return xfe_make(bfe_neg(a.c0), bfe_neg(a.c1), bfe_neg(a.c2)); }
TVM_HD xfe xfe_add_bfe(xfe a, u64 b) { return xfe_make(bfe_add(a.c0, b), a.c1, a.c2); }
TVM_HD xfe xfe_sub_bfe(xfe a, u64 b) { return xfe_make(bfe_sub(a.c0, b), a.c1, a.c2); }
TVM_HD xfe xfe_mul_bfe(xfe a, u64 b) { return xfe_make(bfe_mul(a.c0, b), bfe_mul(a.c1, b), bfe_mul(a.c2, b)); }
// schoolbook, then X^3 = X - 1, X^4 = X^2 - X
TVM_HD xfe xfe_mul(xfe a, xfe b) {
    u64 d0 = bfe_mul(a.c0, b.c0);
    u64 d1 = bfe_add(bfe_mul(a.c0, b.c1), bfe_mul(a.c1, b.c0));
    u64 d2 = bfe_add(bfe_add(bfe_mul(a.c0, b.c2), bfe_mul(a.c1, b.c1)), bfe_mul(a.c2, b.c0));
    u64 d3 = bfe_add(bfe_mul(a.c1, b.c2), bfe_mul(a.c2, b.c1));
    u64 d4 = bfe_mul(a.c2, b.c2);
    return xfe_make(bfe_sub(d0, d3), bfe_sub(bfe_add(d1, d3), d4), bfe_add(d2, d4));
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
