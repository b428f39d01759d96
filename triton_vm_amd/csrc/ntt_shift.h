// ntt_shift.h -- transforms of 2, 4, 8 and 16 points whose twiddles are powers of two.
//
// In F_p, p = 2^64 - 2^32 + 1, two is a 192nd root of unity (2^96 = -1), and the 2^k-th roots of unity the reference's
// domains are built from (BFieldElement::primitive_root_of_unity, twenty-first; the generator of every ArithmeticDomain,
// /root/reference/triton-vm/src/arithmetic_domain.rs:99-108) are powers of two up to order 64:
//     w_4 = 2^48,  w_8 = 2^120 = -2^24,  w_16 = 2^156 = -2^60,  w_32 = 2^78,  w_64 = 2^39.
// A multiplication by 2^s is a 96-bit shift and one folding step with the shape of p -- 9 to 12 VALU instructions against
// the 16 of a general Montgomery multiplication, on the Montgomery word directly ((a R) 2^s = (a 2^s) R) -- and the sign of
// 2^(96 + s) = -2^s is absorbed by exchanging the butterfly's addition and subtraction.  The lowest four layers of every
// decimation-in-time transform (the highest four of a decimation-in-frequency one) are 16-point transforms of this kind.
#pragma once
#include "field.h"

namespace tvm {

// x * 2^S mod p for a canonical x, 0 <= S < 96, canonical result.  With y = x << (S % 32) as three 32-bit limbs y2:y1:y0
// and phi = 2^32 (phi^2 = phi - 1, phi^3 = -1):
//   S < 32:        y0 + y1 phi + y2 phi^2          = (y1:y0) + y2 (2^32 - 1)
//   32 <= S < 64:  y0 phi + y1 phi^2 + y2 phi^3    = (y0 + y1) 2^32 - (y1 + y2)
//   64 <= S < 96:  y0 phi^2 + y1 phi^3 + y2 phi^4  = y0 (2^32 - 1) - (y2:y1)
template <int S>
TVM_HD u64 bfe_mul_pow2(u64 x) {
    static_assert(S >= 0 && S < 96, "exponent out of range");
    constexpr int R = S % 32, Q = S / 32;
    if constexpr (S == 0) return x;
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    const u32 y0 = x0 << R;
    const u32 y1 = R ? ((x1 << R) | (x0 >> (32 - R))) : x1;
    const u32 y2 = R ? (x1 >> (32 - R)) : 0u;
    if constexpr (Q == 0) {
        const u64 lo = ((u64)y1 << 32) | y0;
        u64 t = lo + (u64)y2 * TVM_EPS;       // y2 < 2^31: the product fits; the sum may wrap once
        if (t < lo) t += TVM_EPS;             // 2^64 = 2^32 - 1; no second wrap: t < 2^63 after the first
        return t >= TVM_P ? t - TVM_P : t;
    } else if constexpr (Q == 1) {
        const u64 h = (u64)y0 + y1;           // 33 bits
        const u64 a = (h << 32) + ((h >> 32) ? TVM_EPS : 0);   // (h mod 2^32) 2^32 + [carry] 2^64, canonical either way
        const u64 b = (u64)y1 + y2;           // < 2^33
        return bfe_sub(a, b);
    } else {
        const u64 a = (u64)y0 * TVM_EPS;      // <= (2^32 - 1)^2 < p
        const u64 b = ((u64)y2 << 32) | y1;   // < 2^63
        return bfe_sub(a, b);
    }
}

// u + 2^E v and u - 2^E v for E mod 192 (the sign of E >= 96 exchanges the two)
template <int E>
TVM_HD void bfe_butterfly_pow2(u64& lo, u64& hi) {
    constexpr int EE = ((E % 192) + 192) % 192;
    const u64 u = lo;
    if constexpr (EE < 96) {
        const u64 v = bfe_mul_pow2<EE>(hi);
        lo = bfe_add(u, v);
        hi = bfe_sub(u, v);
    } else {
        const u64 v = bfe_mul_pow2<EE - 96>(hi);
        lo = bfe_sub(u, v);
        hi = bfe_add(u, v);
    }
}
// (u + v, (u - v) 2^E)
template <int E>
TVM_HD void bfe_butterfly_dif_pow2(u64& lo, u64& hi) {
    constexpr int EE = ((E % 192) + 192) % 192;
    const u64 u = lo, v = hi;
    lo = bfe_add(u, v);
    if constexpr (EE < 96) hi = bfe_mul_pow2<EE>(bfe_sub(u, v));
    else hi = bfe_mul_pow2<EE - 96>(bfe_sub(v, u));
}

// bit reversal of a K-bit index, usable in constant expressions
TVM_HD constexpr int brev_k(int e, int k) {
    int r = 0;
    for (int i = 0; i < k; i++) r |= ((e >> i) & 1) << (k - 1 - i);
    return r;
}

// log2 of the primitive 2^K-th root of unity the domains use (K <= 4), forward and inverse
template <int K, bool INVERSE>
struct Pow2Root {
    static constexpr int value = ((INVERSE ? 36 : 156) << (4 - K)) % 192;
};

// In-register transform of 2^K points with the domain's 2^K-th root (INVERSE: its inverse), element e at x[e * STRIDE].
// DIT: bit-reversed order in, natural order out; DIF: natural order in, bit-reversed order out -- the conventions of
// lds_ntt_group (ntt.hip), of which these are the twiddle-free lowest / highest layers.
template <int K, bool DIT, bool INVERSE, int T, int M, int Qi>
struct Ntt2kStep {
    template <typename X>
    TVM_HD static void run(X& x) {
        constexpr int R = 1 << K, h = 1 << T;
        if constexpr (Qi < R / (2 * h)) {
            constexpr int e = Qi * 2 * h + M;
            constexpr int E = (Pow2Root<K, INVERSE>::value * M * ((R / 2) >> T)) % 192;
            if constexpr (DIT) bfe_butterfly_pow2<E>(x[e], x[e + h]);
            else bfe_butterfly_dif_pow2<E>(x[e], x[e + h]);
            Ntt2kStep<K, DIT, INVERSE, T, M, Qi + 1>::run(x);
        } else if constexpr (M + 1 < h) {
            Ntt2kStep<K, DIT, INVERSE, T, M + 1, 0>::run(x);
        } else if constexpr (DIT ? (T + 1 < K) : (T > 0)) {
            Ntt2kStep<K, DIT, INVERSE, DIT ? T + 1 : T - 1, 0, 0>::run(x);
        }
    }
};
template <int K, bool DIT, bool INVERSE>
TVM_HD void ntt_pow2_points(u64 (&x)[1 << K]) {
    if constexpr (K > 0) Ntt2kStep<K, DIT, INVERSE, DIT ? 0 : K - 1, 0, 0>::run(x);
}

}  // namespace tvm
