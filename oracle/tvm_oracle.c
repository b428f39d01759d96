/*
 * tvm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See tvm_oracle.h.
 *
 * Every function cites the reference file:line it restates.  The arithmetic is deliberately the
 * slow, textbook form (generic REDC with unsigned __int128, bit-reversal radix-2 NTT) so that it
 * shares no trick with the HIP kernels it checks.
 */
#include "tvm_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "tip5_constants.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;

#define P 0xFFFFFFFF00000001ull
/* -p^{-1} mod 2^64; p*(2^32+1) = 2^96+1 == 1 (mod 2^64) so p^{-1} = 2^32+1 */
#define NINV 0xFFFFFFFEFFFFFFFFull
/* R^2 mod p with R = 2^64: 2^128 mod p = -2^32 mod p */
#define R2 0xFFFFFFFE00000001ull

/* ------------------------------------------------------------------ base field */
/* Montgomery product a*b*2^-64 mod p, canonical.  tip-0005.md:85-89 (representation), textbook REDC. */
static inline u64 mmul(u64 a, u64 b) {
    u128 t = (u128)a * b;
    u64 m = (u64)t * NINV;
    u128 mp = (u128)m * P;
    u64 carry = ((u64)t != 0); /* low words cancel to exactly 2^64 unless both are 0 */
    u128 r = (u128)(u64)(t >> 64) + (u64)(mp >> 64) + carry;
    if (r >= P) r -= P;
    return (u64)r;
}
static inline u64 madd(u64 a, u64 b) {
    u128 s = (u128)a + b;
    if (s >= P) s -= P;
    return (u64)s;
}
static inline u64 msub(u64 a, u64 b) { return a >= b ? a - b : (u64)((u128)a + P - b); }
static inline u64 mneg(u64 a) { return a ? P - a : 0; }

uint64_t orc_bfe_new(uint64_t v) { return mmul(v % P, R2); }
void orc_bfe_new_array(uint64_t* a, uint64_t n) { for (uint64_t i = 0; i < n; i++) a[i] = orc_bfe_new(a[i]); }
void orc_bfe_value_array(uint64_t* a, uint64_t n) { for (uint64_t i = 0; i < n; i++) a[i] = mmul(a[i], 1); }
uint64_t orc_bfe_value(uint64_t raw) { return mmul(raw, 1); }
uint64_t orc_bfe_add(uint64_t a, uint64_t b) { return madd(a, b); }
uint64_t orc_bfe_sub(uint64_t a, uint64_t b) { return msub(a, b); }
uint64_t orc_bfe_mul(uint64_t a, uint64_t b) { return mmul(a, b); }
uint64_t orc_bfe_pow(uint64_t a, uint64_t e) {
    u64 r = orc_bfe_new(1);
    while (e) {
        if (e & 1) r = mmul(r, a);
        a = mmul(a, a);
        e >>= 1;
    }
    return r;
}
uint64_t orc_bfe_inv(uint64_t a) { return orc_bfe_pow(a, P - 2); }
/* [twenty-first, not in tree; pinned by the proof snapshots, tests/test_proof_snapshot.py] BFieldElement::generator() = 7 */
uint64_t orc_bfe_generator(void) { return orc_bfe_new(7); }
/* [twenty-first, not in tree; pinned by the proof snapshots, tests/test_proof_snapshot.py] primitive_root_of_unity(2^k): the 2^32-th root is
 * 7^((p-1)/2^32) = 1753635133440165772, smaller orders by repeated squaring.  Callers of the
 * product library always pass domain generators explicitly, so this only feeds tests. */
uint64_t orc_bfe_primitive_root(uint64_t order) {
    u64 r = orc_bfe_pow(orc_bfe_new(7), (P - 1) >> 32);
    u64 o = 1ull << 32;
    while (o > order) {
        r = mmul(r, r);
        o >>= 1;
    }
    return r;
}
/* Montgomery's trick; same values as element-wise inversion (master_table.rs:1200 uses batch_inversion) */
void orc_bfe_batch_inv(uint64_t* a, size_t n) {
    if (!n) return;
    u64* pre = (u64*)malloc(n * sizeof(u64));
    u64 acc = orc_bfe_new(1);
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        acc = mmul(acc, a[i]);
    }
    u64 inv = orc_bfe_inv(acc);
    for (size_t i = n; i-- > 0;) {
        u64 t = mmul(inv, pre[i]);
        inv = mmul(inv, a[i]);
        a[i] = t;
    }
    free(pre);
}

/* ------------------------------------------------------------------ extension field */
/* F_p[X]/(X^3 - X + 1), specification/src/isa.md:8 */
void orc_xfe_add(const u64* a, const u64* b, u64* o) {
    for (int i = 0; i < 3; i++) o[i] = madd(a[i], b[i]);
}
void orc_xfe_sub(const u64* a, const u64* b, u64* o) {
    for (int i = 0; i < 3; i++) o[i] = msub(a[i], b[i]);
}
void orc_xfe_mul(const u64* a, const u64* b, u64* o) {
    /* schoolbook product, then X^3 = X - 1 and X^4 = X^2 - X */
    u64 c0 = mmul(a[0], b[0]);
    u64 c1 = madd(mmul(a[0], b[1]), mmul(a[1], b[0]));
    u64 c2 = madd(madd(mmul(a[0], b[2]), mmul(a[1], b[1])), mmul(a[2], b[0]));
    u64 c3 = madd(mmul(a[1], b[2]), mmul(a[2], b[1]));
    u64 c4 = mmul(a[2], b[2]);
    u64 r0 = msub(c0, c3);
    u64 r1 = msub(madd(c1, c3), c4);
    u64 r2 = madd(c2, c4);
    o[0] = r0; o[1] = r1; o[2] = r2;
}
static void xfe_mul_bfe(const u64* a, u64 b, u64* o) {
    for (int i = 0; i < 3; i++) o[i] = mmul(a[i], b);
}
void orc_xfe_pow(const u64* a, u64 e, u64* o) {
    u64 r[3] = {orc_bfe_new(1), 0, 0}, b[3] = {a[0], a[1], a[2]};
    while (e) {
        if (e & 1) orc_xfe_mul(r, b, r);
        orc_xfe_mul(b, b, b);
        e >>= 1;
    }
    memcpy(o, r, 24);
}
/* inverse through the adjugate of the multiplication-by-a matrix (unique answer, any method) */
void orc_xfe_inv(const u64* a, u64* o) {
    u64 a0 = a[0], a1 = a[1], a2 = a[2];
    u64 s = madd(a0, a2);
    u64 c00 = msub(mmul(s, s), mmul(msub(a1, a2), a1));
    u64 c01 = mneg(msub(mmul(a1, s), mmul(msub(a1, a2), a2)));
    u64 c02 = msub(mmul(a1, a1), mmul(s, a2));
    u64 det = msub(msub(mmul(a0, c00), mmul(a2, c01)), mmul(a1, c02));
    u64 di = orc_bfe_inv(det);
    o[0] = mmul(c00, di);
    o[1] = mmul(c01, di);
    o[2] = mmul(c02, di);
}
void orc_xfe_batch_inv(uint64_t* a, size_t n) {
    if (!n) return;
    u64* pre = (u64*)malloc(n * 24);
    u64 acc[3] = {orc_bfe_new(1), 0, 0};
    for (size_t i = 0; i < n; i++) {
        memcpy(pre + 3 * i, acc, 24);
        orc_xfe_mul(acc, a + 3 * i, acc);
    }
    u64 inv[3];
    orc_xfe_inv(acc, inv);
    for (size_t i = n; i-- > 0;) {
        u64 t[3];
        orc_xfe_mul(inv, pre + 3 * i, t);
        orc_xfe_mul(inv, a + 3 * i, inv);
        memcpy(a + 3 * i, t, 24);
    }
    free(pre);
}

/* ------------------------------------------------------------------ domains */
static int ilog2(u64 n) {
    int l = 0;
    while ((1ull << l) < n) l++;
    return l;
}
/* arithmetic_domain.rs:78-85 */
orc_domain orc_domain_of_length(uint64_t length) {
    orc_domain d = {orc_bfe_new(1), orc_bfe_primitive_root(length), length};
    return d;
}
/* arithmetic_domain.rs:280-296 */
orc_domain orc_domain_pow(orc_domain d, uint64_t e) {
    orc_domain r = {orc_bfe_pow(d.offset, e), orc_bfe_pow(d.generator, e), d.length / e ? d.length / e : 1};
    return r;
}
/* arithmetic_domain.rs:227-229 */
uint64_t orc_domain_value(orc_domain d, uint64_t i) { return mmul(orc_bfe_pow(d.generator, i), d.offset); }
/* arithmetic_domain.rs:232-246 */
void orc_domain_values(orc_domain d, uint64_t* out) {
    u64 acc = orc_bfe_new(1);
    for (u64 i = 0; i < d.length; i++) {
        out[i] = mmul(acc, d.offset);
        acc = mmul(acc, d.generator);
    }
}

/* ------------------------------------------------------------------ NTT
 * [twenty-first ntt/intt, not in tree]: natural order in and out, ntt(a)[i] = sum_j a_j w^(ij),
 * w = primitive_root_of_unity(n)  (algebraically pinned by arithmetic_domain.rs:361-393,457-473).
 * Textbook: bit-reversal permutation followed by decimation-in-time butterflies. */
static void ntt_strided(u64* a, u64 n, int stride, u64 omega) {
    int lg = ilog2(n);
    for (u64 i = 0; i < n; i++) {
        u64 j = 0;
        for (int b = 0; b < lg; b++) j |= ((i >> b) & 1) << (lg - 1 - b);
        if (j > i) {
            u64 t = a[i * stride];
            a[i * stride] = a[j * stride];
            a[j * stride] = t;
        }
    }
    for (u64 len = 2; len <= n; len <<= 1) {
        u64 wl = orc_bfe_pow(omega, n / len);
        u64 half = len >> 1;
        u64* tw = (u64*)malloc(half * sizeof(u64));
        u64 w = orc_bfe_new(1);
        for (u64 k = 0; k < half; k++) {
            tw[k] = w;
            w = mmul(w, wl);
        }
        for (u64 s = 0; s < n; s += len)
            for (u64 k = 0; k < half; k++) {
                u64 u = a[(s + k) * stride];
                u64 v = mmul(a[(s + k + half) * stride], tw[k]);
                a[(s + k) * stride] = madd(u, v);
                a[(s + k + half) * stride] = msub(u, v);
            }
        free(tw);
    }
}
void orc_ntt(uint64_t* a, uint64_t n, int fk) {
    if (n <= 1) return;
    u64 w = orc_bfe_primitive_root(n);
    for (int c = 0; c < fk; c++) ntt_strided(a + c, n, fk, w);
}
void orc_intt(uint64_t* a, uint64_t n, int fk) {
    if (n <= 1) return;
    u64 w = orc_bfe_inv(orc_bfe_primitive_root(n));
    u64 ninv = orc_bfe_inv(orc_bfe_new(n));
    for (int c = 0; c < fk; c++) {
        ntt_strided(a + c, n, fk, w);
        for (u64 i = 0; i < n; i++) a[i * fk + c] = mmul(a[i * fk + c], ninv);
    }
}
/* the same two transforms for an explicitly given generator (domains carry their own generator) */
static void ntt_gen(u64* a, u64 n, int fk, u64 gen) {
    if (n <= 1) return;
    for (int c = 0; c < fk; c++) ntt_strided(a + c, n, fk, gen);
}
static void intt_gen(u64* a, u64 n, int fk, u64 gen) {
    if (n <= 1) return;
    u64 w = orc_bfe_inv(gen), ninv = orc_bfe_inv(orc_bfe_new(n));
    for (int c = 0; c < fk; c++) {
        ntt_strided(a + c, n, fk, w);
        for (u64 i = 0; i < n; i++) a[i * fk + c] = mmul(a[i * fk + c], ninv);
    }
}

/* [twenty-first fast_coset_evaluate, not in tree]: coefficient i times offset^i, zero-pad, NTT */
static void fast_coset_evaluate(int fk, const u64* chunk, u64 n_chunk, orc_domain d, u64* out) {
    memset(out, 0, d.length * fk * sizeof(u64));
    u64 s = orc_bfe_new(1);
    for (u64 i = 0; i < n_chunk; i++) {
        for (int c = 0; c < fk; c++) out[i * fk + c] = mmul(chunk[i * fk + c], s);
        s = mmul(s, d.offset);
    }
    ntt_gen(out, d.length, fk, d.generator);
}
/* arithmetic_domain.rs:141-170 (chunk folding at :153-167) */
void orc_coset_evaluate(int fk, const uint64_t* coeffs, uint64_t n_coeffs, orc_domain d, uint64_t* out) {
    u64 len = d.length;
    if (n_coeffs == 0) {
        memset(out, 0, len * fk * sizeof(u64));
        return;
    }
    u64 first = n_coeffs < len ? n_coeffs : len;
    fast_coset_evaluate(fk, coeffs, first, d, out);
    if (n_coeffs <= len) return;
    u64* tmp = (u64*)malloc(len * fk * sizeof(u64));
    for (u64 k = 1; k * len < n_coeffs; k++) {
        u64 cnt = n_coeffs - k * len < len ? n_coeffs - k * len : len;
        fast_coset_evaluate(fk, coeffs + k * len * fk, cnt, d, tmp);
        u64 scaled_offset = orc_bfe_pow(d.offset, k * len);
        for (u64 i = 0; i < len * fk; i++) out[i] = madd(out[i], mmul(tmp[i], scaled_offset));
    }
    free(tmp);
}
/* arithmetic_domain.rs:182-189; [twenty-first fast_coset_interpolate]: iNTT, coefficient i times offset^-i */
void orc_coset_interpolate(int fk, const uint64_t* values, orc_domain d, uint64_t* out) {
    if (out != values) memcpy(out, values, d.length * fk * sizeof(u64));
    intt_gen(out, d.length, fk, d.generator);
    u64 oi = orc_bfe_inv(d.offset), s = orc_bfe_new(1);
    for (u64 i = 0; i < d.length; i++) {
        for (int c = 0; c < fk; c++) out[i * fk + c] = mmul(out[i * fk + c], s);
        s = mmul(s, oi);
    }
}

/* ------------------------------------------------------------------ LDE */
/* master_table.rs:392-403 with arithmetic_domain.rs:262-269: trace domain offset is 1 so
 * zerofier*r = shift_by(n_rows)(r) - r.  out has 2*n_rows coefficients (zero padded). */
void orc_randomized_column_interpolant(int fk, const uint64_t* column, uint64_t n_rows,
                                       const uint64_t* randomizer, uint64_t h, uint64_t* out) {
    memset(out, 0, 2 * n_rows * fk * sizeof(u64));
    memcpy(out, column, n_rows * fk * sizeof(u64));
    intt_gen(out, n_rows, fk, orc_bfe_primitive_root(n_rows));
    for (u64 i = 0; i < h * fk; i++) {
        out[n_rows * fk + i] = madd(out[n_rows * fk + i], randomizer[i]);
        out[i] = msub(out[i], randomizer[i]);
    }
}
/* master_table.rs:258-322: every column interpolated, then evaluated on the evaluation domain into
 * a row-major [eval.length, n_cols] table. */
void orc_lde_table(int fk, const uint64_t* trace, uint64_t n_rows, uint64_t n_cols,
                   const uint64_t* randomizers, uint64_t h, orc_domain eval, uint64_t* out) {
#pragma omp parallel for schedule(dynamic)
    for (u64 c = 0; c < n_cols; c++) {
        u64* poly = (u64*)malloc(2 * n_rows * fk * sizeof(u64));
        u64* cw = (u64*)malloc(eval.length * fk * sizeof(u64));
        orc_randomized_column_interpolant(fk, trace + c * n_rows * fk, n_rows, randomizers + c * h * fk, h, poly);
        orc_coset_evaluate(fk, poly, n_rows + h, eval, cw);
        for (u64 i = 0; i < eval.length; i++)
            for (int k = 0; k < fk; k++) out[(i * n_cols + c) * fk + k] = cw[i * fk + k];
        free(poly);
        free(cw);
    }
}

/* ------------------------------------------------------------------ Tip5 */
/* tip-0005.md:54-76, S-box on raw Montgomery bytes :91-99, MDS circulant (hash.rs:50-54) */
void orc_tip5_permutation(uint64_t st[16]) {
    for (int r = 0; r < 5; r++) {
        for (int i = 0; i < 4; i++) {
            u64 x = st[i], y = 0;
            for (int b = 0; b < 8; b++) y |= (u64)ORACLE_TIP5_LOOKUP[(x >> (8 * b)) & 0xFF] << (8 * b);
            st[i] = y;
        }
        for (int i = 4; i < 16; i++) {
            u64 x = st[i], x2 = mmul(x, x), x4 = mmul(x2, x2);
            st[i] = mmul(mmul(x4, x2), x);
        }
        u64 nx[16];
        for (int i = 0; i < 16; i++) {
            u64 acc = 0;
            for (int j = 0; j < 16; j++)
                acc = madd(acc, mmul(orc_bfe_new(ORACLE_TIP5_MDS_FIRST_COLUMN[(16 + i - j) % 16]), st[j]));
            nx[i] = acc;
        }
        for (int i = 0; i < 16; i++) st[i] = madd(nx[i], ORACLE_TIP5_ROUND_CONSTANTS[16 * r + i]);
    }
}
/* The permutation's trace: the input state and the state after each of the 5 rounds (twenty-first Tip5::trace, used by
 * the VM for the Hash table: /root/reference/triton-vm/src/vm.rs:655-747, table/hash.rs:35-43).  out: [6][16]. */
void orc_tip5_trace(const uint64_t in[16], uint64_t* out) {
    u64 st[16];
    memcpy(st, in, 128);
    memcpy(out, st, 128);
    for (int r = 0; r < 5; r++) {
        for (int i = 0; i < 4; i++) {
            u64 x = st[i], y = 0;
            for (int b = 0; b < 8; b++) y |= (u64)ORACLE_TIP5_LOOKUP[(x >> (8 * b)) & 0xFF] << (8 * b);
            st[i] = y;
        }
        for (int i = 4; i < 16; i++) {
            u64 x = st[i], x2 = mmul(x, x), x4 = mmul(x2, x2);
            st[i] = mmul(mmul(x4, x2), x);
        }
        u64 nx[16];
        for (int i = 0; i < 16; i++) {
            u64 acc = 0;
            for (int j = 0; j < 16; j++)
                acc = madd(acc, mmul(orc_bfe_new(ORACLE_TIP5_MDS_FIRST_COLUMN[(16 + i - j) % 16]), st[j]));
            nx[i] = acc;
        }
        for (int i = 0; i < 16; i++) st[i] = madd(nx[i], ORACLE_TIP5_ROUND_CONSTANTS[16 * r + i]);
        memcpy(out + 16 * (r + 1), st, 128);
    }
}
/* tip-0005.md:82 fixed-length mode: capacity all ones */
void orc_hash_10(const uint64_t in[10], uint64_t out[5]) {
    u64 st[16];
    memcpy(st, in, 80);
    for (int i = 10; i < 16; i++) st[i] = orc_bfe_new(1);
    orc_tip5_permutation(st);
    memcpy(out, st, 40);
}
/* [twenty-first Tip5::hash_pair]: hash_10(left || right).  The fixed-length domain (capacity of ones) and the
 * left/right order are what the pinned AIR's Hash-table and merkle_step constraints enforce on the valid traces of
 * tests/test_vm_tables.py; the function itself has no in-tree vector. */
void orc_hash_pair(const uint64_t l[5], const uint64_t r[5], uint64_t out[5]) {
    u64 in[10];
    memcpy(in, l, 40);
    memcpy(in + 5, r, 40);
    orc_hash_10(in, out);
}
/* tip-0005.md:83 + overwrite-mode absorb (specification/src/hash-table.md:24-26; restated by
 * master_table.rs:667-716 SpongeWithPendingAbsorb): pad with 1 then 0s, capacity zero. */
void orc_hash_varlen(const uint64_t* in, size_t len, uint64_t out[5]) {
    u64 st[16] = {0};
    size_t pos = 0;
    for (;;) {
        size_t rem = len - pos;
        if (rem >= 10) {
            memcpy(st, in + pos, 80);
            orc_tip5_permutation(st);
            pos += 10;
        } else {
            for (size_t i = 0; i < rem; i++) st[i] = in[pos + i];
            st[rem] = orc_bfe_new(1);
            for (size_t i = rem + 1; i < 10; i++) st[i] = 0;
            orc_tip5_permutation(st);
            break;
        }
    }
    memcpy(out, st, 40);
}
/* master_table.rs:455-468 */
void orc_hash_rows(const uint64_t* rows, uint64_t n_rows, uint64_t w, uint64_t* digests) {
#pragma omp parallel for
    for (u64 i = 0; i < n_rows; i++) orc_hash_varlen(rows + i * w, w, digests + 5 * i);
}
/* [twenty-first MerkleTree; layout pinned by the proof snapshots]: heap order, nodes[1] root, children 2i, 2i+1 */
void orc_merkle_tree(const uint64_t* leaves, uint64_t n, uint64_t* nodes) {
    memset(nodes, 0, 40);
    memcpy(nodes + 5 * n, leaves, n * 40);
    for (u64 lvl = n >> 1; lvl >= 1; lvl >>= 1) {
#pragma omp parallel for
        for (u64 i = lvl; i < 2 * lvl; i++) orc_hash_pair(nodes + 10 * i, nodes + 10 * i + 5, nodes + 5 * i);
    }
}
/* fri.rs:343-347 + [twenty-first Digest::from(XFE); pinned by the proof snapshots]: [c0,c1,c2,0,0] */
void orc_xfe_to_digest(const uint64_t* x, uint64_t n, uint64_t* d) {
    for (u64 i = 0; i < n; i++) {
        d[5 * i] = x[3 * i]; d[5 * i + 1] = x[3 * i + 1]; d[5 * i + 2] = x[3 * i + 2];
        d[5 * i + 3] = 0; d[5 * i + 4] = 0;
    }
}

/* ------------------------------------------------------------------ quotient plumbing */
/* master_table.rs:1194-1250 */
void orc_zerofier_inverses(orc_domain trace, orc_domain q, uint64_t* init, uint64_t* cons, uint64_t* tran, uint64_t* term) {
    u64 n = q.length, one = orc_bfe_new(1);
    u64* x = (u64*)malloc(n * sizeof(u64));
    orc_domain_values(q, x);
    u64 gi = orc_bfe_inv(trace.generator);
    for (u64 i = 0; i < n; i++) {
        init[i] = msub(x[i], one);
        cons[i] = msub(orc_bfe_pow(x[i], trace.length), one);
        term[i] = msub(x[i], gi);
    }
    orc_bfe_batch_inv(init, n);
    orc_bfe_batch_inv(cons, n);
    orc_bfe_batch_inv(term, n);
    for (u64 i = 0; i < n; i++) tran[i] = mmul(msub(x[i], gi), cons[i]);
    free(x);
}
/* stark.rs:1224-1263: coset-interpolate, segment k takes coefficients k, k+4, k+8, ... */
void orc_interpolate_quotient_segments(const uint64_t* cw, orc_domain q, uint64_t* seg) {
    u64* poly = (u64*)malloc(q.length * 24);
    orc_coset_interpolate(3, cw, q, poly);
    u64 sl = q.length / 4;
    for (u64 k = 0; k < 4; k++)
        for (u64 j = 0; j < sl; j++) memcpy(seg + (k * sl + j) * 3, poly + (4 * j + k) * 3, 24);
    free(poly);
}
/* stark.rs:1302-1356 (zeta = 3, stark.rs:1801) */
void orc_randomize_quotient_segments(const uint64_t* seg, uint64_t seg_len, const uint64_t* rnd, uint64_t n_rand,
                                     orc_domain ldt, uint64_t* polys, uint64_t poly_len, uint64_t* cws) {
    memset(polys, 0, 5 * poly_len * 24);
    for (u64 k = 0; k < 4; k++) memcpy(polys + k * poly_len * 3, seg + k * seg_len * 3, seg_len * 24);
    memcpy(polys + 4 * poly_len * 3, rnd, n_rand * 24);
    u64 zeta = orc_bfe_new(3), zk = orc_bfe_pow(zeta, 4);
    for (int i = 3; i >= 0; i--) {
        u64 mzi = mneg(orc_bfe_pow(zeta, (u64)i));
        u64 s = orc_bfe_new(1); /* zk^j */
        const u64* nxt = polys + (u64)(i + 1) * poly_len * 3;
        u64* cur = polys + (u64)i * poly_len * 3;
        for (u64 j = 0; j < poly_len; j++) {
            u64 f = mmul(mzi, s);
            for (int c = 0; c < 3; c++) cur[3 * j + c] = madd(cur[3 * j + c], mmul(nxt[3 * j + c], f));
            s = mmul(s, zk);
        }
    }
    u64* col = (u64*)malloc(ldt.length * 24);
    for (u64 k = 0; k < 5; k++) {
        orc_coset_evaluate(3, polys + k * poly_len * 3, poly_len, ldt, col);
        for (u64 i = 0; i < ldt.length; i++) memcpy(cws + (i * 5 + k) * 3, col + 3 * i, 24);
    }
    free(col);
}

/* ------------------------------------------------------------------ combination / DEEP / FRI */
static void cell_times_xfe(int fk, const u64* cell, const u64* w, u64* o) {
    if (fk == 1) xfe_mul_bfe(w, cell[0], o);
    else orc_xfe_mul(cell, w, o);
}
/* master_table.rs:512-542 */
void orc_weighted_sum_of_columns(int fk, const uint64_t* trace, uint64_t n, uint64_t n_cols, const uint64_t* rnd,
                                 uint64_t h, const uint64_t* w, uint64_t* out) {
    memset(out, 0, 2 * n * 24);
    for (u64 i = 0; i < n; i++) {
        u64 acc[3] = {0, 0, 0}, t[3];
        for (u64 c = 0; c < n_cols; c++) {
            cell_times_xfe(fk, trace + (c * n + i) * fk, w + 3 * c, t);
            orc_xfe_add(acc, t, acc);
        }
        memcpy(out + 3 * i, acc, 24);
    }
    intt_gen(out, n, 3, orc_bfe_primitive_root(n));
    for (u64 j = 0; j < h; j++) {
        u64 acc[3] = {0, 0, 0}, t[3];
        for (u64 c = 0; c < n_cols; c++) {
            cell_times_xfe(fk, rnd + (c * h + j) * fk, w + 3 * c, t);
            orc_xfe_add(acc, t, acc);
        }
        orc_xfe_add(out + 3 * (n + j), acc, out + 3 * (n + j));
        orc_xfe_sub(out + 3 * j, acc, out + 3 * j);
    }
}
void orc_poly_eval_xfe(const uint64_t* co, uint64_t n, const uint64_t pt[3], uint64_t out[3]) {
    u64 acc[3] = {0, 0, 0};
    for (u64 i = n; i-- > 0;) {
        orc_xfe_mul(acc, pt, acc);
        orc_xfe_add(acc, co + 3 * i, acc);
    }
    memcpy(out, acc, 24);
}
/* master_table.rs:348-390 (barycentric over the trace domain, offset 1) */
void orc_out_of_domain_row(int fk, const uint64_t* trace, uint64_t n, uint64_t n_cols, const uint64_t* rnd,
                           uint64_t h, const uint64_t pt[3], uint64_t* out) {
    orc_domain td = orc_domain_of_length(n);
    u64* dom = (u64*)malloc(n * 8);
    u64* shift = (u64*)malloc(n * 24);
    orc_domain_values(td, dom);
    for (u64 j = 0; j < n; j++) {
        shift[3 * j] = msub(pt[0], dom[j]);
        shift[3 * j + 1] = pt[1];
        shift[3 * j + 2] = pt[2];
    }
    orc_xfe_batch_inv(shift, n);
    u64 den[3] = {0, 0, 0};
    for (u64 j = 0; j < n; j++) {
        xfe_mul_bfe(shift + 3 * j, dom[j], shift + 3 * j); /* d_j / (alpha - d_j) */
        orc_xfe_add(den, shift + 3 * j, den);
    }
    u64 deni[3], zf[3], one[3] = {orc_bfe_new(1), 0, 0};
    orc_xfe_inv(den, deni);
    orc_xfe_pow(pt, n, zf);
    orc_xfe_sub(zf, one, zf);
    for (u64 c = 0; c < n_cols; c++) {
        u64 num[3] = {0, 0, 0}, t[3], r[3] = {0, 0, 0};
        for (u64 j = 0; j < n; j++) {
            cell_times_xfe(fk, trace + (c * n + j) * fk, shift + 3 * j, t);
            orc_xfe_add(num, t, num);
        }
        for (u64 j = h; j-- > 0;) { /* Horner on the randomizer, lifted into XFE */
            orc_xfe_mul(r, pt, r);
            if (fk == 1) r[0] = madd(r[0], rnd[c * h + j]);
            else orc_xfe_add(r, rnd + (c * h + j) * 3, r);
        }
        orc_xfe_mul(num, deni, num);
        orc_xfe_mul(zf, r, r);
        orc_xfe_add(num, r, out + 3 * c);
    }
    free(dom);
    free(shift);
}
/* stark.rs:1360-1379, 2096-2103 */
void orc_deep_codeword(const uint64_t* cw, orc_domain d, const uint64_t pt[3], const uint64_t val[3], uint64_t* out) {
    u64* x = (u64*)malloc(d.length * 8);
    orc_domain_values(d, x);
    for (u64 i = 0; i < d.length; i++) {
        u64 num[3], den[3] = {msub(x[i], pt[0]), mneg(pt[1]), mneg(pt[2])}, di[3];
        orc_xfe_sub(cw + 3 * i, val, num);
        orc_xfe_inv(den, di);
        orc_xfe_mul(num, di, out + 3 * i);
    }
    free(x);
}
/* fri.rs:349-366 */
void orc_fri_split_and_fold(const uint64_t* cw, orc_domain d, const uint64_t ch[3], uint64_t* out) {
    u64 n = d.length, one = orc_bfe_new(1);
    u64 two_inv = orc_bfe_inv(orc_bfe_new(2));
    u64* x = (u64*)malloc(n * 8);
    orc_domain_values(d, x);
    orc_bfe_batch_inv(x, n);
    for (u64 i = 0; i < n / 2; i++) {
        u64 s[3], a[3], b[3], l[3], r[3];
        xfe_mul_bfe(ch, x[i], s);
        a[0] = madd(one, s[0]); a[1] = s[1]; a[2] = s[2];
        b[0] = msub(one, s[0]); b[1] = mneg(s[1]); b[2] = mneg(s[2]);
        orc_xfe_mul(a, cw + 3 * i, l);
        orc_xfe_mul(b, cw + 3 * (n / 2 + i), r);
        orc_xfe_add(l, r, l);
        xfe_mul_bfe(l, two_inv, out + 3 * i);
    }
    free(x);
}

/* ------------------------------------------------------------------ AIR / quotients
 * master_table.rs:1264-1363 with the generated evaluate_*_constraints replaced by a literal walk of
 * the lowered circuit DAG (air_circuit.h, exported by tools/air/export.py; every node is evaluated as
 * an XFieldElement, base-field values lifted).  Tables are the reference's row-major quotient-domain
 * views: main [q.length][n_main], aux [q.length][n_aux][3]. */
#include "air_circuit.h"

static void eval_section(const uint32_t (*nodes)[3], uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots,
                         const u64* consts, const u64 (*xconsts)[3], const u64* mc, const u64* mn, const u64* ac,
                         const u64* an, const u64* challenges, const u64* weights, u64* acc /* xfe */) {
    u64* val = (u64*)malloc((size_t)n_nodes * 24);
    for (uint32_t i = 0; i < n_nodes; i++) {
        u64* v = val + 3 * i;
        uint32_t k = nodes[i][0], a = nodes[i][1], b = nodes[i][2];
        switch (k) {
            case 0: v[0] = consts[a]; v[1] = v[2] = 0; break;
            case 1: memcpy(v, xconsts[a], 24); break;
            case 2: v[0] = mc[a]; v[1] = v[2] = 0; break;
            case 3: v[0] = mn[a]; v[1] = v[2] = 0; break;
            case 4: memcpy(v, ac + 3 * a, 24); break;
            case 5: memcpy(v, an + 3 * a, 24); break;
            case 6: memcpy(v, challenges + 3 * a, 24); break;
            case 7: orc_xfe_add(val + 3 * a, val + 3 * b, v); break;
            default: orc_xfe_mul(val + 3 * a, val + 3 * b, v); break;
        }
    }
    acc[0] = acc[1] = acc[2] = 0;
    for (uint32_t r = 0; r < n_roots; r++) {
        u64 t[3];
        orc_xfe_mul(val + 3 * roots[r], weights + 3 * r, t);
        orc_xfe_add(acc, t, acc);
    }
    free(val);
}

/* All 604 constraint values on one (current, next) row pair, in the evaluator's order (per section: base-field
 * constraints first, then extension-field ones; sections init, cons, tran, term) -- what the reference's
 * MasterAuxTable::evaluate_{initial,consistency,transition,terminal}_constraints return
 * (master_table.rs:2349-2392 calls them with BFieldElement and with XFieldElement main rows).
 * main_words = 1: main rows are base-field words; 3: main rows are XFieldElements. out: [604][3]. */
static void eval_section_values(const uint32_t (*nodes)[3], uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots,
                                const u64* consts, const u64 (*xconsts)[3], const u64* mc, const u64* mn, const u64* ac,
                                const u64* an, const u64* challenges, int main_words, u64* out) {
    u64* val = (u64*)malloc((size_t)n_nodes * 24);
    for (uint32_t i = 0; i < n_nodes; i++) {
        u64* v = val + 3 * i;
        uint32_t k = nodes[i][0], a = nodes[i][1], b = nodes[i][2];
        switch (k) {
            case 0: v[0] = consts[a]; v[1] = v[2] = 0; break;
            case 1: memcpy(v, xconsts[a], 24); break;
            case 2: case 3: {
                const u64* m = (k == 2 ? mc : mn) + (size_t)main_words * a;
                v[0] = m[0]; v[1] = main_words == 3 ? m[1] : 0; v[2] = main_words == 3 ? m[2] : 0;
                break;
            }
            case 4: memcpy(v, ac + 3 * a, 24); break;
            case 5: memcpy(v, an + 3 * a, 24); break;
            case 6: memcpy(v, challenges + 3 * a, 24); break;
            case 7: orc_xfe_add(val + 3 * a, val + 3 * b, v); break;
            default: orc_xfe_mul(val + 3 * a, val + 3 * b, v); break;
        }
    }
    for (uint32_t r = 0; r < n_roots; r++) memcpy(out + 3 * r, val + 3 * roots[r], 24);
    free(val);
}

void orc_air_constraint_values(const uint64_t* mc, const uint64_t* mn, const uint64_t* ac, const uint64_t* an,
                               const uint64_t* challenges, int main_words, uint64_t* out) {
    u64* o = out;
    eval_section_values(ORACLE_AIR_INIT_NODES, ORACLE_AIR_INIT_NUM_NODES, ORACLE_AIR_INIT_ROOTS, ORACLE_AIR_INIT_NUM_ROOTS,
                        ORACLE_AIR_INIT_CONSTS, ORACLE_AIR_INIT_XCONSTS, mc, mn, ac, an, challenges, main_words, o);
    o += 3 * ORACLE_AIR_INIT_NUM_ROOTS;
    eval_section_values(ORACLE_AIR_CONS_NODES, ORACLE_AIR_CONS_NUM_NODES, ORACLE_AIR_CONS_ROOTS, ORACLE_AIR_CONS_NUM_ROOTS,
                        ORACLE_AIR_CONS_CONSTS, ORACLE_AIR_CONS_XCONSTS, mc, mn, ac, an, challenges, main_words, o);
    o += 3 * ORACLE_AIR_CONS_NUM_ROOTS;
    eval_section_values(ORACLE_AIR_TRAN_NODES, ORACLE_AIR_TRAN_NUM_NODES, ORACLE_AIR_TRAN_ROOTS, ORACLE_AIR_TRAN_NUM_ROOTS,
                        ORACLE_AIR_TRAN_CONSTS, ORACLE_AIR_TRAN_XCONSTS, mc, mn, ac, an, challenges, main_words, o);
    o += 3 * ORACLE_AIR_TRAN_NUM_ROOTS;
    eval_section_values(ORACLE_AIR_TERM_NODES, ORACLE_AIR_TERM_NUM_NODES, ORACLE_AIR_TERM_ROOTS, ORACLE_AIR_TERM_NUM_ROOTS,
                        ORACLE_AIR_TERM_CONSTS, ORACLE_AIR_TERM_XCONSTS, mc, mn, ac, an, challenges, main_words, o);
}

void orc_quotients_combined(const uint64_t* main_rows, uint64_t n_main, const uint64_t* aux_rows, uint64_t n_aux,
                            orc_domain trace, orc_domain q, const uint64_t* challenges, const uint64_t* weights,
                            uint64_t* out) {
    u64 n = q.length, unit = q.length / trace.length;
    u64 *zi = (u64*)malloc(n * 8), *zc = (u64*)malloc(n * 8), *zt = (u64*)malloc(n * 8), *ze = (u64*)malloc(n * 8);
    orc_zerofier_inverses(trace, q, zi, zc, zt, ze);
    const u64* w_init = weights;
    const u64* w_cons = w_init + 3 * ORACLE_AIR_INIT_NUM_ROOTS;
    const u64* w_tran = w_cons + 3 * ORACLE_AIR_CONS_NUM_ROOTS;
    const u64* w_term = w_tran + 3 * ORACLE_AIR_TRAN_NUM_ROOTS;
#pragma omp parallel for
    for (u64 i = 0; i < n; i++) {
        u64 nx = (i + unit) % n;
        const u64 *mc = main_rows + i * n_main, *mn = main_rows + nx * n_main;
        const u64 *ac = aux_rows + i * n_aux * 3, *an = aux_rows + nx * n_aux * 3;
        u64 acc[3], t[3], quot[3] = {0, 0, 0};
        eval_section(ORACLE_AIR_INIT_NODES, ORACLE_AIR_INIT_NUM_NODES, ORACLE_AIR_INIT_ROOTS, ORACLE_AIR_INIT_NUM_ROOTS,
                     ORACLE_AIR_INIT_CONSTS, ORACLE_AIR_INIT_XCONSTS, mc, mn, ac, an, challenges, w_init, acc);
        xfe_mul_bfe(acc, zi[i], t); orc_xfe_add(quot, t, quot);
        eval_section(ORACLE_AIR_CONS_NODES, ORACLE_AIR_CONS_NUM_NODES, ORACLE_AIR_CONS_ROOTS, ORACLE_AIR_CONS_NUM_ROOTS,
                     ORACLE_AIR_CONS_CONSTS, ORACLE_AIR_CONS_XCONSTS, mc, mn, ac, an, challenges, w_cons, acc);
        xfe_mul_bfe(acc, zc[i], t); orc_xfe_add(quot, t, quot);
        eval_section(ORACLE_AIR_TRAN_NODES, ORACLE_AIR_TRAN_NUM_NODES, ORACLE_AIR_TRAN_ROOTS, ORACLE_AIR_TRAN_NUM_ROOTS,
                     ORACLE_AIR_TRAN_CONSTS, ORACLE_AIR_TRAN_XCONSTS, mc, mn, ac, an, challenges, w_tran, acc);
        xfe_mul_bfe(acc, zt[i], t); orc_xfe_add(quot, t, quot);
        eval_section(ORACLE_AIR_TERM_NODES, ORACLE_AIR_TERM_NUM_NODES, ORACLE_AIR_TERM_ROOTS, ORACLE_AIR_TERM_NUM_ROOTS,
                     ORACLE_AIR_TERM_CONSTS, ORACLE_AIR_TERM_XCONSTS, mc, mn, ac, an, challenges, w_term, acc);
        xfe_mul_bfe(acc, ze[i], t); orc_xfe_add(quot, t, quot);
        memcpy(out + 3 * i, quot, 24);
    }
    free(zi); free(zc); free(zt); free(ze);
}
