/*
 * tvm_oracle_fast.c -- CPU BASELINE (test infrastructure, NOT product code).
 *
 * The textbook oracle (tvm_oracle.c) is written to share no trick with the kernels it checks, which makes it a poor
 * stand-in for "the reference's CPU path timed beside the GPU" (its Tip5 does the MDS layer as 256 modular
 * multiplications per round).  This file restates the SAME functions the way a competent CPU implementation does them --
 * the shapes the reference itself uses -- and is what bench.py's `cpu_baseline` leg times (kind "port": the Rust prover
 * cannot be built here).  tests/test_oracle_fast.py holds every function equal, bit for bit, to the textbook oracle.
 *
 *   Tip5            twenty-first's structure: the S-box lookup on the bytes of the Montgomery word, x^7 by 4 products,
 *                   the circulant MDS as two integer convolutions on the 32-bit halves of the state (entries < 2^16, so
 *                   sums stay below 2^53) recombined mod p (tip-0005.md:54-99; triton-air/src/table/hash.rs:50-68)
 *   row hashing     master_table.rs:455-468: parallel over rows (the reference: rayon)
 *   Merkle tree     [twenty-first MerkleTree::par_new]: parallel over the nodes of a level
 *   LDE             master_table.rs:258-322 / arithmetic_domain.rs:141-170: one inverse transform per column, one forward
 *                   transform per (column, coset of the trace domain); parallel over those tasks, table-driven twiddles
 *   AIR             master_table.rs:1264-1363: the lowered circuit walked once per row with every node TYPED -- base-field
 *                   nodes cost one product, mixed products three, only extension-field products nine -- which is what
 *                   the reference's generated evaluator does; parallel over rows
 *   DEEP            stark.rs:1360-1379 with the inversions batched per thread chunk
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "tip5_constants.h"
#include "tvm_oracle.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;

#define P 0xFFFFFFFF00000001ull

/* Montgomery product for p = 2^64 - 2^32 + 1 using the shape of p (2^64 = 2^32 - 1, 2^96 = -1 mod p) */
static inline u64 fmul(u64 a, u64 b) {   /* (branch-free throughout: the conditions are coin flips) */
    const u128 t = (u128)a * b;
    const u64 lo = (u64)t, hi = (u64)(t >> 64);
    const u64 x = lo + (lo << 32);
    const u64 e = x < lo;
    const u64 y = x - (x >> 32) - e;
    return hi - y - (0xFFFFFFFFull & -(u64)(hi < y));
}
static inline u64 fadd(u64 a, u64 b) {
    const u64 s = a + b;
    return s - (P & -(u64)((s < a) | (s >= P)));
}
static inline u64 fsub(u64 a, u64 b) { return a - b + (P & -(u64)(a < b)); }
static inline u64 fnew(u64 v) { return fmul(v % P, 0xFFFFFFFE00000001ull); }
static u64 fpow(u64 a, u64 e) {
    u64 r = fnew(1);
    for (; e; e >>= 1, a = fmul(a, a))
        if (e & 1) r = fmul(r, a);
    return r;
}
static inline u64 reduce128(u128 v) {  /* v < 2^96 -> v mod p */
    const u64 lo = (u64)v, hi = (u64)(v >> 64);       /* hi < 2^32: hi * 2^64 = hi * (2^32 - 1) */
    u64 r = lo, add = (hi << 32) - hi;
    r += add;
    if (r < add || r >= P) r -= P;
    return r;
}

/* ------------------------------------------------------------------ Tip5 */
static u32 MDS_ROWS[16][16];   /* MDS_ROWS[j][i] = M[i][j] = first_column[(i - j) mod 16] */
__attribute__((constructor)) static void make_mds_rows(void) {
    for (int j = 0; j < 16; j++)
        for (int i = 0; i < 16; i++) MDS_ROWS[j][i] = (u32)ORACLE_TIP5_MDS_FIRST_COLUMN[(16 + i - j) & 15];
}
__attribute__((target_clones("avx2", "default"))) void orcf_tip5_permutation(uint64_t st[16]) {
    for (int r = 0; r < 5; r++) {
        for (int i = 0; i < 4; i++) {
            u64 x = st[i], y = 0;
            for (int b = 0; b < 8; b++) y |= (u64)ORACLE_TIP5_LOOKUP[(x >> (8 * b)) & 0xFF] << (8 * b);
            st[i] = y;
        }
        for (int i = 4; i < 16; i++) {
            const u64 x = st[i], x2 = fmul(x, x), x4 = fmul(x2, x2);
            st[i] = fmul(fmul(x4, x2), x);
        }
        /* MDS: y = M x over the integers on the two 32-bit halves (M circulant, first column < 2^16): the matrix written out
         * row by row of x_j so that the inner loop runs over consecutive entries (32 x 32 -> 64 multiplies, vectorisable);
         * the round constants are the initial values of the sums */
        u64 ylo[16], yhi[16];
        for (int i = 0; i < 16; i++) ylo[i] = (u32)ORACLE_TIP5_ROUND_CONSTANTS[16 * r + i], yhi[i] = ORACLE_TIP5_ROUND_CONSTANTS[16 * r + i] >> 32;
        for (int j = 0; j < 16; j++) {
            const u32 xl = (u32)st[j], xh = (u32)(st[j] >> 32);
            const u32* m = MDS_ROWS[j];
            for (int i = 0; i < 16; i++) {
                ylo[i] += (u64)m[i] * xl;
                yhi[i] += (u64)m[i] * xh;
            }
        }
        for (int i = 0; i < 16; i++) st[i] = reduce128((u128)ylo[i] + ((u128)yhi[i] << 32));   /* < 2^54 + 2^86 */
    }
}
static void hash_varlen(const u64* in, size_t len, u64 out[5]) {  /* as orc_hash_varlen */
    u64 st[16] = {0};
    size_t pos = 0;
    const u64 one = fnew(1);
    for (;;) {
        const size_t rem = len - pos;
        if (rem >= 10) {
            memcpy(st, in + pos, 80);
            orcf_tip5_permutation(st);
            pos += 10;
        } else {
            for (size_t i = 0; i < rem; i++) st[i] = in[pos + i];
            st[rem] = one;
            for (size_t i = rem + 1; i < 10; i++) st[i] = 0;
            orcf_tip5_permutation(st);
            break;
        }
    }
    memcpy(out, st, 40);
}
void orcf_hash_rows(const uint64_t* rows, uint64_t n_rows, uint64_t w, uint64_t* digests) {
#pragma omp parallel for schedule(static)
    for (u64 i = 0; i < n_rows; i++) hash_varlen(rows + i * w, w, digests + 5 * i);
}
void orcf_merkle_tree(const uint64_t* leaves, uint64_t n, uint64_t* nodes) {
    const u64 one = fnew(1);
    memset(nodes, 0, 40);
    memcpy(nodes + 5 * n, leaves, n * 40);
    for (u64 lvl = n >> 1; lvl >= 1; lvl >>= 1) {
#pragma omp parallel for schedule(static) if (lvl >= 256)
        for (u64 i = lvl; i < 2 * lvl; i++) {
            u64 st[16];
            memcpy(st, nodes + 10 * i, 80);
            for (int k = 10; k < 16; k++) st[k] = one;
            orcf_tip5_permutation(st);
            memcpy(nodes + 5 * i, st, 40);
        }
    }
}

/* ------------------------------------------------------------------ NTT */
/* in place, natural order in and out; tw: the n/2 powers of the generator (table shared by all transforms of a size) */
static void ntt_with_table(u64* a, u64 n, const u64* tw) {
    for (u64 i = 1, j = 0; i < n; i++) {   /* bit-reversal permutation, j = reverse(i) kept incrementally */
        u64 bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            const u64 t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    }
    for (u64 len = 2; len <= n; len <<= 1) {
        const u64 half = len >> 1, step = n / len;
        for (u64 s = 0; s < n; s += len)
            for (u64 k = 0; k < half; k++) {
                const u64 u = a[s + k], v = fmul(a[s + k + half], tw[k * step]);
                a[s + k] = fadd(u, v);
                a[s + k + half] = fsub(u, v);
            }
    }
}
static u64* twiddle_table(u64 n, u64 gen) {
    u64* tw = (u64*)malloc((n / 2 + 1) * sizeof(u64));
    u64 w = fnew(1);
    for (u64 k = 0; k < n / 2; k++) tw[k] = w, w = fmul(w, gen);
    return tw;
}
void orcf_ntt(uint64_t* a, uint64_t n, uint64_t generator) {
    if (n <= 1) return;
    u64* tw = twiddle_table(n, generator);
    ntt_with_table(a, n, tw);
    free(tw);
}

/* maybe_low_degree_extend_all_columns for base-field columns (master_table.rs:258-322, 392-403): trace [n_cols][n],
 * randomizers [n_cols][h], the evaluation domain eval = offset * <w_L>, L = X n -> the reference's row-major table [L][n_cols].
 * On the coset gamma_k <w_n> (gamma_k = offset w_L^k) the randomized interpolant t + (X^n - 1) r has the n coefficients
 * (t[m] + (gamma_k^n - 1) r[m]) gamma_k^m: one inverse transform per column, X forward transforms of length n. */
void orcf_lde_table(const uint64_t* trace, uint64_t n, uint64_t n_cols, const uint64_t* rnd, uint64_t h, orc_domain eval, uint64_t* out) {
    const u64 L = eval.length, X = L / n, one = fnew(1);
    const u64 w_n = fpow(eval.generator, X), w_n_inv = fpow(w_n, P - 2), n_inv = fpow(fnew(n), P - 2);
    u64 *tw_f = twiddle_table(n, w_n), *tw_i = twiddle_table(n, w_n_inv);
    u64* coeffs = (u64*)malloc(n_cols * n * sizeof(u64));
#pragma omp parallel for schedule(dynamic)
    for (u64 c = 0; c < n_cols; c++) {
        u64* t = coeffs + c * n;
        memcpy(t, trace + c * n, n * sizeof(u64));
        ntt_with_table(t, n, tw_i);
        for (u64 i = 0; i < n; i++) t[i] = fmul(t[i], n_inv);
    }
#pragma omp parallel
    {
        u64* buf = (u64*)malloc(n * sizeof(u64));
#pragma omp for schedule(dynamic) collapse(2)
        for (u64 c = 0; c < n_cols; c++)
            for (u64 k = 0; k < X; k++) {
                const u64 gamma = fmul(eval.offset, fpow(eval.generator, k));
                const u64 zk = fsub(fpow(gamma, n), one);
                const u64* t = coeffs + c * n;
                const u64* r = rnd + c * h;
                u64 s = one;
                for (u64 m = 0; m < n; m++) {
                    const u64 v = m < h ? fadd(t[m], fmul(zk, r[m])) : t[m];
                    buf[m] = fmul(v, s);
                    s = fmul(s, gamma);
                }
                ntt_with_table(buf, n, tw_f);
                for (u64 j = 0; j < n; j++) out[(X * j + k) * n_cols + c] = buf[j];
            }
        free(buf);
    }
    free(coeffs);
    free(tw_f);
    free(tw_i);
}

/* ------------------------------------------------------------------ AIR */
#include "air_circuit.h"

static inline void xmul(const u64* a, const u64* b, u64* o) {
    const u64 c0 = fmul(a[0], b[0]);
    const u64 c1 = fadd(fmul(a[0], b[1]), fmul(a[1], b[0]));
    const u64 c2 = fadd(fadd(fmul(a[0], b[2]), fmul(a[1], b[1])), fmul(a[2], b[0]));
    const u64 c3 = fadd(fmul(a[1], b[2]), fmul(a[2], b[1]));
    const u64 c4 = fmul(a[2], b[2]);
    o[0] = fsub(c0, c3);
    o[1] = fsub(fadd(c1, c3), c4);
    o[2] = fadd(c2, c4);
}
typedef struct {
    const u32 (*nodes)[3];
    u32 n_nodes, n_roots;
    const u32* roots;
    const u64* consts;
    const u64 (*xconsts)[3];
    unsigned char* is_x;  /* per node: extension-field typed? */
} section;
static section SECTIONS[4];
static int sections_ready = 0;
static void prepare_sections(void) {
    if (sections_ready) return;
    const section s[4] = {
        {ORACLE_AIR_INIT_NODES, ORACLE_AIR_INIT_NUM_NODES, ORACLE_AIR_INIT_NUM_ROOTS, ORACLE_AIR_INIT_ROOTS, ORACLE_AIR_INIT_CONSTS, ORACLE_AIR_INIT_XCONSTS, 0},
        {ORACLE_AIR_CONS_NODES, ORACLE_AIR_CONS_NUM_NODES, ORACLE_AIR_CONS_NUM_ROOTS, ORACLE_AIR_CONS_ROOTS, ORACLE_AIR_CONS_CONSTS, ORACLE_AIR_CONS_XCONSTS, 0},
        {ORACLE_AIR_TRAN_NODES, ORACLE_AIR_TRAN_NUM_NODES, ORACLE_AIR_TRAN_NUM_ROOTS, ORACLE_AIR_TRAN_ROOTS, ORACLE_AIR_TRAN_CONSTS, ORACLE_AIR_TRAN_XCONSTS, 0},
        {ORACLE_AIR_TERM_NODES, ORACLE_AIR_TERM_NUM_NODES, ORACLE_AIR_TERM_NUM_ROOTS, ORACLE_AIR_TERM_ROOTS, ORACLE_AIR_TERM_CONSTS, ORACLE_AIR_TERM_XCONSTS, 0}};
    for (int k = 0; k < 4; k++) {
        SECTIONS[k] = s[k];
        unsigned char* x = (unsigned char*)malloc(s[k].n_nodes);
        for (u32 i = 0; i < s[k].n_nodes; i++) {
            const u32 kind = s[k].nodes[i][0], a = s[k].nodes[i][1], b = s[k].nodes[i][2];
            x[i] = kind == 0 || kind == 2 || kind == 3 ? 0 : kind >= 7 ? (x[a] | x[b]) : 1;
        }
        SECTIONS[k].is_x = x;
    }
    sections_ready = 1;
}
/* sum_r weights[r] * root_r of one section on one (current, next) row pair; val: scratch of 3 words per node */
static void eval_section_typed(const section* s, const u64* mc, const u64* mn, const u64* ac, const u64* an, const u64* challenges,
                               const u64* weights, u64* val, u64* acc) {
    for (u32 i = 0; i < s->n_nodes; i++) {
        u64* v = val + 3 * i;
        const u32 k = s->nodes[i][0], a = s->nodes[i][1], b = s->nodes[i][2];
        const u64 *va = val + 3 * a, *vb = val + 3 * b;
        switch (k) {
            case 0: v[0] = s->consts[a]; break;
            case 1: memcpy(v, s->xconsts[a], 24); break;
            case 2: v[0] = mc[a]; break;
            case 3: v[0] = mn[a]; break;
            case 4: memcpy(v, ac + 3 * a, 24); break;
            case 5: memcpy(v, an + 3 * a, 24); break;
            case 6: memcpy(v, challenges + 3 * a, 24); break;
            case 7:
                if (!s->is_x[i]) v[0] = fadd(va[0], vb[0]);
                else if (s->is_x[a] && s->is_x[b]) v[0] = fadd(va[0], vb[0]), v[1] = fadd(va[1], vb[1]), v[2] = fadd(va[2], vb[2]);
                else if (s->is_x[a]) v[0] = fadd(va[0], vb[0]), v[1] = va[1], v[2] = va[2];
                else v[0] = fadd(va[0], vb[0]), v[1] = vb[1], v[2] = vb[2];
                break;
            default:
                if (!s->is_x[i]) v[0] = fmul(va[0], vb[0]);
                else if (s->is_x[a] && s->is_x[b]) xmul(va, vb, v);
                else if (s->is_x[a]) v[0] = fmul(va[0], vb[0]), v[1] = fmul(va[1], vb[0]), v[2] = fmul(va[2], vb[0]);
                else v[0] = fmul(vb[0], va[0]), v[1] = fmul(vb[1], va[0]), v[2] = fmul(vb[2], va[0]);
                break;
        }
    }
    acc[0] = acc[1] = acc[2] = 0;
    for (u32 r = 0; r < s->n_roots; r++) {
        const u32 root = s->roots[r];
        const u64 *v = val + 3 * root, *w = weights + 3 * r;
        u64 t[3];
        if (s->is_x[root]) xmul(v, w, t);
        else t[0] = fmul(w[0], v[0]), t[1] = fmul(w[1], v[0]), t[2] = fmul(w[2], v[0]);
        acc[0] = fadd(acc[0], t[0]), acc[1] = fadd(acc[1], t[1]), acc[2] = fadd(acc[2], t[2]);
    }
}
/* all_quotients_combined (master_table.rs:1264-1363): same arguments and result as orc_quotients_combined */
void orcf_quotients_combined(const uint64_t* main_rows, uint64_t n_main, const uint64_t* aux_rows, uint64_t n_aux, orc_domain trace,
                             orc_domain q, const uint64_t* challenges, const uint64_t* weights, uint64_t* out) {
    prepare_sections();
    const u64 n = q.length, unit = q.length / trace.length;
    u64 *z[4];
    for (int k = 0; k < 4; k++) z[k] = (u64*)malloc(n * 8);
    orc_zerofier_inverses(trace, q, z[0], z[1], z[2], z[3]);
    const u64* w[4];
    w[0] = weights;
    for (int k = 1; k < 4; k++) w[k] = w[k - 1] + 3 * SECTIONS[k - 1].n_roots;
    u32 max_nodes = 0;
    for (int k = 0; k < 4; k++) max_nodes = SECTIONS[k].n_nodes > max_nodes ? SECTIONS[k].n_nodes : max_nodes;
#pragma omp parallel
    {
        u64* val = (u64*)malloc((size_t)max_nodes * 24);
#pragma omp for schedule(static)
        for (u64 i = 0; i < n; i++) {
            const u64 nx = (i + unit) % n;
            const u64 *mc = main_rows + i * n_main, *mn = main_rows + nx * n_main;
            const u64 *ac = aux_rows + i * n_aux * 3, *an = aux_rows + nx * n_aux * 3;
            u64 quot[3] = {0, 0, 0}, acc[3];
            for (int k = 0; k < 4; k++) {
                eval_section_typed(&SECTIONS[k], mc, mn, ac, an, challenges, w[k], val, acc);
                for (int j = 0; j < 3; j++) quot[j] = fadd(quot[j], fmul(acc[j], z[k][i]));
            }
            memcpy(out + 3 * i, quot, 24);
        }
        free(val);
    }
    for (int k = 0; k < 4; k++) free(z[k]);
}

/* ------------------------------------------------------------------ DEEP */
/* deep_codeword (stark.rs:1360-1379): out[i] = (cw[i] - val) / (x_i - pt), the inversions batched per chunk of 1024 */
void orcf_deep_codeword(const uint64_t* cw, orc_domain d, const uint64_t pt[3], const uint64_t val[3], uint64_t* out) {
    const u64 n = d.length;
    enum { CHUNK = 1024 };
#pragma omp parallel for schedule(static)
    for (u64 base = 0; base < n; base += CHUNK) {
        const u64 m = n - base < CHUNK ? n - base : CHUNK;
        u64 den[CHUNK][3], pre[CHUNK][3];
        u64 x = fmul(d.offset, fpow(d.generator, base));
        u64 run[3] = {fnew(1), 0, 0};
        for (u64 i = 0; i < m; i++) {
            den[i][0] = fsub(x, pt[0]), den[i][1] = fsub(0, pt[1]), den[i][2] = fsub(0, pt[2]);
            memcpy(pre[i], run, 24);
            xmul(run, den[i], run);
            x = fmul(x, d.generator);
        }
        u64 inv[3];
        orc_xfe_inv(run, inv);
        for (u64 i = m; i-- > 0;) {
            u64 di[3], num[3];
            xmul(inv, pre[i], di);
            xmul(inv, den[i], inv);
            for (int k = 0; k < 3; k++) num[k] = fsub(cw[3 * (base + i) + k], val[k]);
            xmul(num, di, out + 3 * (base + i));
        }
    }
}
