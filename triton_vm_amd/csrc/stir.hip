// stir.hip -- the device side of the STIR prover (/root/reference/triton-vm/src/low_degree_test/stir.rs:885-993).
//
// Per round the reference (all on the CPU, polynomial arithmetic from twenty-first):
//   StirMerkleTree::new      stack 4 codeword entries taken at distance n/4 into one leaf, hash_varlen, Merkle tree
//                            (stir.rs:1380-1419)                                          -> tvm_stir_merkle_tree
//   fold_polynomial          chunks of 4 coefficients evaluated at the folding randomness (stir.rs:1132-1147)
//                                                                                         -> tvm_fold_polynomial
//   next_round_domain.evaluate, out-of-domain evaluations                                 -> tvm_evaluate, tvm_evaluate_at_points
//   quotient = (folded - Ans) / Zerofier,  next = quotient * (sum_i r^i X^i)   (stir.rs:945-966, Polynomial
//   interpolate / zerofier / division / multiplication: O(n k) on the CPU)                -> tvm_stir_next_polynomial
// The last step never forms Zerofier or the product as polynomials: on a coset that avoids the quotient set it
// evaluates folded, Ans, Zerofier (as a product of linear factors) and the degree-correction series pointwise,
// divides and multiplies pointwise and interpolates once.  The division is exact (folded - Ans vanishes on the
// quotient set) and deg(next) = deg(folded) < |coset|, so the interpolant IS the reference's polynomial.
#include "kernels.h"
#include "tip5.h"

namespace tvm {

TVM_D xfe stir_ld(const u64* p) { return xfe_make(p[0], p[1], p[2]); }

// leaf i = Tip5::hash_varlen(cw[i], cw[i + d], ..., cw[i + (sh-1) d]) with d = n / sh, XFEs flattened c0,c1,c2: the matrix-core
// form of the permutation (tip5.h), four lanes per leaf, sixteen leaves per wavefront, as the table rows are hashed
// (hash.hip: k_hash_rows_mfma).  Lane (n, g) absorbs the words g, g + 4 (and g + 8 for g < 2) of each block of ten.
__global__ void __launch_bounds__(256, 6) k_hash_stacked(const u64* __restrict__ cw, u64 d, int stack_height,
                                                         u64* __restrict__ digests) {
    __shared__ unsigned char lut[256];
    __shared__ int ctab[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16; i += blockDim.x) ctab[i] = d_tip5_mfma_table.v[i];
    tip5_stage_lut_lowered(lut, tid, blockDim.x);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    u64 i = ((u64)blockIdx.x * 4 + (tid >> 6)) * 16 + n;
    const bool live = i < d;  // every lane of a wavefront takes part in the matrix instructions
    if (!live) i = d - 1;
    const Tip5MfmaOperands a = tip5_mfma_matrix_operands(lane);
    const int W = 3 * stack_height;
    u64 st[4] = {0, 0, 0, 0};
    const int n_perms = W / TIP5_RATE + 1;
    for (int perm = 0; perm < n_perms; perm++) {
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
            const int q = g + 4 * t3;  // word of the state, overwritten if it is in the rate part
            const int wi = perm * TIP5_RATE + q;
            if (q < TIP5_RATE) st[t3] = wi < W ? cw[(i + (u64)(wi / 3) * d) * 3 + wi % 3] : (wi == W ? TVM_ONE : 0);  // padding: 1, 0s
        }
        tip5_permute_mfma(st, a, g, lut, ctab);
    }
    if (live) {
        digests[i * 5 + g] = st[0];
        if (g == 0) digests[i * 5 + 4] = st[1];
    }
}

// out[i] = sum_j poly[ff*i + j] * r^j  (Horner from the top coefficient of the chunk; the last chunk may be short)
__global__ void k_fold_polynomial(const u64* __restrict__ poly, u64 n, int ff, u64 r0, u64 r1, u64 r2, u64 n_out,
                                  u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const xfe r = xfe_make(r0, r1, r2);
    const u64 base = i * (u64)ff;
    int len = ff;
    if (base + len > n) len = (int)(n - base);
    xfe acc = xfe_zero();
    for (int j = len - 1; j >= 0; j--) acc = xfe_add(xfe_mul(acc, r), stir_ld(poly + 3 * (base + j)));
    out[3 * i] = acc.c0;
    out[3 * i + 1] = acc.c1;
    out[3 * i + 2] = acc.c2;
}

struct StirQuotientArgs {
    u64* vals;          // in: folded(x_i), out: next(x_i); work_domain.length XFE
    u64 n;
    u64 offset, gen;
    const u64* points;  // k XFE: the quotient set
    const u64* answer;  // k XFE: coefficients of Ans (degree < k)
    const u64* answer_values;  // Ans on the work domain (n XFE), or null: Horner over `answer` per point
    u32 k, kb;          // kb: how many leading points lie in the base field
    u64 r0, r1, r2;     // degree-correction randomness
};
// vals[i] = (vals[i] - Ans(x)) / prod_j (x - p_j) * sum_{e <= k} (r x)^e,  x = offset * gen^i.
// The leading `kb` points of the quotient set are base-field elements (the queried domain values; only the
// out-of-domain points are proper extension elements), so their part of the zerofier is a base-field product; the
// degree-correction series is the geometric sum ((r x)^(k+1) - 1) / (r x - 1), and its denominator shares the one
// inversion with the zerofier.
__global__ void __launch_bounds__(256) k_stir_quotient(StirQuotientArgs a) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u64 x = bfe_mul(a.offset, bfe_pow(a.gen, i));
    xfe ans = xfe_zero();
    if (a.answer_values) ans = stir_ld(a.answer_values + 3 * i);
    else
        for (u32 j = a.k; j-- > 0;) ans = xfe_add(xfe_mul_bfe(ans, x), stir_ld(a.answer + 3 * j));
    u64 zb = TVM_ONE;
    for (u32 j = 0; j < a.kb; j++) zb = bfe_mul(zb, bfe_sub(x, a.points[3 * j]));
    xfe z = xfe_lift(zb);
    for (u32 j = a.kb; j < a.k; j++) z = xfe_mul(z, xfe_bfe_minus(x, stir_ld(a.points + 3 * j)));
    const xfe t = xfe_mul_bfe(xfe_make(a.r0, a.r1, a.r2), x);
    const xfe tm1 = xfe_sub_bfe(t, TVM_ONE);
    xfe q;
    const xfe num = xfe_sub(stir_ld(a.vals + 3 * i), ans);
    if (xfe_eq(tm1, xfe_zero())) {  // r x = 1: the series is k + 1 ones
        q = xfe_mul(xfe_mul_bfe(num, bfe_from_u64((u64)a.k + 1)), xfe_inv(z));
    } else {
        const xfe series_num = xfe_sub_bfe(xfe_pow(t, (u64)a.k + 1), TVM_ONE);
        q = xfe_mul(xfe_mul(num, series_num), xfe_inv(xfe_mul(z, tm1)));
    }
    a.vals[3 * i] = q.c0;
    a.vals[3 * i + 1] = q.c1;
    a.vals[3 * i + 2] = q.c2;
}

// Polynomial::interpolate through k <= 256 pairwise distinct XFE points (the "Ans" polynomial of a STIR round, stir.rs:954; k is
// the number of queries, ~200 at 160 bits) in ONE workgroup: c_i = y_i / prod_{j != i} (x_i - x_j) per work-item, then the sum
// sum_i c_i prod_{j != i} (X - x_j) built point by point as a pair (N, D): N <- N (X - x_i) + c_i D, D <- D (X - x_i), work-item
// j holding coefficient j of both in registers.  The interpolant is unique, so these are the coefficients twenty-first computes.
// All but one or two points of a STIR quotient set are base-field elements (the queried domain values); they are taken FIRST,
// whatever their place in the input: while only such points have been multiplied in, D has base-field coefficients and a step
// costs 7 base-field multiplications per work-item instead of 27 (and the denominators of a base-field point are base-field
// products but for the one or two proper extension points).  One barrier per step: the neighbour's coefficients travel through
// a double-buffered copy in shared memory.  status: 1 = two points coincide.
__global__ void __launch_bounds__(256) k_xfe_interpolate(const u64* __restrict__ points, const u64* __restrict__ values, int k,
                                                         u64* __restrict__ out, int* __restrict__ status) {
    __shared__ u64 sx[3 * 256], sc[3 * 256];
    __shared__ u64 sn[2][3 * 257], sd[2][3 * 257];  // slot j + 1 = coefficient j; slot 0 stays zero (the neighbour of j = 0)
    __shared__ unsigned char in_base_field[256];
    const int tid = threadIdx.x;
    xfe x = xfe_zero(), coeff = xfe_zero();
    bool base = false;
    if (tid < k) {
        x = stir_ld(points + 3 * tid);
        base = x.c1 == 0 && x.c2 == 0;
    }
    in_base_field[tid] = base;
    __syncthreads();
    // sx, sc: the points and their coefficients IN THE ORDER OF THE STEPS (base-field points first, each group in input order)
    int n_base = 0, step_of_mine = 0;
    {
        int before = 0, extension_before = 0;
        for (int j = 0; j < k; j++) {
            n_base += in_base_field[j];
            if (j < tid) before += in_base_field[j], extension_before += !in_base_field[j];
        }
        step_of_mine = base ? before : n_base + extension_before;
        if (tid < k) sx[3 * step_of_mine] = x.c0, sx[3 * step_of_mine + 1] = x.c1, sx[3 * step_of_mine + 2] = x.c2;
    }
    __syncthreads();
    // denominators prod_{j != i} (x_i - x_j).  A base-field point: a base-field product over the other base-field points (one
    // multiplication each) times the few extension factors.  An extension point: k - 1 extension-field factors -- left to its
    // own work-item that is a serial chain of 204 extension multiplications on one lane (0.2 ms, two thirds of the kernel when
    // first measured), so while there are only a handful of such points (STIR: the one or two out-of-domain points) the whole
    // workgroup multiplies the factors of each as a tree.
    xfe prod = xfe_one();
    const bool trees = k - n_base <= 4;
    if (trees) {
        u64* red = &sn[0][0];  // 3 * 256 words of the (not yet used) coefficient buffers
        for (int e = n_base; e < k; e++) {
            xfe f = xfe_one();
            if (tid < k && tid != e) f = xfe_sub(stir_ld(sx + 3 * e), stir_ld(sx + 3 * tid));
            red[3 * tid] = f.c0, red[3 * tid + 1] = f.c1, red[3 * tid + 2] = f.c2;
            __syncthreads();
            for (int d = 128; d >= 1; d >>= 1) {
                if (tid < d) {
                    f = xfe_mul(f, stir_ld(red + 3 * (tid + d)));
                    red[3 * tid] = f.c0, red[3 * tid + 1] = f.c1, red[3 * tid + 2] = f.c2;
                }
                __syncthreads();
            }
            if (tid < k && step_of_mine == e) prod = stir_ld(red);
            __syncthreads();
        }
    }
    if (tid < k) {
        if (base) {
            u64 pb = TVM_ONE;
            for (int j = 0; j < n_base; j++)
                if (j != step_of_mine) pb = bfe_mul(pb, bfe_sub(x.c0, sx[3 * j]));
            for (int j = n_base; j < k; j++) prod = xfe_mul(prod, xfe_sub(x, stir_ld(sx + 3 * j)));
            prod = xfe_mul_bfe(prod, pb);
        } else if (!trees) {
            for (int j = 0; j < k; j++)
                if (j != step_of_mine) prod = xfe_mul(prod, xfe_sub(x, stir_ld(sx + 3 * j)));
        }
        if (xfe_eq(prod, xfe_zero())) *status = 1;
        else coeff = xfe_mul(stir_ld(values + 3 * tid), xfe_inv(prod));
        sc[3 * step_of_mine] = coeff.c0, sc[3 * step_of_mine + 1] = coeff.c1, sc[3 * step_of_mine + 2] = coeff.c2;
    }
    if (tid < 3) sn[0][tid] = sn[1][tid] = sd[0][tid] = sd[1][tid] = 0;  // the neighbour of coefficient 0
    __syncthreads();
    xfe nj = xfe_zero(), dj = tid == 0 ? xfe_one() : xfe_zero();
    xfe xi = k ? stir_ld(sx) : xfe_zero(), ci = k ? stir_ld(sc) : xfe_zero();  // of the step to come: read a step ahead
    int p = 0;
    for (int step = 0; step < k; step++, p ^= 1) {
        u64* pn = sn[p] + 3 * (tid + 1);
        u64* pd = sd[p] + 3 * (tid + 1);
        pn[0] = nj.c0, pn[1] = nj.c1, pn[2] = nj.c2;
        pd[0] = dj.c0, pd[1] = dj.c1, pd[2] = dj.c2;
        const xfe x_now = xi, c_now = ci;
        if (step + 1 < k) xi = stir_ld(sx + 3 * (step + 1)), ci = stir_ld(sc + 3 * (step + 1));
        __syncthreads();
        const xfe nj1 = stir_ld(pn - 3);
        if (step < n_base) {  // D is a base-field polynomial so far
            const u64 dj1 = pd[-3];
            nj = xfe_add(xfe_sub(nj1, xfe_mul_bfe(nj, x_now.c0)), xfe_mul_bfe(c_now, dj.c0));
            dj.c0 = bfe_sub(dj1, bfe_mul(x_now.c0, dj.c0));
        } else {
            const xfe dj1 = stir_ld(pd - 3);
            nj = xfe_add(xfe_sub(nj1, xfe_mul(x_now, nj)), xfe_mul(c_now, dj));
            dj = xfe_sub(dj1, xfe_mul(x_now, dj));
        }
    }
    if (tid < k) out[3 * tid] = nj.c0, out[3 * tid + 1] = nj.c1, out[3 * tid + 2] = nj.c2;
}
int xfe_interpolate(tvm_ctx* c, const u64* d_points, const u64* d_values, int k, u64* d_out, int* d_status) {
    TVM_LAUNCH(k_xfe_interpolate, dim3(1), dim3(256), 0, c->stream, d_points, d_values, k, d_out, d_status);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int stir_hash_stacked(tvm_ctx* c, const u64* cw, u64 n, int stack_height, u64* digests) {
    const u64 d = n / (u64)stack_height;
    TVM_LAUNCH(k_hash_stacked, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, c->stream, cw, d, stack_height, digests);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int stir_fold_polynomial(tvm_ctx* c, const u64* poly, u64 n, int ff, const u64* h_r, u64* out) {
    const u64 n_out = (n + ff - 1) / ff;
    if (!n_out) return TVM_OK;
    TVM_LAUNCH(k_fold_polynomial, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream, poly, n, ff, h_r[0], h_r[1],
               h_r[2], n_out, out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int stir_quotient(tvm_ctx* c, u64* vals, u64 n, u64 offset, u64 gen, const u64* d_points, const u64* d_answer,
                  const u64* d_answer_values, u32 k, u32 kb, const u64* h_r) {
    StirQuotientArgs a;
    a.vals = vals;
    a.n = n;
    a.offset = offset;
    a.gen = gen;
    a.points = d_points;
    a.answer = d_answer;
    a.answer_values = d_answer_values;
    a.k = k;
    a.kb = kb;
    a.r0 = h_r[0];
    a.r1 = h_r[1];
    a.r2 = h_r[2];
    TVM_LAUNCH(k_stir_quotient, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
