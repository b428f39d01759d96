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

// leaf i = Tip5::hash_varlen(cw[i], cw[i + d], ..., cw[i + (sh-1) d]) with d = n / sh, XFEs flattened c0,c1,c2
__global__ void __launch_bounds__(256) k_hash_stacked(const u64* __restrict__ cw, u64 d, int stack_height,
                                                      u64* __restrict__ digests) {
    __shared__ unsigned char lut[256];
    tip5_stage_lut(lut, threadIdx.x, blockDim.x);
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    const int W = 3 * stack_height;
    u64 st[TIP5_STATE];
#pragma unroll
    for (int q = 0; q < TIP5_STATE; q++) st[q] = 0;
    const int n_perms = W / TIP5_RATE + 1;
    int wi = 0;
    for (int perm = 0; perm < n_perms; perm++) {
#pragma unroll
        for (int q = 0; q < TIP5_RATE; q++) {
            u64 v;
            if (wi < W) v = cw[(i + (u64)(wi / 3) * d) * 3 + wi % 3];
            else v = (wi == W) ? TVM_ONE : 0;  // padding: 1 then 0s
            st[q] = v;
            wi++;
        }
        tip5_permute_inline(st, lut);
    }
#pragma unroll
    for (int q = 0; q < TIP5_DIGEST; q++) digests[i * 5 + q] = st[q];
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
// j updating coefficient j.  The interpolant is unique, so these are the coefficients twenty-first computes.  (On the host the same
// takes 0.9 ms at k = 204 with the device idle: tvm_host_xfe_interpolate.)  status: 1 = two points coincide.
__global__ void __launch_bounds__(256) k_xfe_interpolate(const u64* __restrict__ points, const u64* __restrict__ values, int k,
                                                         u64* __restrict__ out, int* __restrict__ status) {
    __shared__ u64 sx[3 * 256], sn[3 * 256], sd[3 * 256];
    const int tid = threadIdx.x;
    xfe x = xfe_zero(), coeff = xfe_zero();
    if (tid < k) {
        x = stir_ld(points + 3 * tid);
        sx[3 * tid] = x.c0, sx[3 * tid + 1] = x.c1, sx[3 * tid + 2] = x.c2;
    }
    sn[3 * tid] = sn[3 * tid + 1] = sn[3 * tid + 2] = 0;
    sd[3 * tid] = tid == 0 ? TVM_ONE : 0;
    sd[3 * tid + 1] = sd[3 * tid + 2] = 0;
    __syncthreads();
    if (tid < k) {
        xfe prod = xfe_one();
        for (int j = 0; j < k; j++)
            if (j != tid) prod = xfe_mul(prod, xfe_sub(x, stir_ld(sx + 3 * j)));
        if (xfe_eq(prod, xfe_zero())) *status = 1;
        else coeff = xfe_mul(stir_ld(values + 3 * tid), xfe_inv(prod));
    }
    // the coefficient c_i and the point x_i of step i reach every work-item through shared memory
    __shared__ u64 sc[3 * 256];
    sc[3 * tid] = coeff.c0, sc[3 * tid + 1] = coeff.c1, sc[3 * tid + 2] = coeff.c2;
    __syncthreads();
    for (int i = 0; i < k; i++) {
        const xfe xi = stir_ld(sx + 3 * i), ci = stir_ld(sc + 3 * i);
        const xfe nj = stir_ld(sn + 3 * tid), dj = stir_ld(sd + 3 * tid);
        const xfe nj1 = tid ? stir_ld(sn + 3 * (tid - 1)) : xfe_zero(), dj1 = tid ? stir_ld(sd + 3 * (tid - 1)) : xfe_zero();
        __syncthreads();
        const xfe nn = xfe_add(xfe_sub(nj1, xfe_mul(xi, nj)), xfe_mul(ci, dj)), dd = xfe_sub(dj1, xfe_mul(xi, dj));
        sn[3 * tid] = nn.c0, sn[3 * tid + 1] = nn.c1, sn[3 * tid + 2] = nn.c2;
        sd[3 * tid] = dd.c0, sd[3 * tid + 1] = dd.c1, sd[3 * tid + 2] = dd.c2;
        __syncthreads();
    }
    if (tid < k) out[3 * tid] = sn[3 * tid], out[3 * tid + 1] = sn[3 * tid + 1], out[3 * tid + 2] = sn[3 * tid + 2];
}
int xfe_interpolate(tvm_ctx* c, const u64* d_points, const u64* d_values, int k, u64* d_out, int* d_status) {
    TVM_LAUNCH(k_xfe_interpolate, dim3(1), dim3(256), 0, c->stream, d_points, d_values, k, d_out, d_status);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int stir_hash_stacked(tvm_ctx* c, const u64* cw, u64 n, int stack_height, u64* digests) {
    const u64 d = n / (u64)stack_height;
    TVM_LAUNCH(k_hash_stacked, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, c->stream, cw, d, stack_height, digests);
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
int stir_quotient(tvm_ctx* c, u64* vals, u64 n, u64 offset, u64 gen, const u64* d_points, const u64* d_answer, u32 k,
                  u32 kb, const u64* h_r) {
    StirQuotientArgs a;
    a.vals = vals;
    a.n = n;
    a.offset = offset;
    a.gen = gen;
    a.points = d_points;
    a.answer = d_answer;
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
