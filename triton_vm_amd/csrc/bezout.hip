// bezout.hip -- the RAM table's Bezout coefficient polynomials on the device.
//
// Replaces  bezout_coefficient_polynomials_coefficients  (/root/reference/triton-vm/src/table/ram.rs:152-207): for the
// square-free rp(X) = prod (X - r_i) over the n unique RAM pointers and its formal derivative fd, the polynomials a, b with
// a * rp + b * fd = 1, returned as n coefficients each.  Same structure as the reference (b from its values 1 / fd(r_i) at
// the roots, a = (1 - b * fd) / rp), with the polynomial arithmetic on the NTT kernels, n log^2 n throughout:
//   * a subproduct tree over the roots, padded with zero roots to a power of two so that every node has degree exactly
//     2^level (the root is then X^pad * rp): chunks of 64 leaves are multiplied out directly by one work-item each, the
//     levels above by batched transforms (ntt_columns) of twice the node's slot length; every level is kept;
//   * fd(r_i) for all i at once -- the reference's par_batch_evaluate -- by a scaled remainder tree: the first N terms of
//     the series fd / M_root in 1 / X (one power-series inversion of the reversed root polynomial, Newton steps on the
//     transforms), then top-down  T_child = middle product of T_parent with the sibling's polynomial  (one cyclic
//     convolution of the parent's length per node, exact: no divisions below the root), and at the chunks
//     fd mod M_chunk = polynomial part of T_chunk * M_chunk, evaluated at the chunk's 64 roots;
//   * a second bottom-up pass carries N = sum_i c_i prod_{j != i} (X - r_j) with c_i = 1 / fd(r_i)^2 (the Lagrange form of
//     b):  N = N_l M_r + N_r M_l, on the stored M's;
//   * a on a coset of >= 2n points that contains no root: values of rp, fd, b, pointwise (1 - b fd) / rp, interpolate.
// Results are field elements, so they are bit-identical to the reference's whatever the algorithm.
#include "context.h"
#include "kernels.h"

namespace tvm {

#define BZ_CHUNK_LOG 6
#define BZ_CHUNK (1 << BZ_CHUNK_LOG)

// c_i = 1 / fd(r_i)^2;  *zero is set when some fd(r_i) vanishes (a repeated root)
__global__ void k_bz_weights(const u64* __restrict__ fd_at_roots, u64 n, u64* __restrict__ c, unsigned* __restrict__ zero) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!fd_at_roots[i]) *zero = 1u;
    const u64 w = bfe_inv(fd_at_roots[i]);
    c[i] = bfe_mul(w, w);
}
// one work-item per chunk of 64 leaves: M = prod (X - r), N = sum c_i prod_{j != i} (X - r_j), into slots of 128 words.
// Padding leaves are the root 0 with weight 0.  c == nullptr: weights 0; M / N == nullptr: not written.
__global__ void k_bz_leaves(const u64* __restrict__ roots, const u64* __restrict__ c, u64 n, u64 n_chunks, u64* __restrict__ M,
                            u64* __restrict__ N) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_chunks) return;
    u64 m[BZ_CHUNK + 1], nn[BZ_CHUNK + 1];
    for (int k = 0; k <= BZ_CHUNK; k++) m[k] = 0, nn[k] = 0;
    m[0] = TVM_ONE;
    for (int deg = 0; deg < BZ_CHUNK; deg++) {
        const u64 leaf = q * BZ_CHUNK + deg;
        const u64 r = leaf < n ? roots[leaf] : 0, w = (leaf < n && c) ? c[leaf] : 0;
        // N <- N * (X - r) + w * M;  M <- M * (X - r)   (highest coefficient first: in place)
        for (int k = deg + 1; k >= 0; k--) {
            const u64 n_below = k ? nn[k - 1] : 0, m_below = k ? m[k - 1] : 0;
            const u64 mk = k <= deg ? m[k] : 0, nk = k <= deg ? nn[k] : 0;
            nn[k] = bfe_add(bfe_sub(n_below, bfe_mul(nk, r)), bfe_mul(w, mk));
            m[k] = bfe_sub(m_below, bfe_mul(mk, r));
        }
    }
    for (int k = 0; k < 2 * BZ_CHUNK; k++) {
        if (M) M[q * (2 * BZ_CHUNK) + k] = k <= BZ_CHUNK ? m[k] : 0;
        if (N) N[q * (2 * BZ_CHUNK) + k] = k <= BZ_CHUNK ? nn[k] : 0;
    }
}
// transforms of the children -> transforms of the parents:  PM = FM_l FM_r,  PN = FN_l FM_r + FN_r FM_l  (either may be null)
__global__ void k_bz_combine(const u64* __restrict__ fm, const u64* __restrict__ fn, u64 n_parents, u64 T, u64* __restrict__ pm,
                             u64* __restrict__ pn) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_parents * T) return;
    const u64 q = e / T, k = e % T;
    const u64 ml = fm[(2 * q) * T + k], mr = fm[(2 * q + 1) * T + k];
    if (pm) pm[e] = bfe_mul(ml, mr);
    if (pn) pn[e] = bfe_add(bfe_mul(fn[(2 * q) * T + k], mr), bfe_mul(fn[(2 * q + 1) * T + k], ml));
}
// dst[i] = src[top - i] when 0 <= top - i < n_src, else 0   (i < count): reversed coefficient order
__global__ void k_bz_reverse(const u64* __restrict__ src, u64 n_src, u64 top, u64 count, u64* __restrict__ dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (i <= top && top - i < n_src) ? src[top - i] : 0;
}
__global__ void k_bz_pointwise(const u64* __restrict__ a, const u64* __restrict__ b, u64 n, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bfe_mul(a[i], b[i]);
}
// e <- 2 - e  (Newton step of the series inversion)
__global__ void k_bz_two_minus(u64* __restrict__ e, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) e[i] = bfe_sub(i ? 0 : bfe_from_u64(2), e[i]);
}
// first `count` (<= 64) terms of 1 / p for a series p with p[0] = 1, by one work-item
__global__ void k_bz_series_inverse_head(const u64* __restrict__ p, int count, u64* __restrict__ g) {
    if (blockIdx.x || threadIdx.x) return;
    u64 loc[BZ_CHUNK];
    loc[0] = TVM_ONE;
    for (int k = 1; k < count; k++) {
        u64 acc = 0;
        for (int j = 1; j <= k; j++) acc = bfe_add(acc, bfe_mul(p[j], loc[k - j]));
        loc[k] = bfe_sub(0, acc);
    }
    for (int k = 0; k < count; k++) g[k] = loc[k];
}
// children's polynomials (slots of D words, degree d = D / 2) reversed and zero-padded to D words: out[c][u] = m_c[d - u]
__global__ void k_bz_reverse_children(const u64* __restrict__ level, u64 n_children, u64 D, u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_children * D) return;
    const u64 cidx = e / D, u = e % D, d = D / 2;
    out[e] = u <= d ? level[cidx * D + d - u] : 0;
}
// transforms: child c of parent q gets  FT[q] * Frev[sibling of c]   (in place on the pair)
__global__ void k_bz_middle(const u64* __restrict__ ft, u64 n_parents, u64 D, u64* __restrict__ frev) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_parents * D) return;
    const u64 q = e / D, k = e % D;
    const u64 t = ft[e], l = frev[(2 * q) * D + k], r = frev[(2 * q + 1) * D + k];
    frev[(2 * q) * D + k] = bfe_mul(t, r);
    frev[(2 * q + 1) * D + k] = bfe_mul(t, l);
}
// T_child[c][i] = conv[c][d + i], i < d
__global__ void k_bz_take_middle(const u64* __restrict__ conv, u64 n_children, u64 D, u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 d = D / 2;
    if (e >= n_children * d) return;
    out[e] = conv[(e / d) * D + d + e % d];
}
// one workgroup of 64 per chunk: R = polynomial part of T_chunk * M_chunk (= fd mod M_chunk), then R at the chunk's roots
__global__ void k_bz_chunk_values(const u64* __restrict__ T, const u64* __restrict__ Mchunks, const u64* __restrict__ roots, u64 n,
                                  u64* __restrict__ out) {
    __shared__ u64 t[BZ_CHUNK], m[BZ_CHUNK + 1], R[BZ_CHUNK];
    const u64 q = blockIdx.x;
    const int j = threadIdx.x;
    t[j] = T[q * BZ_CHUNK + j];
    m[j] = Mchunks[q * (2 * BZ_CHUNK) + j];
    if (j == 0) m[BZ_CHUNK] = Mchunks[q * (2 * BZ_CHUNK) + BZ_CHUNK];
    __syncthreads();
    u64 acc = 0;
    for (int k = 1; k <= BZ_CHUNK - j; k++) acc = bfe_add(acc, bfe_mul(t[k - 1], m[j + k]));
    R[j] = acc;
    __syncthreads();
    const u64 leaf = q * BZ_CHUNK + j;
    if (leaf >= n) return;
    const u64 r = roots[leaf];
    u64 v = 0;
    for (int k = BZ_CHUNK - 1; k >= 0; k--) v = bfe_add(bfe_mul(v, r), R[k]);
    out[leaf] = v;
}
__global__ void k_bz_derivative(const u64* __restrict__ rp, u64 n, u64* __restrict__ fd) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fd[i] = bfe_mul(bfe_from_u64(i + 1), rp[i + 1]);
}
// *hit is set when a root lies on the coset offset * <w_D>:  (r / offset)^D = 1
__global__ void k_bz_root_on_coset(const u64* __restrict__ roots, u64 n, u64 offset_inv, u64 D, unsigned* __restrict__ hit) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && bfe_pow(bfe_mul(roots[i], offset_inv), D) == TVM_ONE) *hit = 1u;
}
// a's values on the coset: (1 - b fd) / rp
__global__ void k_bz_quotient(const u64* __restrict__ rp, const u64* __restrict__ fd, const u64* __restrict__ b, u64 D, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) out[i] = bfe_mul(bfe_sub(TVM_ONE, bfe_mul(b[i], fd[i])), bfe_inv(rp[i]));
}
__global__ void k_bz_copy(const u64* __restrict__ src, u64 n_src, u64 n, u64* __restrict__ dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < n_src ? src[i] : 0;
}

static u64 bz_root_of_unity(u64 order) { return bfe_pow(bfe_from_u64(7), (TVM_P - 1) / order); }
// `count` transforms of length `len` (input: in_len words at stride in_stride, zero-padded; output at stride out_stride),
// at most 2^15 per launch (the transform kernels take the column index from the grid's y dimension)
static int bz_transforms(tvm_ctx* c, const u64* in, u64 in_len, u64 in_stride, u64* out, u64 out_stride, u64 count, u64 len, u64 w,
                         u64 out_mult) {
    for (u64 first = 0; first < count; first += 1u << 15) {
        const u64 batch = count - first < (1u << 15) ? count - first : (1u << 15);
        TVM_TRY(ntt_columns(c, in + first * in_stride, in_len, 1, in_stride, out + first * out_stride, 1, out_stride, 1, 0, (int)batch, len, w,
                            TVM_ONE, TVM_ONE, out_mult));
    }
    return TVM_OK;
}

// d_roots: n pairwise distinct field elements; d_a, d_b: n words each
int bezout_coefficients(tvm_ctx* c, const u64* d_roots, u64 n, u64* d_a, u64* d_b) {
    if (!n) return TVM_OK;
    if (n > (1ull << 24)) return set_error(c, TVM_ERR_UNSUPPORTED, "bezout: more than 2^24 roots");  // transforms of up to 2^26 points
    const unsigned bs = 256;
    auto grid = [&](u64 total) { return dim3((unsigned)((total + bs - 1) / bs)); };
    u64 NP = BZ_CHUNK;  // leaves after padding
    int K = BZ_CHUNK_LOG;
    while (NP < n) NP <<= 1, K++;
    const u64 pad = NP - n, n_chunks = NP / BZ_CHUNK;
    const int n_levels = K - BZ_CHUNK_LOG + 1;
    // buffers: the M tree (2 NP words per level: nodes of degree 2^l in slots of 2^(l+1)), the N tree's current and next
    // level, two transform arrays of 4 NP words, six series of NP words
    u64* tree = (u64*)pool_alloc(c, (size_t)n_levels * 2 * NP * sizeof(u64));
    u64* fdr = (u64*)pool_alloc(c, n * sizeof(u64));
    u64* w = (u64*)pool_alloc(c, n * sizeof(u64));
    u64* N[2] = {(u64*)pool_alloc(c, 2 * NP * sizeof(u64)), (u64*)pool_alloc(c, 2 * NP * sizeof(u64))};
    u64* FA = (u64*)pool_alloc(c, 4 * NP * sizeof(u64));
    u64* FB = (u64*)pool_alloc(c, 4 * NP * sizeof(u64));
    u64* ser = (u64*)pool_alloc(c, 6 * NP * sizeof(u64));
    unsigned* flag = (unsigned*)pool_alloc(c, sizeof(unsigned));
    int rc = TVM_OK;
    auto release = [&]() {
        for (void* p : {(void*)tree, (void*)fdr, (void*)w, (void*)N[0], (void*)N[1], (void*)FA, (void*)FB, (void*)ser, (void*)flag}) pool_release(c, p);
    };
    if (!tree || !fdr || !w || !N[0] || !N[1] || !FA || !FB || !ser || !flag) {
        release();
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "bezout scratch");
    }
    u64 *rev_m = ser, *g = ser + NP, *rev_f = ser + 2 * NP, *T[2] = {ser + 3 * NP, ser + 4 * NP}, *fd = ser + 5 * NP;
    auto level = [&](int l) { return tree + (size_t)(l - BZ_CHUNK_LOG) * 2 * NP; };
    unsigned h_flag = 0;
    auto read_flag = [&]() {
        if (hipMemcpyAsync(&h_flag, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = set_error(c, TVM_ERR_DEVICE, "bezout flag");
    };
    // single transforms of length `len` (the first l_src words of src, zero-padded) and back, for the series products
    auto forward = [&](const u64* src, u64 l_src, u64 len, u64* dst) {
        return ntt_columns(c, src, l_src, 1, 0, dst, 1, 0, 1, 0, 1, len, bz_root_of_unity(len), TVM_ONE, TVM_ONE, TVM_ONE);
    };
    auto inverse = [&](u64* buf, u64 len) {
        return ntt_columns(c, buf, len, 1, 0, buf, 1, 0, 1, 0, 1, len, bfe_inv(bz_root_of_unity(len)), TVM_ONE, TVM_ONE, bfe_inv(bfe_from_u64(len)));
    };

    // -- the M tree, bottom-up ------------------------------------------------------------------------------------------
    TVM_LAUNCH(k_bz_leaves, grid(n_chunks), dim3(bs), 0, c->stream, d_roots, (const u64*)nullptr, n, n_chunks, level(BZ_CHUNK_LOG), (u64*)nullptr);
    for (int l = BZ_CHUNK_LOG; rc == TVM_OK && l < K; l++) {
        const u64 slot = 2ull << l, nodes = NP >> l, len = 2 * slot;  // children; parents have slots of `len` words
        rc = bz_transforms(c, level(l), slot, slot, FA, len, nodes, len, bz_root_of_unity(len), TVM_ONE);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_combine, grid(nodes / 2 * len), dim3(bs), 0, c->stream, (const u64*)FA, (const u64*)nullptr, nodes / 2, len, level(l + 1), (u64*)nullptr);
        rc = bz_transforms(c, level(l + 1), len, len, level(l + 1), len, nodes / 2, len, bfe_inv(bz_root_of_unity(len)), bfe_inv(bfe_from_u64(len)));
    }
    const u64* m_root = level(K);   // X^pad * rp: NP + 1 coefficients, monic
    const u64* rp = m_root + pad;   // n + 1 coefficients
    if (rc == TVM_OK) TVM_LAUNCH(k_bz_derivative, grid(n), dim3(bs), 0, c->stream, rp, n, fd);

    // -- fd at the roots: scaled remainder tree ----------------------------------------------------------------------------
    // With Z = 1 / X:  fd / M_root = Z * rev_f(Z) / rev_m(Z),  rev_m[i] = M_root[NP - i],  rev_f[i] = fd[NP - 1 - i];  T_root[k] =
    // coefficient of X^-(k+1), k < NP, = (rev_f * (1 / rev_m))[k].
    if (rc == TVM_OK) {
        TVM_LAUNCH(k_bz_reverse, grid(NP), dim3(bs), 0, c->stream, m_root, NP + 1, NP, NP, rev_m);
        TVM_LAUNCH(k_bz_reverse, grid(NP), dim3(bs), 0, c->stream, (const u64*)fd, n, NP - 1, NP, rev_f);
        TVM_LAUNCH(k_bz_series_inverse_head, dim3(1), dim3(64), 0, c->stream, (const u64*)rev_m, BZ_CHUNK, g);
    }
    for (u64 m = BZ_CHUNK; rc == TVM_OK && m < NP; m <<= 1) {  // g = 1 / rev_m mod Z^m  ->  mod Z^2m:  g <- g (2 - rev_m g)
        const u64 len = 4 * m;
        rc = forward(rev_m, 2 * m, len, FA);
        if (rc == TVM_OK) rc = forward(g, m, len, FB);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_pointwise, grid(len), dim3(bs), 0, c->stream, (const u64*)FA, (const u64*)FB, len, FA);
        rc = inverse(FA, len);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_two_minus, grid(2 * m), dim3(bs), 0, c->stream, FA, 2 * m);
        rc = forward(FA, 2 * m, len, FA);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_pointwise, grid(len), dim3(bs), 0, c->stream, (const u64*)FA, (const u64*)FB, len, FA);
        rc = inverse(FA, len);
        if (rc == TVM_OK) TVM_LAUNCH(k_bz_copy, grid(2 * m), dim3(bs), 0, c->stream, (const u64*)FA, 2 * m, 2 * m, g);
    }
    int cur = 0;
    if (rc == TVM_OK) {
        const u64 len = 2 * NP;
        rc = forward(rev_f, NP, len, FA);
        if (rc == TVM_OK) rc = forward(g, NP, len, FB);
        if (rc == TVM_OK) {
            TVM_LAUNCH(k_bz_pointwise, grid(len), dim3(bs), 0, c->stream, (const u64*)FA, (const u64*)FB, len, FA);
            rc = inverse(FA, len);
        }
        if (rc == TVM_OK) TVM_LAUNCH(k_bz_copy, grid(NP), dim3(bs), 0, c->stream, (const u64*)FA, NP, NP, T[cur]);
    }
    // top-down: parents of degree D = 2^l with D terms of T each -> children of degree D / 2 with D / 2 terms each
    for (int l = K; rc == TVM_OK && l > BZ_CHUNK_LOG; l--) {
        const u64 D = 1ull << l, parents = NP >> l;
        const u64 wD = bz_root_of_unity(D);
        rc = bz_transforms(c, T[cur], D, D, FA, D, parents, D, wD, TVM_ONE);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_reverse_children, grid(2 * parents * D), dim3(bs), 0, c->stream, (const u64*)level(l - 1), 2 * parents, D, FB);
        rc = bz_transforms(c, FB, D, D, FB, D, 2 * parents, D, wD, TVM_ONE);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_middle, grid(parents * D), dim3(bs), 0, c->stream, (const u64*)FA, parents, D, FB);
        rc = bz_transforms(c, FB, D, D, FB, D, 2 * parents, D, bfe_inv(wD), bfe_inv(bfe_from_u64(D)));
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_take_middle, grid(NP), dim3(bs), 0, c->stream, (const u64*)FB, 2 * parents, D, T[1 - cur]);
        cur = 1 - cur;
    }
    if (rc == TVM_OK) {
        TVM_LAUNCH(k_bz_chunk_values, dim3((unsigned)n_chunks), dim3(BZ_CHUNK), 0, c->stream, (const u64*)T[cur], (const u64*)level(BZ_CHUNK_LOG), d_roots, n, fdr);
        (void)hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream);
        TVM_LAUNCH(k_bz_weights, grid(n), dim3(bs), 0, c->stream, (const u64*)fdr, n, w, flag);
        read_flag();
        if (rc == TVM_OK && h_flag) rc = set_error(c, TVM_ERR_INVALID_ARGUMENT, "bezout: the roots are not pairwise distinct");
    }

    // -- the N tree (b's Lagrange form), bottom-up on the stored M's -----------------------------------------------------------
    cur = 0;
    if (rc == TVM_OK) TVM_LAUNCH(k_bz_leaves, grid(n_chunks), dim3(bs), 0, c->stream, d_roots, (const u64*)w, n, n_chunks, (u64*)nullptr, N[0]);
    for (int l = BZ_CHUNK_LOG; rc == TVM_OK && l < K; l++) {
        const u64 slot = 2ull << l, nodes = NP >> l, len = 2 * slot;
        const u64 wl = bz_root_of_unity(len);
        rc = bz_transforms(c, level(l), slot, slot, FA, len, nodes, len, wl, TVM_ONE);
        if (rc == TVM_OK) rc = bz_transforms(c, N[cur], slot, slot, FB, len, nodes, len, wl, TVM_ONE);
        if (rc != TVM_OK) break;
        TVM_LAUNCH(k_bz_combine, grid(nodes / 2 * len), dim3(bs), 0, c->stream, (const u64*)FA, (const u64*)FB, nodes / 2, len, (u64*)nullptr, N[1 - cur]);
        rc = bz_transforms(c, N[1 - cur], len, len, N[1 - cur], len, nodes / 2, len, bfe_inv(wl), bfe_inv(bfe_from_u64(len)));
        cur = 1 - cur;
    }

    // -- b, and a = (1 - b fd) / rp on a root-free coset -----------------------------------------------------------------------
    if (rc == TVM_OK) {
        const u64* b = N[cur] + pad;  // X^pad * b at the root of the N tree: n coefficients
        TVM_LAUNCH(k_bz_copy, grid(n), dim3(bs), 0, c->stream, b, n, n, d_b);
        if (n == 1) {
            (void)hipMemsetAsync(d_a, 0, sizeof(u64), c->stream);  // rp = X - r, fd = 1, b = 1, a = 0
        } else {
            u64 D = 2;
            while (D < 2 * n) D <<= 1;  // b * fd has 2n - 1 coefficients;  D <= 2 NP
            // a coset offset * <w_D> without a root of rp: offset = 7^t, t = 1, 2, ...
            u64 offset = bfe_from_u64(7);
            for (int attempt = 0; rc == TVM_OK; attempt++, offset = bfe_mul(offset, bfe_from_u64(7))) {
                (void)hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream);
                TVM_LAUNCH(k_bz_root_on_coset, grid(n), dim3(bs), 0, c->stream, d_roots, n, bfe_inv(offset), D, flag);
                read_flag();
                if (!h_flag) break;
                if (attempt == 16) rc = set_error(c, TVM_ERR_UNSUPPORTED, "bezout: no root-free coset found");
            }
            u64* rp_v = FA;
            u64* b_v = FB;
            u64* fd_v = N[1 - cur];
            if (rc == TVM_OK) {
                const u64 wD = bz_root_of_unity(D);
                rc = ntt_columns(c, rp, n + 1, 1, 0, rp_v, 1, 0, 1, 0, 1, D, wD, offset, TVM_ONE, TVM_ONE);
                if (rc == TVM_OK) rc = ntt_columns(c, b, n, 1, 0, b_v, 1, 0, 1, 0, 1, D, wD, offset, TVM_ONE, TVM_ONE);
                if (rc == TVM_OK) rc = ntt_columns(c, fd, n, 1, 0, fd_v, 1, 0, 1, 0, 1, D, wD, offset, TVM_ONE, TVM_ONE);
                if (rc == TVM_OK) {
                    TVM_LAUNCH(k_bz_quotient, grid(D), dim3(bs), 0, c->stream, (const u64*)rp_v, (const u64*)fd_v, (const u64*)b_v, D, rp_v);
                    rc = ntt_columns(c, rp_v, D, 1, 0, rp_v, 1, 0, 1, 0, 1, D, bfe_inv(wD), TVM_ONE, bfe_inv(offset), bfe_inv(bfe_from_u64(D)));
                }
                if (rc == TVM_OK) TVM_LAUNCH(k_bz_copy, grid(n), dim3(bs), 0, c->stream, (const u64*)rp_v, n - 1, n, d_a);
            }
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == TVM_OK) rc = set_error(c, TVM_ERR_DEVICE, "bezout");
    release();
    return rc;
}

}  // namespace tvm
