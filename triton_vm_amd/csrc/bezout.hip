// bezout.hip -- the RAM table's Bezout coefficient polynomials on the device.
//
// Replaces  bezout_coefficient_polynomials_coefficients  (/root/reference/triton-vm/src/table/ram.rs:152-207): for the
// square-free rp(X) = prod (X - r_i) over the n unique RAM pointers and its formal derivative fd, the polynomials a, b with
// a * rp + b * fd = 1, returned as n coefficients each.  Same structure as the reference (b from its values 1 / fd(r_i) at
// the roots, a = (1 - b * fd) / rp), with the polynomial arithmetic on the NTT kernels:
//   * fd(r_i) = prod_{j != i} (r_i - r_j), all pairs, LDS-tiled (n^2 multiplications: 30 ms at 2^18 pointers; a remainder
//     tree would make it n log^2 n -- the reference's par_batch_evaluate -- and is the next step for 2^20 pointers);
//   * a subproduct tree bottom-up carries, per node, M = prod (X - r_i) and N = sum_i c_i prod_{j != i} (X - r_j) with
//     c_i = 1 / fd(r_i)^2 (the Lagrange form of b):  M = M_l M_r,  N = N_l M_r + N_r M_l.  Chunks of 64 leaves are
//     multiplied out directly by one work-item each, the levels above by batched transforms (ntt_columns) of twice the
//     node's slot length and a pointwise combine;
//   * a on a coset of >= 2n points that contains no root: values of rp, fd, b, pointwise (1 - b fd) / rp, interpolate.
// Results are field elements, so they are bit-identical to the reference's whatever the algorithm.
#include "context.h"
#include "kernels.h"

namespace tvm {

#define BZ_CHUNK_LOG 6
#define BZ_CHUNK (1 << BZ_CHUNK_LOG)

// out[i] = prod_{j != i} (roots[i] - roots[j])
__global__ void k_bz_fd_at_roots(const u64* __restrict__ roots, u64 n, u64* __restrict__ out) {
    __shared__ u64 tile[256];
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 ri = i < n ? roots[i] : 0;
    u64 acc = TVM_ONE;
    for (u64 j0 = 0; j0 < n; j0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = (j0 + threadIdx.x < n) ? roots[j0 + threadIdx.x] : 0;
        __syncthreads();
        const u64 m = (n - j0 < 256) ? n - j0 : 256;
        for (u64 j = 0; j < m; j++) {
            const u64 d = bfe_sub(ri, tile[j]);
            acc = bfe_mul(acc, (j0 + j == i) ? TVM_ONE : d);
        }
    }
    if (i < n) out[i] = acc;
}
// c_i = 1 / fd(r_i)^2;  *zero is set when some fd(r_i) vanishes (a repeated root)
__global__ void k_bz_weights(const u64* __restrict__ fd_at_roots, u64 n, u64* __restrict__ c, unsigned* __restrict__ zero) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!fd_at_roots[i]) *zero = 1u;
    const u64 w = bfe_inv(fd_at_roots[i]);
    c[i] = bfe_mul(w, w);
}
// one work-item per chunk of 64 leaves: M = prod (X - r), N = sum c_i prod_{j != i} (X - r_j), into slots of 128 words
__global__ void k_bz_leaves(const u64* __restrict__ roots, const u64* __restrict__ c, u64 n, u64 n_chunks, u64* __restrict__ M,
                            u64* __restrict__ N) {
    const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_chunks) return;
    u64 m[BZ_CHUNK + 1], nn[BZ_CHUNK + 1];
    for (int k = 0; k <= BZ_CHUNK; k++) m[k] = 0, nn[k] = 0;
    m[0] = TVM_ONE;
    int deg = 0;
    for (int e = 0; e < BZ_CHUNK; e++) {
        const u64 leaf = q * BZ_CHUNK + e;
        if (leaf >= n) break;  // padding leaves are the polynomial 1
        const u64 r = roots[leaf], w = c[leaf];
        // N <- N * (X - r) + w * M;  M <- M * (X - r)   (highest coefficient first: in place)
        for (int k = deg + 1; k >= 0; k--) {
            const u64 n_below = k ? nn[k - 1] : 0, m_below = k ? m[k - 1] : 0;
            const u64 mk = k <= deg ? m[k] : 0, nk = k <= deg ? nn[k] : 0;
            nn[k] = bfe_add(bfe_sub(n_below, bfe_mul(nk, r)), bfe_mul(w, mk));
            m[k] = bfe_sub(m_below, bfe_mul(mk, r));
        }
        deg++;
    }
    u64* mo = M + q * (2 * BZ_CHUNK);
    u64* no = N + q * (2 * BZ_CHUNK);
    for (int k = 0; k < 2 * BZ_CHUNK; k++) {
        mo[k] = k <= BZ_CHUNK ? m[k] : 0;
        no[k] = k <= BZ_CHUNK ? nn[k] : 0;
    }
}
// transforms of the children -> transforms of the parents:  PM = FM_l FM_r,  PN = FN_l FM_r + FN_r FM_l
__global__ void k_bz_combine(const u64* __restrict__ fm, const u64* __restrict__ fn, u64 n_parents, u64 T, u64* __restrict__ pm,
                             u64* __restrict__ pn) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_parents * T) return;
    const u64 q = e / T, k = e % T;
    const u64 ml = fm[(2 * q) * T + k], mr = fm[(2 * q + 1) * T + k], nl = fn[(2 * q) * T + k], nr = fn[(2 * q + 1) * T + k];
    pm[e] = bfe_mul(ml, mr);
    pn[e] = bfe_add(bfe_mul(nl, mr), bfe_mul(nr, ml));
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

// d_roots: n pairwise distinct field elements; d_a, d_b: n words each
int bezout_coefficients(tvm_ctx* c, const u64* d_roots, u64 n, u64* d_a, u64* d_b) {
    if (!n) return TVM_OK;
    if (n > (1ull << 21)) return set_error(c, TVM_ERR_UNSUPPORTED, "bezout: more than 2^21 roots");  // 2^15 chunk columns per transform launch
    const unsigned bs = 256;
    auto grid = [&](u64 total) { return dim3((unsigned)((total + bs - 1) / bs)); };
    u64 n_pad = BZ_CHUNK;
    while (n_pad < n) n_pad <<= 1;
    const u64 n_chunks = n_pad / BZ_CHUNK;
    // buffers: two trees of n_pad * 2 words per level (ping-pong), two transform arrays of n_pad * 4 words
    u64* fdr = (u64*)pool_alloc(c, n * sizeof(u64));
    u64* w = (u64*)pool_alloc(c, n * sizeof(u64));
    u64* M[2] = {(u64*)pool_alloc(c, 2 * n_pad * sizeof(u64)), (u64*)pool_alloc(c, 2 * n_pad * sizeof(u64))};
    u64* N[2] = {(u64*)pool_alloc(c, 2 * n_pad * sizeof(u64)), (u64*)pool_alloc(c, 2 * n_pad * sizeof(u64))};
    u64* FM = (u64*)pool_alloc(c, 4 * n_pad * sizeof(u64));
    u64* FN = (u64*)pool_alloc(c, 4 * n_pad * sizeof(u64));
    unsigned* flag = (unsigned*)pool_alloc(c, sizeof(unsigned));
    int rc = TVM_OK;
    auto release = [&]() {
        for (void* p : {(void*)fdr, (void*)w, (void*)M[0], (void*)M[1], (void*)N[0], (void*)N[1], (void*)FM, (void*)FN, (void*)flag}) pool_release(c, p);
    };
    if (!fdr || !w || !M[0] || !M[1] || !N[0] || !N[1] || !FM || !FN || !flag) {
        release();
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "bezout scratch");
    }
    unsigned h_flag = 0;
    auto read_flag = [&]() {
        if (hipMemcpyAsync(&h_flag, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = set_error(c, TVM_ERR_DEVICE, "bezout flag");
    };
    (void)hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream);
    TVM_LAUNCH(k_bz_fd_at_roots, grid(n), dim3(bs), 0, c->stream, d_roots, n, fdr);
    TVM_LAUNCH(k_bz_weights, grid(n), dim3(bs), 0, c->stream, (const u64*)fdr, n, w, flag);
    read_flag();
    if (rc == TVM_OK && h_flag) rc = set_error(c, TVM_ERR_INVALID_ARGUMENT, "bezout: the roots are not pairwise distinct");
    int cur = 0;
    if (rc == TVM_OK) TVM_LAUNCH(k_bz_leaves, grid(n_chunks), dim3(bs), 0, c->stream, d_roots, (const u64*)w, n, n_chunks, M[0], N[0]);
    // levels: children with slots of `slot` words (degree <= slot / 2) -> parents with slots of 2 * slot
    for (u64 slot = 2 * BZ_CHUNK, nodes = n_chunks; rc == TVM_OK && nodes > 1; slot <<= 1, nodes >>= 1) {
        const u64 T = 2 * slot;
        const u64 wT = bz_root_of_unity(T);
        rc = ntt_columns(c, M[cur], slot, 1, slot, FM, 1, T, 1, 0, (int)nodes, T, wT, TVM_ONE, TVM_ONE, TVM_ONE);
        if (rc == TVM_OK) rc = ntt_columns(c, N[cur], slot, 1, slot, FN, 1, T, 1, 0, (int)nodes, T, wT, TVM_ONE, TVM_ONE, TVM_ONE);
        if (rc != TVM_OK) break;
        // the products overwrite the first halves of the transform arrays' partners: PM into M[1 - cur] is too small (it holds
        // nodes / 2 * T = nodes * slot words: exactly the tree buffer's size), so combine into the tree buffers directly
        TVM_LAUNCH(k_bz_combine, grid(nodes / 2 * T), dim3(bs), 0, c->stream, (const u64*)FM, (const u64*)FN, nodes / 2, T, M[1 - cur], N[1 - cur]);
        const u64 wTi = bfe_inv(wT), Tinv = bfe_inv(bfe_from_u64(T));
        rc = ntt_columns(c, M[1 - cur], T, 1, T, M[1 - cur], 1, T, 1, 0, (int)(nodes / 2), T, wTi, TVM_ONE, TVM_ONE, Tinv);
        if (rc == TVM_OK) rc = ntt_columns(c, N[1 - cur], T, 1, T, N[1 - cur], 1, T, 1, 0, (int)(nodes / 2), T, wTi, TVM_ONE, TVM_ONE, Tinv);
        cur = 1 - cur;
    }
    if (rc == TVM_OK) {
        const u64* rp = M[cur];  // n + 1 coefficients, monic
        const u64* b = N[cur];   // n coefficients
        TVM_LAUNCH(k_bz_copy, grid(n), dim3(bs), 0, c->stream, b, n, n, d_b);
        if (n == 1) {
            (void)hipMemsetAsync(d_a, 0, sizeof(u64), c->stream);  // rp = X - r, fd = 1, b = 1, a = 0
        } else {
            u64 D = 2;
            while (D < 2 * n) D <<= 1;  // b * fd has 2n - 1 coefficients
            // a coset offset * <w_D> without a root of rp: offset = 7^t, t = 1, 2, ...
            u64 offset = bfe_from_u64(7);
            for (int attempt = 0; rc == TVM_OK; attempt++, offset = bfe_mul(offset, bfe_from_u64(7))) {
                (void)hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream);
                TVM_LAUNCH(k_bz_root_on_coset, grid(n), dim3(bs), 0, c->stream, d_roots, n, bfe_inv(offset), D, flag);
                read_flag();
                if (!h_flag) break;
                if (attempt == 16) rc = set_error(c, TVM_ERR_UNSUPPORTED, "bezout: no root-free coset found");
            }
            // FM / FN hold 4 * n_pad >= D words each; M[1 - cur] (2 * n_pad >= D words) takes fd's values
            u64* fd = N[1 - cur];
            u64* rp_v = FM;
            u64* b_v = FN;
            u64* fd_v = M[1 - cur];
            if (rc == TVM_OK) {
                const u64 wD = bz_root_of_unity(D);
                TVM_LAUNCH(k_bz_derivative, grid(n), dim3(bs), 0, c->stream, rp, n, fd);
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
