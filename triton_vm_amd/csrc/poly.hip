// poly.hip -- the element-wise / reduction kernels between the big transforms of Prover::prove:
// out-of-domain rows, linear combinations, quotient-segment randomization, DEEP and FRI folding.
//
// Replaces, on the reference's hot path:
//   MasterTable::out_of_domain_row            /root/reference/triton-vm/src/table/master_table.rs:348-390
//   MasterTable::weighted_sum_of_columns      master_table.rs:512-542
//   split_polynomial_into_segments + randomize_quotient_segments   stark.rs:1224-1263, 1302-1356
//   Prover::deep_codeword / Stark::deep_update + the weighted sum   stark.rs:566-625, 1360-1379, 2096-2103
//   ProverRound::split_and_fold               low_degree_test/fri.rs:349-366
//
// XFieldElement vectors are arrays of 3-word elements (c0,c1,c2), exactly the reference's layout.
// All of these are streaming kernels: one pass over their operands, HBM bound.

#include "air_eval.h"   // AirAcc: sums of products with one reduction per coefficient at the end
#include "context.h"

namespace tvm {

TVM_D xfe ld_xfe(const u64* p) { return xfe_make(p[0], p[1], p[2]); }
TVM_D void st_xfe(u64* p, xfe v) { p[0] = v.c0; p[1] = v.c1; p[2] = v.c2; }
// trace cell (column c, row j) of a column-major table of field kind fk, times an XFE
TVM_D xfe cell_times(const u64* __restrict__ trace, int fk, u64 n_rows, u64 c, u64 j, xfe w) {
    const u64* p = trace + (c * n_rows + j) * fk;
    return fk == 1 ? xfe_mul_bfe(w, p[0]) : xfe_mul(ld_xfe(p), w);
}

#define TVM_RED_BLOCK 256
// sum of v over the workgroup, valid in thread 0; smem: 3*TVM_RED_BLOCK words
TVM_D xfe block_sum_xfe(xfe v, u64* smem, int tid, int nt) {
    smem[tid] = v.c0;
    smem[nt + tid] = v.c1;
    smem[2 * nt + tid] = v.c2;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            smem[tid] = bfe_add(smem[tid], smem[tid + s]);
            smem[nt + tid] = bfe_add(smem[nt + tid], smem[nt + tid + s]);
            smem[2 * nt + tid] = bfe_add(smem[2 * nt + tid], smem[2 * nt + tid + s]);
        }
        __syncthreads();
    }
    return xfe_make(smem[0], smem[nt], smem[2 * nt]);
}

// ------------------------------------------------------------------------------------------------
// Out-of-domain rows (master_table.rs:348-390), for n_points indeterminates at once.
// u[p][j] = d_j / (alpha_p - d_j), d_j = gen^j (trace domain, offset 1)
__global__ void k_ood_weights(u64 gen, u64 n, const u64* __restrict__ points, int n_points, u64* __restrict__ u) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u64 d = bfe_pow(gen, j);
    for (int p = 0; p < n_points; p++) {
        xfe a = ld_xfe(points + 3 * p);
        xfe inv = xfe_inv(xfe_sub_bfe(a, d));
        st_xfe(u + ((u64)p * n + j) * 3, xfe_mul_bfe(inv, d));
    }
}
// partial[p][c][chunk] = sum over the rows j of the chunk of cell(c, j) * u[p][j]; column index n_cols is the
// all-ones column (the barycentric denominator).  A workgroup owns G columns and one chunk of rows: each
// work-item reads the weights u[.][j] of its row once and the G cells next to them, so the trace is read once
// per pass and u once per column group (the first version re-read u for every column: 19 GB at 2^20 rows).
// The products are summed UNREDUCED (AirAcc, air_eval.h: 160-bit sums per coefficient, one Montgomery reduction at the end
// of the chunk): 13 instructions per product-accumulate instead of 20 for multiply, reduce, add.  An accumulator is 15 VGPRs
// for a base-field column and 25 for an extension-field one, which sets the columns per workgroup.
#define TVM_DOT_P 2        // points per pass
TVM_D u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = bfe_add(v, __shfl_xor(v, m, 64));
    return v;
}
template <int FK, int G>   // G columns per workgroup
__global__ void __launch_bounds__(TVM_RED_BLOCK) k_column_dot(const u64* __restrict__ trace, u64 n, u64 n_cols,
                                                               const u64* __restrict__ u, int p0, int n_points,
                                                               u64 rows_per_chunk, u64 n_chunks, u64* __restrict__ partial) {
    __shared__ u64 smem[TVM_RED_BLOCK / 64][G * TVM_DOT_P * 3];
    const int tid = threadIdx.x, nt = blockDim.x;
    const u64 c0 = (u64)blockIdx.x * G;
    const u64 chunk = blockIdx.y;
    const u64 r0 = chunk * rows_per_chunk, r1 = (r0 + rows_per_chunk < n) ? r0 + rows_per_chunk : n;
    const int np = (n_points - p0 < TVM_DOT_P) ? n_points - p0 : TVM_DOT_P;
    AirAcc acc[G][TVM_DOT_P];
    xfe ones[TVM_DOT_P];   // the all-ones column (index n_cols): plain sums of the weights
#pragma unroll
    for (int q = 0; q < TVM_DOT_P; q++) {
        ones[q] = xfe_zero();
#pragma unroll
        for (int g = 0; g < G; g++) acc[g][q] = air_acc_zero();
    }
    for (u64 j = r0 + tid; j < r1; j += nt) {
        xfe w[TVM_DOT_P];
#pragma unroll
        for (int q = 0; q < TVM_DOT_P; q++) w[q] = (q < np) ? ld_xfe(u + ((u64)(p0 + q) * n + j) * 3) : xfe_zero();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const u64 c = c0 + g;
            if (c < n_cols) {
                const u64* cp = trace + (c * n + j) * FK;
                if constexpr (FK == 1) {
                    const u64 x = cp[0];
#pragma unroll
                    for (int q = 0; q < TVM_DOT_P; q++) air_acc_b(acc[g][q], w[q], x);
                } else {
                    const xfe x = ld_xfe(cp);
#pragma unroll
                    for (int q = 0; q < TVM_DOT_P; q++) air_acc_x(acc[g][q], w[q], x);
                }
            } else if (c == n_cols) {
#pragma unroll
                for (int q = 0; q < TVM_DOT_P; q++) ones[q] = xfe_add(ones[q], w[q]);
            }
        }
    }
    // wavefront sums by lane exchange, then the (<= 4) wavefronts of the workgroup through LDS
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int q = 0; q < TVM_DOT_P; q++) {
            const xfe v = (c0 + g == n_cols) ? ones[q] : air_acc_value(acc[g][q]);
            const u64 s0 = wave_sum_u64(v.c0), s1 = wave_sum_u64(v.c1), s2 = wave_sum_u64(v.c2);
            if (lane == 0) {
                smem[wave][(g * TVM_DOT_P + q) * 3 + 0] = s0;
                smem[wave][(g * TVM_DOT_P + q) * 3 + 1] = s1;
                smem[wave][(g * TVM_DOT_P + q) * 3 + 2] = s2;
            }
        }
    __syncthreads();
    if (tid < G * TVM_DOT_P * 3) {
        u64 s = 0;
        for (int wv = 0; wv < nt / 64; wv++) s = bfe_add(s, smem[wv][tid]);
        const int comp = tid % 3, q = (tid / 3) % TVM_DOT_P, g = tid / (3 * TVM_DOT_P);
        const u64 c = c0 + g;
        if (c <= n_cols && q < np) partial[(((u64)(p0 + q) * (n_cols + 1) + c) * n_chunks + chunk) * 3 + comp] = s;
    }
}
// row[p][c] = num/den + (alpha^N - 1) * r_c(alpha)
// One WAVEFRONT per (column, point): the randomizer polynomial r_c(alpha) -- h ~ 200 coefficients -- as 64 lanes x a few
// consecutive coefficients each (Horner inside a lane, the lane's share scaled by alpha^(first index), lane sums by exchange).
// (One work-item per element ran the 200 Horner steps serially in twelve workgroups: 0.17 + 0.21 ms per proof at 2^20 rows.)
__global__ void __launch_bounds__(64) k_ood_finalize(const u64* __restrict__ num, const u64* __restrict__ rnd, int fk, u64 n,
                                                     u64 n_cols, u64 h, const u64* __restrict__ points, int n_points,
                                                     u64* __restrict__ rows) {
    const u64 e = blockIdx.x;
    const int lane = threadIdx.x;
    const u64 c = e % n_cols;
    const int p = (int)(e / n_cols);
    const xfe a = ld_xfe(points + 3 * p);
    const u64 per_lane = (h + 63) / 64;
    const u64 j0 = (u64)lane * per_lane;
    xfe r = xfe_zero();
    for (u64 t = per_lane; t-- > 0;) {
        const u64 j = j0 + t;
        r = xfe_mul(r, a);
        if (j < h) {
            const u64* q = rnd + (c * h + j) * fk;
            r = fk == 1 ? xfe_add_bfe(r, q[0]) : xfe_add(r, ld_xfe(q));
        }
    }
    r = xfe_mul(r, xfe_pow(a, j0));
    const u64 s0 = wave_sum_u64(r.c0), s1 = wave_sum_u64(r.c1), s2 = wave_sum_u64(r.c2);
    if (lane) return;
    const xfe den_inv = xfe_inv(ld_xfe(num + ((u64)p * (n_cols + 1) + n_cols) * 3));
    const xfe zf = xfe_sub_bfe(xfe_pow(a, n), TVM_ONE);
    const xfe v = xfe_add(xfe_mul(ld_xfe(num + ((u64)p * (n_cols + 1) + c) * 3), den_inv), xfe_mul(zf, xfe_make(s0, s1, s2)));
    st_xfe(rows + ((u64)p * n_cols + c) * 3, v);
}

// ------------------------------------------------------------------------------------------------
// weighted_sum_of_columns (master_table.rs:512-542): out[j] (+)= sum_c w_c * cell(c, j)
__global__ void k_weighted_row_sum(const u64* __restrict__ trace, int fk, u64 n, u64 n_cols, const u64* __restrict__ w,
                                   int accumulate, u64* __restrict__ out) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    AirAcc acc = air_acc_zero();   // unreduced sums of the products, see k_column_dot
    if (fk == 1) {
        for (u64 c = 0; c < n_cols; c++) air_acc_b(acc, ld_xfe(w + 3 * c), trace[c * n + j]);
    } else {
        for (u64 c = 0; c < n_cols; c++) air_acc_x(acc, ld_xfe(w + 3 * c), ld_xfe(trace + (c * n + j) * 3));
    }
    xfe r = air_acc_value(acc);
    if (accumulate) r = xfe_add(r, ld_xfe(out + 3 * j));
    st_xfe(out + 3 * j, r);
}
// The same for short traces: with a work-item per row, 1024 rows are four workgroups whose lanes walk the 379 columns one after
// another (88 us at 2^10 rows, twice per proof).  Here a workgroup is 64 rows x 16 column groups -- a wavefront reads 64 consecutive
// rows of its columns c = q mod 16 (coalesced) -- and the sixteen partial sums of a row meet in LDS.  Exact sums: the same words.
#define TVM_WRS_GROUPS 16
__global__ void __launch_bounds__(64 * TVM_WRS_GROUPS) k_weighted_row_sum_split(const u64* __restrict__ trace, int fk, u64 n, u64 n_cols,
                                                                                const u64* __restrict__ w, int accumulate, u64* __restrict__ out) {
    __shared__ u64 smem[TVM_WRS_GROUPS][3][64];
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const u64 j = (u64)blockIdx.x * 64 + r;
    xfe acc = xfe_zero();
    if (j < n)
        for (u64 c = q; c < n_cols; c += TVM_WRS_GROUPS) acc = xfe_add(acc, cell_times(trace, fk, n, c, j, ld_xfe(w + 3 * c)));
    smem[q][0][r] = acc.c0;
    smem[q][1][r] = acc.c1;
    smem[q][2][r] = acc.c2;
    __syncthreads();
    if (q || j >= n) return;
    for (int g = 1; g < TVM_WRS_GROUPS; g++) acc = xfe_add(acc, xfe_make(smem[g][0][r], smem[g][1][r], smem[g][2][r]));
    if (accumulate) acc = xfe_add(acc, ld_xfe(out + 3 * j));
    st_xfe(out + 3 * j, acc);
}
// R[j] = sum_c w_c r_c[j], j < h; poly[j] -= R[j]; poly[n + j] += R[j]   (mul_zerofier_with, offset 1)
// (one WAVEFRONT per coefficient j, the columns over its lanes: a work-item per coefficient walked the 379 columns serially in
// four workgroups, 0.21 + 0.09 ms per proof)
__global__ void __launch_bounds__(64) k_randomizer_contribution(const u64* __restrict__ rnd, int fk, u64 n, u64 n_cols, u64 h,
                                                                const u64* __restrict__ w, u64* __restrict__ poly) {
    const u64 j = blockIdx.x;
    const int lane = threadIdx.x;
    xfe acc = xfe_zero();
    for (u64 c = lane; c < n_cols; c += 64) acc = xfe_add(acc, cell_times(rnd, fk, h, c, j, ld_xfe(w + 3 * c)));
    acc = xfe_make(wave_sum_u64(acc.c0), wave_sum_u64(acc.c1), wave_sum_u64(acc.c2));
    if (lane) return;
    st_xfe(poly + 3 * j, xfe_sub(ld_xfe(poly + 3 * j), acc));
    st_xfe(poly + 3 * (n + j), xfe_add(ld_xfe(poly + 3 * (n + j)), acc));
}

// ------------------------------------------------------------------------------------------------
// Quotient segments (stark.rs:1224-1263) and their randomization (stark.rs:1302-1356):
// segment k of Q has coefficients Q[4j + k]; s_4 = randomizer; s_i = q_i - zeta^i * s_{i+1}(zeta^4 X).
// polys: [5][poly_len] XFE
__global__ void k_randomized_segments(const u64* __restrict__ q_coeffs, u64 q_len, const u64* __restrict__ rnd, u64 n_rand,
                                      u64 zeta, u64 poly_len, u64* __restrict__ polys) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= poly_len) return;
    const u64 z4j = bfe_pow(bfe_pow(zeta, 4), j);
    xfe s = j < n_rand ? ld_xfe(rnd + 3 * j) : xfe_zero();
    st_xfe(polys + (4 * poly_len + j) * 3, s);
    u64 zi = bfe_pow(zeta, 3);
    const u64 zeta_inv = bfe_inv(zeta);
    for (int i = 3; i >= 0; i--) {
        const u64 idx = 4 * j + (u64)i;
        xfe q = idx < q_len ? ld_xfe(q_coeffs + 3 * idx) : xfe_zero();
        s = xfe_sub(q, xfe_mul_bfe(s, bfe_mul(zi, z4j)));
        st_xfe(polys + ((u64)i * poly_len + j) * 3, s);
        zi = bfe_mul(zi, zeta_inv);
    }
}

// out[i] = sum_v w_v * table_cell(domain row i*stride, element v): linear combination of the columns of a
// device table (used for the P and R combinations of the randomized quotient segments, stark.rs:520-540)
__global__ void k_table_lincomb(const u64* __restrict__ table, TabView view, int fk, u64 n_cols,
                                const u64* __restrict__ w, u64* __restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= view.n_out) return;
    u64 row, i;
    view.locate(t, row, i);
    xfe acc = xfe_zero();
    for (u64 c = 0; c < n_cols; c++) {
        const xfe wc = ld_xfe(w + 3 * c);
        u64 e[3] = {0, 0, 0};
        for (int k = 0; k < fk; k++) {
            const u64 v = c * fk + k;
            e[k] = table[tvm_tab_idx(row, v, n_cols * (u64)fk)];
        }
        acc = xfe_add(acc, fk == 1 ? xfe_mul_bfe(wc, e[0]) : xfe_mul(xfe_make(e[0], e[1], e[2]), wc));
    }
    st_xfe(out + 3 * i, acc);
}

// The same over a table stored coset-major in the order of the last LDE pass (context.h: domain row X (j1 + n2 j2) + k is
// storage row k pitch + j1 n1 + j2): consecutive storage rows are domain rows X n2 apart, so the kernel above -- storage order
// in, domain order out -- scatters 24-byte results over the output (0.62 ms for 2^23 rows of 15 words where the table is read
// in 0.15).  Here a workgroup takes a tile of 16 consecutive rows j2 (one 128-byte line per word) x 128 (block j1, coset k)
// pairs, combines eight rows per work-item, and hands the results through shared memory: for each j2 the 128 pairs are 128
// CONSECUTIVE domain rows (3 KB of output) when the view has at most eight cosets, runs of eight otherwise.
#define TVM_LINCOMB_TILE_PAIRS 128
// WORDS = fk * n_cols known at compile time (15: the five extension-field columns of the quotient-segment table): the words of a
// row are loaded together and one row AHEAD of the arithmetic (with run-time loop bounds hipcc put each load right before its
// use: fifteen exposed memory latencies per row, 0.6 ms for 2^23 rows); 0: any shape, loops as they come.
template <int WORDS>
__global__ void __launch_bounds__(256) k_table_lincomb_tiles(const u64* __restrict__ table, TabView view, int fk, u64 n_cols,
                                                             const u64* __restrict__ w, u64* __restrict__ out) {
    constexpr int RS = 3 * TVM_LINCOMB_TILE_PAIRS + 1;
    __shared__ u64 so[16 * RS];
    const TabLayout& l = view.l;
    const int tid = threadIdx.x, p = tid & 15, q = tid >> 4;
    const int log_kx = view.log_xv < 3 ? view.log_xv : 3, kx = 1 << log_kx;
    const int j1_per_tile = TVM_LINCOMB_TILE_PAIRS >> log_kx;
    // tile coordinates: j2 group fastest, then j1 group, then coset group
    u64 b = blockIdx.x;
    const u64 P = (b & ((l.n1 >> 4) - 1)) << 4;
    b >>= l.log_n1 - 4;
    const u64 tiles_j1 = l.n2 / j1_per_tile;
    const u64 J = (b % tiles_j1) * j1_per_tile, k0 = (b / tiles_j1) << log_kx;
    const u64 W = n_cols * (u64)fk;
    auto row_of = [&](int it) {
        const int pair = q + 16 * it;
        const u64 j1 = J + (pair >> log_kx), kv = k0 + (pair & (kx - 1));
        return kv * view.stride * l.pitch + j1 * l.n1 + P + p;
    };
    if constexpr (WORDS > 0) {
        static_assert(WORDS % 3 == 0, "extension-field columns");
        xfe wc[WORDS / 3];
#pragma unroll
        for (int c = 0; c < WORDS / 3; c++) wc[c] = ld_xfe(w + 3 * c);
        u64 cur[WORDS], nxt[WORDS];
        {
            const u64 r0 = row_of(0);
#pragma unroll
            for (int v = 0; v < WORDS; v++) cur[v] = table[tvm_tab_idx(r0, v, WORDS)];
        }
#pragma unroll 1
        for (int it = 0; it < TVM_LINCOMB_TILE_PAIRS / 16; it++) {
            if (it + 1 < TVM_LINCOMB_TILE_PAIRS / 16) {
                const u64 r1 = row_of(it + 1);
#pragma unroll
                for (int v = 0; v < WORDS; v++) nxt[v] = table[tvm_tab_idx(r1, v, WORDS)];
            }
            xfe acc = xfe_zero();
#pragma unroll
            for (int c = 0; c < WORDS / 3; c++) acc = xfe_add(acc, xfe_mul(xfe_make(cur[3 * c], cur[3 * c + 1], cur[3 * c + 2]), wc[c]));
            u64* d = so + p * RS + 3 * (q + 16 * it);
            d[0] = acc.c0, d[1] = acc.c1, d[2] = acc.c2;
#pragma unroll
            for (int v = 0; v < WORDS; v++) cur[v] = nxt[v];
        }
    } else {
        for (int it = 0; it < TVM_LINCOMB_TILE_PAIRS / 16; it++) {
            const u64 row = row_of(it);
            xfe acc = xfe_zero();
            for (u64 c = 0; c < n_cols; c++) {
                const xfe wc = ld_xfe(w + 3 * c);
                u64 e[3] = {0, 0, 0};
                for (int k = 0; k < fk; k++) e[k] = table[tvm_tab_idx(row, c * fk + k, W)];
                acc = xfe_add(acc, fk == 1 ? xfe_mul_bfe(wc, e[0]) : xfe_mul(xfe_make(e[0], e[1], e[2]), wc));
            }
            u64* d = so + p * RS + 3 * (q + 16 * it);
            d[0] = acc.c0, d[1] = acc.c1, d[2] = acc.c2;
        }
    }
    __syncthreads();
    for (int e = tid; e < 16 * 3 * TVM_LINCOMB_TILE_PAIRS; e += 256) {
        const int pp = e / (3 * TVM_LINCOMB_TILE_PAIRS), word = e % (3 * TVM_LINCOMB_TILE_PAIRS), pair = word / 3, comp = word % 3;
        const u64 j = J + (pair >> log_kx) + ((P + pp) << l.log_n2);   // coset_index of the row
        const u64 i = (j << view.log_xv) + k0 + (pair & (kx - 1));
        out[3 * i + comp] = so[pp * RS + word];
    }
}

// polynomial (XFE coefficients) at XFE points: partial[p][block] then a final pass
__global__ void __launch_bounds__(TVM_RED_BLOCK) k_poly_eval_partial(const u64* __restrict__ co, u64 n,
                                                                     const u64* __restrict__ points,
                                                                     u64* __restrict__ partial) {
    __shared__ u64 smem[3 * TVM_RED_BLOCK];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int p = blockIdx.y;
    const u64 T = (u64)gridDim.x * nt;            // total work-items per point
    const u64 t = (u64)blockIdx.x * nt + tid;
    const xfe z = ld_xfe(points + 3 * p);
    const xfe zT = xfe_pow(z, T);
    // f(z) = sum_t z^t * sum_k c[t + k*T] (z^T)^k ; Horner in z^T over k descending
    xfe acc = xfe_zero();
    if (t < n) {
        u64 kmax = (n - 1 - t) / T;
        for (u64 k = kmax + 1; k-- > 0;) acc = xfe_add(xfe_mul(acc, zT), ld_xfe(co + 3 * (t + k * T)));
        acc = xfe_mul(acc, xfe_pow(z, t));
    }
    xfe s = block_sum_xfe(acc, smem, tid, nt);
    if (tid == 0) st_xfe(partial + ((u64)p * gridDim.x + blockIdx.x) * 3, s);
}
__global__ void __launch_bounds__(TVM_RED_BLOCK) k_sum_partials(const u64* __restrict__ partial, u64 n_partial,
                                                                u64* __restrict__ out) {
    __shared__ u64 smem[3 * TVM_RED_BLOCK];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int p = blockIdx.x;
    xfe acc = xfe_zero();
    for (u64 i = tid; i < n_partial; i += nt) acc = xfe_add(acc, ld_xfe(partial + ((u64)p * n_partial + i) * 3));
    xfe s = block_sum_xfe(acc, smem, tid, nt);
    if (tid == 0) st_xfe(out + 3 * p, s);
}

// ------------------------------------------------------------------------------------------------
// DEEP (stark.rs:566-625): out[i] = sum_k weight_k * (cw_k[i] - value_k) / (x_i - point_k),
// x_i = offset * gen^i.  One inversion per row: Montgomery's trick over the n_comp denominators.
#define TVM_DEEP_MAX 4
struct DeepArgs {
    const u64* cw[TVM_DEEP_MAX];
    u64 point[TVM_DEEP_MAX][3];
    u64 value[TVM_DEEP_MAX][3];
    u64 weight[TVM_DEEP_MAX][3];
    int n_comp;
    u64 offset, gen, n;
    u64* out;
};
// Four points per work-item, a quarter of the domain apart: x_{i + j n/4} = x_i w^j with w = gen^(n/4) a fourth root of unity, so
// one power of the generator serves four points, and ONE inversion the 4 n_comp denominators of all of them (Montgomery's
// trick; the denominators are recomputed on the way back instead of kept: sixteen prefix products are what the registers hold).
// (Measured: the steps written out for four components -- no run-time indexing of the prefix products, which hipcc keeps in
// 400 bytes of scratch here -- need 225 VGPRs and run 11 % slower, 1.07 vs 0.96 ms at 2^23 points; the kernel issues ~4700
// instructions per point and is within 10 % of that issue time as it stands.)
#define TVM_DEEP_POINTS 4
__global__ void __launch_bounds__(256) k_deep(DeepArgs a) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 quarter = a.n / TVM_DEEP_POINTS;
    if (i >= quarter) return;
    const u64 w = bfe_pow(a.gen, quarter);
    u64 x[TVM_DEEP_POINTS];
    x[0] = bfe_mul(a.offset, bfe_pow(a.gen, i));
#pragma unroll
    for (int j = 1; j < TVM_DEEP_POINTS; j++) x[j] = bfe_mul(x[j - 1], w);
    xfe pre[TVM_DEEP_POINTS][TVM_DEEP_MAX];
    xfe run = xfe_one();
#pragma unroll
    for (int j = 0; j < TVM_DEEP_POINTS; j++) {
#pragma unroll
        for (int k = 0; k < TVM_DEEP_MAX; k++) {
            if (k < a.n_comp) {
                pre[j][k] = run;
                run = xfe_mul(run, xfe_bfe_minus(x[j], xfe_make(a.point[k][0], a.point[k][1], a.point[k][2])));
            }
        }
    }
    xfe inv = xfe_inv(run);
#pragma unroll
    for (int j = TVM_DEEP_POINTS - 1; j >= 0; j--) {
        xfe acc = xfe_zero();
        const u64 row = i + (u64)j * quarter;
#pragma unroll
        for (int k = TVM_DEEP_MAX - 1; k >= 0; k--) {
            if (k < a.n_comp) {
                const xfe den = xfe_bfe_minus(x[j], xfe_make(a.point[k][0], a.point[k][1], a.point[k][2]));
                const xfe dinv = xfe_mul(inv, pre[j][k]);
                inv = xfe_mul(inv, den);
                const xfe num = xfe_sub(ld_xfe(a.cw[k] + 3 * row), xfe_make(a.value[k][0], a.value[k][1], a.value[k][2]));
                const xfe wk = xfe_make(a.weight[k][0], a.weight[k][1], a.weight[k][2]);
                acc = xfe_add(acc, xfe_mul(xfe_mul(num, dinv), wk));
            }
        }
        st_xfe(a.out + 3 * row, acc);
    }
}
// a domain shorter than four points: one point per work-item
__global__ void k_deep_short(DeepArgs a) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const u64 x = bfe_mul(a.offset, bfe_pow(a.gen, i));
    xfe acc = xfe_zero();
    for (int k = 0; k < a.n_comp; k++) {
        const xfe den = xfe_bfe_minus(x, xfe_make(a.point[k][0], a.point[k][1], a.point[k][2]));
        const xfe num = xfe_sub(ld_xfe(a.cw[k] + 3 * i), xfe_make(a.value[k][0], a.value[k][1], a.value[k][2]));
        acc = xfe_add(acc, xfe_mul(xfe_mul(num, xfe_inv(den)), xfe_make(a.weight[k][0], a.weight[k][1], a.weight[k][2])));
    }
    st_xfe(a.out + 3 * i, acc);
}

// ------------------------------------------------------------------------------------------------
// FRI split-and-fold (fri.rs:349-366): out[i] = ((1 + c/x_i) f[i] + (1 - c/x_i) f[i + n/2]) / 2
// (challenge: c0, c1, c2 as arguments, or -- d_challenge != nullptr -- three words in device memory written by an earlier
// kernel of the stream: tvm_fri_commit_phase)
__global__ void k_fri_fold(const u64* __restrict__ f, u64 n, u64 offset_inv, u64 gen_inv, u64 c0, u64 c1, u64 c2,
                           const u64* __restrict__ d_challenge, u64 two_inv, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 half = n >> 1;
    if (i >= half) return;
    if (d_challenge) {
        c0 = d_challenge[0];
        c1 = d_challenge[1];
        c2 = d_challenge[2];
    }
    const u64 xinv = bfe_mul(offset_inv, bfe_pow(gen_inv, i));
    const xfe s = xfe_mul_bfe(xfe_make(c0, c1, c2), xinv);
    const xfe l = xfe_mul(xfe_add_bfe(s, TVM_ONE), ld_xfe(f + 3 * i));
    const xfe r = xfe_mul(xfe_sub(xfe_one(), s), ld_xfe(f + 3 * (half + i)));
    st_xfe(out + 3 * i, xfe_mul_bfe(xfe_add(l, r), two_inv));
}

// ------------------------------------------------------------------------------------------------
// host side
#define TVM_GRID(n, bs) dim3((unsigned)(((n) + (bs)-1) / (bs)))

// h_points: n_points XFE; rows_out (device): [n_points][n_cols] XFE
int out_of_domain_rows(tvm_ctx* c, int fk, const u64* trace, u64 n, u64 n_cols, const u64* rnd, u64 h, u64 trace_gen,
                       const u64* d_points, int n_points, u64* d_rows) {
    u64* u = (u64*)scratch(c, 6, (size_t)n_points * n * 3 * sizeof(u64));
    u64* num = (u64*)scratch(c, 7, (size_t)n_points * (n_cols + 1) * 3 * sizeof(u64));
    if (!u || !num) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "ood scratch");
    TVM_LAUNCH(k_ood_weights, TVM_GRID(n, 256), dim3(256), 0, c->stream, trace_gen, n, d_points, n_points, u);
    // rows in chunks of 2^15 (128 rows per work-item), columns in groups of G (2 columns: the accumulators' registers and the
    // wavefronts per SIMD they leave), points two at a time (2^13 for narrow tables so that the grid still fills the chip)
    // (measured at 2^20 rows, both points, main + aux: 4 / 2 columns per workgroup 4.58 ms, 2 / 2 4.07 ms)
    const u64 G = fk == 1 ? 2 : 1;   // (4 main / 2 aux columns per workgroup: 4.58 against 4.07 ms, round 4)
    const u64 chunk_log = (n_cols + 1 + G - 1) / G >= 64 ? 15 : 13;
    const u64 rows_per_chunk = n < (1ull << chunk_log) ? n : (1ull << chunk_log);
    const u64 n_chunks = (n + rows_per_chunk - 1) / rows_per_chunk;
    const u64 n_sums = (u64)n_points * (n_cols + 1);
    u64* partial = (u64*)scratch(c, 8, (size_t)n_sums * n_chunks * 3 * sizeof(u64));
    if (!partial) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "ood scratch");
    const dim3 grid((unsigned)((n_cols + 1 + G - 1) / G), (unsigned)n_chunks);
    for (int p0 = 0; p0 < n_points; p0 += TVM_DOT_P) {
        if (fk == 1) TVM_LAUNCH((k_column_dot<1, 2>), grid, dim3(TVM_RED_BLOCK), 0, c->stream, trace, n, n_cols, u, p0, n_points, rows_per_chunk, n_chunks, partial);
        else TVM_LAUNCH((k_column_dot<3, 1>), grid, dim3(TVM_RED_BLOCK), 0, c->stream, trace, n, n_cols, u, p0, n_points, rows_per_chunk, n_chunks, partial);
    }
    TVM_LAUNCH(k_sum_partials, dim3((unsigned)n_sums), dim3(TVM_RED_BLOCK), 0, c->stream, partial, n_chunks, num);
    TVM_LAUNCH(k_ood_finalize, dim3((unsigned)(n_cols * n_points)), dim3(64), 0, c->stream, num, rnd, fk, n, n_cols, h,
               d_points, n_points, d_rows);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// d_values: n XFE sums over the trace rows (+ accumulate); no transform here
int weighted_row_sum(tvm_ctx* c, int fk, const u64* trace, u64 n, u64 n_cols, const u64* d_w, int accumulate, u64* d_values) {
    if (n <= 16384)   // fewer than 64 workgroups of a work-item per row: split the columns too
        TVM_LAUNCH(k_weighted_row_sum_split, TVM_GRID(n, 64), dim3(64 * TVM_WRS_GROUPS), 0, c->stream, trace, fk, n, n_cols, d_w, accumulate, d_values);
    else
        TVM_LAUNCH(k_weighted_row_sum, TVM_GRID(n, 256), dim3(256), 0, c->stream, trace, fk, n, n_cols, d_w, accumulate, d_values);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int randomizer_contribution(tvm_ctx* c, int fk, const u64* rnd, u64 n, u64 n_cols, u64 h, const u64* d_w, u64* d_poly) {
    if (!h) return TVM_OK;
    TVM_LAUNCH(k_randomizer_contribution, dim3((unsigned)h), dim3(64), 0, c->stream, rnd, fk, n, n_cols, h, d_w, d_poly);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int randomized_segments(tvm_ctx* c, const u64* d_q_coeffs, u64 q_len, const u64* d_rnd, u64 n_rand, u64 zeta, u64 poly_len,
                        u64* d_polys) {
    TVM_LAUNCH(k_randomized_segments, TVM_GRID(poly_len, 256), dim3(256), 0, c->stream, d_q_coeffs, q_len, d_rnd, n_rand,
               zeta, poly_len, d_polys);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int table_lincomb(tvm_ctx* c, const u64* table, const TabLayout& layout, int fk, u64 n_cols, u64 stride, const u64* d_w, u64* d_out) {
    const TabView view = tab_view(layout, stride);
    const u64 cosets = 1ull << view.log_xv;
    const u64 j1_per_tile = TVM_LINCOMB_TILE_PAIRS / (cosets < 8 ? cosets : 8);
    if (view.by_coset && layout.n1 >= 16 && layout.n2 >= j1_per_tile) {
        const u64 tiles = view.n_out / (16 * TVM_LINCOMB_TILE_PAIRS);
        if (fk == 3 && n_cols == 5)
            TVM_LAUNCH(k_table_lincomb_tiles<15>, dim3((unsigned)tiles), dim3(256), 0, c->stream, table, view, fk, n_cols, d_w, d_out);
        else
            TVM_LAUNCH(k_table_lincomb_tiles<0>, dim3((unsigned)tiles), dim3(256), 0, c->stream, table, view, fk, n_cols, d_w, d_out);
    } else
        TVM_LAUNCH(k_table_lincomb, TVM_GRID(view.n_out, 256), dim3(256), 0, c->stream, table, view, fk, n_cols, d_w, d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int poly_eval(tvm_ctx* c, const u64* d_coeffs, u64 n, const u64* d_points, int n_points, u64* d_out) {
    if (n == 0) {
        TVM_HIP_CHECK(c, hipMemsetAsync(d_out, 0, (size_t)n_points * 3 * sizeof(u64), c->stream));
        return TVM_OK;
    }
    u64 blocks = (n + TVM_RED_BLOCK * 16 - 1) / (TVM_RED_BLOCK * 16);
    if (blocks > 1024) blocks = 1024;
    u64* partial = (u64*)scratch(c, 8, (size_t)n_points * blocks * 3 * sizeof(u64));
    if (!partial) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "poly_eval scratch");
    TVM_LAUNCH(k_poly_eval_partial, dim3((unsigned)blocks, (unsigned)n_points), dim3(TVM_RED_BLOCK), 0, c->stream, d_coeffs,
               n, d_points, partial);
    TVM_LAUNCH(k_sum_partials, dim3((unsigned)n_points), dim3(TVM_RED_BLOCK), 0, c->stream, partial, blocks, d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int deep_sum(tvm_ctx* c, int n_comp, const u64* const* d_cw, const u64* h_points, const u64* h_values, const u64* h_weights,
             u64 offset, u64 gen, u64 n, u64* d_out) {
    if (n_comp < 1 || n_comp > TVM_DEEP_MAX) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "deep: 1..4 components");
    DeepArgs a;
    for (int k = 0; k < TVM_DEEP_MAX; k++) {
        a.cw[k] = k < n_comp ? d_cw[k] : nullptr;
        for (int j = 0; j < 3; j++) {
            a.point[k][j] = k < n_comp ? h_points[3 * k + j] : 0;
            a.value[k][j] = k < n_comp ? h_values[3 * k + j] : 0;
            a.weight[k][j] = k < n_comp ? h_weights[3 * k + j] : 0;
        }
    }
    a.n_comp = n_comp;
    a.offset = offset;
    a.gen = gen;
    a.n = n;
    a.out = d_out;
    if (n % TVM_DEEP_POINTS) TVM_LAUNCH(k_deep_short, TVM_GRID(n, 256), dim3(256), 0, c->stream, a);
    else TVM_LAUNCH(k_deep, TVM_GRID(n / TVM_DEEP_POINTS, 256), dim3(256), 0, c->stream, a);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int fri_fold(tvm_ctx* c, const u64* d_cw, u64 n, u64 offset, u64 gen, const u64* h_challenge, u64* d_out, const u64* d_challenge) {
    if (n < 2) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fold: codeword of length >= 2");
    if (!h_challenge && !d_challenge) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fold: no challenge");
    TVM_LAUNCH(k_fri_fold, TVM_GRID(n / 2, 256), dim3(256), 0, c->stream, d_cw, n, bfe_inv(offset), bfe_inv(gen),
               h_challenge ? h_challenge[0] : 0, h_challenge ? h_challenge[1] : 0, h_challenge ? h_challenge[2] : 0, d_challenge,
               bfe_inv(bfe_from_u64(2)), d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
