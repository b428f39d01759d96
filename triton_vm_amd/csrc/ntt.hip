// ntt.hip -- radix-2 NTT / iNTT over F_p (p = 2^64 - 2^32 + 1) and the master-table LDE for gfx950.
//
// Replaces, on the reference's hot path:
//   ArithmeticDomain::{evaluate, interpolate, low_degree_extension}
//       /root/reference/triton-vm/src/arithmetic_domain.rs:141-212
//   MasterTable::{randomized_column_interpolant, maybe_low_degree_extend_all_columns}
//       /root/reference/triton-vm/src/table/master_table.rs:258-322, 392-403
//
// Design: a length-N transform (N up to 2^24) is split N = N1 x N2 and done in
// two HBM passes whose tiles live in LDS: a "strided" pass (all i1 for a batch of 16 adjacent i2,
// 128-byte coalesced runs) and a "row" pass (whole contiguous rows).  No bit-reversal pass exists
// anywhere: inverse sub-transforms are decimation-in-frequency (natural in, bit-reversed out),
// forward ones decimation-in-time (bit-reversed in, natural out), and the bit-reversed index is
// absorbed into the tile <-> global index map.
//
// The LDE of a table column never forms the 8N-point transform.  With the evaluation domain
// g*<w_L>, L = X*N, row X*j + k of the output is the value on coset gamma_k*<w_N>,
// gamma_k = g*w_L^k, and on that coset X^N = gamma_k^N is constant, so the randomized interpolant
// t(X) + (X^N - 1) r(X) folds to N coefficients: c_k[m] = (t[m] + (gamma_k^N - 1) r[m]) gamma_k^m.
// Three kernels per column chunk:
//   pass1  iNTT columns step (strided, DIF)                      reads trace,   writes Y   (1x)
//   pass2  iNTT rows step (DIF) -> coefficients held in VGPRs -> for each of the X cosets:
//          scale, forward columns step (DIT)                     reads Y,       writes Z   (Xx)
//   pass3  forward rows step (DIT) for 16 table columns at once  reads Z,       writes the table
// Algorithmic HBM bytes per base-field trace cell: 8 (read) + 8*X (write) = 72 at X = 8; the
// scheme moves 8*(1+1+1+X+X+X) = 216.
// The LDS-resident transforms run at their VGPR budgets (128 at 4 wavefronts per SIMD): the carry-out form of bfe_mul
// (field.h) keeps one more register pair live per product and measured 3-7 % slower in the three passes.
#define TVM_MUL_CARRY_FORM 0
#include <cstdlib>
#include <cstring>

#include "context.h"
#include "ntt_shift.h"

namespace tvm {

TVM_D u32 brev_bits(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

struct Pow2 {  // t^e = lo[e & mask] * hi[e >> shift]
    const u64* lo;
    const u64* hi;
    int shift;
};
TVM_D u64 pow2_get(const Pow2& t, u64 e) { return bfe_mul(t.lo[e & ((1ull << t.shift) - 1)], t.hi[e >> t.shift]); }

__global__ void k_pow_table(u64 base, u64 count, u64 scale, u64* out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = bfe_mul(scale, bfe_pow(base, i));
}

// In-LDS transform of a tile: element (a, b) at s[a*SA + b*SB], a < 2^log_n the transform axis,
// b < 2^batch_log independent transforms.  tw[e] = w^e, e < n (all n powers: the twiddle steps of lds_ntt_group use the
// upper half too), w the n-th root to use.
// DIT = false: decimation in frequency, natural order in, bit-reversed order out.
// DIT = true : decimation in time, bit-reversed order in, natural order out.
//
// The log_n butterfly layers are taken four at a time: a work-item loads the 16 elements whose indices
// differ in the four index bits of the group, runs the 32 butterflies of those layers in VGPRs and writes
// the 16 results back in place, so a 1024-point transform makes 3 trips through LDS (4 + 4 + 2 layers)
// instead of 10 and synchronises 3 times instead of 10.  Lanes of a wavefront are consecutive b first
// (SB is either 1 or an odd row pitch), then consecutive groups: at most 2-way bank conflicts.
// Ends with a barrier; the caller must have synchronised the tile before the call.
#define TVM_ROW_PAD 1
// CL / CLOGN >= 0: the group's first layer and the transform length are compile-time constants, for the
// production tile (16 transforms side by side, element (a, b) at s[a + b * (n + TVM_ROW_PAD)]): every LDS address
// of the group is then the work-item's base plus an immediate offset, and the twiddle indices are shifts by
// constants -- the generic form spends ~190 of its ~1050 instructions per group on that arithmetic.
// ROOT: 0 = any root of unity (every twiddle from tw); 1 / 2 = the n-th root is the domains' own (ntt_shift.h) / its
// inverse, all of whose 16th roots are powers of two: the group of the lowest four layers (l == 0) is then a transform with
// shift twiddles, no table access and 9-12-instruction multiplications instead of 16-instruction ones.
template <bool DIT, int K, bool L0, int CL = -1, int CLOGN = -1, int CB = 4, int ROOT = 0>
TVM_D void lds_ntt_group(u64* s, int log_n_rt, int batch_log_rt, int SA_rt, int SB_rt, const u64* __restrict__ tw, int l_rt, int tid, int nt) {
    constexpr int R = 1 << K;
    const int log_n = CLOGN >= 0 ? CLOGN : log_n_rt, batch_log = CLOGN >= 0 ? CB : batch_log_rt, l = CL >= 0 ? CL : l_rt;
    const int SA = CLOGN >= 0 ? 1 : SA_rt, SB = CLOGN >= 0 ? (1 << (CLOGN >= 0 ? CLOGN : 0)) + TVM_ROW_PAD : SB_rt;
    const int n_groups = (1 << (log_n - K)) << batch_log;
    const int bmask = (1 << batch_log) - 1, lmask = (1 << l) - 1;
    for (int gi = tid; gi < n_groups; gi += nt) {
        const int b = gi & bmask, g = gi >> batch_log;
        const int j0 = g & lmask;
        u64* p = s + ((((g >> l) << (l + K)) | j0) * SA + b * SB);
        const int stride = SA << l;
        u64 x[R];
#pragma unroll
        for (int e = 0; e < R; e++) x[e] = p[e * stride];
        if constexpr (L0 && ROOT != 0) {
            ntt_pow2_points<K, DIT, ROOT == 2>(x);
        } else if constexpr (ROOT != 0) {
            // Layers l .. l+K-1 of the transform = a twiddle step and a 2^K-point transform with shift twiddles: the
            // twiddle of layer l+t, w_{2^(l+t+1)}^((m << l) | j0), is w_{2^(t+1)}^m -- a power of two -- times a factor
            // that depends on j0 and t alone, and those factors collect on element e as  w_{2^(l+K)}^(j0 * brev_K(e))
            // (decimation in time: before the butterflies; in frequency: after them).  R - 1 general multiplications
            // and the shifts instead of R/2 * K general ones; tw holds all n powers of the root here.
            const int shift = log_n - l - K;
            if constexpr (DIT) {
#pragma unroll
                for (int e = 1; e < R; e++) x[e] = bfe_mul(x[e], tw[(j0 * brev_k(e, K)) << shift]);
            }
            ntt_pow2_points<K, DIT, ROOT == 2>(x);
            if constexpr (!DIT) {
#pragma unroll
                for (int e = 1; e < R; e++) x[e] = bfe_mul(x[e], tw[(j0 * brev_k(e, K)) << shift]);
            }
        } else {
#pragma unroll
        for (int tt = 0; tt < K; tt++) {
            const int t = DIT ? tt : (K - 1 - tt);  // layer l + t: butterfly span 2^(l+t)
            const int h = 1 << t;
            const int tws = log_n - 1 - (l + t);    // w_{2^(l+t+1)}^j = w_n^(j << tws)
#pragma unroll
            for (int m = 0; m < h; m++) {           // the 2^t distinct twiddles of this layer in the group
                // in the group of the lowest four layers (l == 0) the twiddle of m == 0 is w^0 = 1: 15 of its 32
                // butterflies need no multiplication
                const bool unit = L0 && m == 0;
                const u64 w = unit ? TVM_ONE : tw[((m << l) | j0) << tws];
#pragma unroll
                for (int q = 0; q < R / (2 * h); q++) {
                    const int e = q * 2 * h + m;
                    const u64 u = x[e];
                    if (DIT) {
                        const u64 v = unit ? x[e + h] : bfe_mul(x[e + h], w);
                        x[e] = bfe_add(u, v);
                        x[e + h] = bfe_sub(u, v);
                    } else {
                        const u64 v = x[e + h];
                        x[e] = bfe_add(u, v);
                        x[e + h] = unit ? bfe_sub(u, v) : bfe_mul(bfe_sub(u, v), w);
                    }
                }
            }
        }
        }
#pragma unroll
        for (int e = 0; e < R; e++) p[e * stride] = x[e];
    }
    tvm_lds_barrier();
}

template <bool DIT, int MAXK = 4, int ROOT = 0>
TVM_D void lds_ntt(u64* s, int log_n, int batch_log, int SA, int SB, const u64* __restrict__ tw, int tid, int nt) {
    // DIT runs the layers upwards from span 1, DIF downwards from span n/2; groups of MAXK layers (4, or 3
    // where the caller keeps other per-thread state in VGPRs), the remainder as the last group
    int done = 0;
    while (done < log_n) {
        const int k = (log_n - done) >= MAXK ? MAXK : (log_n - done);
        const int l = DIT ? done : (log_n - done - k);
        if (l == 0) {
            if (MAXK >= 4 && k == 4) lds_ntt_group<DIT, 4, true, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else if (k == 3) lds_ntt_group<DIT, 3, true, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else if (k == 2) lds_ntt_group<DIT, 2, true, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else lds_ntt_group<DIT, 1, true, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
        } else {
            if (MAXK >= 4 && k == 4) lds_ntt_group<DIT, 4, false, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else if (k == 3) lds_ntt_group<DIT, 3, false, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else if (k == 2) lds_ntt_group<DIT, 2, false, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
            else lds_ntt_group<DIT, 1, false, -1, -1, 4, ROOT>(s, log_n, batch_log, SA, SB, tw, l, tid, nt);
        }
        done += k;
    }
}
// the same with the kind of root known only at run time (generic kernels: both forms are compiled)
template <bool DIT, int MAXK = 4>
TVM_D void lds_ntt_rt(int root, u64* s, int log_n, int batch_log, int SA, int SB, const u64* __restrict__ tw, int tid, int nt) {
    if (root == 1) lds_ntt<DIT, MAXK, 1>(s, log_n, batch_log, SA, SB, tw, tid, nt);
    else if (root == 2) lds_ntt<DIT, MAXK, 2>(s, log_n, batch_log, SA, SB, tw, tid, nt);
    else lds_ntt<DIT, MAXK, 0>(s, log_n, batch_log, SA, SB, tw, tid, nt);
}

// The same sequence of groups with everything known at compile time (see lds_ntt_group).
template <bool DIT, int MAXK, int LOGN, int DONE = 0, int CB = 4, int ROOT = 0>
TVM_D void lds_ntt_fixed(u64* s, const u64* __restrict__ tw, int tid, int nt) {
    if constexpr (DONE < LOGN) {
        constexpr int k = (LOGN - DONE) >= MAXK ? MAXK : (LOGN - DONE);
        constexpr int l = DIT ? DONE : (LOGN - DONE - k);
        lds_ntt_group<DIT, k, l == 0, l, LOGN, CB, ROOT>(s, LOGN, CB, 1, (1 << LOGN) + TVM_ROW_PAD, tw, l, tid, nt);
        lds_ntt_fixed<DIT, MAXK, LOGN, DONE + k, CB, ROOT>(s, tw, tid, nt);
    }
}

// ------------------------------------------------------------------------------------------------
// One transform row per WAVEFRONT.  A row of n = 2^LOGN points lives in LDS words only its wavefront touches, element p at
// row[p + (p >> 4)] (one pad word per 16: the stride-16 accesses of the lowest group and the consecutive accesses of the
// others all fall into distinct banks), and the butterfly groups of lds_ntt_group run with the wavefront's 64 lanes taking
// groups lane, lane + 64, ...: between groups the data crosses lanes of the SAME wavefront only, so there is no workgroup
// barrier inside a transform -- the wavefronts of a workgroup drift apart and one's LDS / memory phases run under another's
// butterflies.  Roots: the domains' own (ROOT = 1) or their inverses (2); tw = all n powers of the root.
#ifndef TVM_TW_BATCH
#define TVM_TW_BATCH 0   // twiddle loads in flight per batch in row_ntt_group (0: let the compiler schedule them)
#endif
#define TVM_ROW_SKEW(p) ((p) + ((p) >> 4))
#define TVM_ROW_WORDS(n) ((n) + ((n) >> 4) + 1)   // odd pitch: position p of the 16 rows of a tile falls into 16 different banks
// LPR: lanes per row -- 64 (the row's own wavefront; tvm_wave_sync between groups), 128 (two wavefronts share a row of 2048 points,
// `lane` is the lane number within the pair, and the groups are separated by workgroup barriers: every wavefront of the workgroup runs
// the same sequence), or 32 / 16 (round 6: rows of 512 / 256 points, two / four of them per wavefront, `lane` the number within the
// row's lanes; still wavefront-private: tvm_wave_sync)
// what separates two butterfly groups of a row by default: see LPR below
template <int LPR>
struct RowSync {
    TVM_D void operator()() const {
        if constexpr (LPR <= 64) tvm_wave_sync();
        else tvm_lds_barrier();
    }
};
template <bool DIT, int K, int L, int LOGN, int ROOT, int TWB = TVM_TW_BATCH, int LPR = 64, class Sync = RowSync<LPR>>
TVM_D void row_ntt_group(u64* row, const u64* __restrict__ tw, int lane, Sync& sync) {
    static_assert(ROOT == 1 || ROOT == 2, "the domains' own roots of unity only");
    constexpr int R = 1 << K, NG = 1 << (LOGN - K);
#pragma unroll 1
    for (int g = lane; g < NG; g += LPR) {
        const int j0 = g & ((1 << L) - 1);
        const int base = ((g >> L) << (L + K)) | j0;
        // skew(base + (e << L)) = skew(base) + skew(e << L): no carry into bit 4 -- for L >= 4 the second term is a multiple
        // of 16, below that base mod 16 = j0 < 2^L and (e << L) mod 16 <= 16 - 2^L.  Every LDS address of the group is the
        // lane's base plus an immediate.
        static_assert(L == 0 || L + K >= 4, "no carry into bit 4 of the skewed index");
        u64* const p = row + TVM_ROW_SKEW(base);
        u64 x[R];
#pragma unroll
        for (int e = 0; e < R; e++) x[e] = p[TVM_ROW_SKEW(e << L)];
        // (the twiddles are fetched TWB at a time where the caller says so: left alone, hipcc hoists all 2^K - 1 loads above the
        // multiplications -- 30 VGPRs that the coset loop of pass 2 does not have)
        if constexpr (L > 0 && DIT) {
#pragma unroll
            for (int e = 1; e < R; e++) {
                x[e] = bfe_mul(x[e], tw[(j0 * brev_k(e, K)) << (LOGN - L - K)]);
                if constexpr (TWB > 0) {
                    if (e % TWB == 0) asm volatile("" ::: "memory");
                }
            }
        }
        ntt_pow2_points<K, DIT, ROOT == 2>(x);
        if constexpr (L > 0 && !DIT) {
#pragma unroll
            for (int e = 1; e < R; e++) {
                x[e] = bfe_mul(x[e], tw[(j0 * brev_k(e, K)) << (LOGN - L - K)]);
                if constexpr (TWB > 0) {
                    if (e % TWB == 0) asm volatile("" ::: "memory");
                }
            }
        }
#pragma unroll
        for (int e = 0; e < R; e++) p[TVM_ROW_SKEW(e << L)] = x[e];
    }
    sync();
}
template <bool DIT, int K, int L, int LOGN, int ROOT, int TWB = TVM_TW_BATCH, int LPR = 64>
TVM_D void row_ntt_group(u64* row, const u64* __restrict__ tw, int lane) {
    RowSync<LPR> sync;
    row_ntt_group<DIT, K, L, LOGN, ROOT, TWB, LPR, RowSync<LPR>>(row, tw, lane, sync);
}
// (`sync`: called after every group; the default is the wavefront-level wait for rows of <= 64 lanes and the workgroup barrier for
// rows of 128 -- the kernels with two wavefronts per row pass their pair synchronisation instead)
template <bool DIT, int MAXK, int LOGN, int ROOT, int DONE = 0, int LPR = 64, class Sync = RowSync<LPR>>
TVM_D void row_ntt(u64* row, const u64* __restrict__ tw, int lane, Sync& sync) {
    if constexpr (DONE < LOGN) {
        constexpr int k = (LOGN - DONE) >= MAXK ? MAXK : (LOGN - DONE);
        constexpr int l = DIT ? DONE : (LOGN - DONE - k);
        row_ntt_group<DIT, k, l, LOGN, ROOT, TVM_TW_BATCH, LPR, Sync>(row, tw, lane, sync);
        row_ntt<DIT, MAXK, LOGN, ROOT, DONE + k, LPR, Sync>(row, tw, lane, sync);
    }
}
template <bool DIT, int MAXK, int LOGN, int ROOT, int DONE = 0, int LPR = 64>
TVM_D void row_ntt(u64* row, const u64* __restrict__ tw, int lane) {
    RowSync<LPR> sync;
    row_ntt<DIT, MAXK, LOGN, ROOT, DONE, LPR, RowSync<LPR>>(row, tw, lane, sync);
}

// ------------------------------------------------------------------------------------------------
// Generic two-pass transform of `ncols` columns (grid.y), natural order in and out.
//   X[k1 + N1*k2] = sum_{i2} w_N^(i2 k1) w_N2^(i2 k2) sum_{i1} x[i1*N2 + i2] w_N1^(i1 k1)
// Input column v starts at in + (v / fk) * in_col_stride + v % fk with element stride fk, so an
// array-of-XFE (fk = 3) is three interleaved base-field columns; output likewise with out_fk.
struct Ntt2Args {
    const u64* in;
    u64* tmp;
    u64* out;
    int log_n1, log_n2, batch_log;
    u64 in_len;                  // inputs at index >= in_len are zero (zero padding)
    int in_fk, out_fk;
    u64 in_col_stride, tmp_col_stride, out_col_stride;
    const u64* tw1;              // w_N1^e, e < N1/2
    const u64* tw2;              // w_N2^e, e < N2/2
    Pow2 tw_inter;               // w_N^e
    const u64* pre_hi;           // optional input scaling x[i1*N2+i2] *= pre_hi[i1] * pre_lo[i2]
    const u64* pre_lo;
    const u64* post_lo;          // optional output scaling X[k1+N1*k2] *= post_lo[k1] * post_hi[k2]
    const u64* post_hi;
    u64 out_mul, out_add;        // output element k is stored at index k*out_mul + out_add
    int col0;                    // first virtual column (in/out use col0 + blockIdx.y, tmp uses blockIdx.y)
    int root;                    // 1 / 2: w is the domains' own N-th root of unity / its inverse (shift twiddles, lds_ntt_group)
};


__global__ void __launch_bounds__(1024) k_ntt2_pass1(Ntt2Args a) {
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n1 = 1 << a.log_n1;
    const u64 n2 = 1ull << a.log_n2;
    const int B = 1 << a.batch_log;
    const int vl = blockIdx.y, v = a.col0 + vl;
    const u64 i2_0 = (u64)blockIdx.x * B;
    const u64* in = a.in + (u64)(v / a.in_fk) * a.in_col_stride + (v % a.in_fk);
    const int tile = n1 << a.batch_log;
    for (int idx = tid; idx < tile; idx += nt) {
        const int b = idx & (B - 1), i1 = idx >> a.batch_log;
        const u64 i2 = i2_0 + b;
        const u64 i = (u64)i1 * n2 + i2;
        u64 x = (i2 < n2 && i < a.in_len) ? TVM_LOAD_STREAM(&in[i * a.in_fk]) : 0;
        if (a.pre_lo && x) x = bfe_mul(x, bfe_mul(a.pre_hi[i1], a.pre_lo[i2]));
        s[idx] = x;
    }
    tvm_lds_barrier();
    lds_ntt_rt<false>(a.root, s, a.log_n1, a.batch_log, B, 1, a.tw1, tid, nt);
    u64* tmp = a.tmp + (u64)vl * a.tmp_col_stride;
    for (int idx = tid; idx < tile; idx += nt) {
        const int b = idx & (B - 1), p = idx >> a.batch_log;
        const u64 i2 = i2_0 + b;
        if (i2 >= n2) continue;
        const u64 k1 = brev_bits((u32)p, a.log_n1);
        TVM_STORE_STREAM(&tmp[(u64)p * n2 + i2], bfe_mul(s[idx], pow2_get(a.tw_inter, i2 * k1)));
    }
}

__global__ void __launch_bounds__(1024) k_ntt2_pass2(Ntt2Args a) {
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const u64 n1 = 1ull << a.log_n1;
    const int n2 = 1 << a.log_n2;
    const int B = 1 << a.batch_log;
    const int RS = n2 + TVM_ROW_PAD;
    const int vl = blockIdx.y, v = a.col0 + vl;
    const u64 k1_0 = (u64)blockIdx.x * B;
    const u64* tmp = a.tmp + (u64)vl * a.tmp_col_stride;
    const int tile = n2 << a.batch_log;
    for (int idx = tid; idx < tile; idx += nt) {
        const int i2 = idx & (n2 - 1), b = idx >> a.log_n2;
        const u64 k1 = k1_0 + b;
        u64 x = 0;
        if (k1 < n1) x = tmp[(u64)brev_bits((u32)k1, a.log_n1) * n2 + i2];
        s[b * RS + i2] = x;
    }
    tvm_lds_barrier();
    lds_ntt_rt<false>(a.root, s, a.log_n2, a.batch_log, 1, RS, a.tw2, tid, nt);
    u64* out = a.out + (u64)(v / a.out_fk) * a.out_col_stride + (v % a.out_fk);
    for (int idx = tid; idx < tile; idx += nt) {
        const int b = idx & (B - 1), q = idx >> a.batch_log;
        const u64 k1 = k1_0 + b;
        if (k1 >= n1) continue;
        const u64 k2 = brev_bits((u32)q, a.log_n2);
        u64 x = s[b * RS + q];
        if (a.post_lo) x = bfe_mul(x, bfe_mul(a.post_lo[k1], a.post_hi[k2]));
        const u64 k = k1 + n1 * k2;
        out[(k * a.out_mul + a.out_add) * a.out_fk] = x;
    }
}

// ------------------------------------------------------------------------------------------------
// Master-table LDE, passes 2 and 3 (pass 1 is k_ntt2_pass1 with inverse tables).
#define TVM_LDE_MAX_COSETS 32
#define TVM_LDE_E 16  // coefficients a thread keeps in VGPRs across the coset loop
#define TVM_LDE_INVERSE_ONLY 1
#define TVM_LDE_FORWARD_ONLY 2

struct LdePass2Args {
    const u64* y;        // [cols][N1 positions p][N2]
    u64* z;              // [cols][X][N2 (j1)][N1 positions p]
    const u64* rnd;      // trace randomizers [n_cols][h][fk]
    int log_n1, log_n2, batch_log;
    int fk;
    int col0;            // first virtual column of this chunk (y and z are chunk-local)
    u64 h;
    int n_cosets;
    const u64* tw_a2;    // w_N2^-e
    const u64* tw_b1;    // w_N2^e
    Pow2 tw_inter;       // w_N^e
    const u64* g_lo;     // [X][N1]: gamma_k^m2 / N
    const u64* g_hi;     // [X][N2]: gamma_k^(N1*m1)
    const u64* g_lo_step;  // [N1]: (gamma_{k+1} / gamma_k)^m2       (the row kernels walk the cosets with running
    const u64* g_hi_step;  // [N2]: (gamma_{k+1} / gamma_k)^(N1*m1)    products: no table load inside their coset loop)
    u64 zk[TVM_LDE_MAX_COSETS];  // N * (gamma_k^N - 1)
    int std_roots;       // the trace domain's generator is the domains' own root of unity (shift twiddles, lds_ntt_group)
    // The pass split at the coefficients (the column sharding of SURVEY 8(e): a rank interpolates ITS columns, the coefficients
    // are exchanged, every rank extends all columns onto its cosets).  0: the whole pass.  TVM_LDE_INVERSE_ONLY: the inverse rows
    // step, then the tile goes back to Y in place -- Y then holds N * t[m1*n1 + m2] at [position p of m2][position q of m1], the
    // library's COEFFICIENT FORM of a column -- and the kernel returns.  TVM_LDE_FORWARD_ONLY: Y already holds that form.
    int mode;
    // k_lde_pass2_fused (1024-point axes) only:
    const u64* g_hi_pos; // [X][N2]: gamma_k^(N1*brev(q)) at POSITION q of a row (g_hi in the order the row holds its coefficients)
    const u64* f_tw;     // [N1 rows p][64 lanes][4 it][4 e]: w_N2^(g brev2(e)) * w_N^(brev(p) g), g = lane + 64 it (a lane's 16 values: one line)
    const u64* u_tw;     // [X][N1 rows p][4]: (gamma_k^m2 / N) * w_N^(m2 * 256 e), m2 = brev(p)
};

__global__ void __launch_bounds__(1024) k_lde_pass2(LdePass2Args a) {
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const u64 n1 = 1ull << a.log_n1;
    const int n2 = 1 << a.log_n2;
    const int B = 1 << a.batch_log;
    const int RS = n2 + TVM_ROW_PAD;
    const int vl = blockIdx.y;            // chunk-local virtual column
    const int v = a.col0 + vl;
    const u64 p0 = (u64)blockIdx.x * B;   // tile rows are positions p0..p0+B-1, k1 = brev(p)
    const u64 n = n1 << a.log_n2;
    const u64* y = a.y + (u64)vl * n;
    const int tile = n2 << a.batch_log;

    for (int idx = tid; idx < tile; idx += nt) {
        const int i2 = idx & (n2 - 1), b = idx >> a.log_n2;
        s[b * RS + i2] = (p0 + b < n1) ? y[(p0 + b) * n2 + i2] : 0;
    }
    tvm_lds_barrier();
    // inverse rows step: position q of row b now holds N * t[k1 + N1*k2], k2 = brev(q)
    if (a.mode != TVM_LDE_FORWARD_ONLY) lds_ntt_rt<false>(a.std_roots ? 2 : 0, s, a.log_n2, a.batch_log, 1, RS, a.tw_a2, tid, nt);
    if (a.mode == TVM_LDE_INVERSE_ONLY) {
        u64* yw = const_cast<u64*>(y);
        for (int idx = tid; idx < tile; idx += nt) {
            const int i2 = idx & (n2 - 1), b = idx >> a.log_n2;
            if (p0 + b < n1) yw[(p0 + b) * n2 + i2] = s[b * RS + i2];
        }
        return;
    }

    // The N coefficients of this tile stay in VGPRs for the whole coset loop (the only per-thread
    // state: 16 words); coset factors come from two small L2-resident tables per coset.
    u64 coef[TVM_LDE_E];
    const u64* rnd = a.rnd + (u64)(v / a.fk) * a.h * a.fk + (v % a.fk);
#pragma unroll
    for (int e = 0; e < TVM_LDE_E; e++) {
        const int idx = tid + e * nt;
        coef[e] = 0;
        if (idx < tile) coef[e] = s[(idx >> a.log_n2) * RS + (idx & (n2 - 1))];
    }
    for (int k = 0; k < a.n_cosets; k++) {
        tvm_lds_barrier();
        const u64 zk = a.zk[k];
        const u64* g_lo = a.g_lo + (u64)k * n1;
        const u64* g_hi = a.g_hi + (u64)k * n2;
#pragma unroll
        for (int e = 0; e < TVM_LDE_E; e++) {
            const int idx = tid + e * nt;
            if (idx < tile) {
                const int q = idx & (n2 - 1), b = idx >> a.log_n2;
                const u64 m1 = brev_bits((u32)q, a.log_n2);
                const u64 m2 = brev_bits((u32)(p0 + b), a.log_n1);
                const u64 m = m1 * n1 + m2;
                u64 c = coef[e];
                if (m < a.h && p0 + b < n1) c = bfe_add(c, bfe_mul(zk, rnd[m * a.fk]));  // few lanes
                s[b * RS + q] = bfe_mul(c, bfe_mul(g_lo[m2 & (n1 - 1)], g_hi[m1]));
            }
        }
        tvm_lds_barrier();
        // forward columns step over m1 (bit-reversed in position q): natural j1 out
        lds_ntt_rt<true, 3>(a.std_roots ? 1 : 0, s, a.log_n2, a.batch_log, 1, RS, a.tw_b1, tid, nt);  // coef[] stays live: 8-element groups
        u64* z = a.z + ((u64)vl * a.n_cosets + k) * n;
        for (int idx = tid; idx < tile; idx += nt) {
            const int b = idx & (B - 1), j1 = idx >> a.batch_log;
            if (p0 + b >= n1) continue;
            const u64 m2 = brev_bits((u32)(p0 + b), a.log_n1);
            z[(u64)j1 * n1 + p0 + b] = bfe_mul(s[b * RS + j1], pow2_get(a.tw_inter, m2 * (u64)j1));
        }
    }
}

// Pass 3 writes the table (context.h: coset-major, inside a coset in the order of this pass).  A workgroup owns one virtual
// column and 2^rows_log (<= 16) of the (k, j1) rows of Z, enumerated rho = X*j1 + k: it transforms them over m2 and stores, for
// every row, its n1 results j2 = 0 .. n1-1 into the n1 CONSECUTIVE storage rows k*pitch + j1*n1 + j2 -- consecutive lanes
// write consecutive rows, i.e. full 128-byte lines of the row-block-major table whatever the number of rows in the tile.
struct LdePass3Args {
    const u64* z;        // [cols][X][N2 (j1)][N1 positions p]
    u64* table;          // row-block-major over storage rows, W words per row
    int log_n1, log_n2;
    int n_cosets;
    int col0;            // first virtual column of the chunk
    int tiles;           // k_lde_pass3_v3 / _rows: consecutive row tiles per workgroup
    int W;               // words per table row
    u64 L;
    u64 pitch;           // storage rows per coset
    const u64* tw_b2;    // w_N1^e
    int rows_log;        // rows per workgroup tile
    int std_roots;       // see LdePass2Args
    const u64* fb_tw;    // k_lde_pass3_halves: the second half's last-group factors (k_pass3_halves_table)
};

__global__ void __launch_bounds__(1024) k_lde_pass3(LdePass3Args a) {
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n1 = 1 << a.log_n1;
    const u64 n2 = 1ull << a.log_n2;
    const int RS = n1 + TVM_ROW_PAD;
    const int RB = 1 << a.rows_log;
    const u64 X = (u64)a.n_cosets;
    const u64 period = X * n2;                 // rows per j2
    const u64 rho0 = (u64)blockIdx.x * RB;     // first local row of the tile
    const int vl = blockIdx.y;
    const u64 n = (u64)n1 << a.log_n2;
    const int tile = n1 << a.rows_log;
    for (int idx = tid; idx < tile; idx += nt) {
        const int p = idx & (n1 - 1), b = idx >> a.log_n1;
        const u64 rho = rho0 + b;
        u64 x = 0;
        if (rho < period) {
            const u64 j1 = rho / X, k = rho % X;
            x = a.z[(((u64)vl * X + k) * n2 + j1) * n1 + p];
        }
        s[b * RS + p] = x;
    }
    tvm_lds_barrier();
    lds_ntt_rt<true>(a.std_roots ? 1 : 0, s, a.log_n1, a.rows_log, 1, RS, a.tw_b2, tid, nt);
    const u64 v = (u64)(a.col0 + vl);
    for (int idx = tid; idx < tile; idx += nt) {
        const int j2 = idx & (n1 - 1), b = idx >> a.log_n1;
        const u64 rho = rho0 + b;
        if (rho >= period) continue;
        const u64 j1 = rho / X, k = rho % X;
        a.table[tvm_tab_idx(k * a.pitch + j1 * (u64)n1 + (u64)j2, v, (u64)a.W)] = s[b * RS + j2];
    }
}

// ------------------------------------------------------------------------------------------------
// The same two passes for an LDS-resident axis LONGER than a workgroup: 2^LOGN points on 2^TLOG work-items, i.e.
// PPT = 2^(LOGN - TLOG) positions per work-item and tiles of 16 / PPT rows, so that a work-item still owns 16 elements and
// the tile still holds 16 * 2^TLOG words (2048-point axes on 1024 work-items with 8-row tiles: traces of 2^21 and 2^22
// rows, which the production kernels above -- one position per work-item -- cannot take; 4096-point axes with 4-row tiles:
// 2^23 and 2^24 rows; <7, 6> and <8, 6> are the same shapes at sizes the CPU suite can run).  Element e of a work-item: row e % ROWS, position tid + (e / ROWS) * 2^TLOG.
// Same results as the generic k_lde_pass2 / k_lde_pass3; what the tile ownership buys is that every index a work-item needs is
// either its own constant or uniform across the workgroup, so the scale and store phases carry no per-element index arithmetic:
//   * a work-item owns the same positions q of all rows of the tile across the coset loop: m1 = brev(q) is its constant,
//     m2 = brev(p0 + row) is uniform, and only work-items with m1*n1 < h ever see a randomizer coefficient;
//   * gamma_k^m / N = gamma_k^(n1*m1) * (gamma_k^m2 / N): the first factor is applied before the column step (a running product
//     over the cosets, one multiplication per element), the second is a per-row constant, commutes with the column step and is
//     merged into the inter-pass twiddle of the store phase, which itself is a running product  T(j1 + step) = T(j1) * w_N^(m2*step)
//     instead of two table loads per element.
template <int LOGN, int TLOG>
__global__ void __launch_bounds__(1 << TLOG) k_lde_pass2_v3(LdePass2Args a) {
    constexpr int NT = 1 << TLOG, RLOG = 4 - (LOGN - TLOG), ROWS = 1 << RLOG, PPT = 16 / ROWS;
    constexpr int n2 = 1 << LOGN, RS = n2 + TVM_ROW_PAD;
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x;
    const u64 n1 = 1ull << a.log_n1;
    const int vl = blockIdx.y, v = a.col0 + vl;
    const u64 p0 = (u64)blockIdx.x * ROWS;
    const u64 n = n1 << LOGN;
    const u64* y = a.y + (u64)vl * n + p0 * n2;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int r = e & (ROWS - 1), q = tid + (e >> RLOG) * NT;
        s[r * RS + q] = TVM_LOAD_STREAM(&y[(u64)r * n2 + q]);
    }
    tvm_lds_barrier();
    if (a.mode != TVM_LDE_FORWARD_ONLY) lds_ntt_fixed<false, 4, LOGN, 0, RLOG, 2>(s, a.tw_a2, tid, NT);
    if (a.mode == TVM_LDE_INVERSE_ONLY) {
        u64* yw = const_cast<u64*>(y);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = e & (ROWS - 1), q = tid + (e >> RLOG) * NT;
            yw[(u64)r * n2 + q] = s[r * RS + q];
        }
        return;
    }
    u64 coef[16];
#pragma unroll
    for (int e = 0; e < 16; e++) coef[e] = s[(e & (ROWS - 1)) * RS + tid + (e >> RLOG) * NT];
    // the forward twiddles (all n2 powers) behind the tile; a 4096-point axis leaves no room: global
    const u64* tw_fwd = a.tw_b1;
    if constexpr (LOGN < 12) {
        u64* tw_lds = s + ROWS * RS;
        for (int i = tid; i < n2; i += NT) tw_lds[i] = a.tw_b1[i];
        tw_fwd = tw_lds;
    }
    u64 m1[PPT], gh[PPT], gh_step[PPT];
    bool has_rnd = false;
#pragma unroll
    for (int hh = 0; hh < PPT; hh++) {
        m1[hh] = brev_bits((u32)(tid + hh * NT), LOGN);
        has_rnd = has_rnd || m1[hh] * n1 < a.h;
        gh[hh] = a.g_hi[m1[hh]];
        gh_step[hh] = a.g_hi_step[m1[hh]];
    }
    const u64* rnd = a.rnd + (u64)(v / a.fk) * a.h * a.fk + (v % a.fk);
    // h <= n1: only position 0 of each row (work-item 0's first ROWS coefficients) sees a randomizer -- parked in LDS, written
    // into the tile by work-items 0 .. ROWS-1 in every coset: no global load in the coset loop
    const bool single = a.h <= n1;
    u64* c0 = s + ROWS * RS + (LOGN < 12 ? n2 : 0);
    u64* r0 = c0 + 16;
    if (single) {
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < ROWS; e++) c0[e] = coef[e];
        }
        if (tid < ROWS) {
            const u64 m = brev_bits((u32)(p0 + tid), a.log_n1);
            r0[tid] = m < a.h ? rnd[m * a.fk] : 0;
        }
    }
    // store phase: this work-item writes row b = tid % ROWS, columns j1 = tid / ROWS + i * NT / ROWS, i < 16
    // (the column slots of a wavefront are ROWS apart: LDS bank conflicts, see k_lde_pass3_v3)
    constexpr int LPW = 64 >> RLOG;
    const int b_out = tid & (ROWS - 1), q_ = tid >> RLOG;
    const int j1_0 = (NT >> RLOG) >= 64 ? ((q_ & ~63) | ((q_ & (LPW - 1)) << RLOG) | ((q_ & 63) >> (6 - RLOG))) : q_;
    constexpr int j1_step = NT >> RLOG;
    const u64 m2_out = brev_bits((u32)(p0 + b_out), a.log_n1);
    const u64 t_step = pow2_get(a.tw_inter, (m2_out * (u64)j1_step) & (n - 1));
    u64 t_first = bfe_mul(pow2_get(a.tw_inter, m2_out * (u64)j1_0), a.g_lo[m2_out]);
    const u64 gl_step = a.g_lo_step[m2_out];
    for (int k = 0; k < a.n_cosets; k++) {
        tvm_lds_barrier();
        const u64 zk = a.zk[k];
        if (single) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = e & (ROWS - 1), hh = e >> RLOG;
                if (hh || tid) s[r * RS + tid + hh * NT] = bfe_mul(coef[e], gh[hh]);
            }
            if (tid < ROWS) s[tid * RS] = bfe_add(c0[tid], bfe_mul(zk, r0[tid]));
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int r = e & (ROWS - 1), hh = e >> RLOG;
                u64 c = coef[e];
                if (has_rnd) {
                    const u64 m = m1[hh] * n1 + brev_bits((u32)(p0 + r), a.log_n1);
                    if (m < a.h) c = bfe_add(c, bfe_mul(zk, rnd[m * a.fk]));
                }
                s[r * RS + tid + hh * NT] = bfe_mul(c, gh[hh]);
            }
        }
        tvm_lds_barrier();
        lds_ntt_fixed<true, 3, LOGN, 0, RLOG, 1>(s, tw_fwd, tid, NT);
        u64* z = a.z + ((u64)vl * a.n_cosets + k) * n + p0 + b_out;
        u64 t = t_first;
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const int j1 = j1_0 + i * j1_step;
            TVM_STORE_STREAM(&z[(u64)j1 * n1], bfe_mul(s[b_out * RS + j1], t));
            t = bfe_mul(t, t_step);
        }
#pragma unroll
        for (int hh = 0; hh < PPT; hh++) gh[hh] = bfe_mul(gh[hh], gh_step[hh]);
        t_first = bfe_mul(t_first, gl_step);
    }
}

template <int LOGN, int TLOG>
__global__ void __launch_bounds__(1 << TLOG) k_lde_pass3_v3(LdePass3Args a) {
    constexpr int NT = 1 << TLOG, RLOG = 4 - (LOGN - TLOG), ROWS = 1 << RLOG;
    constexpr int n1 = 1 << LOGN, RS = n1 + TVM_ROW_PAD;
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x;
    const u64 n2 = 1ull << a.log_n2;
    const u64 X = (u64)a.n_cosets;
    const int log_x = 31 - __builtin_clz((unsigned)a.n_cosets);
    const int vl = blockIdx.x;
    const u64* zc = a.z + (u64)vl * X * (n2 << LOGN) + tid;
    const u64* tw_fwd = a.tw_b2;
    if constexpr (LOGN < 12) {
        u64* tw_lds = s + ROWS * RS;
        for (int i = tid; i < n1; i += NT) tw_lds[i] = a.tw_b2[i];
        tw_fwd = tw_lds;
    }
    // store phase: work-item tid writes positions j2 = tid + hh * NT of every row of the tile (consecutive lanes =
    // consecutive storage rows and consecutive LDS words: full lines, no bank conflicts)
    const u64 W = (u64)a.W;
    u64* const out_t = a.table + ((((u64)(tid >> TVM_RB_LOG)) * W + (u64)(a.col0 + vl)) << TVM_RB_LOG) + (tid & (TVM_RB - 1));
    u64 nxt[16];
    u64 rho0 = (u64)blockIdx.y * a.tiles * ROWS;  // first local row of the tile
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const u64 rho = rho0 + (e & (ROWS - 1)), j1 = rho >> log_x, k = rho & (X - 1);
        nxt[e] = TVM_LOAD_STREAM(&zc[((k * n2 + j1) << LOGN) + (u64)(e >> RLOG) * NT]);
    }
    for (int it = 0; it < a.tiles; it++, rho0 += ROWS) {
        if (it) tvm_lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; e++) s[(e & (ROWS - 1)) * RS + tid + (e >> RLOG) * NT] = nxt[e];
        tvm_lds_barrier();
        if (it + 1 < a.tiles) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const u64 rho = rho0 + ROWS + (e & (ROWS - 1)), j1 = rho >> log_x, k = rho & (X - 1);
                nxt[e] = TVM_LOAD_STREAM(&zc[((k * n2 + j1) << LOGN) + (u64)(e >> RLOG) * NT]);
            }
        }
        lds_ntt_fixed<true, 4, LOGN, 0, RLOG, 1>(s, tw_fwd, tid, NT);
#pragma unroll 4
        for (int e = 0; e < 16; e++) {
            const int r = e & (ROWS - 1), hh = e >> RLOG;
            const u64 rho = rho0 + r, j1 = rho >> log_x, k = rho & (X - 1);
            const u64 blk = ((k * a.pitch + (j1 << LOGN)) >> TVM_RB_LOG) + (u64)(hh * (NT >> TVM_RB_LOG));
            TVM_STORE_STREAM(&out_t[(blk * W) << TVM_RB_LOG], s[r * RS + tid + hh * NT]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Passes 2 and 3 with one row per wavefront (row_ntt), for 1024-point axes (traces of 2^19 and 2^20 rows).
//
// Pass 3: a wavefront owns one (k, j1) row of Z at a time -- loads its 1024 words (16 coalesced loads per lane), transforms
// them in its own LDS words and stores the 1024 results into 1024 consecutive storage rows of the table (context.h): NO
// workgroup barrier after the twiddle table is staged.  Work-item = lane of wavefront w of WAVES; the workgroup walks a.tiles
// tiles of WAVES rows, every wavefront with the loads of its next row in flight under the butterflies of the current one.
// (2048-point rows, round 4: 32 elements per lane, 17 KB of LDS per wavefront -- eight wavefronts fill a CU's LDS, two per SIMD, so
// the kernel may use 256 VGPRs and keeps the next row's 32 loads in flight)
// LOGN = 9 / 8 (round 6: the axes of 2^16 .. 2^19-row traces, which ran the generic tile kernel k_lde_pass3 until then): still one row per
// wavefront -- 8 / 4 points per lane -- with radix-8 butterfly groups, so that every group of a 512-point row has a butterfly for
// every lane (3 + 3 + 3 layers; 256 points: 3 + 3 + 2, half the lanes idle in the first two groups).
template <int LOGN, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, LOGN <= 10 ? 4 : 2) k_lde_pass3_rows(LdePass3Args a) {   // up to 1024 points: room for 4 wavefronts per SIMD (128 VGPRs)
    constexpr int n1 = 1 << LOGN, ROWW = TVM_ROW_WORDS(n1), E = n1 / 64, MAXK = LOGN >= 10 ? 4 : 3;
    static_assert(E == 4 || E == 8 || E == 16 || E == 32, "4 .. 32 elements per lane");
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 n2 = 1ull << a.log_n2;
    const u64 X = (u64)a.n_cosets;
    const int log_x = 31 - __builtin_clz((unsigned)a.n_cosets);
    const int vl = blockIdx.x;
    const u64* zc = a.z + (u64)vl * X * (n2 << LOGN) + lane;
    u64* row = s + w * ROWW;
    u64* const rowl = row + TVM_ROW_SKEW(lane);
    u64* tw_lds = s + WAVES * ROWW;
    for (int i = tid; i < n1; i += 64 * WAVES) tw_lds[i] = a.tw_b2[i];
    __syncthreads();   // the only workgroup barrier: the twiddles are read-only from here on
    const u64 W = (u64)a.W;
    u64* const out_l = a.table + ((((u64)(lane >> TVM_RB_LOG)) * W + (u64)(a.col0 + vl)) << TVM_RB_LOG) + (lane & (TVM_RB - 1));
    u64 rho = ((u64)blockIdx.y * a.tiles) * WAVES + w;   // this wavefront's row of the first tile
    u64 nxt[E];
    {
        const u64 j1 = rho >> log_x, k = rho & (X - 1);
#pragma unroll
        for (int e = 0; e < E; e++) nxt[e] = TVM_LOAD_STREAM(&zc[((k * n2 + j1) << LOGN) + 64 * e]);
    }
    for (int it = 0; it < a.tiles; it++, rho += WAVES) {
#pragma unroll
        for (int e = 0; e < E; e++) rowl[68 * e] = nxt[e];   // position lane + 64 e (skew: + 4 e)
        tvm_wave_sync();
        if (it + 1 < a.tiles) {
            const u64 r2 = rho + WAVES, j1 = r2 >> log_x, k = r2 & (X - 1);
#pragma unroll
            for (int e = 0; e < E; e++) nxt[e] = TVM_LOAD_STREAM(&zc[((k * n2 + j1) << LOGN) + 64 * e]);
        }
        row_ntt<true, MAXK, LOGN, 1>(row, tw_lds, lane);
        const u64 j1 = rho >> log_x, k = rho & (X - 1);
        const u64 blk = (k * a.pitch + (j1 << LOGN)) >> TVM_RB_LOG;   // the row's first 16-row block of the table
#pragma unroll 4
        for (int e = 0; e < E; e++)   // j2 = lane + 64 e: consecutive lanes = consecutive storage rows, 4 full lines per store
            TVM_STORE_STREAM(&out_l[((blk + 4 * e) * W) << TVM_RB_LOG], rowl[68 * e]);
        tvm_wave_sync();   // (the next tile overwrites the row)
    }
}

// Pass 3 on 2048-point rows (2^22 and 2^23-row traces) as TWO 1024-point halves through ONE 1024-word LDS region per wavefront (round 5).
// k_lde_pass3_rows<11, 8> keeps a whole row of 2048 points in LDS: 17 KB per wavefront, eight of them fill a CU, two wavefronts per SIMD
// -- and runs 36 % less efficiently per element than the 1024-point kernel at four.  A decimation-in-time transform of 2048 points in
// bit-reversed order is two independent 1024-point transforms of its halves (layers 0 .. 9 never cross the middle, and their twiddles
// w_(2^(l+1))^(pos mod 2^l) do not know which half they are in) followed by ONE layer across the halves, X[j], X[j + 1024] = A[j] +-
// w_2048^j B[j].  So: half A through the wavefront's LDS region (row_ntt, as for 1024-point rows), its 16 results per lane back in
// VGPRs; half B through the same region, with w_2048^j folded into its last butterfly group the way pass 2 folds the inter-pass
// twiddle -- w_2048^(g + 256 e) = w_2048^g * w_8^e: the first factor rides on the group's input twiddles (fb_tw: 1024 row-independent
// values, read at their use), the second is a power of two -- and the last layer is an addition and a subtraction.  8.7 KB of LDS per
// wavefront: two workgroups of eight per CU, four wavefronts per SIMD, 128 VGPRs.
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 4) k_lde_pass3_halves(LdePass3Args a) {
    constexpr int LOGH = 10, nh = 1 << LOGH, ROWW = TVM_ROW_WORDS(nh);
    TVM_DYN_SMEM(u64, s);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 n2 = 1ull << a.log_n2;
    const u64 X = (u64)a.n_cosets;
    const int log_x = 31 - __builtin_clz((unsigned)a.n_cosets);
    const int vl = blockIdx.x;
    u64* const row = s + w * ROWW;
    u64* tw_lds = s + WAVES * ROWW;
    for (int i = tid; i < nh; i += 64 * WAVES) tw_lds[i] = a.tw_b2[2 * i];   // all 1024 powers of w_1024 = w_2048^2
    __syncthreads();   // the only workgroup barrier
    const u64 W = (u64)a.W;
    u64 rho = ((u64)blockIdx.y * a.tiles) * WAVES + w;
    {   // the first row's half A into the region
        const u64 j1 = rho >> log_x, k = rho & (X - 1);
        const u64* const z_row = a.z + ((((u64)vl * X + k) * n2 + j1) << 11) + lane;
        u64* const rl = row + TVM_ROW_SKEW(lane);
#pragma unroll
        for (int e = 0; e < 16; e++) rl[68 * e] = TVM_LOAD_STREAM(&z_row[64 * e]);
    }
    for (int it = 0; it < a.tiles; it++, rho += WAVES) {
        const u64 j1 = rho >> log_x, k = rho & (X - 1);
        const int ln = tvm_opaque(lane);   // (addresses formed where they are used: kept across the loop they went to scratch)
        const u64* const z_row = a.z + ((((u64)vl * X + k) * n2 + j1) << 11) + ln;
        u64* const rl = row + TVM_ROW_SKEW(ln);
        tvm_wave_sync();   // half A is in the region (parked there at the end of the previous row's work)
        u64 in_b[16];      // half B: requested now, in flight under A's transform
#pragma unroll
        for (int e = 0; e < 16; e++) in_b[e] = TVM_LOAD_STREAM(&z_row[nh + 64 * e]);
        row_ntt_group<true, 4, 0, LOGH, 1, 4>(row, tw_lds, ln);
        row_ntt_group<true, 4, 4, LOGH, 1, 4>(row, tw_lds, ln);
        row_ntt_group<true, 2, 8, LOGH, 1, 4>(row, tw_lds, ln);
        u64 ra[16];
#pragma unroll
        for (int e = 0; e < 16; e++) ra[e] = rl[68 * e];
        tvm_wave_sync();
        // half B
#pragma unroll
        for (int e = 0; e < 16; e++) rl[68 * e] = in_b[e];
        tvm_wave_sync();
        row_ntt_group<true, 4, 0, LOGH, 1, 4>(row, tw_lds, ln);
        row_ntt_group<true, 4, 4, LOGH, 1, 4>(row, tw_lds, ln);
        // the next row's half A is requested once this row's results have left the region and parked there at once: requested any
        // earlier (before the last group, before the stores) it costs 108 B of scratch and 17 % of the kernel (15.8 against 13.5 ms
        // per chunk, profiles/r05_o_*); carried across the loop's back edge in registers it went to scratch as well
        const bool more = it + 1 < a.tiles;
        u64 nxt[16];
        auto request_next = [&] {
            const u64 r2 = rho + WAVES, j1n = r2 >> log_x, kn = r2 & (X - 1);
            const u64* const z_next = a.z + ((((u64)vl * X + kn) * n2 + j1n) << 11) + tvm_opaque(lane);
#pragma unroll
            for (int e = 0; e < 16; e++) nxt[e] = TVM_LOAD_STREAM(&z_next[64 * e]);
        };
        const u64x2* const fb = (const u64x2*)(a.fb_tw + 16 * ln);   // the lane's 16 factors: one line
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {   // layers 8 and 9 of the half on positions g + 256 e, g = lane + 64 g4, times w_2048^(g + 256 e)
            u64* const q = row + TVM_ROW_SKEW(ln + 64 * g4);   // TVM_ROW_SKEW(g + 256 e) = TVM_ROW_SKEW(g) + 272 e
            const u64x2 f01 = fb[2 * g4], f23 = fb[2 * g4 + 1];
            u64 y[4] = {bfe_mul(q[0], f01.x), bfe_mul(q[272], f01.y), bfe_mul(q[544], f23.x), bfe_mul(q[816], f23.y)};
            ntt_pow2_points<2, true, false>(y);
            q[0] = y[0];
            q[272] = bfe_neg(bfe_mul_pow2<24>(y[1]));    // w_8   = 2^120 = -2^24
            q[544] = bfe_mul_pow2<48>(y[2]);             // w_8^2 = 2^48
            q[816] = bfe_neg(bfe_mul_pow2<72>(y[3]));    // w_8^3 = 2^168 = -2^72
        }
        tvm_wave_sync();
        const u64 blk = (k * a.pitch + (j1 << 11)) >> TVM_RB_LOG;   // the row's first 16-row block of the table
        u64* const out_l = a.table + ((((u64)(ln >> TVM_RB_LOG)) * W + (u64)(a.col0 + vl)) << TVM_RB_LOG) + (ln & (TVM_RB - 1));
#pragma unroll 4
        for (int e = 0; e < 16; e++) {   // j2 = lane + 64 e and j2 + 1024: consecutive lanes = consecutive storage rows
            const u64 b = rl[68 * e];
            TVM_STORE_STREAM(&out_l[((blk + 4 * e) * W) << TVM_RB_LOG], bfe_add(ra[e], b));
            TVM_STORE_STREAM(&out_l[((blk + 64 + 4 * e) * W) << TVM_RB_LOG], bfe_sub(ra[e], b));
        }
        tvm_wave_sync();   // (the region has been read)
        if (more) {
            request_next();
#pragma unroll
            for (int e = 0; e < 16; e++) rl[68 * e] = nxt[e];
        }
    }
}

// fb[lane*16 + g4*4 + e] = w_1024^(g brev2(e)) * w_2048^g, g = lane + 64 g4 (k_lde_pass3_halves)
__global__ void k_pass3_halves_table(const u64* __restrict__ tw_2048, u64* __restrict__ fb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 1024) return;
    const int e = i & 3, g4 = (i >> 2) & 3, lane = i >> 4, g = lane + 64 * g4;
    fb[i] = bfe_mul(tw_2048[(2 * g * brev_k(e, 2)) & 2047], tw_2048[g]);
}

// Pass 1 (the inverse transform's column step) in the same form: a tile is 16 adjacent columns i2 of the N1 x N2 view, i.e.
// 16 rows of 1024 points i1 (128-byte runs in memory); wavefront w transforms row w (decimation in frequency, inverse roots),
// and the store applies the inter-pass twiddle w_N^-(i2 * k1) as a running product over k1 = brev(position) -- the positions
// of a work-item are visited in bit-reversed order so that k1 advances by one -- instead of two table loads per element.
template <int LOGN, int ROWS>
__global__ void __launch_bounds__((1 << (LOGN - 4)) * ROWS, 4) k_lde_pass1_rows(Ntt2Args a) {
    // LOGN = 10: one wavefront per row, 16 rows (128-byte runs of the input).  LOGN = 11 (round 5; 2^22 and 2^23-row traces): TWO
    // wavefronts per row of 2048 points, 8 rows (LDS holds no more of them: 64-byte runs), the butterfly groups separated by
    // workgroup barriers (row_ntt_group, LPR = 128) -- instead of the generic k_ntt2_pass1.
    // LOGN = 9 / 8 (round 6; 2^16 .. 2^19-row traces): 32 / 16 lanes per row, two / four rows per wavefront, 16 rows.
    constexpr int n1 = 1 << LOGN, ROWW = TVM_ROW_WORDS(n1), LPR = n1 / 16, NT = LPR * ROWS, RLOG = ROWS == 16 ? 4 : 3;
    static_assert((LOGN >= 8 && LOGN <= 10 && (ROWS == 16 || ROWS == 8)) || (LOGN == 11 && ROWS == 8), "16 .. 128 lanes per row of 256 .. 2048 points");
    TVM_DYN_SMEM(u64, s_all);
    u64* const s = s_all + (LPR > 64 ? 512 : 0);   // (two wavefronts per row: sixteen blocks of 64 pair flags first, tvm_pair_sync)
    const int tid = threadIdx.x;
    const u64 n2 = 1ull << a.log_n2;
    const int vl = blockIdx.y, v = a.col0 + vl;
    const u64 i2_0 = (u64)blockIdx.x * ROWS;
    const int b = tid & (ROWS - 1), q0 = tid >> RLOG;   // this work-item loads / stores row b, positions q0 + LPR * it
    const u64* in = a.in + (u64)(v / a.in_fk) * a.in_col_stride + (v % a.in_fk) + (i2_0 + b) * a.in_fk;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int i1 = q0 + LPR * it;
        s[b * ROWW + TVM_ROW_SKEW(i1)] = TVM_LOAD_STREAM(&in[(u64)i1 * n2 * a.in_fk]);
    }
    u64* tw_lds = s + ROWS * ROWW;
    for (int i = tid; i < n1; i += NT) tw_lds[i] = a.tw1[i];   // all n1 powers of the inverse root
    if constexpr (LPR > 64) {   // two wavefronts per row: the pair meets between the butterfly groups (tvm_pair_sync), not the workgroup
        unsigned* const pair_flags = (unsigned*)s_all;
        pair_flags[tid] = 0;   // (NT = 16 x 64)
        tvm_lds_barrier();
        unsigned pair_epoch = 0;
        const int wave = tvm_uniform(tid >> 6);
        int row_lane = tid % LPR;   // (tvm_pair_sync borrows its register: 64 * (wave % 2) + lane)
        auto pair = [&] { tvm_pair_sync(pair_flags, wave, wave ^ 1, ++pair_epoch, row_lane, (wave & 1) * 64); };
        row_ntt<false, 4, LOGN, 2, 0, LPR>(s + (tid / LPR) * ROWW, tw_lds, row_lane, pair);
    } else {
        tvm_lds_barrier();
        row_ntt<false, 4, LOGN, 2, 0, LPR>(s + (tid / LPR) * ROWW, tw_lds, tid % LPR);
    }
    tvm_lds_barrier();
    // position p = q0 + LPR * brev4(c) holds index k1 = brev(p) = brev(q0) * 16 + c
    const u64 i2 = i2_0 + b;
    const u64 k1_0 = (u64)brev_bits((u32)q0, LOGN - 4) << 4;
    u64 t = pow2_get(a.tw_inter, (i2 * k1_0) & ((n2 << LOGN) - 1));
    const u64 t_step = pow2_get(a.tw_inter, i2);
    u64* tmp = a.tmp + (u64)vl * a.tmp_col_stride + i2;
#pragma unroll 4
    for (int c = 0; c < 16; c++) {
        const int p = q0 + LPR * (int)brev_bits((u32)c, 4);
        TVM_STORE_STREAM(&tmp[(u64)p * n2], bfe_mul(s[b * ROWW + TVM_ROW_SKEW(p)], t));
        t = bfe_mul(t, t_step);
    }
}

// Pass 2 for 1024-point axes, round 5: every wavefront keeps ITS row from the inverse transform to the end of the coset loop.
//   * The inverse rows step leaves position q of row w in the wavefront's own LDS words; lane l takes the 16 consecutive
//     positions 16 l .. 16 l + 15 into VGPRs -- exactly the 16 points of its first forward butterfly group -- and keeps them
//     across the coset loop.  The coset factor gamma_k^(n1 m1) is a table value at the POSITION (g_hi_pos, 8 KB per coset, four
//     16-byte loads per lane) and the scaled coefficients go straight into the group's butterflies: the scale phase has no LDS
//     round trip and no barrier (the pass-2 kernel of rounds 3-4, k_lde_pass2_rows, wrote the scaled tile position-major and re-read
//     it row-major between two barriers), and the randomizer term touches lane 0's first coefficient only (h <= n1: m1 = 0).
//   * The inter-pass twiddle w_N^(m2 j1) gamma_k^m2 / N is folded into the LAST butterfly group (radix 4 over positions
//     j1 = g + 256 e, g = lane + 64 it): w_N^(m2 g) is common to the four outputs of a butterfly, so it rides on the group's own
//     twiddle step (f_tw: one table value per input, four multiplications where the step alone has three), and what is left,
//     gamma_k^m2 / N * w_N^(256 m2 e), is uniform over the wavefront (u_tw: scalar loads).  2 multiplications per element for
//     twiddle step + inter-pass twiddle where the running product of k_lde_pass2_rows' store phase paid 0.75 + 2.
//   Measured against k_lde_pass2_rows on one box (profiles/r05_b_*, 96 columns at 2^20 rows): 8 % fewer VALU instructions, 29 % fewer
//   LDS instructions, 36 instead of 200 bytes of scratch, WRITE_SIZE 66.5 instead of 77.9 B per cell, 4.72 against 4.78 ms on average
//   (minimum 4.27 against 4.43): the four wavefronts per SIMD do not hide the table loads at their uses, and every variant that
//   requests them earlier pays more in scratch than it gains (profiles/r05_d_*: 160 B -> 5.3 ms, 288 B -> 6.0 ms).
//   * The store phase is what is left of the transposition: a copy of the tile, row-major in LDS (written by the rows'
//     wavefronts) to 64-byte runs of 8 adjacent rows in Z.  Two workgroup barriers per coset (rows complete / tile read), none
//     inside a row's work.  Row pitch = 8 (mod 32) words: the store phase's 8 rows x 8 positions per wavefront fall into 64
//     different banks (the odd pitch of TVM_ROW_WORDS put b + j1 = const into one).
// 8 rows, 512 work-items, 78 KB of LDS: two workgroups per CU.
#define TVM_P2F_ROWW(logn) ((logn) == 8 ? 296 : (logn) == 9 ? 552 : (logn) == 10 ? 1096 : 2184)   // >= TVM_ROW_WORDS, = 8 (mod 32)
#define TVM_P2F_TW2_WORDS 272   // 16 x 17: the middle group's twiddles
#define TVM_P2F_LDS_WORDS(logn) (8 * TVM_P2F_ROWW(logn) + TVM_P2F_TW2_WORDS + TVM_ROW_WORDS(1 << (logn)) + 8)
#define TVM_P2F_FLAG_WORDS 512   // u64 words in front of the tile at 2048 points: sixteen blocks of 64 pair flags (tvm_pair_sync)
#define TVM_P2F_LDS_BYTES(logn) ((size_t)(TVM_P2F_LDS_WORDS(logn) + ((logn) == 11 ? TVM_P2F_FLAG_WORDS : 0)) * sizeof(u64))   // tile + twiddles + one coset's factors + a randomizer word per row: 79.2 / 159.4 KB
#ifndef TVM_P2F_FT_EARLY_11
#define TVM_P2F_FT_EARLY_11 0   // (2048-point rows: the radix-8 last group leaves no registers for them)
#endif
#ifndef TVM_P2F_FT_EARLY
#define TVM_P2F_FT_EARLY 2   // 16-byte loads of the last group's factors requested BEFORE the middle group (4 registers each)
#endif
#ifndef TVM_P2F_TWB
#define TVM_P2F_TWB 4   // twiddle loads in flight per batch in the middle group of the coset loop (registers)
#endif
// LOGN = 10: one wavefront per row, 512 work-items, 79 KB of LDS -- two workgroups per CU.  LOGN = 11 (round 5, the shape of 2^21 and
// 2^22-row traces -- BASELINE configs[2]'s height): TWO wavefronts per row of 2048 points, 16 positions per lane as before, the butterfly
// groups separated by workgroup barriers, the last group radix 8; 1024 work-items, 159 KB: one workgroup per CU (as the tile kernel
// k_lde_pass2_v3<11, 10> it replaces, which kept the position-major tile, three multiplications outside the butterflies and 192 B of
// scratch).
// LOGN = 9 / 8 (round 6: the 512- / 256-point axes of 2^16 .. 2^19-row traces, on which pass 2 ran the generic tile kernel k_lde_pass2 --
// 344 B of scratch, 2.4 times the time per cell -- or, at 256 points, the 64-work-item stand-in k_lde_pass2_v3<8, 6>): 32 / 16 lanes
// per row, TWO / FOUR rows per wavefront (everything between two butterfly groups of a row stays inside its wavefront, as at 1024
// points), the last group radix 2 / nothing but the inter-pass factors; 256 / 128 work-items, 42 / 23 KB of LDS.
template <int LOGN>
// (below 1024 points the LDS leaves room for three wavefronts per SIMD: the register budget is set for three, and nothing spills)
__global__ void __launch_bounds__(8 << (LOGN - 4), LOGN >= 10 ? 4 : 3) k_lde_pass2_fused(LdePass2Args a) {
    constexpr int n2 = 1 << LOGN, ROWS = 8, LPR = n2 / 16, WPR = LPR / 64, NT = ROWS * LPR, RLOG = 3;
    constexpr int ROWW = TVM_P2F_ROWW(LOGN), K3 = LOGN - 8, R3 = 1 << K3, ITS = 16 / R3, LSTEP = LPR + LPR / 16;
    constexpr int FE = LOGN == 11 ? TVM_P2F_FT_EARLY_11 : TVM_P2F_FT_EARLY;   // 16-byte loads of the last group's factors requested early
    static_assert(LOGN >= 8 && LOGN <= 11 && ROWW >= TVM_ROW_WORDS(n2) && ROWW % 32 == 8, "256 .. 2048 points, bank-spread row pitch");
    TVM_DYN_SMEM(u64, s_all);
    // (two wavefronts per row: the pair flags of tvm_pair_sync come FIRST -- they must lie in the first 64 KB of LDS)
    u64* const s = s_all + (WPR > 1 ? TVM_P2F_FLAG_WORDS : 0);
    const int tid = threadIdx.x;
    // this lane's row of the tile and its number within the row's lanes (a row per wavefront, per pair of wavefronts, or -- below
    // 1024 points -- per 32 / 16 lanes; with whole wavefronts per row the row number is uniform, and told so)
    int r;
    if constexpr (WPR >= 1) r = tvm_uniform(tid >> 6) / WPR;
    else r = tid / LPR;
    int rl = tid % LPR;   // (not const: with two wavefronts per row tvm_pair_sync borrows its register and puts the value back)
    const u64 n1 = 1ull << a.log_n1, n = n1 << LOGN;
    const int vl = blockIdx.y, v = a.col0 + vl;
    const u64 p0 = (u64)blockIdx.x * ROWS, p = p0 + (u64)r;
    // (grid x = tile, y = column.  With the columns side by side instead -- so that the workgroups running together share their rows'
    // factor table in L2: it streams through once per coset and column, 76 B per cell fetched at 2^22 rows where 8 are due,
    // profiles/r05_j_pmc_lde_hash_2p22.txt -- the kernel takes the same time at 2^22 rows (20.75 against 20.92 ms) and 1 % longer at
    // 2^20: the Infinity Cache serves the table, the fetches were never what the kernel waited for; profiles/r05_k_*.)
    u64* const row = s + r * ROWW;
    // between two butterfly groups of a row: its lanes exchange data through the row's LDS words.  One wavefront (or less) per row:
    // a wait on the wavefront's own LDS traffic; two wavefronts per row (2048 points): the PAIR meets (tvm_pair_sync, round 6 -- a
    // workgroup barrier here stopped all sixteen wavefronts of the CU's one workgroup five times per coset)
    unsigned* const pair_flags = (unsigned*)s_all;   // a block of 64 words per wavefront
    unsigned pair_epoch = 0;   // (uniform, like the wavefront's number: scalar registers)
    const int wave = tvm_uniform(tid >> 6);
    auto row_sync = [&] {
        if constexpr (WPR <= 1) tvm_wave_sync();
        else tvm_pair_sync(pair_flags, wave, wave ^ 1, pair_epoch = (unsigned)tvm_uniform((int)pair_epoch + 1), rl, (wave & 1) * 64);
    };
    if constexpr (WPR > 1) pair_flags[tid] = 0;   // (NT = 16 x 64 words; visible after the workgroup barrier below, before the first meeting)
    // behind the tile: the 15 x 16 twiddles the middle butterfly group uses, tw2[17 j0 + e] = w_n2^((j0 brev4(e)) << (LOGN - 8)) (pitch
    // 17: sixteen j0 in sixteen banks), and the coset factors of the current coset at their positions, skewed like a row
    u64* const tw2 = s + ROWS * ROWW;
    u64* const ghl = tw2 + TVM_P2F_TW2_WORDS;
    {
        const u64* y = a.y + (u64)vl * n + p * n2 + rl;
        u64* const rowl = row + TVM_ROW_SKEW(rl);
#pragma unroll
        for (int e = 0; e < 16; e++) rowl[LSTEP * e] = TVM_LOAD_STREAM(&y[LPR * e]);   // position rl + LPR e
    }
    for (int i = tid; i < 256; i += NT) tw2[17 * (i >> 4) + (i & 15)] = a.tw_b1[((i >> 4) * brev_bits((u32)(i & 15), 4)) << (LOGN - 8)];
#pragma unroll
    for (int hh = 0; hh < n2 / NT; hh++) ghl[TVM_ROW_SKEW(tid + hh * NT)] = a.g_hi_pos[tid + hh * NT];
    if constexpr (WPR <= 1) tvm_wave_sync();
    else tvm_lds_barrier();   // (the row is loaded, the pair flags are zero)
    // inverse rows step: position q of the row then holds N * t[m1*n1 + m2], m1 = brev(q), m2 = brev(p)
    if (a.mode != TVM_LDE_FORWARD_ONLY) row_ntt<false, 4, LOGN, 2, 0, LPR>(row, a.tw_a2, rl, row_sync);
    if (a.mode == TVM_LDE_INVERSE_ONLY) {
        u64* yw = const_cast<u64*>(a.y) + (u64)vl * n + p * n2 + rl;
        const u64* const rowl = row + TVM_ROW_SKEW(rl);
#pragma unroll
        for (int e = 0; e < 16; e++) yw[LPR * e] = rowl[LSTEP * e];
        return;
    }
    u64 coef[16];
#pragma unroll
    for (int e = 0; e < 16; e++) coef[e] = row[17 * rl + e];   // TVM_ROW_SKEW(16 * rl + e) = 17 * rl + e
    const u64 m2 = brev_bits((u32)p, a.log_n1);   // uniform over the wavefront
    // row-lane 0: the randomizer coefficient that meets coefficient m = m2 (m1 = 0: position 0) -- parked in an LDS word of the row's
    // own, re-read in every coset (a global load at its use stalled the wavefront, and with it the workgroup's barrier, once per
    // coset: every 8-row tile has a row with m2 < 128)
    const bool has_rnd = m2 < a.h;
    u64* const r0 = ghl + TVM_ROW_WORDS(n2) + r;
    if (rl == 0) *r0 = has_rnd ? a.rnd[((u64)(v / a.fk) * a.h + m2) * a.fk + (v % a.fk)] : 0;
    tvm_lds_barrier();   // the twiddle tables and the first coset's factors are staged
    for (int k = 0; k < a.n_cosets; k++) {
        // (lane and work-item number through opaque moves: the addresses below are cheap to form and expensive to keep -- hoisted out
        // of the coset loop they went to scratch)
        const int ln = tvm_opaque(rl);
        u64 x[16];
        {
            const u64* const gh = ghl + 17 * ln;   // the coset's factors at positions 16 rl + e (staged by the workgroup, below)
#pragma unroll
            for (int e = 0; e < 16; e++) x[e] = bfe_mul(coef[e], gh[e]);
        }
        if (has_rnd && ln == 0) x[0] = bfe_add(coef[0], bfe_mul(a.zk[k], *r0));   // gamma_k^0 = 1 at position 0
        ntt_pow2_points<4, true, false>(x);
        {
            u64* const out = row + 17 * ln;
#pragma unroll
            for (int e = 0; e < 16; e++) out[e] = x[e];
        }
        const u64x2* ft = (const u64x2*)(a.f_tw + (((p * LPR) + ln) << 4));
        u64x2 f[8];
#pragma unroll
        for (int j = 0; j < FE; j++) f[j] = ft[j];   // the last group's first factors: in flight under the middle group
        row_sync();
        {   // layers 4 .. 7 (row_ntt_group<true, 4, 4>: one group per row-lane), the twiddle step from the compact table
            const int j0 = ln & 15;
            u64* const q = row + TVM_ROW_SKEW(((ln >> 4) << 8) | j0);   // positions base + 16 e: TVM_ROW_SKEW adds 17 e
            const u64* const t2 = tw2 + 17 * j0;
            u64 x2[16];
#pragma unroll
            for (int e = 0; e < 16; e++) x2[e] = q[17 * e];
#pragma unroll
            for (int e = 1; e < 16; e++) {
                x2[e] = bfe_mul(x2[e], t2[e]);
                if (e % TVM_P2F_TWB == 0) asm volatile("" ::: "memory");
            }
            ntt_pow2_points<4, true, false>(x2);
#pragma unroll
            for (int e = 0; e < 16; e++) q[17 * e] = x2[e];
        }
        row_sync();
        u64 u[R3];
#pragma unroll
        for (int e = 0; e < R3; e++) u[e] = a.u_tw[((u64)k * n1 + p) * R3 + e];
#pragma unroll
        for (int j = FE; j < 8; j++) f[j] = ft[j];
#pragma unroll
        for (int it = 0; it < ITS; it++) {   // layers 8 .. LOGN - 1 on positions g + 256 e, g = rl + LPR it, with the inter-pass twiddle
            u64* const q = row + TVM_ROW_SKEW(ln + LPR * it);   // TVM_ROW_SKEW(g + 256 e) = TVM_ROW_SKEW(g) + 272 e
            u64 y8[R3];
#pragma unroll
            for (int e = 0; e < R3; e++) {
                const int j = it * R3 + e;   // (a constant after unrolling: the factor's place in the eight 16-byte loads)
                y8[e] = bfe_mul(q[272 * e], (j & 1) ? f[j >> 1].y : f[j >> 1].x);
            }
            if constexpr (K3 > 0) ntt_pow2_points<K3, true, false>(y8);
#pragma unroll
            for (int e = 0; e < R3; e++) q[272 * e] = bfe_mul(y8[e], u[e]);
        }
        // (timing experiments of round 5 with deliberately wrong results -- no store phase: -11 %, the same words as contiguous 64 KB
        // blocks: -0 %, no workgroup barriers: -3 %; profiles/r05_c_* -- are in the history, commit a103ba1, not in this source)
        tvm_lds_barrier();   // the rows are complete: the store phase reads across them
        {
            const int t2 = tvm_opaque(tid), b_out = t2 & (ROWS - 1), j1_0 = t2 >> RLOG;
            // the next coset's factors: requested here, parked in LDS at the end of the store phase (every wavefront has read this
            // coset's before the barrier above) -- their latency hides behind the store phase and costs two registers
            const bool more = k + 1 < a.n_cosets;
            u64 g_next[n2 / NT];
#pragma unroll
            for (int hh = 0; hh < n2 / NT; hh++) g_next[hh] = more ? a.g_hi_pos[(u64)(k + 1) * n2 + t2 + hh * NT] : 0;
            // a lane copies TWO adjacent rows of a position with one 16-byte store: 8 store instructions per lane and coset instead of
            // 16 for the same 64-byte runs (-3 % of the kernel, profiles/r05_k_*)
            (void)b_out, (void)j1_0;
            const int bp = t2 & (ROWS / 2 - 1), j1_p = t2 >> (RLOG - 1);
            u64* zk = a.z + ((u64)vl * a.n_cosets + k) * n + p0 + 2 * bp;
            const u64* src = s + 2 * bp * ROWW;
#pragma unroll 4
            for (int i = 0; i < 8; i++) {
                const int j1 = j1_p + i * (NT >> (RLOG - 1));
                TVM_STORE_STREAM_X2(&zk[(u64)j1 * n1], src[TVM_ROW_SKEW(j1)], src[ROWW + TVM_ROW_SKEW(j1)]);
            }
            if (more) {
#pragma unroll
                for (int hh = 0; hh < n2 / NT; hh++) ghl[TVM_ROW_SKEW(t2 + hh * NT)] = g_next[hh];
            }
        }
        tvm_lds_barrier();   // the tile has been read: the next coset's first group overwrites the rows
    }
}

// the tables of k_lde_pass2_fused (LdePass2Args), for rows of n2 = 1024 or 2048 points: LPR = n2 / 16 lanes per row, the last
// butterfly group of radix R3 = n2 / 256 in 16 / R3 iterations.  hi_pos[k][q] = hi[k][brev(q)];
// f[((p*LPR + rl)*16 + it*R3 + e] = tw_n2[g * brev(e)] * w_N^(brev(p) * g), g = rl + LPR it;
// u[(k*n1 + p)*R3 + e] = lo[k][brev(p)] * w_N^(brev(p) * 256 e)
__global__ void k_pass2_fused_tables(const u64* __restrict__ lo, const u64* __restrict__ hi, Pow2 tw_inter, const u64* __restrict__ tw_n2,
                                     int log_n1, int log_n2, u64 X, u64* __restrict__ hi_pos, u64* __restrict__ f, u64* __restrict__ u) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n1 = 1ull << log_n1, n2 = 1ull << log_n2, n = n1 * n2;
    const int k3 = log_n2 - 8;
    const u64 r3 = 1ull << k3, lpr = n2 >> 4;
    if (i < X * n2) hi_pos[i] = hi[(i / n2) * n2 + brev_bits((u32)(i % n2), log_n2)];
    if (f && i < n) {
        const u64 e = i & (r3 - 1), it = (i >> k3) & ((16 >> k3) - 1), rl = (i >> 4) & (lpr - 1), p = i >> log_n2, g = rl + lpr * it;
        const u64 m2 = brev_bits((u32)p, log_n1);
        f[i] = bfe_mul(tw_n2[(g * (u64)brev_bits((u32)e, k3)) & (n2 - 1)], pow2_get(tw_inter, (m2 * g) & (n - 1)));
    }
    if (i < X * n1 * r3) {
        const u64 e = i & (r3 - 1), p = (i >> k3) % n1, k = (i >> k3) / n1;
        const u64 m2 = brev_bits((u32)p, log_n1);
        u[i] = bfe_mul(lo[k * n1 + m2], pow2_get(tw_inter, (m2 * 256 * e) & (n - 1)));
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// hipMalloc allocates on the calling thread's CURRENT device, which another context (or the application) may have
// changed since tvm_ctx_create: every allocating path re-selects the context's device first.
static bool bind_device(tvm_ctx* c) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == c->device) return true;
    return hipSetDevice(c->device) == hipSuccess;
}

const u64* pow_table(tvm_ctx* c, u64 base, u64 count, u64 scale) {
    auto key = std::make_tuple(base, count, scale);
    auto it = c->tables.find(key);
    if (it != c->tables.end()) return it->second;
    u64* d = nullptr;
    if (!bind_device(c)) return nullptr;
    if (hipMalloc((void**)&d, (count ? count : 1) * sizeof(u64)) != hipSuccess) return nullptr;
    const int bs = 256;
    TVM_LAUNCH(k_pow_table, dim3((unsigned)((count + bs - 1) / bs)), dim3(bs), 0, c->stream, base, count, scale, d);
    c->tables[key] = d;
    return d;
}

void* scratch(tvm_ctx* c, int slot, size_t bytes) {
    if ((size_t)slot >= c->scratch.size()) {
        c->scratch.resize(slot + 1, nullptr);
        c->scratch_bytes.resize(slot + 1, 0);
    }
    if (c->scratch_bytes[slot] < bytes) {
        if (c->scratch[slot]) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipFree(c->scratch[slot]);
            c->scratch[slot] = nullptr;
            c->scratch_bytes[slot] = 0;
        }
        void* p = nullptr;
        if (!bind_device(c) || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        c->scratch[slot] = p;
        c->scratch_bytes[slot] = bytes;
    }
    return c->scratch[slot];
}

static size_t pool_round(size_t bytes) {
    const size_t g = bytes < (1u << 20) ? 256 : (2u << 20);
    return (bytes + g - 1) / g * g;
}
void* pool_alloc(tvm_ctx* c, size_t bytes) {
    const size_t want = pool_round(bytes ? bytes : 1);
    auto it = c->pool_free.lower_bound(want);
    if (it != c->pool_free.end() && it->first <= want + want / 4) {  // at most 25 % slack
        void* p = it->second;
        c->pool_live[p] = it->first;
        c->pool_free.erase(it);
        return p;
    }
    void* p = nullptr;
    if (!bind_device(c)) return nullptr;
    if (c->pool_limit && c->pool_bytes + want > c->pool_limit) {
        pool_trim(c);  // cached blocks count against the limit: give them back first
        if (c->pool_bytes + want > c->pool_limit) return nullptr;
    }
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        pool_trim(c);
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    c->pool_live[p] = want;
    c->pool_bytes += want;
    return p;
}
void pool_release(tvm_ctx* c, void* p) {
    if (!p) return;
    auto it = c->pool_live.find(p);
    if (it == c->pool_live.end()) {  // not ours (should not happen): hand it to the driver
        (void)hipStreamSynchronize(c->stream);
        (void)hipFree(p);
        return;
    }
    c->pool_free.emplace(it->second, p);
    c->pool_live.erase(it);
}
void pool_trim(tvm_ctx* c) {
    if (c->pool_free.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool_free) {
        (void)hipFree(kv.second);
        c->pool_bytes -= kv.first;
    }
    c->pool_free.clear();
}

// what a pool_alloc could still obtain: the device's free memory plus this context's cached blocks, capped by the limit
int h2d_small(tvm_ctx* c, void* d, const void* h, size_t bytes) {
    if (!bytes) return TVM_OK;
    constexpr size_t RING = (size_t)4 << 20;
    if (!c->pin && !c->pin_unavailable) {
        void* p = nullptr;
        if (bind_device(c) && hipHostMalloc(&p, RING, 0) == hipSuccess) {
            c->pin = (char*)p;
            c->pin_bytes = RING;
        } else {
            (void)hipGetLastError();
            c->pin_unavailable = true;
        }
    }
    if (!c->pin || bytes > c->pin_bytes / 4) {
        TVM_HIP_CHECK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
        TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        return TVM_OK;
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (c->pin_head + need > c->pin_bytes) {   // wrap: every copy out of the ring so far has been issued on this stream
        TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        c->pin_head = 0;
    }
    char* slot = c->pin + c->pin_head;
    c->pin_head += need;
    std::memcpy(slot, h, bytes);
    TVM_HIP_CHECK(c, hipMemcpyAsync(d, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return TVM_OK;
}
size_t pool_available(tvm_ctx* c, size_t* device_total) {
    size_t free_b = 0, total_b = 0;
    if (!bind_device(c) || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    size_t cached = 0;
    for (const auto& kv : c->pool_free) cached += kv.first;
    size_t avail = free_b + cached;
    if (c->pool_limit) {
        const size_t live = c->pool_bytes - cached;
        avail = live >= c->pool_limit ? 0 : (avail < c->pool_limit - live ? avail : c->pool_limit - live);
    }
    if (device_total) *device_total = total_b;
    return avail;
}

int set_error(tvm_ctx* c, int code, const char* what) {
    if (c) c->last_error = what;
    return code;
}

static int threads_for_tile(int tile) {
    int t = tile / 16;
    t = (t + 63) / 64 * 64;
    if (t < 64) t = 64;
    if (t > 1024) t = 1024;
    return t;
}
// Tile = 2^log_axis points x 2^batch transforms, at most 2^14 words = 128 KiB of LDS (one 1024-thread workgroup per
// CU).  Measured alternative: 64 KiB tiles with two 512-thread workgroups per CU, so that one workgroup's global
// traffic overlaps the other's butterflies -- no gain for the generic passes in round 1 (24.9 vs 25.5 ms for 128
// columns), but worth 4 % for the LDE's pass 3 once its arithmetic had been trimmed (lde_table: k_lde_pass3_v3<10, 9>).
static int tile_words_log() { return 14; }
static int batch_log_for(int log_axis) {
    int b = tile_words_log() - log_axis;
    if (b < 0) b = 0;
    return b > 4 ? 4 : b;
}

static bool g_attr_done = false;
static void set_lds_attributes() {
    if (g_attr_done) return;
    g_attr_done = true;
    const int max_lds = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)k_ntt2_pass1, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_ntt2_pass2, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass1_rows<8, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass1_rows<9, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass1_rows<10, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass1_rows<11, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_fused<8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_fused<9>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_fused<10>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_fused<11>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_rows<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_rows<9, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_rows<10, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_rows<11, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_halves<8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_v3<11, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass2_v3<12, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    (void)hipFuncSetAttribute((const void*)k_lde_pass3_v3<12, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
}

// lo[k][i] = scale * gamma_k^i (i < n1), hi[k][i] = gamma_k^(n1*i) (i < n2), gamma_k = offset * gen^k
__global__ void k_coset_tables(u64 offset, u64 gen, u64 X, u64 n1, u64 n2, u64 scale, u64* lo, u64* hi) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= X * (n1 + n2)) return;
    const u64 k = e / (n1 + n2), i = e % (n1 + n2);
    const u64 gamma = bfe_mul(offset, bfe_pow(gen, k));
    if (i < n1) lo[k * n1 + i] = bfe_mul(scale, bfe_pow(gamma, i));
    else hi[k * n2 + (i - n1)] = bfe_pow(bfe_pow(gamma, n1), i - n1);
}
static int coset_tables(tvm_ctx* c, u64 offset, u64 gen, u64 X, u64 n1, u64 n2, u64 scale, const u64** lo, const u64** hi) {
    auto key = std::make_tuple(offset ^ 0xC05E7C05E7ull, gen, (X << 56) | (n1 << 28) | n2);
    auto it = c->tables.find(key);
    u64* d = nullptr;
    if (it != c->tables.end()) {
        d = it->second;
    } else {
        if (hipMalloc((void**)&d, X * (n1 + n2) * sizeof(u64)) != hipSuccess)
            return set_error(c, TVM_ERR_OUT_OF_MEMORY, "coset tables");
        const u64 total = X * (n1 + n2);
        TVM_LAUNCH(k_coset_tables, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, offset, gen, X, n1, n2,
                   scale, d, d + X * n1);
        c->tables[key] = d;
    }
    *lo = d;
    *hi = d + X * n1;
    return TVM_OK;
}

// the three tables of k_lde_pass2_fused, cached per context: f depends on the trace domain alone, hi_pos and u on the cosets too
static int pass2_fused_tables(tvm_ctx* c, u64 trace_gen, u64 offset, u64 gen, u64 X, int log_n1, int log_n2, const u64* lo, const u64* hi,
                              const Pow2& tw_inter, const u64* tw_n2, const u64** hi_pos, const u64** f, const u64** u) {
    const u64 n1 = 1ull << log_n1, n2 = 1ull << log_n2, n = n1 * n2;
    const auto key_f = std::make_tuple(trace_gen ^ 0xF05EDF05EDull, n1, n2);
    const auto key_k = std::make_tuple(offset ^ 0xF05EDC05E7ull, gen, (X << 56) | (n1 << 28) | n2);
    auto it_f = c->tables.find(key_f);
    auto it_k = c->tables.find(key_k);
    u64 *d_f = it_f != c->tables.end() ? it_f->second : nullptr, *d_k = it_k != c->tables.end() ? it_k->second : nullptr;
    const bool new_f = !d_f, new_k = !d_k;
    if (!bind_device(c)) return set_error(c, TVM_ERR_DEVICE, "bind device");
    if (new_f && hipMalloc((void**)&d_f, n * sizeof(u64)) != hipSuccess) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "pass-2 twiddle table");
    const u64 r3 = n2 >> 8;   // radix of the last butterfly group
    if (new_k && hipMalloc((void**)&d_k, X * (n2 + r3 * n1) * sizeof(u64)) != hipSuccess) {
        if (new_f) (void)hipFree(d_f);
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "pass-2 coset tables");
    }
    if (new_f || new_k) {
        const u64 total = new_f ? (n > X * (n2 + r3 * n1) ? n : X * (n2 + r3 * n1)) : X * (n2 > r3 * n1 ? n2 : r3 * n1);
        // (entries that exist already are simply written again with the same values when only one of the two is new)
        TVM_LAUNCH(k_pass2_fused_tables, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, lo, hi, tw_inter, tw_n2, log_n1,
                   log_n2, X, d_k, new_f ? d_f : (u64*)nullptr, d_k + X * n2);
        if (new_f) c->tables[key_f] = d_f;
        if (new_k) c->tables[key_k] = d_k;
    }
    *f = d_f;
    *hi_pos = d_k;
    *u = d_k + X * n2;
    return TVM_OK;
}

struct Split {
    int log_n, log_n1, log_n2, shift;
};
static Split split_for(u64 n) {
    Split s;
    s.log_n = ilog2(n);
    s.log_n1 = s.log_n / 2;
    s.log_n2 = s.log_n - s.log_n1;
    s.shift = (s.log_n + 1) / 2;
    return s;
}
// the layout lde_table writes (context.h): X cosets of n2 blocks of n1 rows, plus one successor block per coset
TabLayout lde_table_layout(u64 n_rows, u64 L) {
    const Split sp = split_for(n_rows);
    TabLayout l;
    l.X = L / n_rows;
    l.log_x = ilog2(l.X);
    l.n1 = 1ull << sp.log_n1;
    l.n2 = 1ull << sp.log_n2;
    l.log_n1 = sp.log_n1;
    l.log_n2 = sp.log_n2;
    l.pitch = (l.n2 + 1) * l.n1;
    return l;
}
static int make_inter(tvm_ctx* c, u64 w, const Split& sp, Pow2* out) {
    out->shift = sp.shift;
    out->lo = pow_table(c, w, 1ull << sp.shift);
    out->hi = pow_table(c, bfe_pow(w, 1ull << sp.shift), 1ull << (sp.log_n - sp.shift));
    return (out->lo && out->hi) ? TVM_OK : TVM_ERR_OUT_OF_MEMORY;
}

// Is w (w^n = 1, n >= 2 a power of two) the domains' own n-th root of unity [twenty-first primitive_root_of_unity:
// 1753635133440165772^(2^32 / n)] (-> 1), its inverse (-> 2), or some other root (-> 0)?  Decided on its power of order
// min(n, 16), which must be 2^156 = w_16 (2^36 for the inverse) or the corresponding root of lower order (ntt_shift.h).
static int classify_root(u64 w, u64 n) {
    const int k = ilog2(n) < 4 ? ilog2(n) : 4;
    const u64 r = bfe_pow(w, n >> k), two = bfe_from_u64(2);
    if (r == bfe_pow(two, (u64)((156 << (4 - k)) % 192))) return 1;
    if (r == bfe_pow(two, (u64)((36 << (4 - k)) % 192))) return 2;
    return 0;
}

// Transform `ncols` columns of length n with root `w` (w^n = 1; pass the inverse root for an
// inverse transform).  in_scale/out_scale: optional x[i] *= in_scale^i and X[k] *= out_mult*out_scale^k.
int ntt_columns(tvm_ctx* c, const u64* in, u64 in_len, int in_fk, u64 in_col_stride, u64* out, int out_fk,
                u64 out_col_stride, u64 out_mul, u64 out_add, int ncols, u64 n, u64 w, u64 in_scale, u64 out_scale,
                u64 out_mult) {
    if (!is_pow2(n) || n < 2) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "ntt length must be a power of two >= 2");
    if (n > (1ull << 26)) return set_error(c, TVM_ERR_UNSUPPORTED, "ntt length above 2^26");  // tiles: 2^13 points x 2
    set_lds_attributes();
    Split sp = split_for(n);
    const u64 n1 = 1ull << sp.log_n1, n2 = 1ull << sp.log_n2;
    Ntt2Args a;
    a.in = in;
    a.out = out;
    a.tmp = (u64*)scratch(c, 0, (size_t)ncols * n * sizeof(u64));
    if (!a.tmp) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "ntt scratch");
    a.log_n1 = sp.log_n1;
    a.log_n2 = sp.log_n2;
    a.in_len = in_len;
    a.in_fk = in_fk;
    a.out_fk = out_fk;
    a.in_col_stride = in_col_stride;
    a.tmp_col_stride = n;
    a.out_col_stride = out_col_stride;
    a.tw1 = pow_table(c, bfe_pow(w, n2), n1);
    a.tw2 = pow_table(c, bfe_pow(w, n1), n2);
    TVM_TRY(make_inter(c, w, sp, &a.tw_inter));
    a.pre_hi = a.pre_lo = a.post_lo = a.post_hi = nullptr;
    if (in_scale != TVM_ONE) {
        a.pre_lo = pow_table(c, in_scale, n2);
        a.pre_hi = pow_table(c, bfe_pow(in_scale, n2), n1);
        if (!a.pre_lo || !a.pre_hi) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "tables");
    }
    if (out_scale != TVM_ONE || out_mult != TVM_ONE) {
        a.post_lo = pow_table(c, out_scale, n1, out_mult);
        a.post_hi = pow_table(c, bfe_pow(out_scale, n1), n2);
        if (!a.post_lo || !a.post_hi) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "tables");
    }
    if (!a.tw1 || !a.tw2) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "tables");
    a.out_mul = out_mul;
    a.out_add = out_add;
    a.col0 = 0;
    a.root = classify_root(w, n);
    {
        a.batch_log = batch_log_for(sp.log_n1);
        const int B = 1 << a.batch_log;
        const int tile = (int)n1 << a.batch_log;
        dim3 grid((unsigned)((n2 + B - 1) / B), (unsigned)ncols);
        TVM_LAUNCH(k_ntt2_pass1, grid, dim3(threads_for_tile(tile)), (size_t)tile * sizeof(u64), c->stream, a);
    }
    {
        a.batch_log = batch_log_for(sp.log_n2);
        const int B = 1 << a.batch_log;
        const int tile = (int)n2 << a.batch_log;
        dim3 grid((unsigned)((n1 + B - 1) / B), (unsigned)ncols);
        const size_t lds = (size_t)B * (n2 + TVM_ROW_PAD) * sizeof(u64);
        TVM_LAUNCH(k_ntt2_pass2, grid, dim3(threads_for_tile(tile)), lds, c->stream, a);
    }
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// master_table.rs:258-322.  trace: column-major [n_cols][n_rows][fk]; rnd: [n_cols][h][fk];
// table: row-block-major over the storage rows of lde_table_layout(n_rows, L), n_cols*fk words per row (context.h); the
// successor blocks are the caller's (fill_successor_blocks).
// split (optional): the pass split at the coefficients for a range of virtual columns (LdeSplit, kernels.h) -- mode
// TVM_LDE_INVERSE_ONLY writes the coefficient form of the range (n_rows words per virtual column) to split->coeffs and ignores
// the evaluation domain (pass L = n_rows); TVM_LDE_FORWARD_ONLY reads it from there and writes the range's columns of the table.
int lde_table(tvm_ctx* c, int fk, const u64* trace, u64 n_rows, u64 n_cols, const u64* rnd, u64 h, u64 trace_gen,
              u64 eval_offset, u64 eval_gen, u64 L, u64* table, int chunk_cols, const LdeSplit* split) {
    if (!is_pow2(n_rows) || !is_pow2(L) || n_rows < 2 || L < n_rows || (fk != 1 && fk != 3))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "lde: lengths must be powers of two, L >= n_rows >= 2");
    const u64 X = L / n_rows;
    if (X > TVM_LDE_MAX_COSETS) return set_error(c, TVM_ERR_UNSUPPORTED, "lde: expansion above 32");
    if (h > n_rows) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "lde: more randomizers than rows");
    if (n_rows > (1ull << 24)) return set_error(c, TVM_ERR_UNSUPPORTED, "lde: trace above 2^24 rows");
    if (bfe_pow(eval_gen, X) != trace_gen)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "lde: evaluation generator ^ (L/N) must equal the trace generator");
    set_lds_attributes();
    const u64 N = n_rows;
    const Split sp = split_for(N);
    const u64 n1 = 1ull << sp.log_n1, n2 = 1ull << sp.log_n2;
    const int W = (int)(n_cols * fk);
    const int mode = split ? split->mode : 0;
    const int first = split ? split->first_vcol : 0, last = split ? split->first_vcol + split->n_vcols : W;
    if (split && (first < 0 || split->n_vcols < 0 || last > W || !split->coeffs || (mode != TVM_LDE_INVERSE_ONLY && mode != TVM_LDE_FORWARD_ONLY)))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "lde: column range / mode of the split");
    // columns per chunk: 96 while the chunk's intermediates (96 * (1 + X) * N words, from the pool: they count against the
    // context's memory limit) stay below 32 GiB AND below a third of what the context can still obtain, else 32.  Measured at 2^20
    // rows (main table, with 8 row tiles per pass-3 workgroup): 16 -> 48.0 ms, 32 -> 47.0, 96 -> 45.6, 192 -> 45.5, 379 -> 45.1; at 2^22
    // rows (intermediate 25.8 GB; round 4, whole proof): 32 -> 836.9 ms, 64 -> 821.8, 96 -> 820.6; at 2^21 rows 398.1 -> 392.0 --
    // 2 % that are not worth the coset-wise fallback on a device (or under a limit) that 25.8 GB push over the edge.
    // c->lde_chunk_columns (TVM_OPTION_LDE_CHUNK_COLUMNS) overrides.
    const auto intermediates = [&](int cols) { return (size_t)cols * (1 + X) * n_rows * sizeof(u64); };
    if (chunk_cols <= 0) chunk_cols = c->lde_chunk_columns;
    // (short traces: every column in ONE chunk while its intermediates stay below 512 MB -- traces of up to 2^14 rows: three launches per
    // table instead of twelve, and grids four times as wide on a chip they do not fill)
    if (chunk_cols <= 0 && !split && intermediates(W) <= ((size_t)512 << 20)) chunk_cols = W > 96 ? W : 96;
    if (chunk_cols <= 0) chunk_cols = (intermediates(96) <= ((size_t)36 << 30) && intermediates(96) <= pool_available(c, nullptr) / 3) ? 96 : 32;
    // (a chunk wider than the columns there are buys nothing; not below 96, so that narrow tables keep sharing the blocks of the wide ones)
    if (chunk_cols > 96 && chunk_cols > last - first) chunk_cols = last - first > 96 ? last - first : 96;

    const u64 w = trace_gen, wi = bfe_inv(trace_gen);
    const u64 n_inv = bfe_inv(bfe_from_u64(N));
    Ntt2Args p1;
    p1.in = trace;
    p1.out = nullptr;
    p1.log_n1 = sp.log_n1;
    p1.log_n2 = sp.log_n2;
    p1.batch_log = batch_log_for(sp.log_n1);
    p1.in_len = N;
    p1.in_fk = fk;
    p1.out_fk = 1;
    p1.in_col_stride = N * fk;
    p1.tmp_col_stride = N;
    p1.out_col_stride = 0;
    p1.tw1 = pow_table(c, bfe_pow(wi, n2), n1);
    p1.tw2 = nullptr;
    TVM_TRY(make_inter(c, wi, sp, &p1.tw_inter));
    p1.pre_hi = p1.pre_lo = p1.post_lo = p1.post_hi = nullptr;
    p1.out_mul = 1;
    p1.out_add = 0;
    p1.col0 = 0;
    const bool std_roots = classify_root(w, N) == 1;  // every ArithmeticDomain's generator is; any other root takes the generic kernels
    p1.root = std_roots ? 2 : 0;

    LdePass2Args p2;
    p2.rnd = rnd;
    p2.log_n1 = sp.log_n1;
    p2.log_n2 = sp.log_n2;
    p2.batch_log = batch_log_for(sp.log_n2);
    p2.fk = fk;
    p2.h = h;
    p2.n_cosets = (int)X;
    p2.std_roots = std_roots ? 1 : 0;
    p2.tw_a2 = pow_table(c, bfe_pow(wi, n1), n2);
    p2.tw_b1 = pow_table(c, bfe_pow(w, n1), n2);
    TVM_TRY(make_inter(c, w, sp, &p2.tw_inter));
    TVM_TRY(coset_tables(c, eval_offset, eval_gen, X, n1, n2, n_inv, &p2.g_lo, &p2.g_hi));
    p2.g_lo_step = pow_table(c, eval_gen, n1);
    p2.g_hi_step = pow_table(c, bfe_pow(eval_gen, n1), n2);
    p2.g_hi_pos = p2.f_tw = p2.u_tw = nullptr;
    // (256- and 512-point axes -- 2^16 .. 2^19-row traces -- since round 6; TVM_OPTION_LDE_PASS2_TILES restores the tile kernels on all of them)
    const bool short_rows = c->lde_pass2_tiles == 0;   // the row kernels of passes 1 and 3 on 256- / 512-point axes, likewise
    const bool fused = std_roots && sp.log_n2 >= 8 && sp.log_n2 <= 11 && n1 % 16 == 0 && h <= n1 && c->lde_pass2_tiles == 0;
    if (fused)
        TVM_TRY(pass2_fused_tables(c, w, eval_offset, eval_gen, X, sp.log_n1, sp.log_n2, p2.g_lo, p2.g_hi, p2.tw_inter, p2.tw_b1, &p2.g_hi_pos,
                                   &p2.f_tw, &p2.u_tw));
    const u64 n_mont = bfe_from_u64(N);
    for (u64 k = 0; k < X; k++) {
        const u64 gamma = bfe_mul(eval_offset, bfe_pow(eval_gen, k));
        p2.zk[k] = bfe_mul(n_mont, bfe_sub(bfe_pow(gamma, N), TVM_ONE));
    }
    for (u64 k = X; k < TVM_LDE_MAX_COSETS; k++) p2.zk[k] = 0;

    LdePass3Args p3;
    p3.table = table;
    p3.log_n1 = sp.log_n1;
    p3.log_n2 = sp.log_n2;
    p3.n_cosets = (int)X;
    p3.L = L;
    p3.W = W;
    p3.pitch = lde_table_layout(N, L).pitch;
    p3.std_roots = std_roots ? 1 : 0;
    p3.tw_b2 = pow_table(c, bfe_pow(w, n2), n1);
    p3.fb_tw = nullptr;
    if (!p1.tw1 || !p2.tw_a2 || !p2.tw_b1 || !p2.g_lo || !p2.g_hi || !p2.g_lo_step || !p2.g_hi_step || !p3.tw_b2)
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "lde tables");

    // the intermediates of one chunk: Y (pass 1 -> pass 2) and Z (pass 2 -> pass 3), pool blocks (the next call of the same
    // shape gets the same blocks back from the cache, in stream order)
    // (with a split, Y is the caller's coefficient array: one chunk's worth at a time; the inverse-only mode needs no Z)
    PoolBlock y_block(c), z_block(c);
    u64* y = split ? split->coeffs : (u64*)y_block.alloc((size_t)chunk_cols * N * sizeof(u64));
    u64* z = mode == TVM_LDE_INVERSE_ONLY ? y : y ? (u64*)z_block.alloc((size_t)chunk_cols * X * N * sizeof(u64)) : nullptr;
    if ((!y || !z) && chunk_cols > 32) {   // the wide chunk does not fit: the narrow one before giving up
        chunk_cols = 32;
        z_block.alloc(0);
        if (!split) y = (u64*)y_block.alloc((size_t)chunk_cols * N * sizeof(u64));
        z = y ? (u64*)z_block.alloc((size_t)chunk_cols * X * N * sizeof(u64)) : nullptr;
    }
    if (!y || !z) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "lde intermediates");
    p2.mode = mode;
    p2.z = z;
    p3.z = z;

    for (int col0 = first; col0 < last; col0 += chunk_cols) {
        const int nc = (last - col0 < chunk_cols) ? (last - col0) : chunk_cols;
        p1.tmp = split ? y + (size_t)(col0 - first) * N : y;
        p2.y = p1.tmp;
        if (mode != TVM_LDE_FORWARD_ONLY) {
            Ntt2Args a = p1;
            a.col0 = col0;
            const int B = 1 << a.batch_log;
            const int tile = (int)n1 << a.batch_log;
            dim3 grid((unsigned)((n2 + B - 1) / B), (unsigned)nc);
            if (std_roots && short_rows && (sp.log_n1 == 8 || sp.log_n1 == 9) && n2 % 16 == 0) {   // 256 / 512 points: four / two rows per wavefront
                const size_t lds_r = (size_t)(16 * TVM_ROW_WORDS(n1) + n1) * sizeof(u64);
                if (sp.log_n1 == 8) TVM_LAUNCH((k_lde_pass1_rows<8, 16>), dim3((unsigned)(n2 / 16), (unsigned)nc), dim3(256), lds_r, c->stream, a);
                else TVM_LAUNCH((k_lde_pass1_rows<9, 16>), dim3((unsigned)(n2 / 16), (unsigned)nc), dim3(512), lds_r, c->stream, a);
            } else
            if (std_roots && sp.log_n1 == 10 && n2 % 16 == 0) {   // 1024-point axis: one row per wavefront
                // 16-row tiles: 128-byte runs of the input.  (8-row tiles -- two workgroups per CU -- measured the same time and
                // fetch every input line twice: 16 instead of 8 B per cell, profiles/r03_q_pmc_lde.txt.)
                const size_t lds_r = (size_t)(16 * TVM_ROW_WORDS(n1) + n1) * sizeof(u64);
                TVM_LAUNCH((k_lde_pass1_rows<10, 16>), dim3((unsigned)(n2 / 16), (unsigned)nc), dim3(1024), lds_r, c->stream, a);
            } else if (std_roots && sp.log_n1 == 11 && n2 % 8 == 0 && c->lde_pass2_tiles == 0) {   // 2048-point axis: two wavefronts per row
                const size_t lds_r = (size_t)(8 * TVM_ROW_WORDS(n1) + n1 + 512) * sizeof(u64);   // (+ sixteen blocks of 64 pair flags)
                TVM_LAUNCH((k_lde_pass1_rows<11, 8>), dim3((unsigned)(n2 / 8), (unsigned)nc), dim3(1024), lds_r, c->stream, a);
            } else
                TVM_LAUNCH(k_ntt2_pass1, grid, dim3(threads_for_tile(tile)), (size_t)tile * sizeof(u64), c->stream, a);
        }
        {
            LdePass2Args a = p2;
            a.col0 = col0;
            const int B = 1 << a.batch_log;
            const int tile = (int)n2 << a.batch_log;
            dim3 grid((unsigned)((n1 + B - 1) / B), (unsigned)nc);
            const size_t lds = (size_t)B * (n2 + TVM_ROW_PAD) * sizeof(u64);
            // axes longer than a workgroup: 2048 / 4096 points on 1024 work-items (2^21 .. 2^24 rows), and the same shapes at a
            // size the CPU suite can run (128 / 256 points on 64 work-items); rows per tile = 16 / positions per work-item
            const int ppt_log = (sp.log_n2 == 11 || sp.log_n2 == 7) ? 1 : (sp.log_n2 == 12 || sp.log_n2 == 8) ? 2 : 0;
            const u64 rows3 = 16 >> ppt_log;
            const size_t lds_v3 = (size_t)(rows3 * (n2 + TVM_ROW_PAD) + (sp.log_n2 < 12 ? n2 : 0) + 32) * sizeof(u64);
            const dim3 g2((unsigned)(n1 / rows3), (unsigned)nc);
            if (fused) {
                // 1024- / 2048-point axis: every wavefront (pair of wavefronts) keeps its row across the coset loop (k_lde_pass2_fused);
                // more trace randomizers than n1 (never the case for a STARK's parameters) take the kernels below
                const size_t lds_r = TVM_P2F_LDS_BYTES(sp.log_n2);
                if (sp.log_n2 == 8) TVM_LAUNCH((k_lde_pass2_fused<8>), dim3((unsigned)(n1 / 8), (unsigned)nc), dim3(128), lds_r, c->stream, a);
                else if (sp.log_n2 == 9) TVM_LAUNCH((k_lde_pass2_fused<9>), dim3((unsigned)(n1 / 8), (unsigned)nc), dim3(256), lds_r, c->stream, a);
                else if (sp.log_n2 == 10) TVM_LAUNCH((k_lde_pass2_fused<10>), dim3((unsigned)(n1 / 8), (unsigned)nc), dim3(512), lds_r, c->stream, a);
                else TVM_LAUNCH((k_lde_pass2_fused<11>), dim3((unsigned)(n1 / 8), (unsigned)nc), dim3(1024), lds_r, c->stream, a);
            }
            else if (std_roots && ppt_log && n1 % rows3 == 0) {
                if (sp.log_n2 == 11) TVM_LAUNCH((k_lde_pass2_v3<11, 10>), g2, dim3(1024), lds_v3, c->stream, a);
                else if (sp.log_n2 == 12) TVM_LAUNCH((k_lde_pass2_v3<12, 10>), g2, dim3(1024), lds_v3, c->stream, a);
                else if (sp.log_n2 == 7) TVM_LAUNCH((k_lde_pass2_v3<7, 6>), g2, dim3(64), lds_v3, c->stream, a);
                else TVM_LAUNCH((k_lde_pass2_v3<8, 6>), g2, dim3(64), lds_v3, c->stream, a);
            }
            else {
                // Short axes (traces below 2^13 rows): a 16-row tile is ONE wavefront that walks its 16 coefficients per lane through
                // nine transforms, and the grid is n1 / 16 x columns of them -- 384 wavefronts for 96 columns of a 2^12-row trace, 200 us
                // of latency.  Fewer rows per tile until the grid has ~2048 wavefronts (or a tile is two rows): the same kernel, the
                // same words (the tile height is a launch shape: every row is transformed on its own), 200 -> 60 us.
                int bl = a.batch_log;
                const auto waves = [&](int b) { return (u64)((n1 + (1u << b) - 1) >> b) * (u64)nc * (u64)(threads_for_tile((int)n2 << b) / 64); };
                while (bl > 1 && waves(bl) < 2048) bl--;
                a.batch_log = bl;
                const int tile_s = (int)n2 << bl;
                const dim3 grid_s((unsigned)((n1 + (1u << bl) - 1) >> bl), (unsigned)nc);
                const size_t lds_s = ((size_t)(n2 + TVM_ROW_PAD) << bl) * sizeof(u64);
                TVM_LAUNCH(k_lde_pass2, grid_s, dim3(threads_for_tile(tile_s)), lds_s, c->stream, a);
            }
        }
        if (mode != TVM_LDE_INVERSE_ONLY) {
            LdePass3Args a = p3;
            a.col0 = col0;
            a.rows_log = batch_log_for(sp.log_n1);
            const int RB = 1 << a.rows_log;
            const int tile = (int)n1 << a.rows_log;
            dim3 grid((unsigned)((X * n2 + RB - 1) / RB), (unsigned)nc);
            const size_t lds = ((size_t)(n1 + TVM_ROW_PAD) << a.rows_log) * sizeof(u64);
            const int ppt_log = (sp.log_n1 == 11 || sp.log_n1 == 7) ? 1 : (sp.log_n1 == 12 || sp.log_n1 == 8) ? 2 : 0;
            const u64 rows3 = 16 >> ppt_log, tiles3 = X * n2 / rows3;  // see pass 2
            // (pass 3 has no coset loop and no workgroup barrier: here the 2048-point row form is 8 % faster than k_lde_pass3_v3<11, 10>,
            // 15.5 against 16.8 ms per 96 columns at 2^22 rows, even at two wavefronts per SIMD)
            if (std_roots && (sp.log_n1 == 10 || sp.log_n1 == 11 || (short_rows && (sp.log_n1 == 8 || sp.log_n1 == 9))) && (X * n2) % 8 == 0) {
                // 1024- / 2048-point axis: one (k, j1) row per wavefront, no workgroup barrier (k_lde_pass3_rows): 8 wavefronts per
                // workgroup -- 78 KB of LDS, two workgroups per CU at 1024 points (4 wavefronts per workgroup: +2 %, 16: +3 %,
                // profiles/r03_g_lde_ab.txt); one workgroup per CU at 2048
                const u64 tiles_w = X * n2 / 8;
                a.tiles = tiles_w % 16 == 0 ? 16 : tiles_w % 8 == 0 ? 8 : tiles_w % 4 == 0 ? 4 : 1;
                if (tiles_w / a.tiles >= 65536) return set_error(c, TVM_ERR_UNSUPPORTED, "lde: too many row tiles");
                const size_t lds_w = (size_t)(8 * TVM_ROW_WORDS(n1) + n1) * sizeof(u64);
                const dim3 g3((unsigned)nc, (unsigned)(tiles_w / a.tiles));
                if (sp.log_n1 == 11 && c->lde_pass2_tiles == 0) {
                    // 2048-point rows as two 1024-point halves through one LDS region per wavefront (k_lde_pass3_halves)
                    const auto key = std::make_tuple(bfe_pow(w, n2) ^ 0xFB7AB1EFB7ull, (u64)2048, (u64)0);
                    auto found = c->tables.find(key);
                    u64* fb = found != c->tables.end() ? found->second : nullptr;
                    if (!fb) {
                        if (!bind_device(c) || hipMalloc((void**)&fb, 1024 * sizeof(u64)) != hipSuccess)
                            return set_error(c, TVM_ERR_OUT_OF_MEMORY, "pass-3 twiddle table");
                        TVM_LAUNCH(k_pass3_halves_table, dim3(4), dim3(256), 0, c->stream, a.tw_b2, fb);
                        c->tables[key] = fb;
                    }
                    a.fb_tw = fb;
                    const size_t lds_h = (size_t)(8 * TVM_ROW_WORDS(1024) + 1024) * sizeof(u64);
                    TVM_LAUNCH((k_lde_pass3_halves<8>), g3, dim3(512), lds_h, c->stream, a);
                } else if (sp.log_n1 == 11) TVM_LAUNCH((k_lde_pass3_rows<11, 8>), g3, dim3(512), lds_w, c->stream, a);
                else if (sp.log_n1 == 8) TVM_LAUNCH((k_lde_pass3_rows<8, 8>), g3, dim3(512), lds_w, c->stream, a);
                else if (sp.log_n1 == 9) TVM_LAUNCH((k_lde_pass3_rows<9, 8>), g3, dim3(512), lds_w, c->stream, a);
                else TVM_LAUNCH((k_lde_pass3_rows<10, 8>), g3, dim3(512), lds_w, c->stream, a);
            } else
            if (std_roots && ppt_log && (X * n2) % 16 == 0) {
                a.tiles = tiles3 % 8 == 0 ? 8 : tiles3 % 4 == 0 ? 4 : 1;
                const dim3 g3((unsigned)nc, (unsigned)(tiles3 / a.tiles));
                const size_t lds_v3 = (size_t)(rows3 * (n1 + TVM_ROW_PAD) + (sp.log_n1 < 12 ? n1 : 0)) * sizeof(u64);
                if (g3.y >= 65536) return set_error(c, TVM_ERR_UNSUPPORTED, "lde: too many row tiles");
                if (sp.log_n1 == 12) TVM_LAUNCH((k_lde_pass3_v3<12, 10>), g3, dim3(1024), lds_v3, c->stream, a);
                else if (sp.log_n1 == 7) TVM_LAUNCH((k_lde_pass3_v3<7, 6>), g3, dim3(64), lds_v3, c->stream, a);
                else TVM_LAUNCH((k_lde_pass3_v3<8, 6>), g3, dim3(64), lds_v3, c->stream, a);
            }
            else
                TVM_LAUNCH(k_lde_pass3, grid, dim3(threads_for_tile(tile)), lds, c->stream, a);
        }
    }
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
