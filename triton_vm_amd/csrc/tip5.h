// tip5.h -- the Tip5 permutation on gfx950 (one permutation per work-item, state in VGPRs).
//
// Specification: /root/reference/tips/tip-0005/tip-0005.md:31-83.  Replaces, on the hot path,
// twenty-first's Tip5 as used by MasterTable::hash_all_ldt_domain_rows
// (/root/reference/triton-vm/src/table/master_table.rs:455-503) and MerkleTree::par_new (:449).
//
// All words are Montgomery representatives.  The split-and-lookup S-box acts on the bytes of the
// Montgomery word itself (tip-0005.md:91-99), so it is a pure byte lookup; the MDS matrix has small
// integer entries, so it commutes with the Montgomery factor and is evaluated over the integers on
// the 32-bit halves of each word, with one reduction per output (tip-0005.md:101-103).
#pragma once
#include "field.h"
#include "tip5_tables.h"

#define TIP5_STATE 16
#define TIP5_RATE 10
#define TIP5_ROUNDS 5
#define TIP5_DIGEST 5

// Constant tables.  The round constants are read with wave-uniform indices (scalar loads); the
// 256-byte S-box table is copied into LDS by each workgroup (tip5_stage_lut) because its index is
// per-lane data.
#ifdef TVM_EMU
#define TVM_CONST_TABLE static const
#else
#define TVM_CONST_TABLE static __device__ const
#endif
TVM_CONST_TABLE u64 d_tip5_rc[80] = {TVM_TIP5_RC_LIST};
TVM_CONST_TABLE unsigned char d_tip5_lut[256] = {TVM_TIP5_LUT_LIST};

TVM_D void tip5_stage_lut(unsigned char* lds_lut, int tid, int nt) {
    for (int i = tid; i < 256; i += nt) lds_lut[i] = d_tip5_lut[i];
    __syncthreads();
}

// x = hi*2^64 + lo with hi < 2^32  ->  x mod p, canonical
TVM_HD u64 bfe_reduce96(u64 lo, u64 hi) {
    u64 t = hi * TVM_EPS;          // hi < 2^32: no overflow
    u64 r = lo + t;
    if (r < t) r += TVM_EPS;       // wrapped once: 2^64 = EPS (mod p); cannot wrap again
    return r >= TVM_P ? r - TVM_P : r;
}

TVM_HD u64 tip5_sbox_lookup(u64 x, const unsigned char* lut) {
    u64 y = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) y |= (u64)lut[(x >> (8 * b)) & 0xFF] << (8 * b);
    return y;
}
TVM_HD u64 tip5_pow7(u64 x) {
    u64 x2 = bfe_sqr(x);
    u64 x4 = bfe_sqr(x2);
    return bfe_mul(bfe_mul(x4, x2), x);
}

// y = M x + rc for the circulant M with first column TVM_TIP5_MDS_FIRST_COLUMN, over the integers on 32-bit
// halves: every partial sum is below 16 * 2^16 * 2^32 = 2^52.  The round constants (canonical words) are the
// initial values of the integer sums, so adding them costs nothing: M x + rc < 2^69 is reduced once.
TVM_HD void tip5_mds_add(u64 (&st)[TIP5_STATE], const u64* rc) {
    const u32 c[16] = {TVM_TIP5_MDS_LIST};
    u64 lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        lo[i] = (u32)rc[i];
        hi[i] = rc[i] >> 32;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u64 xl = (u32)st[j], xh = st[j] >> 32;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u64 m = c[(16 + i - j) & 15];
            lo[i] += m * xl;
            hi[i] += m * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        // lo + hi*2^32 as a 96-bit integer
        u64 l = lo[i] + (hi[i] << 32);
        u64 h = (hi[i] >> 32) + (l < lo[i] ? 1 : 0);
        st[i] = bfe_reduce96(l, h);
    }
}
TVM_HD void tip5_mds(u64 (&st)[TIP5_STATE]) {
    const u64 zero[16] = {0};
    tip5_mds_add(st, zero);
}

TVM_D void tip5_permute_inline(u64 (&st)[TIP5_STATE], const unsigned char* lut) {
    const u64* rc = d_tip5_rc;
    for (int r = 0; r < TIP5_ROUNDS; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) st[i] = tip5_sbox_lookup(st[i], lut);
#pragma unroll
        for (int i = 4; i < 16; i++) st[i] = tip5_pow7(st[i]);
        tip5_mds_add(st, rc + 16 * r);
    }
}

// ------------------------------------------------------------------------------------------------
// Lane-parallel form: one permutation spread over 16 adjacent lanes, lane `pos` holds state word `pos`.
// A dependent chain of permutations (the upper levels of a Merkle tree, one tree per FRI round) is bound by
// the latency of ONE permutation, ~8.8k dependent-ish instructions when a single lane does all 16 words;
// here a round is one S-box per lane, 16 lane rotations for the circulant MDS and one reduction: ~1k
// instructions per permutation.  Throughput per wavefront is lower (4 permutations instead of 64), so this
// form is used only where a level has too few nodes to fill the chip anyway.
TVM_D u64 tip5_permute_lanes(u64 x, int pos, int lane, const unsigned char* lut) {
    const u32 c[16] = {TVM_TIP5_MDS_LIST};
    const int base = lane & ~15;
    for (int r = 0; r < TIP5_ROUNDS; r++) {
        if (pos < 4) x = tip5_sbox_lookup(x, lut);
        else x = tip5_pow7(x);
        // out[pos] = sum_k M[k] * x[(pos - k) mod 16], over the integers on 32-bit halves
        const u64 rc = d_tip5_rc[16 * r + pos];  // the round constant rides in the integer sums (tip5_mds_add)
        u64 lo = (u32)rc, hi = rc >> 32;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64 xk = k ? __shfl(x, base | ((pos - k) & 15), 64) : x;
            lo += (u64)c[k] * (u32)xk;
            hi += (u64)c[k] * (xk >> 32);
        }
        const u64 l = lo + (hi << 32);
        const u64 h = (hi >> 32) + (l < lo ? 1 : 0);
        x = bfe_reduce96(l, h);
    }
    return x;
}

// ------------------------------------------------------------------------------------------------
// Matrix-core form: one permutation spread over FOUR lanes, sixteen permutations per wavefront.
//
// Lane l = (n = l % 16, g = l / 16) of a wavefront holds the words {g, g + 4, g + 8, g + 12} of the state of
// permutation n (st[t] = word g + 4t), so every lane has one split-and-lookup word (t = 0) and three power-map
// words: the S-box layer is uniform across lanes.  The MDS layer -- 512 of the ~725 integer multiplications
// of a round when one lane does everything -- goes to the matrix cores as ten v_mfma_i32_16x16x64_i8:
//
//   y_i = sum_j M_ij x_j over the integers, M_ij = m0 + 2^8 m1 + 2^16 m2 in balanced digits (m0, m1 in
//   [-128, 127], m2 in {0, 1}), x_j = sum_b 2^(8b) x_jb in bytes.  For c = 0..9:
//       T_c[i] = sum_j sum_a m_ija x_{j, c-a},           y_i = sum_c 2^(8c) T_c[i].
//   T_c is a 16 x 48 by 48 x 16 product: A[r][k] holds the digits (a constant operand), B[k][n] the byte windows
//   (x_{c-2}, x_{c-1}, x_c) of the four words a lane owns -- every B operand is built from the lane's OWN
//   registers with byte-align instructions, and the 16x16 result puts rows 4g..4g+3 of column n into lane (n, g),
//   which with the row order r -> word r/4 + 4(r%4) are exactly the words that lane owns: no cross-lane traffic.
//   The i8 operands are signed, so bytes travel as x - 128 and the accumulator input C carries the correction
//   128 * sum(digits), a bias 2^21 that keeps every T_c positive, and byte c of the (adjusted) round constant.
//
// The remaining VALU work per word is the recombination of ten 22-bit sums into one field element (~20
// instructions), against ~62 for the all-VALU form.
#ifdef TVM_EMU
struct tvm_v4i {
    int v[4];
    int& operator[](int i) { return v[i]; }
    const int& operator[](int i) const { return v[i]; }
};
// v_mfma_i32_16x16x64_i8 as this file relies on it: lane (i = l % 16, q = l / 16) supplies 16 signed bytes of row i
// of A and of column i of B for the same sixteen k-indices (which sixteen is irrelevant to a sum over k), and
// receives D[4q + v][i] = C + sum_k A[4q + v][k] B[k][i] in element v.  The GPU parity tests of tvm_hash_rows
// (tests/test_kernels_hash.py) run the same kernel on the hardware: they fail if it differs from this model.
static inline tvm_v4i emu_mfma_i32_16x16x64_i8(tvm_v4i a, tvm_v4i b, tvm_v4i c) {
    struct { tvm_v4i a, b; } mine = {a, b}, all[64];
    emu_wave_gather(&mine, sizeof(mine), all);
    const int lane = emu::lane_id(), col = lane & 15, q = lane >> 4;
    tvm_v4i d = c;
    for (int v = 0; v < 4; v++)
        for (int kq = 0; kq < 4; kq++) {
            const signed char* pa = (const signed char*)&all[16 * kq + 4 * q + v].a;
            const signed char* pb = (const signed char*)&all[16 * kq + col].b;
            for (int k = 0; k < 16; k++) d[v] += (int)pa[k] * (int)pb[k];
        }
    return d;
}
#define TVM_MFMA_I8(a, b, c) emu_mfma_i32_16x16x64_i8((a), (b), (c))
static inline u32 tvm_alignbyte(u32 hi, u32 lo, u32 shift) { return (u32)((((u64)hi << 32) | lo) >> (8 * shift)); }
#else
typedef int tvm_v4i __attribute__((ext_vector_type(4)));
#define TVM_MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_16x16x64_i8((a), (b), (c), 0, 0, 0)
static __device__ __forceinline__ u32 tvm_alignbyte(u32 hi, u32 lo, u32 shift) { return __builtin_amdgcn_alignbyte(hi, lo, shift); }
#endif

#define TIP5_MFMA_POSITIONS 10
#define TIP5_MFMA_BIAS_LOG 21

// balanced base-256 digits of an MDS entry
struct Tip5Digits { int d0, d1, d2; };
constexpr Tip5Digits tip5_digits(int m) {
    const int d0 = ((m + 128) & 255) - 128;
    const int rem = (m - d0) >> 8;
    const int d1 = ((rem + 128) & 255) - 128;
    return Tip5Digits{d0, d1, (rem - d1) >> 8};
}

// Accumulator inputs: ctab[((round * 10 + c) * 4 + g) * 4 + v] for row r = 4g + v, i.e. word g + 4v.
struct Tip5MfmaTable { int v[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16]; };
constexpr Tip5MfmaTable tip5_make_mfma_table() {
    Tip5MfmaTable t{};
    const u64 rc[80] = {TVM_TIP5_RC_LIST};
    const int mds[16] = {TVM_TIP5_MDS_LIST};
    int digit_sum = 0;  // the same for every row of a circulant matrix
    for (int j = 0; j < 16; j++) {
        const Tip5Digits d = tip5_digits(mds[j]);
        digit_sum += d.d0 + d.d1 + d.d2;
    }
    // the biases add up to 2^21 * sum_c 2^(8c); the round constants are lowered by that amount (mod p)
    unsigned __int128 bias_total = 0;
    for (int c = 0; c < TIP5_MFMA_POSITIONS; c++) bias_total += (unsigned __int128)1 << (8 * c + TIP5_MFMA_BIAS_LOG);
    const u64 k0 = (u64)(bias_total % TVM_P);
    for (int r = 0; r < TIP5_ROUNDS; r++)
        for (int g = 0; g < 4; g++)
            for (int v = 0; v < 4; v++) {
                const u64 word = rc[16 * r + g + 4 * v];
                const u64 adj = word >= k0 ? word - k0 : word + (TVM_P - k0);
                for (int c = 0; c < TIP5_MFMA_POSITIONS; c++)
                    t.v[((r * TIP5_MFMA_POSITIONS + c) * 4 + g) * 4 + v] =
                        128 * digit_sum + (1 << TIP5_MFMA_BIAS_LOG) + (c < 8 ? (int)((adj >> (8 * c)) & 0xFF) : 0);
            }
    return t;
}
TVM_CONST_TABLE Tip5MfmaTable d_tip5_mfma_table = tip5_make_mfma_table();
TVM_CONST_TABLE int d_tip5_mds[16] = {TVM_TIP5_MDS_LIST};

// The constant A operand of lane (r = lane % 16, g = lane / 16): for the four words j = g + 4jj that the lanes
// (., g) own, the digits of M[r/4 + 4(r%4)][j], laid out against the window bytes (pad, x_{c-2}, x_{c-1}, x_c).
TVM_D tvm_v4i tip5_mfma_matrix_operand(int lane) {
    const int r = lane & 15, g = lane >> 4;
    const int i_out = (r >> 2) + 4 * (r & 3);
    tvm_v4i a;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const Tip5Digits d = tip5_digits(d_tip5_mds[(16 + i_out - (g + 4 * jj)) & 15]);
        a[jj] = (int)(((u32)(d.d2 & 0xFF) << 8) | ((u32)(d.d1 & 0xFF) << 16) | ((u32)(d.d0 & 0xFF) << 24));
    }
    return a;
}

// ten 22-bit sums D_c -> sum_c 2^(8c) D_c mod p, canonical
TVM_D u64 tip5_mfma_recombine(const tvm_v4i (&d)[TIP5_MFMA_POSITIONS], int v) {
    const u32 u0 = (u32)d[0][v] + ((u32)d[1][v] << 8), v0 = (u32)d[2][v] + ((u32)d[3][v] << 8);  // < 2^31
    const u32 u1 = (u32)d[4][v] + ((u32)d[5][v] << 8), v1 = (u32)d[6][v] + ((u32)d[7][v] << 8);
    const u64 p0 = (u64)v0 * 65536u + u0;  // bits 0..46
    const u64 p1 = (u64)v1 * 65536u + u1;  // weight 2^32
    const u32 p2 = (u32)d[8][v] + ((u32)d[9][v] << 8);  // weight 2^64
    // 2^64 = EPS (mod p): p0 + 2^32 lo(p1) + (hi(p1) + p2) * EPS
    const u32 h = (u32)(p1 >> 32) + p2;
    const u64 t = (u64)h * 0xFFFFFFFFu + p0;  // < 2^63
    const u64 s = t + ((u64)(u32)p1 << 32);
    const u64 y = s < t ? s + TVM_EPS : s;  // wrapped once (then s < t < 2^63: adding EPS cannot wrap again)
    return y >= TVM_P ? y - TVM_P : y;
}

// st[t] = word g + 4t of the state of permutation n; every lane of the wavefront must take part.
TVM_D void tip5_permute_mfma(u64 (&st)[4], const tvm_v4i a, int g, const unsigned char* lut, const int* ctab) {
    for (int r = 0; r < TIP5_ROUNDS; r++) {
        // accumulator inputs first: their LDS latency hides behind the S-box layer
        tvm_v4i d[TIP5_MFMA_POSITIONS];
#pragma unroll
        for (int c = 0; c < TIP5_MFMA_POSITIONS; c++) {
            const int* cp = ctab + ((r * TIP5_MFMA_POSITIONS + c) * 4 + g) * 4;
#pragma unroll
            for (int v = 0; v < 4; v++) d[c][v] = cp[v];
        }
        st[0] = tip5_sbox_lookup(st[0], lut);
#pragma unroll
        for (int t = 1; t < 4; t++) st[t] = tip5_pow7(st[t]);
        const u32 pad = 0x80808080u;
        u32 x0[4], x1[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            x0[t] = (u32)st[t] ^ pad;
            x1[t] = (u32)(st[t] >> 32) ^ pad;
        }
#pragma unroll
        for (int c = 0; c < TIP5_MFMA_POSITIONS; c++) {
            tvm_v4i b;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                // bytes c-3 .. c of (pad pad pad | x | pad pad pad)
                u32 w;
                if (c < 3) w = tvm_alignbyte(x0[t], pad, c + 1);
                else if (c == 3) w = x0[t];
                else if (c < 7) w = tvm_alignbyte(x1[t], x0[t], c - 3);
                else if (c == 7) w = x1[t];
                else w = tvm_alignbyte(pad, x1[t], c - 7);
                b[t] = (int)w;
            }
            d[c] = TVM_MFMA_I8(a, b, d[c]);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) st[t] = tip5_mfma_recombine(d, t);
    }
}
