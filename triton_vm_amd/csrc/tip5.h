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
// the table with every entry lowered by 128 (as a signed byte), for tip5_permute_mfma
TVM_D void tip5_stage_lut_lowered(unsigned char* lds_lut, int tid, int nt) {
    for (int i = tid; i < 256; i += nt) lds_lut[i] = d_tip5_lut[i] ^ 0x80;
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
#if !defined(__HIP_DEVICE_COMPILE__)
// host code: the circulant matrix written out, row j = the entries that multiply x_j, so that the inner loop of tip5_mds_add
// runs over consecutive 32-bit entries (the compiler vectorises it: 32 x 32 -> 64 multiplies)
struct Tip5MdsRows { u32 m[16][16]; };
constexpr Tip5MdsRows tip5_make_mds_rows() {
    Tip5MdsRows r{};
    const u32 c[16] = {TVM_TIP5_MDS_LIST};
    for (int j = 0; j < 16; j++)
        for (int i = 0; i < 16; i++) r.m[j][i] = c[(16 + i - j) & 15];
    return r;
}
static constexpr Tip5MdsRows tip5_mds_rows = tip5_make_mds_rows();
#endif
TVM_HD void tip5_mds_add(u64 (&st)[TIP5_STATE], const u64* rc) {
    u64 lo[16], hi[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        lo[i] = (u32)rc[i];
        hi[i] = rc[i] >> 32;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    for (int j = 0; j < 16; j++) {
        const u32 xl = (u32)st[j], xh = (u32)(st[j] >> 32);
        const u32* m = tip5_mds_rows.m[j];
        for (int i = 0; i < 16; i++) {
            lo[i] += (u64)m[i] * xl;
            hi[i] += (u64)m[i] * xh;
        }
    }
#else
    const u32 c[16] = {TVM_TIP5_MDS_LIST};
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
#endif
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
// of a round when one lane does everything -- goes to the matrix cores as twelve v_mfma_i32_16x16x64_i8:
//
//   y_i = sum_j M_ij x_j over the integers, M_ij = m0 + 2^8 m1 + 2^16 m2 in balanced digits (m0, m1 in
//   [-128, 127], m2 in {0, 1}), x_j = sum_b 2^(8b) x_jb in bytes.  For c = 0..9:
//       T_c[i] = sum_j sum_k m_ijk x_{j, c-k},           y_i = sum_c 2^(8c) T_c[i].
//   The B operands are the two 32-bit halves of the four words a lane owns, exactly as they lie in its registers; the
//   byte shifts are in the CONSTANT operands: A_s (s = 0..5) holds digit s - b against byte b of a half, so
//       T_c = A_c * (low halves)  [c <= 5]  +  A_{c-4} * (high halves)  [c >= 4]
//   -- no byte-align instruction anywhere.  The 16x16 result puts rows 4g..4g+3 of column n into lane (n, g), which with
//   the row order r -> word r/4 + 4(r%4) are exactly the words that lane owns: no cross-lane traffic.
//   The i8 operands are signed, so bytes travel as x - 128 and the accumulator input C carries the correction
//   128 * sum(digits present at that position), a bias 2^21 that keeps every T_c positive, and byte c of the (adjusted)
//   round constant.
//
// The remaining VALU work per word is the recombination of ten 22-bit sums into one field element (17 instructions:
// tip5_mfma_recombine), against ~62 for the all-VALU form.
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
#else
typedef int tvm_v4i __attribute__((ext_vector_type(4)));
#define TVM_MFMA_I8(a, b, c) __builtin_amdgcn_mfma_i32_16x16x64_i8((a), (b), (c), 0, 0, 0)
#endif

#define TIP5_MFMA_POSITIONS 10
#define TIP5_MFMA_BIAS_LOG 21

// balanced base-256 digits of an MDS entry
struct Tip5Digits { int d[3]; };
constexpr Tip5Digits tip5_digits(int m) {
    const int d0 = ((m + 128) & 255) - 128;
    const int rem = (m - d0) >> 8;
    const int d1 = ((rem + 128) & 255) - 128;
    return Tip5Digits{{d0, d1, (rem - d1) >> 8}};
}

// Accumulator inputs: ctab[((round * 10 + c) * 4 + g) * 4 + v] for row r = 4g + v, i.e. word g + 4v.
struct Tip5MfmaTable { int v[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16]; };
constexpr Tip5MfmaTable tip5_make_mfma_table() {
    Tip5MfmaTable t{};
    const u64 rc[80] = {TVM_TIP5_RC_LIST};
    const int mds[16] = {TVM_TIP5_MDS_LIST};
    int digit_sum[3] = {0, 0, 0};  // per digit position; the same for every row of a circulant matrix
    for (int j = 0; j < 16; j++) {
        const Tip5Digits d = tip5_digits(mds[j]);
        for (int k = 0; k < 3; k++) digit_sum[k] += d.d[k];
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
                for (int c = 0; c < TIP5_MFMA_POSITIONS; c++) {
                    // position c sums digit k of the matrix against byte c - k of the state (where that byte exists): every
                    // such product was taken with the byte lowered by 128
                    int digits_here = 0;
                    for (int k = 0; k < 3; k++)
                        if (c - k >= 0 && c - k < 8) digits_here += digit_sum[k];
                    t.v[((r * TIP5_MFMA_POSITIONS + c) * 4 + g) * 4 + v] =
                        128 * digits_here + (1 << TIP5_MFMA_BIAS_LOG) + (c < 8 ? (int)((adj >> (8 * c)) & 0xFF) : 0);
                }
            }
    return t;
}
TVM_CONST_TABLE Tip5MfmaTable d_tip5_mfma_table = tip5_make_mfma_table();
TVM_CONST_TABLE int d_tip5_mds[16] = {TVM_TIP5_MDS_LIST};

// The constant A operands of lane (r = lane % 16, g = lane / 16).  The B operands are the two 32-bit halves of the four
// words j = g + 4jj that the lanes (., g) own, AS THEY LIE IN THE REGISTERS (bytes 0..3 and 4..7); operand s in 0..5 holds,
// against byte b of such a half, digit s - b of M[r/4 + 4(r%4)][j] (nothing where s - b is not a digit position): the
// product with the low halves is the share of the bytes 0..3 in position s, the product with the high halves the share
// of the bytes 4..7 in position s + 4.
#define TIP5_MFMA_SHIFTS 6
struct Tip5MfmaOperands { tvm_v4i a[TIP5_MFMA_SHIFTS]; };
TVM_D Tip5MfmaOperands tip5_mfma_matrix_operands(int lane) {
    const int r = lane & 15, g = lane >> 4;
    const int i_out = (r >> 2) + 4 * (r & 3);
    Tip5MfmaOperands o;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        const Tip5Digits d = tip5_digits(d_tip5_mds[(16 + i_out - (g + 4 * jj)) & 15]);
#pragma unroll
        for (int s = 0; s < TIP5_MFMA_SHIFTS; s++) {
            u32 w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (s - b >= 0 && s - b < 3) w |= (u32)(d.d[s - b] & 0xFF) << (8 * b);
            o.a[s][jj] = (int)w;
        }
    }
    return o;
}

// Ten 22-bit sums D_c per word -> sum_c 2^(8c) D_c mod p, canonical, for the four words of a lane.
//   P0 = D_0 + 2^8 D_1 + 2^16 D_2 + 2^24 D_3 (< 2^47) and P1 likewise from D_4..D_7 (weight 2^32): chains of v_mad_u64_u32,
//   the first with a zero addend, so that no 32-bit value has to be widened into an aligned register pair;
//   2^64 = EPS (mod p):  value = P0 + 2^32 lo(P1) + h * EPS,  h = hi(P1) + D_8 + 2^8 D_9 (< 2^31);
//   t = h * EPS + P0 < 2^63;  s = t + 2^32 lo(P1) (carry c): the value is s + 2^64 c, and it is canonical after ONE
//   subtraction of p (= addition of EPS modulo 2^64) exactly when c is set (then s < t < 2^63) or s >= p, i.e. when
//   z = s + EPS carries (c2):  result = (c | c2) ? z : s  -- the shape of bfe_add's tail.
// The tail of the four words is one instruction stream, chain by chain (carries in VCC and seven SGPR pairs), so that
// every carry consumer has at least two instructions between it and its producer (field.h: TVM_VCC_WAIT).
// `lean` (uniform over the wavefront): the words v = 0, 1 are NOT owed -- the caller overwrites them (tip5_permute_mfma with rate_is_overwritten) --
// so their multiply-add chains are skipped (18 instructions); the carry tail below stays the one four-chain block and turns
// whatever it is given for them into values nobody reads.
#ifndef TVM_TIP5_LEAN_ROUND
#define TVM_TIP5_LEAN_ROUND 1   // 0: the last round recombines the rate words all the same (A/B: profiles/r06_n_*, 43.0-43.3 against 43.4-43.6 ms)
#endif
TVM_D void tip5_mfma_recombine(const tvm_v4i (&d)[TIP5_MFMA_POSITIONS], u64 (&st)[4], bool lean = false) {
    u64 t[4];
    u32 pl[4];
    // the shifts as multiplications by values the compiler cannot see through (it would widen every term into a register
    // pair and use 64-bit shifts and additions instead of one v_mad_u64_u32 per term)
    u32 w8 = 1u << 8, w16 = 1u << 16, w24 = 1u << 24;
#ifdef TVM_FIELD_ASM
    asm("" : "+s"(w8), "+s"(w16), "+s"(w24));
#endif
#pragma unroll
    for (int v = 0; v < 4; v++) {
        if (v < 2 && lean) {   // (any defined value will do)
            t[v] = (u64)(u32)d[0][v];
            pl[v] = (u32)d[4][v];
            continue;
        }
        u64 p0 = (u64)(u32)d[1][v] * w8;
        p0 += (u64)(u32)d[2][v] * w16;
        p0 += (u64)(u32)d[3][v] * w24;
        p0 += (u32)d[0][v];
        u64 p1 = (u64)(u32)d[5][v] * w8;
        p1 += (u64)(u32)d[6][v] * w16;
        p1 += (u64)(u32)d[7][v] * w24;
        p1 += (u32)d[4][v];
        const u32 h = (u32)(p1 >> 32) + (u32)d[8][v] + ((u32)d[9][v] << 8);
        t[v] = (u64)h * 0xFFFFFFFFu + p0;
        pl[v] = (u32)p1;
    }
#ifdef TVM_FIELD_ASM
    u32 sh0, sh1, sh2, sh3, zl0, zl1, zl2, zl3, zh0, zh1, zh2, zh3;
    u64 k0, k1, k2, k3, c1, c2, c3;
#define TIP5_R1(i, K) "v_add_co_u32_e64 %[sh" #i "], " K ", %[th" #i "], %[pl" #i "]\n\t"
#define TIP5_R2(i, C) "v_add_co_u32_e64 %[zl" #i "], " C ", -1, %[tl" #i "]\n\t"
#define TIP5_R3(i, C) "v_addc_co_u32_e64 %[zh" #i "], " C ", 0, %[sh" #i "], " C "\n\t"
#define TIP5_R4(i, C, K) "s_or_b64 " C ", " C ", " K "\n\t"
#define TIP5_R5(i, C) "v_cndmask_b32_e64 %[zl" #i "], %[tl" #i "], %[zl" #i "], " C "\n\t"
#define TIP5_R6(i, C) "v_cndmask_b32_e64 %[sh" #i "], %[sh" #i "], %[zh" #i "], " C "\n\t"
    asm(TIP5_R1(0, "%[k0]") TIP5_R1(1, "%[k1]") TIP5_R1(2, "%[k2]") TIP5_R1(3, "%[k3]")
        TIP5_R2(0, "vcc") TIP5_R2(1, "%[c1]") TIP5_R2(2, "%[c2]") TIP5_R2(3, "%[c3]")
        TIP5_R3(0, "vcc") TIP5_R3(1, "%[c1]") TIP5_R3(2, "%[c2]") TIP5_R3(3, "%[c3]")
        TIP5_R4(0, "vcc", "%[k0]") TIP5_R4(1, "%[c1]", "%[k1]") TIP5_R4(2, "%[c2]", "%[k2]") TIP5_R4(3, "%[c3]", "%[k3]")
        TIP5_R5(0, "vcc") TIP5_R5(1, "%[c1]") TIP5_R5(2, "%[c2]") TIP5_R5(3, "%[c3]")
        TIP5_R6(0, "vcc") TIP5_R6(1, "%[c1]") TIP5_R6(2, "%[c2]") TIP5_R6(3, "%[c3]")
        : [sh0] "=&v"(sh0), [sh1] "=&v"(sh1), [sh2] "=&v"(sh2), [sh3] "=&v"(sh3), [zl0] "=&v"(zl0), [zl1] "=&v"(zl1),
          [zl2] "=&v"(zl2), [zl3] "=&v"(zl3), [zh0] "=&v"(zh0), [zh1] "=&v"(zh1), [zh2] "=&v"(zh2), [zh3] "=&v"(zh3),
          [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)
        : [tl0] "v"((u32)t[0]), [th0] "v"((u32)(t[0] >> 32)), [pl0] "v"(pl[0]), [tl1] "v"((u32)t[1]), [th1] "v"((u32)(t[1] >> 32)),
          [pl1] "v"(pl[1]), [tl2] "v"((u32)t[2]), [th2] "v"((u32)(t[2] >> 32)), [pl2] "v"(pl[2]), [tl3] "v"((u32)t[3]),
          [th3] "v"((u32)(t[3] >> 32)), [pl3] "v"(pl[3])
        : "vcc", "scc");
#undef TIP5_R1
#undef TIP5_R2
#undef TIP5_R3
#undef TIP5_R4
#undef TIP5_R5
#undef TIP5_R6
    st[0] = ((u64)sh0 << 32) | zl0;
    st[1] = ((u64)sh1 << 32) | zl1;
    st[2] = ((u64)sh2 << 32) | zl2;
    st[3] = ((u64)sh3 << 32) | zl3;
#else
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const u64 s = t[v] + ((u64)pl[v] << 32), z = s + TVM_EPS;
        st[v] = ((s < t[v]) | (z < s)) ? z : s;
    }
#endif
}

// Round 6, measured and NOT adopted (the code is in the history; profiles/r06_b_*, r06_c_*, r06_d_*):
//  * commit 71bc48c -- the split-and-lookup S-box two bytes at a time from a 65536-entry table in LDS: 250 instead of 260 VALU
//    instructions per round, but 128 KB of LDS mean one workgroup of sixteen wavefronts per CU, and at four wavefronts per SIMD instead
//    of six the wait states of the carry chains are no longer hidden: 52.8 against 44.3 ms for the main table's rows.  (The d16 loads
//    that would assemble the result for free zero the other half of their register under SRAM-ECC: hipcc never selects them.)
//  * commit 77412a9 -- tools/ubench/tip5_floor.hip puts the product's row hashing 6.5-7 % above its own permutations run on registers
//    alone.  (a) The next block of a row requested a whole permutation ahead through LDS-DMA loads (no register holds it): no scratch
//    at all, and no gain (43.9-44.1 against 43.7-44.0 ms) -- the gap is NOT the latency of the absorb's loads.  (b) A LEAN last round
//    that does not recombine the rate words the next block overwrites (34 of a permutation's 1300 VALU instructions): as a branch
//    between two tails it spills one matrix operand per round, as two pairs of words it loses the four-chain interleave: 44.3-44.7 ms.
//    What the 6.5-7 % are, by count: the absorb's own instructions 1.7 %, workgroup set-up and relaunch ~1.2 %, the launch's last
//    wave of workgroups (85.3 rounds of them: one in 86) ~1.2 %; the rest (~2.5 %) moves with the memory traffic, not with the code.
//
// st[t] = word g + 4t of the state of permutation n; every lane of the wavefront must take part.  `lut` is the S-box table
// LOWERED by 128 (tip5_stage_lut_lowered): the looked-up word goes to the matrix cores only, where bytes travel that way.
// `rate_is_overwritten` (uniform): the caller absorbs the next block in overwrite mode right after this permutation, i.e. replaces the
// words 0 .. 9 -- st[0], st[1] of every lane and st[2] of the lanes g < 2 -- so the last round need not produce st[0], st[1].
TVM_D void tip5_permute_mfma(u64 (&st)[4], const Tip5MfmaOperands& m, int g, const unsigned char* lut, const int* ctab,
                             bool rate_is_overwritten = false) {
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
        const u32 pad = 0x80808080u;  // bytes travel lowered by 128
        tvm_v4i lo, hi;
        lo[0] = (int)(u32)st[0];
        hi[0] = (int)(u32)(st[0] >> 32);
#pragma unroll
        for (int t = 1; t < 4; t++) {
            lo[t] = (int)((u32)st[t] ^ pad);
            hi[t] = (int)((u32)(st[t] >> 32) ^ pad);
        }
#pragma unroll
        for (int s = 0; s < TIP5_MFMA_SHIFTS; s++) d[s] = TVM_MFMA_I8(m.a[s], lo, d[s]);
#pragma unroll
        for (int s = 0; s < TIP5_MFMA_SHIFTS; s++) d[s + 4] = TVM_MFMA_I8(m.a[s], hi, d[s + 4]);
        tip5_mfma_recombine(d, st, TVM_TIP5_LEAN_ROUND && rate_is_overwritten && r + 1 == TIP5_ROUNDS);
    }
}
