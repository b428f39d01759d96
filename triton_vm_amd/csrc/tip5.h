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
