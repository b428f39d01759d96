// hash.hip -- Tip5 row hashing and Merkle trees on gfx950.
//
// Replaces, on the reference's hot path:
//   MasterTable::hash_all_ldt_domain_rows   /root/reference/triton-vm/src/table/master_table.rs:455-468
//   MasterTable::merkle_tree (MerkleTree::par_new, twenty-first)              master_table.rs:443-453
//   quotient-segment row hashing + tree     /root/reference/triton-vm/src/stark.rs:425-446
//   ProverRound::merkle_tree_from_codeword  /root/reference/triton-vm/src/low_degree_test/fri.rs:343-347
//
// Row hashing: four lanes per row, the sponge state in VGPRs, the MDS layer of Tip5 on the matrix cores
// (tip5.h, "matrix-core form"); Merkle levels: one lane per parent while a level fills the chip, sixteen
// lanes per parent below that.  Hashing is integer-ALU bound (38 + 28 permutations per
// LDT row; ~60% of a permutation's instructions are the x^7 S-boxes), not HBM bound.

#include "context.h"
#include "tip5.h"

namespace tvm {


#ifndef TVM_HASH_BLOCK
#define TVM_HASH_BLOCK 256
#endif
#ifndef TVM_HASH_SPLIT_ABSORB
#define TVM_HASH_SPLIT_ABSORB 1   // 0: one loop over all blocks with the padding logic in every one (A/B, profiles/r06_o_*: proof 182.3 -> 180.1-180.5 ms on one box)
#endif

// digests[r] = Tip5::hash_varlen(domain row r*stride of the table), W words per row, with the permutation's MDS layer on
// the matrix cores (tip5_permute_mfma): four lanes per row, sixteen rows per wavefront.  Lane (n, g) absorbs the
// words g, g + 4 (and g + 8 for g < 2) of each block of ten: with consecutive rows in a wavefront (stride 1) every
// load instruction touches four full 128-byte lines of the row-block-major table.
__global__ void __launch_bounds__(TVM_HASH_BLOCK, 6) k_hash_rows_mfma(const u64* __restrict__ table, TabView view, int W,
                                                                    u64* __restrict__ digests) {
    __shared__ unsigned char lut[256];
    __shared__ int ctab[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16; i += blockDim.x) ctab[i] = d_tip5_mfma_table.v[i];
    tip5_stage_lut_lowered(lut, tid, blockDim.x);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    // the view's rows in the order that walks storage contiguously (context.h: TabView); digest r goes to the row's index
    // in the domain
    const u64 t = ((u64)blockIdx.x * (TVM_HASH_BLOCK / 64) + (tid >> 6)) * 16 + n;
    const bool live = t < view.n_out;  // every lane of a wavefront takes part in the matrix instructions
    u64 row, r;
    view.locate(live ? t : view.n_out - 1, row, r);
    const u64* base = table + (row >> TVM_RB_LOG) * (u64)W * TVM_RB + (row & (TVM_RB - 1));
    const Tip5MfmaOperands a = tip5_mfma_matrix_operands(lane);
    u64 st[4] = {0, 0, 0, 0};
#if TVM_HASH_SPLIT_ABSORB
    // The W / 10 FULL blocks of a row need no padding logic: the lane's words g, g + 4 (, g + 8) at fixed offsets from a pointer that
    // advances by a block (round 6: the compares and selects of the general form were 18 of the 22 VALU instructions a permutation
    // spends outside its rounds); the last block -- W % 10 words, then the padding 1, 0, ... -- keeps the general form.
    const int n_full = W / TIP5_RATE;
    const u64* p = base + (u64)g * TVM_RB;
    for (int perm = 0; perm < n_full; perm++) {
        st[0] = TVM_LOAD_STREAM(p);
        st[1] = TVM_LOAD_STREAM(p + 4 * TVM_RB);
        if (g < 2) st[2] = TVM_LOAD_STREAM(p + 8 * TVM_RB);
        p += TIP5_RATE * TVM_RB;
        tip5_permute_mfma(st, a, g, lut, ctab, true);   // (a block always follows: the one with the padding)
    }
#pragma unroll
    for (int t3 = 0; t3 < 3; t3++) {
        const int q = g + 4 * t3, wi = n_full * TIP5_RATE + q;
        if (q < TIP5_RATE) st[t3] = wi < W ? TVM_LOAD_STREAM(&base[(u64)wi * TVM_RB]) : (wi == W ? TVM_ONE : 0);  // padding: 1 then 0s
    }
    tip5_permute_mfma(st, a, g, lut, ctab, false);
#else
    const int n_perms = W / TIP5_RATE + 1;
    for (int perm = 0; perm < n_perms; perm++) {
#pragma unroll
        for (int t3 = 0; t3 < 3; t3++) {
            const int q = g + 4 * t3;  // word of the state, overwritten if it is in the rate part
            const int wi = perm * TIP5_RATE + q;
            if (q < TIP5_RATE) st[t3] = wi < W ? TVM_LOAD_STREAM(&base[(u64)wi * TVM_RB]) : (wi == W ? TVM_ONE : 0);  // padding: 1 then 0s
        }
        tip5_permute_mfma(st, a, g, lut, ctab, perm + 1 < n_perms);
    }
#endif
    if (live) {
        digests[r * 5 + g] = st[0];
        if (g == 0) digests[r * 5 + 4] = st[1];
    }
}

// nodes[i] = hash_pair(nodes[2i], nodes[2i+1]) for i in [first, first + count): the matrix-core form of the
// permutation (four lanes per parent, sixteen parents per wavefront; tip5.h), for the levels that fill the chip.
__global__ void __launch_bounds__(256, 6) k_merkle_level(u64* __restrict__ nodes, u64 first, u64 count, int reps) {
    __shared__ unsigned char lut[256];
    __shared__ int ctab[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16; i += blockDim.x) ctab[i] = d_tip5_mfma_table.v[i];
    tip5_stage_lut_lowered(lut, tid, blockDim.x);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const Tip5MfmaOperands a = tip5_mfma_matrix_operands(lane);
    // `reps` groups of 64 parents per workgroup, one after the other: the tables above and the matrix operands are set up once
    // (a wide level is ONE permutation per lane quadruple: the set-up was a fifth of the kernel).  (Round 6, measured and not adopted:
    // the children of parent rep + 1 requested before the permutation of parent rep through LDS-DMA staging words -- every
    // permutation here starts from cold loads -- changed nothing: 130.7 against 127.8 us per launch on average,
    // profiles/r06_j_kernels_merkle_level_children_staged.txt.  The kernel is not waiting for its loads.)
    for (int rep = 0; rep < reps; rep++) {
        u64 j = (((u64)blockIdx.x * reps + rep) * 4 + (tid >> 6)) * 16 + n;
        const bool live = j < count;  // every lane of a wavefront takes part in the matrix instructions
        if (!live) j = count - 1;
        const u64 i = first + j;
        u64 st[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int q = g + 4 * t;
            st[t] = q < 10 ? nodes[10 * i + q] : TVM_ONE;  // fixed-length domain: capacity all ones (tip-0005.md:82)
        }
        tip5_permute_mfma(st, a, g, lut, ctab);
        if (live) {
            nodes[5 * i + g] = st[0];
            if (g == 0) nodes[5 * i + 4] = st[1];
        }
    }
}

// Levels too narrow to fill the chip: 16 lanes per parent (tip5_permute_lanes), nodes[i] = hash_pair(nodes[2i],
// nodes[2i+1]) for i in [first, first + count).  Every lane of a wavefront takes part in the lane rotations,
// so out-of-range parents are clamped and their result dropped.
__global__ void __launch_bounds__(256) k_merkle_level_lanes(u64* __restrict__ nodes, u64 first, u64 count) {
    __shared__ unsigned char lut[256];
    tip5_stage_lut(lut, threadIdx.x, blockDim.x);
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const int pos = (int)(t & 15);
    u64 j = t >> 4;
    const bool live = j < count;
    if (!live) j = count - 1;
    const u64 i = first + j;
    u64 x = pos < 10 ? nodes[10 * i + pos] : TVM_ONE;  // fixed-length domain: capacity all ones (tip-0005.md:82)
    x = tip5_permute_lanes(x, pos, (int)(threadIdx.x & 63), lut);
    if (live && pos < 5) nodes[5 * i + pos] = x;
}

// the last levels of a tree (<= 64 parents on the widest: one pass per level) inside one workgroup, 16 lanes per parent
__global__ void __launch_bounds__(1024) k_merkle_top(u64* __restrict__ nodes, u64 widest) {
    __shared__ unsigned char lut[256];
    tip5_stage_lut(lut, threadIdx.x, blockDim.x);
    const int pos = (int)(threadIdx.x & 15), lane = (int)(threadIdx.x & 63);
    if (threadIdx.x < 5) nodes[threadIdx.x] = 0;   // node 0 of the heap is unused: zero (a memset of its own before round 6)
    for (u64 lvl = widest; lvl >= 1; lvl >>= 1) {
        for (u64 j0 = 0; j0 < lvl; j0 += blockDim.x / 16) {
            // The rotations of tip5_permute_lanes stay inside a wavefront: every lane of a wavefront that has a parent joins
            // them, a wavefront without one sits the level out (all sixteen wavefronts permuting clamped duplicates made every
            // level cost what the widest one does: 62 us per tree, seventeen trees per proof).
            if (j0 + ((threadIdx.x & ~63u) >> 4) >= lvl) continue;
            u64 j = j0 + (threadIdx.x >> 4);
            const bool live = j < lvl;
            if (!live) j = lvl - 1;
            const u64 i = lvl + j;
            u64 x = pos < 10 ? nodes[10 * i + pos] : TVM_ONE;
            x = tip5_permute_lanes(x, pos, lane, lut);
            if (live && pos < 5) nodes[5 * i + pos] = x;
        }
        __syncthreads();  // same workgroup wrote the children: workgroup-scope visibility suffices
    }
}

// Fiat-Shamir between two FRI rounds, on the device (one wavefront; lanes 0..15 hold the sponge state, the other lanes
// follow along so that every lane joins the rotations of tip5_permute_lanes):
//   ProofStream::enqueue(MerkleRoot(root)): the item's encoding [0, root] is 6 words, padded with 1, 0, 0, 0 to one block of the
//   rate; the sponge absorbs in overwrite mode, then permutes (proof_stream.rs:40-59, twenty-first Sponge::pad_and_absorb_all);
//   sample_scalars(1): squeeze -- the first 3 words of the state are the scalar -- and permute (proof_stream.rs:81-84).
__global__ void __launch_bounds__(64) k_sponge_root_and_sample(u64* __restrict__ state, const u64* __restrict__ root,
                                                                u64* __restrict__ challenge) {
    __shared__ unsigned char lut[256];
    tip5_stage_lut(lut, threadIdx.x, blockDim.x);
    const int lane = (int)threadIdx.x, pos = lane & 15;
    u64 x = state[pos];
    if (pos == 0) x = 0;                       // the discriminant of ProofItem::MerkleRoot
    else if (pos <= 5) x = root[pos - 1];
    else if (pos == 6) x = TVM_ONE;            // padding: 1, then zeros
    else if (pos < TIP5_RATE) x = 0;
    x = tip5_permute_lanes(x, pos, lane, lut);
    if (challenge) {
        if (lane < 3) challenge[lane] = x;
        x = tip5_permute_lanes(x, pos, lane, lut);
    }
    if (lane < 16) state[pos] = x;
}
int sponge_absorb_root_and_sample(tvm_ctx* c, u64* d_state, const u64* d_root, u64* d_challenge) {
    TVM_LAUNCH(k_sponge_root_and_sample, dim3(1), dim3(64), 0, c->stream, d_state, d_root, d_challenge);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// FRI leaves: Digest::from(xfe) = [c0, c1, c2, 0, 0], no hashing (fri.rs:343-347).
// codeword planar: c0[n], c1[n], c2[n] at stride `plane`.
__global__ void k_xfe_leaves(const u64* __restrict__ cw, u64 plane, u64 n, u64* __restrict__ leaves) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    leaves[5 * i + 0] = cw[i];
    leaves[5 * i + 1] = cw[plane + i];
    leaves[5 * i + 2] = cw[2 * plane + i];
    leaves[5 * i + 3] = 0;
    leaves[5 * i + 4] = 0;
}

// rows[j][0..W) = table row idx[j] (a row of the domain), row-major out (reveal_rows, master_table.rs:548-555)
__global__ void k_gather_rows(const u64* __restrict__ table, TabLayout l, int W, const u64* __restrict__ idx, u64 n,
                              u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * (u64)W) return;
    const u64 j = e / W;
    const int v = (int)(e % W);
    out[e] = table[tvm_tab_idx(l.storage_row(idx[j]), (u64)v, (u64)W)];
}

// whole table to the reference's row-major [L][W] layout in domain order (tests, and hosts that want the cache)
__global__ void k_table_to_row_major(const u64* __restrict__ table, TabLayout l, int W, u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= l.rows() * (u64)W) return;
    const u64 row = e / W;
    const int v = (int)(e % W);
    out[e] = table[tvm_tab_idx(l.storage_row(row), (u64)v, (u64)W)];
}

// the successor block of every coset (context.h): row j2 of block n2 = row (j2 + 1) mod n1 of block 0
__global__ void k_successor_blocks(u64* __restrict__ table, TabLayout l, int W) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= l.X * l.n1 * (u64)W) return;
    const u64 j2 = e % l.n1, v = (e / l.n1) % (u64)W, k = e / (l.n1 * (u64)W);
    const u64 src = k * l.pitch + (j2 + 1 == l.n1 ? 0 : j2 + 1), dst = k * l.pitch + l.n2 * l.n1 + j2;
    table[tvm_tab_idx(dst, v, (u64)W)] = table[tvm_tab_idx(src, v, (u64)W)];
}

// planar columns [W][L] -> row-block-major table in natural row order (used for the quotient-segment table)
__global__ void k_columns_to_table(const u64* __restrict__ cols, u64 col_stride, u64 L, int W, u64* __restrict__ table) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = tvm_tab_words(L, (u64)W);
    if (e >= total) return;
    const u64 r16 = e % TVM_RB;
    const u64 v = (e / TVM_RB) % (u64)W;
    const u64 row = (e / (TVM_RB * (u64)W)) * TVM_RB + r16;
    table[e] = (row < L) ? cols[v * col_stride + row] : 0;
}

// ------------------------------------------------------------------------------------------------
int hash_rows(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, u64 stride, u64* digests) {
    if (!stride || layout.rows() % stride) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "hash_rows: stride must divide the number of rows");
    const TabView view = tab_view(layout, stride);
    const u64 rows_per_block = TVM_HASH_BLOCK / 4;
    TVM_LAUNCH(k_hash_rows_mfma, dim3((unsigned)((view.n_out + rows_per_block - 1) / rows_per_block)), dim3(TVM_HASH_BLOCK), 0, c->stream,
               table, view, W, digests);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// nodes: [2*n_leaves][5], leaves already at nodes[n_leaves..2*n_leaves)
int merkle_subtrees(tvm_ctx* c, u64* nodes, u64 widest, int levels);   // merkle_subtrees.hip

int merkle_tree_from_leaves(tvm_ctx* c, u64* nodes, u64 n_leaves) {
    if (!is_pow2(n_leaves)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "merkle: leaf count must be a power of two");
    u64 lvl = n_leaves >> 1;
    if (lvl < 1) TVM_HIP_CHECK(c, hipMemsetAsync(nodes, 0, 5 * sizeof(u64), c->stream));   // (node 0 is unused: zero; k_merkle_top writes it otherwise)
    for (; lvl > 32768; lvl >>= 1) {  // wide levels: four lanes per parent on the matrix cores (throughput form)
        const u64 groups = (lvl + 63) / 64;   // of 64 parents; up to 8 per workgroup while >= 4096 workgroups remain
        const u64 min_wgs = c->merkle_min_workgroups;   // 4096; TVM_OPTION_MERKLE_MIN_WORKGROUPS (the tests lower it to reach this path)
        const int reps = groups >= 8 * min_wgs ? 8 : groups >= 4 * min_wgs ? 4 : groups >= 2 * min_wgs ? 2 : 1;
        TVM_LAUNCH(k_merkle_level, dim3((unsigned)((groups + reps - 1) / reps)), dim3(256), 0, c->stream, nodes, lvl, lvl, reps);
    }
    if (c->merkle_subtrees) {
        // narrow levels, 16 lanes per parent (latency form), up to seven levels per launch, down to the level of 64 parents
        while (lvl > 64) {
            const int levels = ilog2(lvl) - 5 < 7 ? ilog2(lvl) - 5 : 7;
            TVM_TRY(merkle_subtrees(c, nodes, lvl, levels));   // merkle_subtrees.hip
            lvl >>= levels;
        }
    } else
        for (; lvl > 64; lvl >>= 1)     // one level per launch (TVM_OPTION_MERKLE_SUBTREES 0: A/B)
            TVM_LAUNCH(k_merkle_level_lanes, dim3((unsigned)((lvl * 16 + 255) / 256)), dim3(256), 0, c->stream, nodes, lvl, lvl);
    if (lvl >= 1) TVM_LAUNCH(k_merkle_top, dim3(1), dim3(1024), 0, c->stream, nodes, lvl);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int xfe_leaves(tvm_ctx* c, const u64* cw, u64 plane, u64 n, u64* leaves) {
    TVM_LAUNCH(k_xfe_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, cw, plane, n, leaves);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int gather_rows(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, const u64* d_idx, u64 n, u64* d_out) {
    const u64 total = n * (u64)W;
    if (!total) return TVM_OK;
    TVM_LAUNCH(k_gather_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, table, layout, W, d_idx, n, d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int fill_successor_blocks(tvm_ctx* c, u64* table, const TabLayout& layout, int W) {
    const u64 total = layout.X * layout.n1 * (u64)W;
    if (layout.pitch < (layout.n2 + 1) * layout.n1) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "the table has no successor blocks");
    TVM_LAUNCH(k_successor_blocks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, table, layout, W);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int table_to_row_major(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, u64* d_out) {
    const u64 total = layout.rows() * (u64)W;
    TVM_LAUNCH(k_table_to_row_major, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, table, layout, W, d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int columns_to_table(tvm_ctx* c, const u64* cols, u64 col_stride, u64 L, int W, u64* table) {
    const u64 total = tvm_tab_words(L, (u64)W);
    TVM_LAUNCH(k_columns_to_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, cols, col_stride, L, W, table);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
