// merkle_subtrees.hip -- the narrow levels of a Merkle tree, several to a launch (hash.hip: merkle_tree_from_leaves).
//
// A translation unit of its own: a kernel added to hash.hip moves that code object's text by its descriptor and symbols, and the row
// hashing kernel -- 43 % of a proof, bound by instruction issue -- measured 1 % slower at its new place (same-box alternation,
// profiles/r06_ab_*: main + aux Merkle 78.3 against 79.1 ms at 2^20 rows).
//
// Replaces, with hash.hip: MerkleTree::par_new (twenty-first) behind MasterTable::merkle_tree,
// /root/reference/triton-vm/src/table/master_table.rs:443-453, and ProverRound::merkle_tree_from_codeword,
// /root/reference/triton-vm/src/low_degree_test/fri.rs:343-347.
#include "kernels.h"
#include "tip5.h"

namespace tvm {

// Several narrow levels in ONE launch: a workgroup owns 64 consecutive parents of the widest level and everything above them that only
// they feed -- 64, 32, ..., 1 parents on up to seven consecutive levels, sixteen lanes per parent as above.  A workgroup's subtree depends
// on nothing outside it, so the levels are separated by workgroup barriers instead of launches: what a tree costs between its wide levels
// and its top is the latency of its permutations either way, but a proof of a short trace is bounded by the NUMBER of dependent
// dispatches (36 of these levels per proof at 2^10 rows).
__global__ void __launch_bounds__(1024) k_merkle_subtrees(u64* __restrict__ nodes, u64 widest, int levels) {
    __shared__ unsigned char lut[256];
    tip5_stage_lut(lut, threadIdx.x, blockDim.x);
    const int pos = (int)(threadIdx.x & 15), lane = (int)(threadIdx.x & 63);
    u64 lvl = widest;
    int per = 64;   // this workgroup's parents on the current level
    for (int t = 0; t < levels; t++, lvl >>= 1, per >>= 1) {
        // (a wavefront without a parent sits the level out; one with fewer than four clamps the rest: every lane joins the rotations)
        if ((int)((threadIdx.x & ~63u) >> 4) < per) {
            int jl = (int)(threadIdx.x >> 4);
            const bool live = jl < per;
            if (!live) jl = per - 1;
            const u64 i = lvl + (u64)blockIdx.x * (u64)per + (u64)jl;
            u64 x = pos < 10 ? nodes[10 * i + pos] : TVM_ONE;  // fixed-length domain: capacity all ones (tip-0005.md:82)
            x = tip5_permute_lanes(x, pos, lane, lut);
            if (live && pos < 5) nodes[5 * i + pos] = x;
        }
        __syncthreads();  // the same workgroup wrote the children of the next level: workgroup-scope visibility suffices
    }
}

int merkle_subtrees(tvm_ctx* c, u64* nodes, u64 widest, int levels) {
    if (widest < 64 || widest % 64 || levels < 1 || levels > 7) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "merkle subtrees: 64 parents per workgroup, 1 .. 7 levels");
    TVM_LAUNCH(k_merkle_subtrees, dim3((unsigned)(widest / 64)), dim3(1024), 0, c->stream, nodes, widest, levels);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
