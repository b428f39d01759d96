// verify.hip -- the verifier's batch work over the revealed rows.
//
// Replaces, in Verifier::verify (/root/reference/triton-vm/src/stark.rs:1388-1763):
//   the leaf digests of the revealed main / aux / quotient rows            stark.rs:1598-1601, 1620-1660
//   linearly_sum_main_and_aux_row, the quotient-segment sums, the four Stark::deep_update calls and the weighted
//   sum that must equal the value the low-degree test revealed             stark.rs:1678-1755, 2096-2103
// for all `num_first_round_queries` rows at once.  One workgroup per revealed row: the 470 columns are summed with
// their weights across the work-items (XFE x BFE products for the main row, XFE x XFE for the auxiliary row), one
// work-item finishes the row.  The checks themselves (equality with the revealed value, Merkle inclusion) stay with
// the caller, like every other Fiat-Shamir-dependent decision.
#include "context.h"
#include "kernels.h"

namespace tvm {

TVM_D xfe vf_ld(const u64* p) { return xfe_make(p[0], p[1], p[2]); }

struct VerifyArgs {
    const u64* main_rows;   // [q][n_main]
    const u64* aux_rows;    // [q][n_aux][3]
    const u64* quot_rows;   // [q][5][3]
    const u64* row_idx;     // [q]
    const u64* w_ma;        // [n_main + n_aux][3]
    const u64* small;       // weights_quot[5][3], weights_deep[4][3], ood points[4][3], ood values[4][3]
    u64 offset, gen;        // low-degree-test domain
    int n_main, n_aux;
    u64* out;               // [q][3]
};

#define VF_BLOCK 256
__global__ void __launch_bounds__(VF_BLOCK) k_verifier_deep_values(VerifyArgs a) {
    __shared__ u64 smem[3 * VF_BLOCK];
    const int tid = threadIdx.x;
    const u64 j = blockIdx.x;
    const u64* mrow = a.main_rows + j * (u64)a.n_main;
    const u64* arow = a.aux_rows + j * (u64)a.n_aux * 3;
    // linearly_sum_main_and_aux_row (stark.rs:1765-1787)
    xfe acc = xfe_zero();
    for (int c = tid; c < a.n_main; c += VF_BLOCK) acc = xfe_add(acc, xfe_mul_bfe(vf_ld(a.w_ma + 3 * c), mrow[c]));
    for (int c = tid; c < a.n_aux; c += VF_BLOCK) acc = xfe_add(acc, xfe_mul(vf_ld(a.w_ma + 3 * (a.n_main + c)), vf_ld(arow + 3 * c)));
    // sum over the workgroup
    smem[tid] = acc.c0, smem[VF_BLOCK + tid] = acc.c1, smem[2 * VF_BLOCK + tid] = acc.c2;
    __syncthreads();
    for (int s = VF_BLOCK >> 1; s > 0; s >>= 1) {
        if (tid < s)
            for (int k = 0; k < 3; k++) smem[k * VF_BLOCK + tid] = bfe_add(smem[k * VF_BLOCK + tid], smem[k * VF_BLOCK + tid + s]);
        __syncthreads();
    }
    if (tid) return;
    const xfe ma = xfe_make(smem[0], smem[VF_BLOCK], smem[2 * VF_BLOCK]);
    const u64* wq = a.small;
    const u64* wd = a.small + 15;
    const u64* pts = a.small + 27;
    const u64* vals = a.small + 39;
    const u64* q = a.quot_rows + j * 15;
    // quotient segments: P uses segments 0..3, R segments 1..4 (stark.rs:1700-1717)
    xfe shared = xfe_zero();
    for (int k = 1; k < 4; k++) shared = xfe_add(shared, xfe_mul(vf_ld(q + 3 * k), vf_ld(wq + 3 * k)));
    const xfe for_p = xfe_add(xfe_mul(vf_ld(wq), vf_ld(q)), shared);
    const xfe for_r = xfe_add(xfe_mul(vf_ld(wq + 12), vf_ld(q + 12)), shared);
    // deep_update (stark.rs:2096-2103): (value - ood value) / (domain point - ood point)
    const u64 x = bfe_mul(a.offset, bfe_pow(a.gen, a.row_idx[j]));
    const xfe elems[4] = {ma, ma, for_p, for_r};
    xfe total = xfe_zero();
    for (int k = 0; k < 4; k++) {
        const xfe num = xfe_sub(elems[k], vf_ld(vals + 3 * k));
        const xfe den = xfe_bfe_minus(x, vf_ld(pts + 3 * k));
        total = xfe_add(total, xfe_mul(vf_ld(wd + 3 * k), xfe_mul(num, xfe_inv(den))));
    }
    u64* o = a.out + 3 * j;
    o[0] = total.c0, o[1] = total.c1, o[2] = total.c2;
}

// rows [n][w] row-major -> the row-block-major layout the row-hashing kernel reads
__global__ void k_rows_to_table(const u64* __restrict__ rows, u64 n, int W, u64* __restrict__ table) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * (u64)W) return;
    const u64 row = e / (u64)W, v = e % (u64)W;
    table[tvm_tab_idx(row, v, (u64)W)] = rows[e];
}

int hash_varlen_rows(tvm_ctx* c, const u64* d_rows, u64 n, int W, u64* d_digests) {
    const u64 padded = (n + TVM_RB - 1) / TVM_RB * TVM_RB;
    PoolBlock block(c, (size_t)tvm_tab_words(padded, (u64)W) * sizeof(u64));  // released on every exit path
    u64* table = (u64*)block.p;
    if (!table) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "hash_varlen_rows scratch");
    TVM_HIP_CHECK(c, hipMemsetAsync(table, 0, (size_t)tvm_tab_words(padded, (u64)W) * sizeof(u64), c->stream));
    const u64 total = n * (u64)W;
    TVM_LAUNCH(k_rows_to_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_rows, n, W, table);
    return hash_rows(c, table, tab_layout_natural(n), W, 1, d_digests);
}

int verifier_deep_values(tvm_ctx* c, const u64* d_main_rows, int n_main, const u64* d_aux_rows, int n_aux, const u64* d_quot_rows,
                         const u64* d_row_idx, u64 q, u64 offset, u64 gen, const u64* d_w_ma, const u64* d_small, u64* d_out) {
    VerifyArgs a;
    a.main_rows = d_main_rows, a.aux_rows = d_aux_rows, a.quot_rows = d_quot_rows, a.row_idx = d_row_idx;
    a.w_ma = d_w_ma, a.small = d_small, a.offset = offset, a.gen = gen, a.n_main = n_main, a.n_aux = n_aux, a.out = d_out;
    TVM_LAUNCH(k_verifier_deep_values, dim3((unsigned)q), dim3(VF_BLOCK), 0, c->stream, a);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm

// ---------------------------------------------------------------------------------------------- the AIR at ONE row pair
// Verifier::verify evaluates every constraint on the out-of-domain row pair, with XFieldElement main rows (stark.rs:1466-1523;
// the generated evaluators are generic over the main row's field, codegen.rs:141-367).  One row pair: a walk over the lowered
// multicircuit's DAG on the host (csrc/air_circuit_data.h, the same export as the device kernels and the oracle's header).
#include "air_circuit_data.h"
namespace tvm {
static void air_dag_section(const uint32_t (*nodes)[3], uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots, const u64* consts,
                            const u64 (*xconsts)[3], const xfe* mc, const xfe* mn, const xfe* ac, const xfe* an, const xfe* ch,
                            std::vector<xfe>& val, xfe* out) {
    val.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const uint32_t kind = nodes[i][0], a = nodes[i][1], b = nodes[i][2];
        switch (kind) {
            case 0: val[i] = xfe_lift(consts[a]); break;
            case 1: val[i] = xfe_make(xconsts[a][0], xconsts[a][1], xconsts[a][2]); break;
            case 2: val[i] = mc[a]; break;
            case 3: val[i] = mn[a]; break;
            case 4: val[i] = ac[a]; break;
            case 5: val[i] = an[a]; break;
            case 6: val[i] = ch[a]; break;
            case 7: val[i] = xfe_add(val[a], val[b]); break;
            default: val[i] = xfe_mul(val[a], val[b]); break;
        }
    }
    for (uint32_t r = 0; r < n_roots; r++) out[r] = val[roots[r]];
}
}  // namespace tvm

extern "C" {
int32_t tvm_host_air_constraints(const uint64_t* h_main_cur, const uint64_t* h_aux_cur, const uint64_t* h_main_next,
                                 const uint64_t* h_aux_next, const uint64_t* h_challenges, uint64_t* h_out) {
    using namespace tvm;
    if (!h_main_cur || !h_aux_cur || !h_main_next || !h_aux_next || !h_challenges || !h_out) return TVM_ERR_INVALID_ARGUMENT;
    auto rows = [](const uint64_t* w, size_t n) {
        std::vector<xfe> v(n);
        for (size_t i = 0; i < n; i++) v[i] = xfe_make(w[3 * i], w[3 * i + 1], w[3 * i + 2]);
        return v;
    };
    const std::vector<xfe> mc = rows(h_main_cur, TVM_NUM_MAIN_COLUMNS), ac = rows(h_aux_cur, TVM_NUM_AUX_COLUMNS),
                           mn = rows(h_main_next, TVM_NUM_MAIN_COLUMNS), an = rows(h_aux_next, TVM_NUM_AUX_COLUMNS),
                           ch = rows(h_challenges, TVM_NUM_CHALLENGES);
    std::vector<xfe> val, out(TVM_NUM_QUOTIENT_WEIGHTS);
    xfe* o = out.data();
#define TVM_AIR_DAG_SECTION(S)                                                                                              \
    air_dag_section(TVM_AIR_DAG_##S##_NODES, TVM_AIR_DAG_##S##_NUM_NODES, TVM_AIR_DAG_##S##_ROOTS, TVM_AIR_DAG_##S##_NUM_ROOTS, \
                    TVM_AIR_DAG_##S##_CONSTS, TVM_AIR_DAG_##S##_XCONSTS, mc.data(), mn.data(), ac.data(), an.data(), ch.data(), val, o); \
    o += TVM_AIR_DAG_##S##_NUM_ROOTS;
    TVM_AIR_DAG_SECTION(INIT) TVM_AIR_DAG_SECTION(CONS) TVM_AIR_DAG_SECTION(TRAN) TVM_AIR_DAG_SECTION(TERM)
#undef TVM_AIR_DAG_SECTION
    static_assert(TVM_AIR_DAG_INIT_NUM_ROOTS + TVM_AIR_DAG_CONS_NUM_ROOTS + TVM_AIR_DAG_TRAN_NUM_ROOTS + TVM_AIR_DAG_TERM_NUM_ROOTS ==
                      TVM_NUM_QUOTIENT_WEIGHTS, "604 constraints");
    for (size_t i = 0; i < out.size(); i++) h_out[3 * i] = out[i].c0, h_out[3 * i + 1] = out[i].c1, h_out[3 * i + 2] = out[i].c2;
    return TVM_OK;
}
}  // extern "C"

