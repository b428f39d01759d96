// air.hip -- per-row AIR evaluation and quotient combination on gfx950.
//
// Replaces all_quotients_combined (/root/reference/triton-vm/src/table/master_table.rs:1264-1363)
// together with the build-time generated MasterAuxTable::evaluate_{initial,consistency,transition,
// terminal}_constraints (generator: triton-constraint-builder/src/codegen.rs:141-367) and the four
// zerofier-inverse codewords (master_table.rs:1194-1250).
//
// The 604 constraint polynomials (8.8k circuit nodes) are compiled code: tools/air/export.py schedules
// the degree-lowered circuit (a restatement of the reference's AIR and of its deterministic lowering)
// into straight-line single-assignment HIP, one kernel per part (air_gen_<p>.hip: the initial,
// consistency and terminal constraints together, then the transition constraints in groups of 100), and
// hipcc allocates the registers.  Each part adds  zerofier_inverse * sum_k w_k c_k  of its constraints to
// the quotient codeword; all arithmetic is exact, so the order of the parts does not matter.
// The kernels are integer-ALU bound (~9k modular multiplications per row), not HBM bound (air_eval.h).
#include "air_gen.h"
#include "kernels.h"

namespace tvm {

// The four zerofier-inverse codewords over the quotient domain (master_table.rs:1194-1250):
//   z[0] = 1/(x - 1), z[1] = 1/(x^N - 1), z[2] = (x - w^-1)/(x^N - 1), z[3] = 1/(x - w^-1),  x = offset * gen^i.
// A thread owns ZB rows t, t + T, ... (T = number of threads) and inverts their 3*ZB factors with one
// field inversion (Montgomery's trick).
#define AIR_ZB 8
struct ZerofierArgs {
    u64 q_offset, q_gen, q_len;
    u64 trace_len, trace_gen_inv;
    u64 n_threads;
    u64 step, step_n;  // gen^T, gen^(T*N)
    u64* zinv;         // [4][q_len]
};
__global__ void __launch_bounds__(256) k_zerofier_inverses(ZerofierArgs a) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_threads) return;
    u64 x = bfe_mul(a.q_offset, bfe_pow(a.q_gen, t));
    u64 xn = bfe_pow(x, a.trace_len);
    u64 f[3 * AIR_ZB], pre[3 * AIR_ZB];
    u64 run = TVM_ONE;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < AIR_ZB; j++) {
        if (t + (u64)j * a.n_threads < a.q_len) {
            f[3 * j] = bfe_sub(x, TVM_ONE);
            f[3 * j + 1] = bfe_sub(xn, TVM_ONE);
            f[3 * j + 2] = bfe_sub(x, a.trace_gen_inv);
#pragma unroll
            for (int e = 0; e < 3; e++) {
                pre[3 * j + e] = run;
                run = bfe_mul(run, f[3 * j + e]);
            }
            cnt = j + 1;
        }
        x = bfe_mul(x, a.step);
        xn = bfe_mul(xn, a.step_n);
    }
    u64 inv = bfe_inv(run);
#pragma unroll
    for (int j = AIR_ZB - 1; j >= 0; j--) {
        if (j < cnt) {
            u64 r[3];
#pragma unroll
            for (int e = 2; e >= 0; e--) {
                r[e] = bfe_mul(inv, pre[3 * j + e]);
                inv = bfe_mul(inv, f[3 * j + e]);
            }
            const u64 i = t + (u64)j * a.n_threads;
            a.zinv[i] = r[0];
            a.zinv[a.q_len + i] = r[1];
            a.zinv[2 * a.q_len + i] = bfe_mul(f[3 * j + 2], r[1]);
            a.zinv[3 * a.q_len + i] = r[2];
        }
    }
}

int all_quotients_combined(tvm_ctx* c, const u64* main_table, u64 main_rows, u64 wrap_rows, u64 main_w, const u64* aux_table,
                           u64 aux_w, u64 trace_len, u64 trace_gen, u64 q_offset, u64 q_gen, u64 q_len,
                           const u64* d_challenges, const u64* d_weights, u64* d_out, int part_select, int accumulate) {
    // part_select: 0 = every part, 1 = the parts with consistency / transition constraints only ("low degree"),
    // 2 = the others (initial / terminal constraints); accumulate: the first launched part adds to d_out as well
    if (!is_pow2(q_len) || !is_pow2(trace_len) || q_len < trace_len || main_rows % q_len)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "quotients: domain lengths");
    // the next row of quotient-domain row i is table row (i + q_len/trace_len) * (main_rows/q_len): the tables must
    // carry that many wrap rows (tables made by tvm_lde_table do) and a workgroup's rows must fit 32-bit byte offsets
    if (wrap_rows < main_rows / trace_len || (main_rows / q_len) * (AIR_BLOCK + q_len / trace_len) * main_w * 8 >= (1ull << 32))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "quotients: tables lack wrap rows for this trace length");
    u64* zinv = (u64*)scratch(c, 14, (size_t)4 * q_len * sizeof(u64));
    if (!zinv) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "zerofier inverses");
    {
        ZerofierArgs z;
        z.q_offset = q_offset;
        z.q_gen = q_gen;
        z.q_len = q_len;
        z.trace_len = trace_len;
        z.trace_gen_inv = bfe_inv(trace_gen);
        z.n_threads = (q_len + AIR_ZB - 1) / AIR_ZB;
        z.step = bfe_pow(q_gen, z.n_threads);
        z.step_n = bfe_pow(z.step, trace_len);
        z.zinv = zinv;
        TVM_LAUNCH(k_zerofier_inverses, dim3((unsigned)((z.n_threads + 255) / 256)), dim3(256), 0, c->stream, z);
    }
    AirArgs a;
    a.main_table = main_table;
    a.aux_table = aux_table;
    a.main_w = main_w;
    a.aux_w = aux_w;
    a.stride = main_rows / q_len;
    a.q_len = q_len;
    a.unit = q_len / trace_len;
    a.challenges = d_challenges;
    a.weights = d_weights;
    a.zinv = zinv;
    a.out = d_out;
    const dim3 grid((unsigned)((q_len + AIR_BLOCK - 1) / AIR_BLOCK));
    a.accumulate = accumulate;
    for (int p = 0; p < TVM_AIR_NUM_PARTS; p++) {
        if (part_select == 1 && !TVM_AIR_PART_LOW_DEGREE[p]) continue;
        if (part_select == 2 && TVM_AIR_PART_LOW_DEGREE[p]) continue;
        TVM_LAUNCH(TVM_AIR_PARTS[p], grid, dim3(AIR_BLOCK), 0, c->stream, a);
        a.accumulate = 1;
    }
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
