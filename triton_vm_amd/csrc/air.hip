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
//   z[0] = 1/(x - 1), z[1] = 1/(x^N - 1), z[2] = (x - w^-1)/(x^N - 1), z[3] = 1/(x - w^-1),  x = offset * gen^i,
// stored in the WORK order of the AIR kernels (air_eval.h: air_locate), so that their reads are coalesced.  A thread owns
// AIR_ZB consecutive work items and inverts their 3*ZB factors with one field inversion (Montgomery's trick); consecutive
// work items are almost always consecutive rows j2 of one block, i.e. domain indices `index_step` apart: x is a running
// product, recomputed from the index where the run breaks.
#define AIR_ZB 8
struct ZerofierArgs {
    AirArgs air;       // the mapping work item -> domain index
    u64 q_offset, q_gen;
    u64 trace_len, trace_gen_inv;
    u64 index_step, x_step;  // domain-index distance of rows j2, j2 + 1 of a block (X' * n2); gen^index_step
    u64* zinv;         // [4][q_len]
};
__global__ void __launch_bounds__(256) k_zerofier_inverses(ZerofierArgs a) {
    const u64 z = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 t0 = z * AIR_ZB;
    if (t0 >= a.air.q_len) return;
    u64 f[3 * AIR_ZB], pre[3 * AIR_ZB];
    u64 run = TVM_ONE, x = 0, xn = 0, prev = 0;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < AIR_ZB; j++) {
        const u64 t = t0 + (u64)j;
        if (t < a.air.q_len) {
            bool active;
            u64 tt, s_base, index;
            u32 rel;
            air_locate(a.air, t / AIR_BLOCK, (int)(t % AIR_BLOCK), tt, active, s_base, rel, index);
            if (j > 0 && index == prev + a.index_step) {
                x = bfe_mul(x, a.x_step);   // x^N is unchanged: gen^(index_step * N) = 1
            } else {
                x = bfe_mul(a.q_offset, bfe_pow(a.q_gen, index));
                xn = bfe_pow(x, a.trace_len);
            }
            prev = index;
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
            const u64 i = t0 + (u64)j;
            a.zinv[i] = r[0];
            a.zinv[a.air.q_len + i] = r[1];
            a.zinv[2 * a.air.q_len + i] = bfe_mul(f[3 * j + 2], r[1]);
            a.zinv[3 * a.air.q_len + i] = r[2];
        }
    }
}

// the quotient values from the work order of the parts into the order of the domain: out[index(t)] (+)= acc[t]
// (n_acc accumulators 3 * q_len words apart: the parts of a forked evaluation -- all_quotients_combined below -- add into one
// accumulator per lane; field addition is exact, the sum does not depend on how the parts were grouped)
__global__ void __launch_bounds__(256) k_air_scatter(AirArgs a, const u64* __restrict__ acc, int n_acc, u64* __restrict__ out, int accumulate) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.q_len) return;
    bool active;
    u64 tt, s_base, index;
    u32 rel;
    air_locate(a, t / AIR_BLOCK, (int)(t % AIR_BLOCK), tt, active, s_base, rel, index);
    u64* o = out + 3 * index;
    const u64* v = acc + 3 * t;
    u64 v0 = v[0], v1 = v[1], v2 = v[2];
    for (int k = 1; k < n_acc; k++) {
        v += 3 * a.q_len;
        v0 = bfe_add(v0, v[0]), v1 = bfe_add(v1, v[1]), v2 = bfe_add(v2, v[2]);
    }
    if (accumulate) {
        o[0] = bfe_add(o[0], v0);
        o[1] = bfe_add(o[1], v1);
        o[2] = bfe_add(o[2], v2);
    } else {
        o[0] = v0;
        o[1] = v1;
        o[2] = v2;
    }
}

// The same for the tiled work order through shared memory: a wavefront's 64 rows j2 are domain rows n2 X' apart, so the kernel
// above writes 24 bytes here, 24 bytes there (0.39 ms for 2^23 values that are read in 0.03).  A workgroup takes 16 rows j2 x 128
// (block j1, coset kq) pairs -- in the work order 16 consecutive values per pair, 384 bytes -- and writes, for each j2, the 128
// pairs as 128 CONSECUTIVE domain rows (runs of eight when the quotient domain has more than eight cosets).
#define AIR_SCATTER_PAIRS 128
__global__ void __launch_bounds__(256) k_air_scatter_tiles(AirArgs a, const u64* __restrict__ acc, int n_acc, u64* __restrict__ out, int accumulate) {
    constexpr int WPB_LOG = TVM_AIR_BLOCK_LOG - 6;
    constexpr int RS = 3 * AIR_SCATTER_PAIRS + 1;
    __shared__ u64 so[16 * RS];
    const int tid = threadIdx.x, p = tid & 15, q = tid >> 4;
    const int log_kx = a.log_xq < 3 ? a.log_xq : 3, kx = 1 << log_kx;
    const int j1_per_tile = AIR_SCATTER_PAIRS >> log_kx;
    u64 b = blockIdx.x;
    const u64 P = (b & ((a.n1 >> 4) - 1)) << 4;   // first of the 16 rows j2
    b >>= a.log_n1 - 4;
    const u64 tiles_j1 = (1ull << a.log_n2) / j1_per_tile;
    const u64 J = (b % tiles_j1) * j1_per_tile, k0 = (b / tiles_j1) << log_kx;
    const int log_tiles = a.log_n - TVM_AIR_BLOCK_LOG, log_groups = a.log_n2 - WPB_LOG;
    for (int it = 0; it < AIR_SCATTER_PAIRS / 16; it++) {
        const int pair = q + 16 * it;
        const u64 j1 = J + (pair >> log_kx), kq = k0 + (pair & (kx - 1)), j2 = P + p;
        const u64 block = (kq << log_tiles) + ((j2 >> 6) << log_groups) + (j1 >> WPB_LOG);   // air_locate, inverted
        const u64 t = block * AIR_BLOCK + ((j1 & ((1 << WPB_LOG) - 1)) << 6) + (j2 & 63);
        u64* d = so + p * RS + 3 * pair;
        const u64* v = acc + 3 * t;
        u64 v0 = v[0], v1 = v[1], v2 = v[2];
        for (int k = 1; k < n_acc; k++) {
            v += 3 * a.q_len;
            v0 = bfe_add(v0, v[0]), v1 = bfe_add(v1, v[1]), v2 = bfe_add(v2, v[2]);
        }
        d[0] = v0, d[1] = v1, d[2] = v2;
    }
    __syncthreads();
    for (int e = tid; e < 16 * 3 * AIR_SCATTER_PAIRS; e += 256) {
        const int pp = e / (3 * AIR_SCATTER_PAIRS), word = e % (3 * AIR_SCATTER_PAIRS), pair = word / 3, comp = word % 3;
        const u64 j = J + (pair >> log_kx) + ((P + pp) << a.log_n2);
        const u64 i = (j << a.log_xq) + k0 + (pair & (kx - 1));
        u64* o = out + 3 * i + comp;
        *o = accumulate ? bfe_add(*o, so[pp * RS + word]) : so[pp * RS + word];
    }
}

bool air_parts_fork(const tvm_ctx* c, u64 q_len) { return (q_len + AIR_BLOCK - 1) / AIR_BLOCK <= c->air_fork_max_workgroups; }

bool fork_lanes(tvm_ctx* c) {
    if (c->fork_ready) return true;
    int cur = -1;   // streams belong to the device that is current when they are created
    if ((hipGetDevice(&cur) != hipSuccess || cur != c->device) && hipSetDevice(c->device) != hipSuccess) return false;
    hipStream_t s[3] = {};
    hipEvent_t e[4] = {};
    bool ok = true;
    for (int k = 0; k < 3 && ok; k++) ok = hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < 4 && ok; k++) ok = hipEventCreateWithFlags(&e[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        for (hipStream_t x : s)
            if (x) hipStreamDestroy(x);
        for (hipEvent_t x : e)
            if (x) hipEventDestroy(x);
        return false;
    }
    for (int k = 0; k < 3; k++) c->fork[k] = s[k], c->fork_done[k] = e[k];
    c->fork_ready = e[3];
    return true;
}

int all_quotients_combined(tvm_ctx* c, const u64* main_table, const TabLayout& layout, u64 main_w, const u64* aux_table,
                           u64 aux_w, u64 trace_len, u64 trace_gen, u64 q_offset, u64 q_gen, u64 q_len,
                           const u64* d_challenges, const u64* d_weights, u64* d_out, int part_select, int accumulate) {
    // part_select: 0 = every part, else a mask of part classes (air_gen.h: TVM_AIR_PART_CLASS; bit c = the parts of class c);
    // accumulate: the quotient values are added to d_out
    const u64 rows = layout.rows();
    if (!is_pow2(q_len) || !is_pow2(trace_len) || q_len < trace_len || rows % q_len)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "quotients: domain lengths");
    // the quotient domain must be whole cosets of the trace domain as the tables hold them (tables made by tvm_lde_table
    // from a trace of this length do), with the successor blocks behind every coset
    const u64 kstep = rows / q_len;
    if (layout.n1 * layout.n2 != trace_len || layout.n2 < 2 || kstep > layout.X || layout.X % kstep || layout.pitch < (layout.n2 + 1) * layout.n1)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "quotients: the tables are not extensions of a trace of this length");
    AirArgs a;
    a.main_table = main_table;
    a.aux_table = aux_table;
    a.main_w = main_w;
    a.aux_w = aux_w;
    a.q_len = q_len;
    a.n1 = layout.n1;
    a.log_n1 = layout.log_n1;
    a.log_n2 = layout.log_n2;
    a.log_n = layout.log_n1 + layout.log_n2;
    a.log_xq = ilog2(q_len / trace_len);
    a.coset_rows = kstep * layout.pitch;
    constexpr u64 WPB = AIR_BLOCK / 64;
    a.tiled = (layout.n1 % 64 == 0 && layout.n2 % WPB == 0) ? 1 : 0;
    const u64 wider = main_w > aux_w ? main_w : aux_w;
    const u64 reach = a.tiled ? (WPB + 1) * layout.n1 + TVM_RB : layout.storage_rows() + layout.n1 + TVM_RB;  // rows a 32-bit lane offset must span
    if (reach * wider * 8 >= (1ull << 32))
        return set_error(c, TVM_ERR_UNSUPPORTED, "quotients: table shape beyond the 32-bit lane offsets of the AIR kernels");
    // Fork: a part on a short quotient domain is a few workgroups that run for the latency of its ~1000 dependent multiplications
    // (80-130 us) whatever their number; ten of them one behind the other are a millisecond of a proof that takes eight.  While one
    // part leaves most of the chip empty (grid <= air_fork_max_workgroups) the selected parts go out on the context's stream and its
    // three fork lanes, each lane adding into an accumulator of its own (longest part first, each to the lane with the least work so
    // far: TVM_AIR_PART_COST of air_gen.h), and the scatter adds the accumulators.  Exact arithmetic: the same words.
    const dim3 grid((unsigned)((q_len + AIR_BLOCK - 1) / AIR_BLOCK));
    int selected[TVM_AIR_NUM_PARTS], n_selected = 0;
    for (int p = 0; p < TVM_AIR_NUM_PARTS; p++)
        if (!part_select || (part_select >> TVM_AIR_PART_CLASS[p] & 1)) selected[n_selected++] = p;
    int n_lanes = 1;
    if (n_selected > 1 && air_parts_fork(c, q_len) && fork_lanes(c)) n_lanes = n_selected < 4 ? n_selected : 4;
    u64* zinv = (u64*)scratch(c, 14, (size_t)4 * q_len * sizeof(u64));
    u64* acc = (u64*)scratch(c, 23, (size_t)n_lanes * 3 * q_len * sizeof(u64));
    if (!zinv || !acc) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "zerofier inverses");
    a.challenges = d_challenges;
    a.weights = d_weights;
    a.zinv = zinv;
    a.out = acc;
    {
        ZerofierArgs z;
        z.air = a;
        z.q_offset = q_offset;
        z.q_gen = q_gen;
        z.trace_len = trace_len;
        z.trace_gen_inv = bfe_inv(trace_gen);
        z.index_step = (q_len / trace_len) << layout.log_n2;
        z.x_step = bfe_pow(q_gen, z.index_step);
        z.zinv = zinv;
        const u64 n_threads = (q_len + AIR_ZB - 1) / AIR_ZB;
        TVM_LAUNCH(k_zerofier_inverses, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, c->stream, z);
    }
    if (n_lanes == 1) {
        a.accumulate = 0;
        for (int s = 0; s < n_selected; s++) {
            TVM_LAUNCH(TVM_AIR_PARTS[selected[s]], grid, dim3(AIR_BLOCK), 0, c->stream, a);
            a.accumulate = 1;
        }
        if (!n_selected) TVM_HIP_CHECK(c, hipMemsetAsync(acc, 0, (size_t)3 * q_len * sizeof(u64), c->stream));
    } else {
        for (int i = 1; i < n_selected; i++)   // by cost, descending (insertion sort of at most ten)
            for (int j = i; j > 0 && TVM_AIR_PART_COST[selected[j]] > TVM_AIR_PART_COST[selected[j - 1]]; j--) {
                const int t = selected[j];
                selected[j] = selected[j - 1];
                selected[j - 1] = t;
            }
        int load[4] = {0, 0, 0, 0}, started[4] = {0, 0, 0, 0};
        // (a failure between fork and join must not leave a lane running over tables the caller is about to release: the lanes are
        // drained before the error goes up)
        hipError_t fe = hipEventRecord(c->fork_ready, c->stream);   // the tables, the challenges, the zerofier inverses
        for (int l = 1; l < n_lanes && fe == hipSuccess; l++) fe = hipStreamWaitEvent(c->fork[l - 1], c->fork_ready, 0);
        for (int s = 0; s < n_selected && fe == hipSuccess; s++) {
            int l = 0;
            for (int k = 1; k < n_lanes; k++)
                if (load[k] < load[l]) l = k;
            load[l] += TVM_AIR_PART_COST[selected[s]];
            a.out = acc + (u64)l * 3 * q_len;
            a.accumulate = started[l];
            started[l] = 1;
            TVM_LAUNCH(TVM_AIR_PARTS[selected[s]], grid, dim3(AIR_BLOCK), 0, l ? c->fork[l - 1] : c->stream, a);
            fe = hipGetLastError();
        }
        a.out = acc;
        for (int l = 1; l < n_lanes && fe == hipSuccess; l++) {
            fe = hipEventRecord(c->fork_done[l - 1], c->fork[l - 1]);
            if (fe == hipSuccess) fe = hipStreamWaitEvent(c->stream, c->fork_done[l - 1], 0);
        }
        if (fe != hipSuccess) {
            for (int l = 1; l < n_lanes; l++) (void)hipStreamSynchronize(c->fork[l - 1]);
            return set_error(c, fe == hipErrorOutOfMemory ? TVM_ERR_OUT_OF_MEMORY : TVM_ERR_DEVICE, "quotients: fork lanes");
        }
    }
    const u64 cosets = q_len / trace_len;
    if (a.tiled && (1ull << a.log_n2) >= AIR_SCATTER_PAIRS / (cosets < 8 ? cosets : 8))
        TVM_LAUNCH(k_air_scatter_tiles, dim3((unsigned)(q_len / (16 * AIR_SCATTER_PAIRS))), dim3(256), 0, c->stream, a, (const u64*)acc,
                   n_lanes, d_out, accumulate);
    else
        TVM_LAUNCH(k_air_scatter, dim3((unsigned)((q_len + 255) / 256)), dim3(256), 0, c->stream, a, (const u64*)acc, n_lanes, d_out, accumulate);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
