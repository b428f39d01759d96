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
__global__ void __launch_bounds__(256) k_air_scatter(AirArgs a, const u64* __restrict__ acc, u64* __restrict__ out, int accumulate) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.q_len) return;
    bool active;
    u64 tt, s_base, index;
    u32 rel;
    air_locate(a, t / AIR_BLOCK, (int)(t % AIR_BLOCK), tt, active, s_base, rel, index);
    u64* o = out + 3 * index;
    const u64* v = acc + 3 * t;
    if (accumulate) {
        o[0] = bfe_add(o[0], v[0]);
        o[1] = bfe_add(o[1], v[1]);
        o[2] = bfe_add(o[2], v[2]);
    } else {
        o[0] = v[0];
        o[1] = v[1];
        o[2] = v[2];
    }
}

// The same for the tiled work order through shared memory: a wavefront's 64 rows j2 are domain rows n2 X' apart, so the kernel
// above writes 24 bytes here, 24 bytes there (0.39 ms for 2^23 values that are read in 0.03).  A workgroup takes 16 rows j2 x 128
// (block j1, coset kq) pairs -- in the work order 16 consecutive values per pair, 384 bytes -- and writes, for each j2, the 128
// pairs as 128 CONSECUTIVE domain rows (runs of eight when the quotient domain has more than eight cosets).
#define AIR_SCATTER_PAIRS 128
__global__ void __launch_bounds__(256) k_air_scatter_tiles(AirArgs a, const u64* __restrict__ acc, u64* __restrict__ out, int accumulate) {
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
        d[0] = acc[3 * t], d[1] = acc[3 * t + 1], d[2] = acc[3 * t + 2];
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
    u64* zinv = (u64*)scratch(c, 14, (size_t)4 * q_len * sizeof(u64));
    u64* acc = (u64*)scratch(c, 23, (size_t)3 * q_len * sizeof(u64));
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
    const dim3 grid((unsigned)((q_len + AIR_BLOCK - 1) / AIR_BLOCK));
    a.accumulate = 0;
    bool any = false;
    for (int p = 0; p < TVM_AIR_NUM_PARTS; p++) {
        if (part_select && !(part_select >> TVM_AIR_PART_CLASS[p] & 1)) continue;
        TVM_LAUNCH(TVM_AIR_PARTS[p], grid, dim3(AIR_BLOCK), 0, c->stream, a);
        a.accumulate = 1;
        any = true;
    }
    if (!any) TVM_HIP_CHECK(c, hipMemsetAsync(acc, 0, (size_t)3 * q_len * sizeof(u64), c->stream));
    const u64 cosets = q_len / trace_len;
    if (a.tiled && (1ull << a.log_n2) >= AIR_SCATTER_PAIRS / (cosets < 8 ? cosets : 8))
        TVM_LAUNCH(k_air_scatter_tiles, dim3((unsigned)(q_len / (16 * AIR_SCATTER_PAIRS))), dim3(256), 0, c->stream, a, (const u64*)acc,
                   d_out, accumulate);
    else
        TVM_LAUNCH(k_air_scatter, dim3((unsigned)((q_len + 255) / 256)), dim3(256), 0, c->stream, a, (const u64*)acc, d_out, accumulate);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
