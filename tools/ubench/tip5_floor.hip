// tip5_floor.hip -- how far is the product's row hashing from the floor of its own instruction stream?  (round 6; the closing
// measurement the review of round 5 asked for.)
//
// One process, one box, three things timed:
//   (1) the Tip5 permutation of csrc/tip5.h (tip5_permute_mfma: four lanes per permutation, MDS layer on the matrix cores) on a
//       state that never leaves the registers -- no table, no absorb, no digest store: the pure issue time of the round's
//       instruction stream at the product's occupancy (256 work-items per workgroup, six wavefronts per SIMD);
//   (2) the three parts of that round on their own -- the twelve x^7 chains' share of a lane (three per lane), the MDS layer
//       (operand preparation + twelve v_mfma_i32_16x16x64_i8 + the recombination of the ten partial sums), the split-and-lookup
//       word -- to show that the parts add up to the whole, i.e. that nothing in the round waits for anything;
//   (3) the product itself: tvm_hash_rows (k_hash_rows_mfma) of libtriton_hip.so over a 2^23-row, 379-column table made by
//       tvm_lde_table from synthetic data -- the main-table launch of a 2^20-row proof.
// Printed: nanoseconds of one SIMD per wavefront-round for each, and (3) / (1) - 1 = what the product pays on top of the bare permutations (loads of the rows, padding, digest stores,
// launch ramp).  The x^7 chains and the recombination are the instruction counts of DESIGN.md 4.2; what this file adds is
// that the sum is reached.
//
// build + run (tools/r06_tip5_floor.sh):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I triton_vm_amd/csrc \
//     -mllvm -amdgpu-mfma-vgpr-form tools/ubench/tip5_floor.hip -L triton_vm_amd -ltriton_hip -Wl,-rpath,$PWD/triton_vm_amd -o tools/ubench/bin/tip5_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tip5.h"
#include "triton_hip.h"

#define CHECK(e)                                                                        \
    do {                                                                                \
        hipError_t err_ = (e);                                                          \
        if (err_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)

// MODE 0: the whole round.  1: the power maps only.  2: the MDS layer only.  3: the lookup word only.
template <int MODE>
__device__ __forceinline__ void round_parts(u64 (&st)[4], const Tip5MfmaOperands& m, int g, const unsigned char* lut, const int* ctab, int r) {
    if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 1; t < 4; t++) st[t] = tip5_pow7(st[t]);
        return;
    }
    if constexpr (MODE == 3) {
        st[0] = tip5_sbox_lookup(st[0], lut) ^ 0x8080808080808080ull;   // (the table is the lowered one: keep the word a word)
        return;
    }
    tvm_v4i d[TIP5_MFMA_POSITIONS];
#pragma unroll
    for (int c = 0; c < TIP5_MFMA_POSITIONS; c++) {
        const int* cp = ctab + ((r * TIP5_MFMA_POSITIONS + c) * 4 + g) * 4;
#pragma unroll
        for (int v = 0; v < 4; v++) d[c][v] = cp[v];
    }
    const u32 pad = 0x80808080u;
    tvm_v4i lo, hi;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        lo[t] = (int)((u32)st[t] ^ pad);
        hi[t] = (int)((u32)(st[t] >> 32) ^ pad);
    }
#pragma unroll
    for (int s = 0; s < TIP5_MFMA_SHIFTS; s++) d[s] = TVM_MFMA_I8(m.a[s], lo, d[s]);
#pragma unroll
    for (int s = 0; s < TIP5_MFMA_SHIFTS; s++) d[s + 4] = TVM_MFMA_I8(m.a[s], hi, d[s + 4]);
    tip5_mfma_recombine(d, st);
}

template <int MODE>
__global__ void __launch_bounds__(256, 6) k_rounds(u64* __restrict__ out, int n_perm) {
    __shared__ unsigned char lut[256];
    __shared__ int ctab[TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16];
    const int tid = threadIdx.x;
    for (int i = tid; i < TIP5_ROUNDS * TIP5_MFMA_POSITIONS * 16; i += blockDim.x) ctab[i] = d_tip5_mfma_table.v[i];
    tip5_stage_lut_lowered(lut, tid, blockDim.x);
    const int lane = tid & 63, g = lane >> 4;
    const Tip5MfmaOperands a = tip5_mfma_matrix_operands(lane);
    u64 st[4];
    for (int t = 0; t < 4; t++) st[t] = (u64)(blockIdx.x * 256 + tid) * 0x9E3779B97F4A7C15ull + t;
    for (int p = 0; p < n_perm; p++) {
        if constexpr (MODE == 0) tip5_permute_mfma(st, a, g, lut, ctab);
        else
            for (int r = 0; r < TIP5_ROUNDS; r++) round_parts<MODE>(st, a, g, lut, ctab, r);
    }
    out[(u64)blockIdx.x * 256 + tid] = st[0] ^ st[1] ^ st[2] ^ st[3];
}

struct Timing {
    double ms, simd_ns_per_wave_round;
};
template <int MODE>
Timing run(int workgroups, int n_perm, int n_simds) {
    u64* out;
    CHECK(hipMalloc(&out, (size_t)workgroups * 256 * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_rounds<MODE>, dim3(workgroups), dim3(256), 0, 0, out, n_perm);   // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_rounds<MODE>, dim3(workgroups), dim3(256), 0, 0, out, n_perm);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CHECK(hipFree(out));
    const double wave_rounds = (double)workgroups * 4 * n_perm * TIP5_ROUNDS;
    Timing t;
    t.ms = best;
    t.simd_ns_per_wave_round = best * 1e6 * n_simds / wave_rounds;
    return t;
}

int main(int argc, char** argv) {
    const int log2_rows = argc > 1 ? std::atoi(argv[1]) : 20;   // trace rows of the product measurement (extended 8x)
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount, n_simds = 4 * n_cus;
    std::printf("device: %s, %d CUs, clock %d MHz (reported)\n", prop.name, n_cus, prop.clockRate / 1000);
    // (1), (2): 24 wavefronts per CU resident (six workgroups of four), 16 waves of workgroups per CU, 32 permutations each
    const int workgroups = n_cus * 6 * 16, n_perm = 32;
    const Timing whole = run<0>(workgroups, n_perm, n_simds), x7 = run<1>(workgroups, n_perm, n_simds), mds = run<2>(workgroups, n_perm, n_simds),
                 look = run<3>(workgroups, n_perm, n_simds);
    std::printf("registers-only, %d workgroups x 256 work-items x %d permutations, best of 5:\n", workgroups, n_perm);
    std::printf("  whole round (tip5_permute_mfma)        %8.3f ms   %7.1f SIMD-ns per wavefront-round\n", whole.ms, whole.simd_ns_per_wave_round);
    std::printf("  three x^7 per lane                     %8.3f ms   %7.1f\n", x7.ms, x7.simd_ns_per_wave_round);
    std::printf("  MDS layer (prep + 12 MFMA + recombine) %8.3f ms   %7.1f\n", mds.ms, mds.simd_ns_per_wave_round);
    std::printf("  split-and-lookup word                  %8.3f ms   %7.1f\n", look.ms, look.simd_ns_per_wave_round);
    const double parts = x7.simd_ns_per_wave_round + mds.simd_ns_per_wave_round + look.simd_ns_per_wave_round;
    std::printf("  sum of the parts                                    %7.1f   (whole / parts = %.3f)\n", parts, whole.simd_ns_per_wave_round / parts);

    // (3): the product
    tvm_ctx* ctx = nullptr;
    if (tvm_ctx_create(0, nullptr, &ctx) != TVM_OK) return std::fprintf(stderr, "tvm_ctx_create failed\n"), 1;
    const uint64_t n = 1ull << log2_rows, cols = 379, h = 198, L = 8 * n;
    uint64_t *trace = nullptr, *rnd = nullptr, *digests = nullptr;
    if (tvm_malloc(ctx, n * cols * 8, (void**)&trace) != TVM_OK || tvm_malloc(ctx, h * cols * 8, (void**)&rnd) != TVM_OK ||
        tvm_malloc(ctx, L * 5 * 8, (void**)&digests) != TVM_OK)
        return std::fprintf(stderr, "tvm_malloc: %s\n", tvm_last_error(ctx)), 1;
    tvm_synthetic_fill(ctx, trace, n * cols, 1);
    tvm_synthetic_fill(ctx, rnd, h * cols, 2);
    uint64_t gen_n = 0, gen_l = 0, off = 0;
    {   // the domains: the subgroup generators from the library's own field helpers (tvm_field_op cannot make them: use the host side)
        // p = 2^64 - 2^32 + 1, generator 7, primitive 2^32-th root 1753635133440165772 (twenty-first); Montgomery words via R = 2^64 mod p
        auto mulmod = [](unsigned __int128 a, unsigned __int128 b) { return (uint64_t)(a * b % (unsigned __int128)0xFFFFFFFF00000001ull); };
        auto powmod = [&](uint64_t b, uint64_t e) {
            uint64_t r = 1;
            for (; e; e >>= 1, b = mulmod(b, b))
                if (e & 1) r = mulmod(r, b);
            return r;
        };
        const uint64_t root32 = 1753635133440165772ull, R = 0xFFFFFFFFull;
        auto root_of = [&](uint64_t len) { return powmod(root32, (1ull << 32) / len); };
        gen_n = mulmod(root_of(n), R);
        gen_l = mulmod(root_of(L), R);
        off = mulmod(7, R);
    }
    const tvm_domain trace_dom{0xFFFFFFFFull, gen_n, n}, eval_dom{off, gen_l, L};
    tvm_table* table = nullptr;
    if (tvm_lde_table(ctx, 1, trace, n, cols, rnd, h, trace_dom, eval_dom, &table) != TVM_OK)
        return std::fprintf(stderr, "tvm_lde_table: %s\n", tvm_last_error(ctx)), 1;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        float ms = 0;
        tvm_timer_start(ctx);
        if (tvm_hash_rows(ctx, table, L, digests) != TVM_OK) return std::fprintf(stderr, "tvm_hash_rows: %s\n", tvm_last_error(ctx)), 1;
        tvm_timer_stop(ctx, &ms);
        if (rep && ms < best) best = ms;
    }
    const double perms = (double)(cols / 10 + 1), wave_rounds = (double)L / 16 * perms * TIP5_ROUNDS;
    const double prod = best * 1e6 * n_simds / wave_rounds;
    std::printf("product: tvm_hash_rows of %llu rows x %llu columns (%.0f permutations per row), best of 3: %.3f ms   %7.1f SIMD-ns per wavefront-round\n",
                (unsigned long long)L, (unsigned long long)cols, perms, best, prod);
    std::printf("product / registers-only permutation = %.4f  (the absorb's loads, padding, digest stores and the launch's ramp: %+.1f %%)\n",
                prod / whole.simd_ns_per_wave_round, (prod / whole.simd_ns_per_wave_round - 1) * 100);
    tvm_table_free(ctx, table);
    tvm_ctx_destroy(ctx);
    return 0;
}
