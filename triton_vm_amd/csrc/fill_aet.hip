// fill_aet.hip -- MasterMainTable::new's table fills on the device (the `fill` half): from the
// algebraic execution trace as the VM recorded it to the 149 original columns of the main trace.
//
// Replaces /root/reference/triton-vm/src/table/master_table.rs:881-931 with the per-table `fill`s of
// table/op_stack.rs:186-203, ram.rs:64-84 + 214-262, jump_stack.rs:93-142, processor.rs:44-68, program.rs:33-75,
// hash.rs:249-278, cascade.rs:42-58, lookup.rs:84-112, u32.rs:101-125 + 196-291.  The AET arrives the way
// AlgebraicExecutionTrace holds it (aet.rs:41-96): row-major trace arrays of Montgomery words, plain-integer
// multiplicities.  What is sequential in the reference becomes:
//   * the memory-like tables (op stack, RAM, jump stack) are STABLE radix sorts by one key -- the traces are recorded
//     in clock order, so a stable sort by stack pointer / RAM pointer / jump-stack pointer is the reference's
//     (pointer, clock) order -- followed by row-parallel kernels for the clock-jump differences (a histogram that
//     becomes the processor table's lookup multiplicities), the RAM pointer-difference inverses and the Bezout
//     coefficients (an exclusive prefix count of "pointer changed" selects the coefficient);
//   * the u32 table's variable-length sections are one work-item per entry at host-computed offsets;
//   * everything else is a transpose into the column-major trace.
// The Bezout coefficient POLYNOMIALS of the RAM table (ram.rs:152-207: zerofier, formal derivative, interpolation,
// division -- twenty-first's fast polynomial arithmetic) stay on the host and are an input.
#include "air_columns.h"
#include "context.h"
#include "kernels.h"
#include "tip5_tables.h"

#ifndef TVM_EMU
#include <hipcub/hipcub.hpp>
#else
#include <algorithm>
#include <numeric>
#include <vector>
#endif

namespace tvm {

#define FA_MONT(v) ((u64)(v) * 0xFFFFFFFFull)
TVM_D u64 fa_value(u64 w) { return bfe_mul(w, 1); }
#ifdef TVM_EMU
static const unsigned char d_fa_lut[256] = {TVM_TIP5_LUT_LIST};
static inline void fa_atomic_add(u64* p, u64 v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }  // the emulation runs workgroups on several threads
static inline void fa_atomic_inc_local(unsigned* p) { *p += 1; }                                  // ... and a workgroup's work-items on one
#else
static __device__ const unsigned char d_fa_lut[256] = {TVM_TIP5_LUT_LIST};
static __device__ __forceinline__ void fa_atomic_add(u64* p, u64 v) { atomicAdd((unsigned long long*)p, (unsigned long long)v); }
static __device__ __forceinline__ void fa_atomic_inc_local(unsigned* p) { atomicAdd(p, 1u); }
#endif

// row-major [len][w] -> columns col0 .. col0+w-1 of the column-major trace, rows row0 .. row0+len-1
__global__ void k_fa_transpose(const u64* __restrict__ src, u64 len, int w, u64* __restrict__ main, u64 n, int col0, u64 row0) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= len * (u64)w) return;
    const u64 r = e / (u64)w;
    const int k = (int)(e % (u64)w);
    main[(u64)(col0 + k) * n + row0 + r] = src[e];
}
__global__ void k_fa_fill_column(u64* __restrict__ main, u64 n, int col, u64 row0, u64 len, u64 value) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < len) main[(u64)col * n + row0 + r] = value;
}
// program table (program.rs:33-75)
__global__ void k_fa_program(const u64* __restrict__ words, const u32* __restrict__ mult, u64 program_len, u64 table_len,
                             u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= table_len) return;
    const u64 idx = r % 10;
    main[(u64)MC_PROGRAM_ADDRESS * n + r] = bfe_from_u64(r);
    main[(u64)MC_PROGRAM_INSTRUCTION * n + r] = r < program_len ? words[r] : (r == program_len ? FA_MONT(1) : 0);
    main[(u64)MC_PROGRAM_LOOKUP_MULTIPLICITY * n + r] = r < program_len ? bfe_from_u64(mult[r]) : 0;
    main[(u64)MC_PROGRAM_INDEX_IN_CHUNK * n + r] = FA_MONT(idx);
    main[(u64)MC_PROGRAM_MAX_MINUS_INDEX_IN_CHUNK_INV * n + r] = idx == 9 ? 0 : bfe_inv(FA_MONT(9 - idx));
    main[(u64)MC_PROGRAM_IS_HASH_INPUT_PADDING * n + r] = r < program_len ? 0 : FA_MONT(1);
    main[(u64)MC_PROGRAM_IS_TABLE_PADDING * n + r] = 0;
}
// sort keys: the canonical value of one column of a row-major trace
__global__ void k_fa_keys(const u64* __restrict__ src, u64 len, int w, int key_col, u64* __restrict__ keys, u64* __restrict__ idx) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    keys[r] = fa_value(src[r * (u64)w + key_col]);
    idx[r] = r;
}
// sorted gather of `count` columns of a row-major trace into the main trace (column map: dst col = col0 + k <- src col k)
__global__ void k_fa_gather(const u64* __restrict__ src, const u64* __restrict__ order, u64 len, int w, int count,
                            u64* __restrict__ main, u64 n, int col0) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= len * (u64)count) return;
    const u64 r = e / (u64)count;
    const int k = (int)(e % (u64)count);
    main[(u64)(col0 + k) * n + r] = src[order[r] * (u64)w + k];
}
// jump stack rows (clk, ci, jsp, jso, jsd) from the processor trace in sorted order (jump_stack.rs:93-126)
__global__ void k_fa_jump_stack(const u64* __restrict__ proc, const u64* __restrict__ order, u64 len, u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    const u64* p = proc + order[r] * 39;
    main[(u64)MC_JUMPSTACK_CLK * n + r] = p[MC_PROCESSOR_CLK - MC_PROCESSOR_CLK];
    main[(u64)MC_JUMPSTACK_CI * n + r] = p[MC_PROCESSOR_CI - MC_PROCESSOR_CLK];
    main[(u64)MC_JUMPSTACK_JSP * n + r] = p[MC_PROCESSOR_JSP - MC_PROCESSOR_CLK];
    main[(u64)MC_JUMPSTACK_JSO * n + r] = p[MC_PROCESSOR_JSO - MC_PROCESSOR_CLK];
    main[(u64)MC_JUMPSTACK_JSD * n + r] = p[MC_PROCESSOR_JSD - MC_PROCESSOR_CLK];
}
// clock jump differences of a sorted memory-like table: rows r-1, r with the same pointer contribute clk_r - clk_{r-1}
// (op_stack.rs:261-281, ram.rs:236-248, jump_stack.rs:128-141); hist[d] counts them
// Most differences are tiny and equal (a loop body touches the same stack depth every few cycles): one global atomic per
// row put 600 000 increments on a handful of addresses (8 ms at 2^20 rows).  Each workgroup counts the small differences in
// LDS first and adds its non-zero bins to the global histogram once.
#define FA_HIST_LOCAL 1024
__global__ void k_fa_clock_jump_differences(const u64* __restrict__ main, u64 n, int clk_col, int ptr_col, u64 len, u64* __restrict__ hist,
                                            u64 hist_len) {
    __shared__ unsigned local[FA_HIST_LOCAL];
    for (int i = threadIdx.x; i < FA_HIST_LOCAL; i += blockDim.x) local[i] = 0;
    __syncthreads();
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r != 0 && r < len && main[(u64)ptr_col * n + r] == main[(u64)ptr_col * n + r - 1]) {
        const u64 d = fa_value(bfe_sub(main[(u64)clk_col * n + r], main[(u64)clk_col * n + r - 1]));
        if (d < FA_HIST_LOCAL && d < hist_len) fa_atomic_inc_local(local + d);
        else if (d < hist_len) fa_atomic_add(hist + d, 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FA_HIST_LOCAL; i += blockDim.x)
        if (local[i]) fa_atomic_add(hist + i, local[i]);
}
__global__ void k_fa_multiplicities(const u64* __restrict__ hist, u64 len, u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < len) main[(u64)MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY * n + r] = bfe_from_u64(hist[r]);
}
// RAM: flags[r] = 1 where the pointer differs from the previous row (r >= 1)
__global__ void k_fa_ram_flags(const u64* __restrict__ main, u64 n, u64 len, u64* __restrict__ flags) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    flags[r] = (r > 0 && main[(u64)MC_RAM_RAM_POINTER * n + r] != main[(u64)MC_RAM_RAM_POINTER * n + r - 1]) ? 1 : 0;
}
// the distinct RAM pointers of the sorted table, in table order: row r starts pointer number changes_inclusive[r]
__global__ void k_fa_distinct_pointers(const u64* __restrict__ main, u64 n, u64 len, const u64* __restrict__ flags,
                                       const u64* __restrict__ changes_inclusive, u64* __restrict__ out) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < len && (r == 0 || flags[r])) out[changes_inclusive[r]] = main[(u64)MC_RAM_RAM_POINTER * n + r];
}
// make_ram_table_consistent (ram.rs:214-262): the row's Bezout coefficients are popped from the END of the coefficient
// vectors, one pair per distinct pointer; the inverse of the pointer difference to the NEXT row
__global__ void k_fa_ram_consistent(u64* __restrict__ main, u64 n, u64 len, const u64* __restrict__ changes_inclusive,
                                    const u64* __restrict__ bc0, const u64* __restrict__ bc1, u64 n_unique) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    const u64 k = n_unique - 1 - changes_inclusive[r];
    main[(u64)MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT0 * n + r] = bc0[k];
    main[(u64)MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT1 * n + r] = bc1[k];
    u64 inv = 0;
    if (r + 1 < len) {
        const u64 d = bfe_sub(main[(u64)MC_RAM_RAM_POINTER * n + r + 1], main[(u64)MC_RAM_RAM_POINTER * n + r]);
        inv = d ? bfe_inv(d) : 0;
    }
    main[(u64)MC_RAM_INVERSE_OF_RAMP_DIFFERENCE * n + r] = inv;
}
// cascade table (cascade.rs:42-58): entries [len][2] = (16-bit limb, multiplicity), plain integers
__global__ void k_fa_cascade(const u64* __restrict__ entries, u64 len, u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= len) return;
    const u64 limb = entries[2 * r], lo = limb & 0xFF, hi = (limb >> 8) & 0xFF;
    main[(u64)MC_CASCADE_IS_PADDING * n + r] = 0;
    main[(u64)MC_CASCADE_LOOK_IN_HI * n + r] = FA_MONT(hi);
    main[(u64)MC_CASCADE_LOOK_IN_LO * n + r] = FA_MONT(lo);
    main[(u64)MC_CASCADE_LOOK_OUT_HI * n + r] = FA_MONT(d_fa_lut[hi]);
    main[(u64)MC_CASCADE_LOOK_OUT_LO * n + r] = FA_MONT(d_fa_lut[lo]);
    main[(u64)MC_CASCADE_LOOKUP_MULTIPLICITY * n + r] = bfe_from_u64(entries[2 * r + 1]);
}
// lookup table (lookup.rs:84-112)
__global__ void k_fa_lookup(const u64* __restrict__ mult, u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= 256) return;
    main[(u64)MC_LOOKUP_IS_PADDING * n + r] = 0;
    main[(u64)MC_LOOKUP_LOOK_IN * n + r] = FA_MONT(r);
    main[(u64)MC_LOOKUP_LOOK_OUT * n + r] = FA_MONT(d_fa_lut[r]);
    main[(u64)MC_LOOKUP_LOOKUP_MULTIPLICITY * n + r] = bfe_from_u64(mult[r]);
}
// u32 table (u32.rs:101-125, 196-291): entry e = (opcode, lhs, rhs, multiplicity) with lhs / rhs Montgomery words; its
// section starts at row offsets[e].  Rows top-down (operands shifted right bit by bit), results bottom-up.
#define U32C(col, row) main[(u64)(col) * n + (row)]
__global__ void k_fa_u32(const u64* __restrict__ entries, const u64* __restrict__ offsets, u64 n_entries, u64* __restrict__ main, u64 n) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const u64 op = entries[4 * e], mult = entries[4 * e + 3];
    u64 lhs = fa_value(entries[4 * e + 1]), rhs = fa_value(entries[4 * e + 2]);
    const u64 lhs_word = entries[4 * e + 1];
    const bool is_pow = op == OP_POW;
    const u64 row0 = offsets[e];
    const u64 ci = FA_MONT(op);
    u64 r = row0;
    int bits = 0;
    for (;; bits++, r++) {
        U32C(MC_U32_COPY_FLAG, r) = bits == 0 ? FA_MONT(1) : 0;
        U32C(MC_U32_BITS, r) = FA_MONT(bits);
        U32C(MC_U32_CI, r) = ci;
        U32C(MC_U32_LHS, r) = is_pow ? lhs_word : bfe_from_u64(lhs);
        U32C(MC_U32_RHS, r) = bfe_from_u64(rhs);
        U32C(MC_U32_LOOKUP_MULTIPLICITY, r) = bits == 0 ? bfe_from_u64(mult) : 0;
        if ((lhs == 0 || is_pow) && rhs == 0) break;
        if (!is_pow) lhs >>= 1;
        rhs >>= 1;
    }
    // last row
    u64 res;
    switch (op) {
        case OP_LT: res = bits == 0 ? 0 : FA_MONT(2); break;
        case OP_LOG2_FLOOR: res = bfe_neg(FA_MONT(1)); break;
        case OP_POW: res = FA_MONT(1); break;
        default: res = 0; break;   // split, and, pop_count
    }
    U32C(MC_U32_RESULT, r) = res;
    // the rows above it, bottom-up (the three inverse columns are k_fa_u32_inverses' job: one work-item per ROW)
    while (r > row0) {
        r--;
        const u64 lw = U32C(MC_U32_LHS, r), rw = U32C(MC_U32_RHS, r);
        const u64 lv = is_pow ? 0 : fa_value(lw), rv = fa_value(rw);
        const u64 lhs_lsb = lv & 1, rhs_lsb = rv & 1;
        const u64 next = U32C(MC_U32_RESULT, r + 1);
        u64 out;
        switch (op) {
            case OP_SPLIT: out = next; break;
            case OP_LT:
                if (next == 0 || next == FA_MONT(1)) out = next;
                else if (lhs_lsb == 0 && rhs_lsb == 1) out = FA_MONT(1);
                else if (lhs_lsb == 1 && rhs_lsb == 0) out = 0;
                else out = r == row0 ? 0 : FA_MONT(2);
                break;
            case OP_AND: out = bfe_add(bfe_add(next, next), FA_MONT(lhs_lsb & rhs_lsb)); break;
            case OP_LOG2_FLOOR:
                if (lw == 0) out = bfe_neg(FA_MONT(1));
                else if (U32C(MC_U32_LHS, r + 1) != 0) out = next;
                else out = U32C(MC_U32_BITS, r);
                break;
            case OP_POW: out = rhs_lsb ? bfe_mul(bfe_mul(next, next), lw) : bfe_mul(next, next); break;
            default: out = bfe_add(next, FA_MONT(lhs_lsb)); break;   // pop_count
        }
        U32C(MC_U32_RESULT, r) = out;
    }
}

// LhsInv, RhsInv and BitsMinus33Inv of every section row: the inversions (a 64-step power each) are the bulk of the
// table's arithmetic; per row they fill the chip, per entry (31 775 sections of 33 rows at 2^20 rows) they took 35 ms
__global__ void k_fa_u32_inverses(u64 u32_len, u64* __restrict__ main, u64 n) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= u32_len) return;
    const u64 lw = U32C(MC_U32_LHS, r), rw = U32C(MC_U32_RHS, r);
    U32C(MC_U32_LHS_INV, r) = lw ? bfe_inv(lw) : 0;
    U32C(MC_U32_RHS_INV, r) = rw ? bfe_inv(rw) : 0;
    U32C(MC_U32_BITS_MINUS33_INV, r) = bfe_inv(bfe_sub(U32C(MC_U32_BITS, r), FA_MONT(33)));
}

// ------------------------------------------------------------------------------------------------ host side
// stable sort of len (key, index) pairs by key: d_order receives the source row of every sorted position
static int stable_sort_by_key(tvm_ctx* c, u64* d_keys, u64* d_idx, u64* d_keys_out, u64* d_order, u64 len) {
    if (!len) return TVM_OK;
#ifdef TVM_EMU
    std::vector<u64> order(len);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](u64 a, u64 b) { return d_keys[a] < d_keys[b]; });
    for (u64 i = 0; i < len; i++) d_order[i] = d_idx[order[i]], d_keys_out[i] = d_keys[order[i]];
    return TVM_OK;
#else
    size_t tmp_bytes = 0;
    TVM_HIP_CHECK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys_out, d_idx, d_order, (int)len, 0, 64, c->stream));
    void* tmp = pool_alloc(c, tmp_bytes ? tmp_bytes : 8);
    if (!tmp) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "sort scratch");
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, d_keys, d_keys_out, d_idx, d_order, (int)len, 0, 64, c->stream);
    pool_release(c, tmp);
    TVM_HIP_CHECK(c, e);
    return TVM_OK;
#endif
}
static int inclusive_sum(tvm_ctx* c, u64* d_in, u64* d_out, u64 len) {
    if (!len) return TVM_OK;
#ifdef TVM_EMU
    u64 acc = 0;
    for (u64 i = 0; i < len; i++) d_out[i] = (acc += d_in[i]);
    return TVM_OK;
#else
    size_t tmp_bytes = 0;
    TVM_HIP_CHECK(c, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_in, d_out, (int)len, c->stream));
    void* tmp = pool_alloc(c, tmp_bytes ? tmp_bytes : 8);
    if (!tmp) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "scan scratch");
    const hipError_t e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, d_in, d_out, (int)len, c->stream);
    pool_release(c, tmp);
    TVM_HIP_CHECK(c, e);
    return TVM_OK;
#endif
}

static inline dim3 fa_grid(u64 count) { return dim3((unsigned)((count + 255) / 256)); }
// The AET's arrays may live in host memory (the reference's AlgebraicExecutionTrace) or already on the device (a host that
// runs the VM next to the GPU and keeps the trace resident: nothing crosses PCIe inside the call).
static bool fa_on_device(const void* p) {
#ifdef TVM_EMU
    (void)p;
    return false;
#else
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain (unregistered) host memory on runtimes that report it as an error
        return false;
    }
    return at.type == hipMemoryTypeDevice;
#endif
}
struct FaUpload {  // host array -> device copy from the pool, released on scope exit; a device array is used where it lies
    tvm_ctx* c;
    u64* d = nullptr;
    bool owned = true;
    FaUpload(tvm_ctx* c_, const void* h, size_t bytes) : c(c_) {
        if (bytes && fa_on_device(h)) {
            d = (u64*)h;
            owned = false;
            return;
        }
        d = (u64*)pool_alloc(c, bytes ? bytes : 8);
        if (d && bytes && hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            pool_release(c, d);
            d = nullptr;
        }
    }
    ~FaUpload() {
        if (owned) pool_release(c, d);
    }
    FaUpload(const FaUpload&) = delete;   // (the emulation's launch macro captures its arguments by value)
    FaUpload& operator=(const FaUpload&) = delete;
};

int fill_main_table(tvm_ctx* c, const tvm_aet* aet, u64* d_main, u64 n, u64* h_lengths) {
    const u64 program_table_len = (aet->program_len + 1 + 9) / 10 * 10;   // aet.rs:174-181
    const u64 hash_len = aet->program_hash_len + aet->sponge_len + aet->hash_len;
    u64 u32_len = 0;
    std::vector<u64> u32_offsets(aet->u32_len ? aet->u32_len : 1);
    std::vector<u64> u32_host;                 // the section offsets are computed on the host: a device-resident entry list comes back
    const u64* u32_entries = aet->u32_entries;
    if (aet->u32_len && fa_on_device(aet->u32_entries)) {
        u32_host.resize(aet->u32_len * 4);
        TVM_HIP_CHECK(c, hipMemcpyAsync(u32_host.data(), aet->u32_entries, u32_host.size() * 8, hipMemcpyDeviceToHost, c->stream));
        TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        u32_entries = u32_host.data();
    }
    for (u64 e = 0; e < aet->u32_len; e++) {   // U32TableEntry::table_height_contribution (u32.rs:53-64)
        const u64 op = u32_entries[4 * e];
        auto value = [](u64 w) { return bfe_mul(w, 1); };
        const u64 lhs = value(u32_entries[4 * e + 1]), rhs = value(u32_entries[4 * e + 2]);
        const u64 dominant = op == OP_POW ? rhs : (lhs > rhs ? lhs : rhs);
        u32_offsets[e] = u32_len;
        u32_len += dominant ? 2 + (63 - __builtin_clzll(dominant)) : 1;
    }
    const u64 lengths[9] = {program_table_len, aet->processor_len, aet->op_stack_len, aet->ram_len, aet->processor_len, hash_len,
                            aet->cascade_len, 256, u32_len};
    for (int t = 0; t < 9; t++) {
        h_lengths[t] = lengths[t];
        if (lengths[t] > n) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fill: a table is longer than the padded height");
    }
    // the Bezout coefficient polynomials of the RAM table: given by the host (num_ram_pointers of them) or, when both
    // pointers are null, computed here from the sorted table's distinct RAM pointers (csrc/bezout.hip)
    const bool device_bezout = aet->ram_len && !aet->bezout_coefficients_0 && !aet->bezout_coefficients_1;
    if (aet->processor_len < 1 || (aet->ram_len && !device_bezout && !aet->num_ram_pointers))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fill: empty processor trace or missing Bezout coefficients");
    TVM_HIP_CHECK(c, hipMemsetAsync(d_main, 0, (size_t)TVM_NUM_ORIGINAL_MAIN_COLUMNS * n * sizeof(u64), c->stream));

    // uploads (the AET lives on the host)
    FaUpload program(c, aet->program_words, aet->program_len * 8), imult(c, aet->instruction_multiplicities, aet->program_len * 4);
    FaUpload proc(c, aet->processor_trace, aet->processor_len * 39 * 8), ops(c, aet->op_stack_trace, aet->op_stack_len * 4 * 8);
    FaUpload ram(c, aet->ram_trace, aet->ram_len * 7 * 8), bc0(c, aet->bezout_coefficients_0, aet->num_ram_pointers * 8),
        bc1(c, aet->bezout_coefficients_1, aet->num_ram_pointers * 8);
    FaUpload ph(c, aet->program_hash_trace, aet->program_hash_len * 67 * 8), sp(c, aet->sponge_trace, aet->sponge_len * 67 * 8),
        hs(c, aet->hash_trace, aet->hash_len * 67 * 8);
    FaUpload u32e(c, aet->u32_entries, aet->u32_len * 4 * 8), u32o(c, u32_offsets.data(), u32_offsets.size() * 8);
    FaUpload casc(c, aet->cascade_entries, aet->cascade_len * 2 * 8), lkm(c, aet->lookup_multiplicities, 256 * 8);
    const u64 max_mem = aet->processor_len > aet->op_stack_len ? (aet->processor_len > aet->ram_len ? aet->processor_len : aet->ram_len)
                                                               : (aet->op_stack_len > aet->ram_len ? aet->op_stack_len : aet->ram_len);
    PoolBlock w_block(c, (4 * max_mem + aet->processor_len + 8) * sizeof(u64)), bz_block(c);  // released on every exit path
    u64* w = (u64*)w_block.p;
    u64* bz = nullptr;  // device-computed Bezout coefficients (device_bezout)
    const bool ok = program.d && imult.d && proc.d && ops.d && ram.d && bc0.d && bc1.d && ph.d && sp.d && hs.d && u32e.d && u32o.d && casc.d &&
                    lkm.d && w;
    if (!ok) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "fill: device staging");
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));   // the host arrays may be caller temporaries
    const u64 *d_program = program.d, *d_proc = proc.d, *d_ops = ops.d, *d_ram = ram.d, *d_bc0 = bc0.d, *d_bc1 = bc1.d, *d_u32e = u32e.d,
              *d_u32o = u32o.d, *d_casc = casc.d, *d_lkm = lkm.d;
    const u32* d_imult = (const u32*)imult.d;
    u64 *keys = w, *idx = w + max_mem, *keys_out = w + 2 * max_mem, *order = w + 3 * max_mem, *hist = w + 4 * max_mem;
    TVM_HIP_CHECK(c, hipMemsetAsync(hist, 0, aet->processor_len * sizeof(u64), c->stream));
    int rc = TVM_OK;

    // Program, Hash (three sections with their modes), Cascade, Lookup, U32
    TVM_LAUNCH(k_fa_program, fa_grid(program_table_len), dim3(256), 0, c->stream, d_program, d_imult, aet->program_len, program_table_len, d_main, n);
    const u64 sec_len[3] = {aet->program_hash_len, aet->sponge_len, aet->hash_len};
    const u64* sec_src[3] = {ph.d, sp.d, hs.d};
    u64 row0 = 0;
    for (int s = 0; s < 3; s++) {
        if (sec_len[s]) {
            TVM_LAUNCH(k_fa_transpose, fa_grid(sec_len[s] * 67), dim3(256), 0, c->stream, sec_src[s], sec_len[s], 67, d_main, n, MC_HASH_MODE, row0);
            TVM_LAUNCH(k_fa_fill_column, fa_grid(sec_len[s]), dim3(256), 0, c->stream, d_main, n, MC_HASH_MODE, row0, sec_len[s], FA_MONT(s + 1));
        }
        row0 += sec_len[s];
    }
    if (aet->cascade_len) TVM_LAUNCH(k_fa_cascade, fa_grid(aet->cascade_len), dim3(256), 0, c->stream, d_casc, aet->cascade_len, d_main, n);
    TVM_LAUNCH(k_fa_lookup, fa_grid(256), dim3(256), 0, c->stream, d_lkm, d_main, n);
    if (aet->u32_len) {
        TVM_LAUNCH(k_fa_u32, fa_grid(aet->u32_len), dim3(256), 0, c->stream, d_u32e, d_u32o, aet->u32_len, d_main, n);
        TVM_LAUNCH(k_fa_u32_inverses, fa_grid(u32_len), dim3(256), 0, c->stream, u32_len, d_main, n);
    }

    // OpStack: stable sort by stack pointer (column 2 of the trace rows)
    if (aet->op_stack_len) {
        const u64 len = aet->op_stack_len;
        TVM_LAUNCH(k_fa_keys, fa_grid(len), dim3(256), 0, c->stream, d_ops, len, 4, 2, keys, idx);
        if ((rc = stable_sort_by_key(c, keys, idx, keys_out, order, len)) != TVM_OK) goto done;
        TVM_LAUNCH(k_fa_gather, fa_grid(len * 4), dim3(256), 0, c->stream, d_ops, order, len, 4, 4, d_main, n, MC_OPSTACK_CLK);
        TVM_LAUNCH(k_fa_clock_jump_differences, fa_grid(len), dim3(256), 0, c->stream, d_main, n, MC_OPSTACK_CLK, MC_OPSTACK_STACK_POINTER, len, hist,
                   aet->processor_len);
    }
    // Ram: stable sort by RAM pointer (column 2), then consistency
    if (aet->ram_len) {
        const u64 len = aet->ram_len;
        TVM_LAUNCH(k_fa_keys, fa_grid(len), dim3(256), 0, c->stream, d_ram, len, 7, 2, keys, idx);
        if ((rc = stable_sort_by_key(c, keys, idx, keys_out, order, len)) != TVM_OK) goto done;
        TVM_LAUNCH(k_fa_gather, fa_grid(len * 4), dim3(256), 0, c->stream, d_ram, order, len, 7, 4, d_main, n, MC_RAM_CLK);
        TVM_LAUNCH(k_fa_ram_flags, fa_grid(len), dim3(256), 0, c->stream, d_main, n, len, keys);
        if ((rc = inclusive_sum(c, keys, keys_out, len)) != TVM_OK) goto done;
        if (device_bezout) {
            u64 changes = 0;  // pointer changes up to the last row: one less than the number of distinct pointers
            if (hipMemcpyAsync(&changes, keys_out + len - 1, sizeof(u64), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) {
                rc = set_error(c, TVM_ERR_DEVICE, "fill: distinct RAM pointers");
                goto done;
            }
            const u64 n_unique = changes + 1;
            bz = (u64*)bz_block.alloc(3 * n_unique * sizeof(u64));  // the distinct pointers, then the two coefficient vectors
            if (!bz) {
                rc = set_error(c, TVM_ERR_OUT_OF_MEMORY, "fill: Bezout coefficients");
                goto done;
            }
            TVM_LAUNCH(k_fa_distinct_pointers, fa_grid(len), dim3(256), 0, c->stream, (const u64*)d_main, n, len, (const u64*)keys, (const u64*)keys_out, bz);
            if ((rc = bezout_coefficients(c, bz, n_unique, bz + n_unique, bz + 2 * n_unique)) != TVM_OK) goto done;
            TVM_LAUNCH(k_fa_ram_consistent, fa_grid(len), dim3(256), 0, c->stream, d_main, n, len, keys_out, (const u64*)(bz + n_unique),
                       (const u64*)(bz + 2 * n_unique), n_unique);
        } else {
            TVM_LAUNCH(k_fa_ram_consistent, fa_grid(len), dim3(256), 0, c->stream, d_main, n, len, keys_out, d_bc0, d_bc1, aet->num_ram_pointers);
        }
        TVM_LAUNCH(k_fa_clock_jump_differences, fa_grid(len), dim3(256), 0, c->stream, d_main, n, MC_RAM_CLK, MC_RAM_RAM_POINTER, len, hist, aet->processor_len);
    }
    // JumpStack: the processor rows, stable-sorted by jump-stack pointer
    {
        const u64 len = aet->processor_len;
        TVM_LAUNCH(k_fa_keys, fa_grid(len), dim3(256), 0, c->stream, d_proc, len, 39, MC_PROCESSOR_JSP - MC_PROCESSOR_CLK, keys, idx);
        if ((rc = stable_sort_by_key(c, keys, idx, keys_out, order, len)) != TVM_OK) goto done;
        TVM_LAUNCH(k_fa_jump_stack, fa_grid(len), dim3(256), 0, c->stream, d_proc, order, len, d_main, n);
        TVM_LAUNCH(k_fa_clock_jump_differences, fa_grid(len), dim3(256), 0, c->stream, d_main, n, MC_JUMPSTACK_CLK, MC_JUMPSTACK_JSP, len, hist, len);
        // Processor: the trace itself plus the lookup multiplicities of the clock jump differences (processor.rs:44-68)
        TVM_LAUNCH(k_fa_transpose, fa_grid(len * 39), dim3(256), 0, c->stream, d_proc, len, 39, d_main, n, MC_PROCESSOR_CLK, (u64)0);
        TVM_LAUNCH(k_fa_multiplicities, fa_grid(len), dim3(256), 0, c->stream, hist, len, d_main, n);
    }
    if (hipGetLastError() != hipSuccess) rc = set_error(c, TVM_ERR_DEVICE, "fill kernels");
done:
    return rc;
}

}  // namespace tvm
