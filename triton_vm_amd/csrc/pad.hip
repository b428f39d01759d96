// pad.hip -- MasterMainTable::pad on the device (the `pad` half): the nine table-specific padding
// rules that bring every table of the main trace to the padded height, in place on the column-major trace
// tvm_lde_table takes.
//
// Replaces /root/reference/triton-vm/src/table/master_table.rs:932-983 with the per-table rules of
// table/program.rs:77-127, processor.rs:70-96, op_stack.rs:205-219, ram.rs:86-101, jump_stack.rs:144-199,
// hash.rs:280-309, cascade.rs:60-67, lookup.rs:114-118, u32.rs:127-152.  (The degree-lowering fill that ends the
// reference's `pad` is tvm_fill_derived_main_columns.)  One work-item per row; a padding row is a function of the
// table's last (or, for the jump stack, largest-clock) row and its own index, so nothing is sequential.
#include "air_columns.h"
#include "context.h"
#include "kernels.h"
#include "tip5_tables.h"

namespace tvm {

#define PAD_MONT(v) ((u64)(v) * 0xFFFFFFFFull)   // Montgomery word of a small integer
struct PadArgs {
    u64* main;         // [379][n]
    u64 n;
    u64 len[9];        // unpadded lengths: Program, Processor, OpStack, Ram, JumpStack, Hash, Cascade, Lookup, U32
    u64 js_pivot;      // jump stack: index of the row with the largest clock (jump_stack.rs:151-160)
    const u64* js_tail; // copy of the jump-stack rows after the pivot, [5][len - pivot - 1]
};
#define PM(col, row) (a.main[(u64)(col) * a.n + (row)])
#ifdef TVM_EMU
static const u64 d_pad_rc[80] = {TVM_TIP5_RC_LIST};
#else
static __device__ const u64 d_pad_rc[80] = {TVM_TIP5_RC_LIST};
#endif

__global__ void k_pad_main_table(PadArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    // ---- Program (program.rs:77-127)
    if (r >= a.len[0]) {
        const u64 idx = r % 10;
        PM(MC_PROGRAM_ADDRESS, r) = bfe_from_u64(r);
        PM(MC_PROGRAM_INSTRUCTION, r) = 0;
        PM(MC_PROGRAM_LOOKUP_MULTIPLICITY, r) = 0;
        PM(MC_PROGRAM_INDEX_IN_CHUNK, r) = PAD_MONT(idx);
        PM(MC_PROGRAM_MAX_MINUS_INDEX_IN_CHUNK_INV, r) = idx == 9 ? 0 : bfe_inv(PAD_MONT(9 - idx));
        PM(MC_PROGRAM_IS_HASH_INPUT_PADDING, r) = PAD_MONT(1);
        PM(MC_PROGRAM_IS_TABLE_PADDING, r) = PAD_MONT(1);
    }
    // ---- Processor (processor.rs:70-96): copies of the last row with the clock running on
    {
        const u64 len = a.len[1];
        if (r >= len) {
            for (int c = MC_PROCESSOR_CLK; c <= MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY; c++) PM(c, r) = PM(c, len - 1);
            PM(MC_PROCESSOR_IS_PADDING, r) = PAD_MONT(1);
            PM(MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY, r) = 0;
            PM(MC_PROCESSOR_CLK, r) = bfe_from_u64(r);
        }
    }
    // ---- OpStack (op_stack.rs:205-219)
    {
        const u64 len = a.len[2];
        if (r >= len) {
            for (int c = MC_OPSTACK_CLK; c <= MC_OPSTACK_FIRST_UNDERFLOW_ELEMENT; c++) PM(c, r) = len ? PM(c, len - 1) : 0;
            PM(MC_OPSTACK_IB1_SHRINK_STACK, r) = PAD_MONT(2);
            if (!len) PM(MC_OPSTACK_STACK_POINTER, r) = PAD_MONT(16);
        }
    }
    // ---- Ram (ram.rs:86-101)
    {
        const u64 len = a.len[3];
        if (r >= len) {
            for (int c = MC_RAM_CLK; c <= MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT1; c++) PM(c, r) = len ? PM(c, len - 1) : 0;
            PM(MC_RAM_INSTRUCTION_TYPE, r) = PAD_MONT(2);
            if (!len) PM(MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT1, r) = PAD_MONT(1);
        }
    }
    // ---- JumpStack (jump_stack.rs:144-199): the padding rows go right after the row with the largest clock, the rows
    // behind it move to the end of the table
    {
        const u64 len = a.len[4], pivot = a.js_pivot, n_pad = a.n - len, n_tail = len - pivot - 1;
        if (r > pivot && r <= pivot + n_pad) {
            for (int c = MC_JUMPSTACK_CLK; c <= MC_JUMPSTACK_JSD; c++) PM(c, r) = PM(c, pivot);
            PM(MC_JUMPSTACK_CLK, r) = bfe_from_u64(len + (r - pivot - 1));
        } else if (r > pivot + n_pad) {
            const u64 k = r - pivot - n_pad - 1;   // < n_tail
            for (int c = 0; c < 5; c++) PM(MC_JUMPSTACK_CLK + c, r) = a.js_tail[(u64)c * n_tail + k];
        }
    }
    // ---- Hash (hash.rs:280-309)
    if (r >= a.len[5]) {
        for (int c = MC_HASH_MODE; c <= MC_HASH_CONSTANT15; c++) PM(c, r) = 0;
        // inverse_or_zero_of_highest_2_limbs(0) = 1 / (2^32 - 1)
        const u64 inv = bfe_inv(bfe_from_u64(0xFFFFFFFFull));
        for (int k = 0; k < 4; k++) PM(MC_HASH_STATE0_INV + k, r) = inv;
        for (int k = 0; k < 16; k++) PM(MC_HASH_CONSTANT0 + k, r) = d_pad_rc[k]   /* the constants of round 0 */;
        PM(MC_HASH_CI, r) = PAD_MONT(OP_HASH);
    }
    // ---- Cascade, Lookup (cascade.rs:60-67, lookup.rs:114-118)
    if (r >= a.len[6]) {
        for (int c = MC_CASCADE_IS_PADDING; c <= MC_CASCADE_LOOKUP_MULTIPLICITY; c++) PM(c, r) = 0;
        PM(MC_CASCADE_IS_PADDING, r) = PAD_MONT(1);
    }
    if (r >= a.len[7]) {
        for (int c = MC_LOOKUP_IS_PADDING; c <= MC_LOOKUP_LOOKUP_MULTIPLICITY; c++) PM(c, r) = 0;
        PM(MC_LOOKUP_IS_PADDING, r) = PAD_MONT(1);
    }
    // ---- U32 (u32.rs:127-152)
    {
        const u64 len = a.len[8];
        if (r >= len) {
            for (int c = MC_U32_COPY_FLAG; c <= MC_U32_LOOKUP_MULTIPLICITY; c++) PM(c, r) = 0;
            PM(MC_U32_CI, r) = PAD_MONT(OP_SPLIT);
            PM(MC_U32_BITS_MINUS33_INV, r) = bfe_inv(bfe_neg(PAD_MONT(33)));
            if (len) {
                PM(MC_U32_CI, r) = PM(MC_U32_CI, len - 1);
                PM(MC_U32_LHS, r) = PM(MC_U32_LHS, len - 1);
                PM(MC_U32_LHS_INV, r) = PM(MC_U32_LHS_INV, len - 1);
                PM(MC_U32_RESULT, r) = PM(MC_U32_CI, len - 1) == PAD_MONT(OP_LT) ? PAD_MONT(2) : PM(MC_U32_RESULT, len - 1);
            }
        }
    }
}
// the multiplicity of clock value 1: the jump stack's padding rows look it up once each (processor.rs:84-95)
__global__ void k_pad_processor_row1(PadArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    PM(MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY, 1) =
        bfe_add(PM(MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY, 1), bfe_from_u64(a.n - a.len[1]));
}
// jump stack: find the row whose clock is len - 1 (the largest before padding), and save the rows behind it
__global__ void k_pad_find_js_pivot(PadArgs a, u64* pivot_out) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.len[4]) return;
    if (PM(MC_JUMPSTACK_CLK, r) == bfe_from_u64(a.len[4] - 1)) *pivot_out = r;
}
__global__ void k_pad_save_js_tail(PadArgs a, u64* tail) {
    const u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n_tail = a.len[4] - a.js_pivot - 1;
    if (k >= n_tail) return;
    for (int c = 0; c < 5; c++) tail[(u64)c * n_tail + k] = PM(MC_JUMPSTACK_CLK + c, a.js_pivot + 1 + k);
}

int pad_main_table(tvm_ctx* c, u64* d_main, u64 n, const u64* lengths) {
    PadArgs a;
    a.main = d_main, a.n = n, a.js_pivot = 0, a.js_tail = nullptr;
    for (int t = 0; t < 9; t++) {
        a.len[t] = lengths[t];
        if (lengths[t] > n) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "pad: a table is longer than the padded height");
    }
    if (a.len[1] < 1 || a.len[4] != a.len[1] || n < 2)  // a one-row table (`halt`): row 1 is the first padding row (processor.rs:92-94)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "pad: the processor table needs a row and the jump stack table its length");
    const int bs = 256;
    u64* d_pivot = (u64*)scratch(c, 22, sizeof(u64));
    if (!d_pivot) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "pad scratch");
    // the pivot is only written by the row that holds the largest clock: start from a sentinel, so that a malformed or
    // unsorted trace (no such row) is an error instead of stale scratch contents
    TVM_HIP_CHECK(c, hipMemsetAsync(d_pivot, 0xFF, sizeof(u64), c->stream));
    TVM_LAUNCH(k_pad_find_js_pivot, dim3((unsigned)((a.len[4] + bs - 1) / bs)), dim3(bs), 0, c->stream, a, d_pivot);
    TVM_HIP_CHECK(c, hipMemcpyAsync(&a.js_pivot, d_pivot, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (a.js_pivot >= a.len[4])
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "pad: no jump-stack row carries the last clock value (malformed execution trace)");
    const u64 n_tail = a.len[4] - a.js_pivot - 1;
    PoolBlock tail_block(c, (size_t)(5 * n_tail + 1) * sizeof(u64));  // released on every exit path
    u64* tail = (u64*)tail_block.p;
    if (!tail) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "pad scratch");
    a.js_tail = tail;
    if (n_tail) TVM_LAUNCH(k_pad_save_js_tail, dim3((unsigned)((n_tail + bs - 1) / bs)), dim3(bs), 0, c->stream, a, tail);
    TVM_LAUNCH(k_pad_main_table, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, c->stream, a);
    TVM_LAUNCH(k_pad_processor_row1, dim3(1), dim3(64), 0, c->stream, a);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
