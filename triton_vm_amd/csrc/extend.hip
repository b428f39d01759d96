// extend.hip -- the auxiliary table's `extend` on the device (first half): the 49 cross-table-argument
// columns (running products, running evaluations, logarithmic-derivative sums) of the nine tables, computed from the
// padded main table and the challenges.
//
// Replaces MasterMainTable::extend's per-table loops (/root/reference/triton-vm/src/table/master_table.rs:1006-1075;
// table/program.rs:129-269, processor.rs:98-561, op_stack.rs:108-174, ram.rs:264-399, jump_stack.rs:31-91,
// hash.rs:311-600, cascade.rs:66-124, lookup.rs:24-76, u32.rs:154-193).  In the reference each column is a sequential
// O(n) scan on one core, parallel only across the <= 11 columns of a table.  Here every column is
//      y_i = a_i * y_{i-1} + b_i          (XFieldElement, y_{-1} = 1)
// with (a_i, b_i) a function of main-table rows i-1 and i only: a running product has b = 0, a logarithmic-derivative
// sum a = 1, a running evaluation a in {1, x, x^k}.  Step 1 (k_ext_terms_*, one work-item per row, all tables in one
// launch each) writes the terms; step 2 is ONE generic three-phase prefix scan over the affine maps (composition is
// associative) for all columns at once (grid.y = column).  Two columns read another column's scanned values
// (Program.SendChunk reads PrepareChunk, Ram.FormalDerivative reads RunningProductOfRAMP): they get their terms and
// their scan in a second round.  Tables are the column-major traces tvm_lde_table takes (main [379][n] words,
// aux [91][n][3]), so a wavefront's accesses are runs of consecutive words; the result stays on the device for
// tvm_fill_derived_aux_columns and the LDE.
#include "air_columns.h"
#include "context.h"
#include "kernels.h"

namespace tvm {

#define EXT_MONT(v) ((u64)(v) * 0xFFFFFFFFull)   // Montgomery word of a small integer v < 2^32: v * (2^64 mod p)
#define EXT_KIND_PROD 0
#define EXT_KIND_SUM 1
#define EXT_KIND_AFFINE 2
#define EXT_KIND_SKIP 3
#define EXT_NUM_COLS TVM_NUM_ORIGINAL_AUX_COLUMNS   // 49
#define EXT_NUM_A_SLOTS 16

struct ExtendArgs {
    const u64* main;   // [379][n]
    u64* aux;          // [91][n][3]
    u64* a_buf;        // [EXT_NUM_A_SLOTS][n][3]: the factors a_i of the affine columns
    const u64* ch;     // [63][3]
    u64 n;
};
struct ExtendCols {
    signed char kind[EXT_NUM_COLS];
    signed char slot[EXT_NUM_COLS];
};

// a-slot of each affine column
#define SLOT_PROGRAM_PREPARE 0
#define SLOT_PROGRAM_SEND 1
#define SLOT_PROC_INPUT 2
#define SLOT_PROC_OUTPUT 3
#define SLOT_PROC_HASH_INPUT 4
#define SLOT_PROC_HASH_DIGEST 5
#define SLOT_PROC_SPONGE 6
#define SLOT_RAM_FD 7
#define SLOT_RAM_BC0 8
#define SLOT_RAM_BC1 9
#define SLOT_HASH_RECEIVE 10
#define SLOT_HASH_INPUT 11
#define SLOT_HASH_DIGEST 12
#define SLOT_HASH_SPONGE 13
#define SLOT_LOOKUP_PUBLIC 14

static ExtendCols extend_cols(int round) {
    ExtendCols d;
    for (int c = 0; c < EXT_NUM_COLS; c++) d.kind[c] = EXT_KIND_SKIP, d.slot[c] = -1;
    auto set = [&](int col, int kind, int slot = -1) { d.kind[col] = (signed char)kind, d.slot[col] = (signed char)slot; };
    if (round == 1) {  // the two columns whose terms need scanned values of round 0
        set(AX_PROGRAM_SEND_CHUNK_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_PROGRAM_SEND);
        set(AX_RAM_FORMAL_DERIVATIVE, EXT_KIND_AFFINE, SLOT_RAM_FD);
        return d;
    }
    set(AX_PROGRAM_INSTRUCTION_LOOKUP_SERVER_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_PROGRAM_PREPARE_CHUNK_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_PROGRAM_PREPARE);
    set(AX_PROCESSOR_INPUT_TABLE_EVAL_ARG, EXT_KIND_AFFINE, SLOT_PROC_INPUT);
    set(AX_PROCESSOR_OUTPUT_TABLE_EVAL_ARG, EXT_KIND_AFFINE, SLOT_PROC_OUTPUT);
    set(AX_PROCESSOR_INSTRUCTION_LOOKUP_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_PROCESSOR_OP_STACK_TABLE_PERM_ARG, EXT_KIND_PROD);
    set(AX_PROCESSOR_RAM_TABLE_PERM_ARG, EXT_KIND_PROD);
    set(AX_PROCESSOR_JUMP_STACK_TABLE_PERM_ARG, EXT_KIND_PROD);
    set(AX_PROCESSOR_HASH_INPUT_EVAL_ARG, EXT_KIND_AFFINE, SLOT_PROC_HASH_INPUT);
    set(AX_PROCESSOR_HASH_DIGEST_EVAL_ARG, EXT_KIND_AFFINE, SLOT_PROC_HASH_DIGEST);
    set(AX_PROCESSOR_SPONGE_EVAL_ARG, EXT_KIND_AFFINE, SLOT_PROC_SPONGE);
    set(AX_PROCESSOR_U32_LOOKUP_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_SERVER_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_OPSTACK_RUNNING_PRODUCT_PERM_ARG, EXT_KIND_PROD);
    set(AX_OPSTACK_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_RAM_RUNNING_PRODUCT_OF_RAMP, EXT_KIND_PROD);
    set(AX_RAM_BEZOUT_COEFFICIENT0, EXT_KIND_AFFINE, SLOT_RAM_BC0);
    set(AX_RAM_BEZOUT_COEFFICIENT1, EXT_KIND_AFFINE, SLOT_RAM_BC1);
    set(AX_RAM_RUNNING_PRODUCT_PERM_ARG, EXT_KIND_PROD);
    set(AX_RAM_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_JUMPSTACK_RUNNING_PRODUCT_PERM_ARG, EXT_KIND_PROD);
    set(AX_JUMPSTACK_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_HASH_RECEIVE_CHUNK_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_HASH_RECEIVE);
    set(AX_HASH_HASH_INPUT_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_HASH_INPUT);
    set(AX_HASH_HASH_DIGEST_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_HASH_DIGEST);
    set(AX_HASH_SPONGE_RUNNING_EVALUATION, EXT_KIND_AFFINE, SLOT_HASH_SPONGE);
    for (int k = 0; k < 16; k++) set(AX_HASH_CASCADE_STATE0_HIGHEST_CLIENT_LOG_DERIVATIVE + k, EXT_KIND_SUM);
    set(AX_CASCADE_HASH_TABLE_SERVER_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_CASCADE_LOOKUP_TABLE_CLIENT_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_LOOKUP_CASCADE_TABLE_SERVER_LOG_DERIVATIVE, EXT_KIND_SUM);
    set(AX_LOOKUP_PUBLIC_EVALUATION_ARGUMENT, EXT_KIND_AFFINE, SLOT_LOOKUP_PUBLIC);
    set(AX_U32_LOOKUP_SERVER_LOG_DERIVATIVE, EXT_KIND_SUM);
    return d;
}

// ------------------------------------------------------------------------------------------------ accessors
#define EM(col, row) (a.main[(u64)(col) * a.n + (row)])
TVM_D xfe ext_ch(const ExtendArgs& a, int k) { return xfe_make(a.ch[3 * k], a.ch[3 * k + 1], a.ch[3 * k + 2]); }
TVM_D void ext_put(const ExtendArgs& a, int col, u64 r, xfe v) {
    u64* p = a.aux + ((u64)col * a.n + r) * 3;
    p[0] = v.c0, p[1] = v.c1, p[2] = v.c2;
}
TVM_D xfe ext_get(const ExtendArgs& a, int col, u64 r) {
    const u64* p = a.aux + ((u64)col * a.n + r) * 3;
    return xfe_make(p[0], p[1], p[2]);
}
TVM_D void ext_put_a(const ExtendArgs& a, int slot, u64 r, xfe v) {
    u64* p = a.a_buf + ((u64)slot * a.n + r) * 3;
    p[0] = v.c0, p[1] = v.c1, p[2] = v.c2;
}
TVM_D void ext_affine(const ExtendArgs& a, int col, int slot, u64 r, xfe fa, xfe fb) {
    ext_put_a(a, slot, r, fa);
    ext_put(a, col, r, fb);
}
TVM_D u64 ext_value(u64 w) { return bfe_mul(w, 1); }   // canonical value of a Montgomery word
// acc + challenge[k] * v
TVM_D xfe ext_acc(const ExtendArgs& a, xfe acc, int k, u64 v) { return xfe_add(acc, xfe_mul_bfe(ext_ch(a, k), v)); }
// 1 / (indeterminate - compressed)
TVM_D xfe ext_inv_of(xfe indeterminate, xfe compressed) { return xfe_inv(xfe_sub(indeterminate, compressed)); }

// instruction of a processor row (table/processor.rs:760-770): opcode and argument as integers; ok = a known
// instruction whose argument is legal
struct ExtInstr {
    int op;
    u64 arg;
    bool ok;
};
TVM_D bool ext_has_word_count_arg(int op) {
    return op == OP_POP || op == OP_DIVINE || op == OP_READ_MEM || op == OP_WRITE_MEM || op == OP_READ_IO || op == OP_WRITE_IO;
}
TVM_D bool ext_has_stack_arg(int op) { return op == OP_PICK || op == OP_PLACE || op == OP_DUP || op == OP_SWAP; }
TVM_D bool ext_known_opcode(u64 op) {
    switch (op) {
        case OP_POP: case OP_PUSH: case OP_DIVINE: case OP_PICK: case OP_PLACE: case OP_DUP: case OP_SWAP: case OP_HALT:
        case OP_NOP: case OP_SKIZ: case OP_CALL: case OP_RETURN: case OP_RECURSE: case OP_RECURSE_OR_RETURN: case OP_ASSERT:
        case OP_READ_MEM: case OP_WRITE_MEM: case OP_HASH: case OP_ASSERT_VECTOR: case OP_SPONGE_INIT: case OP_SPONGE_ABSORB:
        case OP_SPONGE_ABSORB_MEM: case OP_SPONGE_SQUEEZE: case OP_ADD: case OP_ADD_I: case OP_MUL: case OP_INVERT: case OP_EQ:
        case OP_SPLIT: case OP_LT: case OP_AND: case OP_XOR: case OP_LOG2_FLOOR: case OP_POW: case OP_DIV_MOD: case OP_POP_COUNT:
        case OP_XX_ADD: case OP_XX_MUL: case OP_XINVERT: case OP_XB_MUL: case OP_READ_IO: case OP_WRITE_IO: case OP_MERKLE_STEP:
        case OP_MERKLE_STEP_MEM: case OP_BHORNER_STEP: case OP_XHORNER_STEP: return true;
        default: return false;
    }
}
TVM_D ExtInstr ext_instruction(const ExtendArgs& a, u64 r) {
    ExtInstr in;
    const u64 op = ext_value(EM(MC_PROCESSOR_CI, r));
    in.ok = ext_known_opcode(op);
    in.op = (int)op;
    in.arg = 0;
    if (in.ok && (ext_has_word_count_arg(in.op) || ext_has_stack_arg(in.op))) {
        in.arg = ext_value(EM(MC_PROCESSOR_NIA, r));
        if (ext_has_word_count_arg(in.op) && (in.arg < 1 || in.arg > 5)) in.ok = false;
        if (ext_has_stack_arg(in.op) && in.arg > 15) in.ok = false;
    }
    return in;
}
// Instruction::op_stack_size_influence (triton-isa/src/instruction.rs:496-545)
TVM_D int ext_stack_delta(ExtInstr in) {
    switch (in.op) {
        case OP_POP: case OP_WRITE_MEM: case OP_WRITE_IO: return -(int)in.arg;
        case OP_DIVINE: case OP_READ_MEM: case OP_READ_IO: return (int)in.arg;
        case OP_PUSH: case OP_DUP: case OP_SPLIT: return 1;
        case OP_SKIZ: case OP_ASSERT: case OP_ADD: case OP_MUL: case OP_EQ: case OP_LT: case OP_AND: case OP_XOR: case OP_POW:
        case OP_XB_MUL: return -1;
        case OP_HASH: case OP_ASSERT_VECTOR: return -5;
        case OP_SPONGE_ABSORB: return -10;
        case OP_SPONGE_SQUEEZE: return 10;
        case OP_XX_ADD: case OP_XX_MUL: return -3;
        default: return 0;
    }
}
#define EXT_ST(k) (MC_PROCESSOR_ST0 + (k))
#define EXT_HV(k) (MC_PROCESSOR_HV0 + (k))
// sum_k StackWeight_k * values_k over columns given by a small functor
template <class F>
TVM_D xfe ext_weighted(const ExtendArgs& a, int count, F value) {
    xfe acc = xfe_zero();
    for (int k = 0; k < count; k++) acc = ext_acc(a, acc, CH_STACK_WEIGHT0 + k, value(k));
    return acc;
}

// ------------------------------------------------------------------------------------------------ Program table
__global__ void k_ext_terms_program(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const u64 instr = EM(MC_PROGRAM_INSTRUCTION, r);
    // instruction-lookup server: the value of row r is the sum over the rows BEFORE r (program.rs:148-163)
    xfe term = xfe_zero();
    if (r > 0 && EM(MC_PROGRAM_IS_HASH_INPUT_PADDING, r - 1) != EXT_MONT(1)) {
        xfe cr = xfe_mul_bfe(ext_ch(a, CH_PROGRAM_ADDRESS_WEIGHT), EM(MC_PROGRAM_ADDRESS, r - 1));
        cr = ext_acc(a, cr, CH_PROGRAM_INSTRUCTION_WEIGHT, EM(MC_PROGRAM_INSTRUCTION, r - 1));
        cr = ext_acc(a, cr, CH_PROGRAM_NEXT_INSTRUCTION_WEIGHT, instr);
        term = xfe_mul_bfe(ext_inv_of(ext_ch(a, CH_INSTRUCTION_LOOKUP_INDETERMINATE), cr), EM(MC_PROGRAM_LOOKUP_MULTIPLICITY, r - 1));
    }
    ext_put(a, AX_PROGRAM_INSTRUCTION_LOOKUP_SERVER_LOG_DERIVATIVE, r, term);
    // prepare-chunk running evaluation: resets at the start of a chunk (program.rs:236-250)
    const xfe x = ext_ch(a, CH_PROGRAM_ATTESTATION_PREPARE_CHUNK_INDETERMINATE);
    if (EM(MC_PROGRAM_INDEX_IN_CHUNK, r) == 0)
        ext_affine(a, AX_PROGRAM_PREPARE_CHUNK_RUNNING_EVALUATION, SLOT_PROGRAM_PREPARE, r, xfe_zero(), xfe_add_bfe(x, instr));
    else
        ext_affine(a, AX_PROGRAM_PREPARE_CHUNK_RUNNING_EVALUATION, SLOT_PROGRAM_PREPARE, r, x, xfe_lift(instr));
}
// second round: send-chunk running evaluation reads the scanned prepare-chunk column (program.rs:252-269)
__global__ void k_ext_terms_program_send(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const bool update = EM(MC_PROGRAM_IS_TABLE_PADDING, r) != EXT_MONT(1) && EM(MC_PROGRAM_INDEX_IN_CHUNK, r) == EXT_MONT(9);
    if (update)
        ext_affine(a, AX_PROGRAM_SEND_CHUNK_RUNNING_EVALUATION, SLOT_PROGRAM_SEND, r,
                   ext_ch(a, CH_PROGRAM_ATTESTATION_SEND_CHUNK_INDETERMINATE), ext_get(a, AX_PROGRAM_PREPARE_CHUNK_RUNNING_EVALUATION, r));
    else
        ext_affine(a, AX_PROGRAM_SEND_CHUNK_RUNNING_EVALUATION, SLOT_PROGRAM_SEND, r, xfe_one(), xfe_zero());
}

// ------------------------------------------------------------------------------------------------ Processor table
TVM_D xfe ext_u32_term(const ExtendArgs& a, u64 lhs, u64 rhs, u64 ci, bool with_result, u64 result) {
    xfe cr = xfe_mul_bfe(ext_ch(a, CH_U32_LHS_WEIGHT), lhs);
    cr = ext_acc(a, cr, CH_U32_RHS_WEIGHT, rhs);
    cr = ext_acc(a, cr, CH_U32_CI_WEIGHT, ci);
    if (with_result) cr = ext_acc(a, cr, CH_U32_RESULT_WEIGHT, result);
    return ext_inv_of(ext_ch(a, CH_U32_INDETERMINATE), cr);
}
TVM_D xfe ext_ram_factor(const ExtendArgs& a, u64 clk, u64 type, u64 pointer, u64 value) {
    xfe cr = xfe_mul_bfe(ext_ch(a, CH_RAM_CLK_WEIGHT), clk);
    cr = ext_acc(a, cr, CH_RAM_INSTRUCTION_TYPE_WEIGHT, type);
    cr = ext_acc(a, cr, CH_RAM_POINTER_WEIGHT, pointer);
    cr = ext_acc(a, cr, CH_RAM_VALUE_WEIGHT, value);
    return xfe_sub(ext_ch(a, CH_RAM_INDETERMINATE), cr);
}
__global__ void k_ext_terms_processor(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const bool has_prev = r > 0;
    const u64 p = has_prev ? r - 1 : 0;
    ExtInstr pi;
    pi.ok = false, pi.op = -1, pi.arg = 0;
    if (has_prev) pi = ext_instruction(a, p);
    const int pci = has_prev ? (int)ext_value(EM(MC_PROCESSOR_CI, p)) : -1;   // the raw opcode of the previous row
    const bool padding = EM(MC_PROCESSOR_IS_PADDING, r) == EXT_MONT(1);

    // input / output evaluation arguments (processor.rs:133-175)
    {
        xfe fa = xfe_one(), fb = xfe_zero();
        if (pi.ok && pi.op == OP_READ_IO) {
            const xfe x = ext_ch(a, CH_STANDARD_INPUT_INDETERMINATE);
            for (int k = (int)pi.arg - 1; k >= 0; k--) fa = xfe_mul(fa, x), fb = xfe_add_bfe(xfe_mul(fb, x), EM(EXT_ST(k), r));
        }
        ext_affine(a, AX_PROCESSOR_INPUT_TABLE_EVAL_ARG, SLOT_PROC_INPUT, r, fa, fb);
        fa = xfe_one(), fb = xfe_zero();
        if (pi.ok && pi.op == OP_WRITE_IO) {
            const xfe x = ext_ch(a, CH_STANDARD_OUTPUT_INDETERMINATE);
            for (int k = 0; k < (int)pi.arg; k++) fa = xfe_mul(fa, x), fb = xfe_add_bfe(xfe_mul(fb, x), EM(EXT_ST(k), p));
        }
        ext_affine(a, AX_PROCESSOR_OUTPUT_TABLE_EVAL_ARG, SLOT_PROC_OUTPUT, r, fa, fb);
    }
    // instruction lookup client (processor.rs:177-208)
    {
        xfe term = xfe_zero();
        if (!padding) {
            xfe cr = xfe_mul_bfe(ext_ch(a, CH_PROGRAM_ADDRESS_WEIGHT), EM(MC_PROCESSOR_IP, r));
            cr = ext_acc(a, cr, CH_PROGRAM_INSTRUCTION_WEIGHT, EM(MC_PROCESSOR_CI, r));
            cr = ext_acc(a, cr, CH_PROGRAM_NEXT_INSTRUCTION_WEIGHT, EM(MC_PROCESSOR_NIA, r));
            term = ext_inv_of(ext_ch(a, CH_INSTRUCTION_LOOKUP_INDETERMINATE), cr);
        }
        ext_put(a, AX_PROCESSOR_INSTRUCTION_LOOKUP_CLIENT_LOG_DERIVATIVE, r, term);
    }
    // op-stack permutation argument (processor.rs:563-611)
    {
        xfe factor = xfe_one();
        if (has_prev && !padding && pi.ok) {
            const int delta = ext_stack_delta(pi);
            const u64 shorter = delta > 0 ? p : r;
            const int count = delta < 0 ? -delta : delta;
            for (int off = 0; off < count; off++) {
                xfe cr = xfe_mul_bfe(ext_ch(a, CH_OP_STACK_CLK_WEIGHT), EM(MC_PROCESSOR_CLK, p));
                cr = ext_acc(a, cr, CH_OP_STACK_IB1_WEIGHT, EM(MC_PROCESSOR_IB1, p));
                cr = ext_acc(a, cr, CH_OP_STACK_POINTER_WEIGHT, bfe_add(EM(MC_PROCESSOR_OP_STACK_POINTER, shorter), EXT_MONT(off)));
                cr = ext_acc(a, cr, CH_OP_STACK_FIRST_UNDERFLOW_ELEMENT_WEIGHT, EM(EXT_ST(15 - off), shorter));
                factor = xfe_mul(factor, xfe_sub(ext_ch(a, CH_OP_STACK_INDETERMINATE), cr));
            }
        }
        ext_put(a, AX_PROCESSOR_OP_STACK_TABLE_PERM_ARG, r, factor);
    }
    // RAM permutation argument (processor.rs:613-735)
    {
        xfe factor = xfe_one();
        if (has_prev && !padding && pi.ok) {
            const u64 clk = EM(MC_PROCESSOR_CLK, p);
            const u64 read = EXT_MONT(1), write = 0;
            if (pi.op == OP_READ_MEM || pi.op == OP_WRITE_MEM) {
                const bool rd = pi.op == OP_READ_MEM;
                const u64 longer = rd ? r : p;
                for (int off = 0; off < (int)pi.arg; off++) {
                    const u64 pointer = bfe_add(EM(EXT_ST(0), longer), EXT_MONT(off + (rd ? 1 : 0)));
                    factor = xfe_mul(factor, ext_ram_factor(a, clk, rd ? read : write, pointer, EM(EXT_ST(off + 1), longer)));
                }
            } else if (pi.op == OP_SPONGE_ABSORB_MEM) {
                const u64 base = EM(EXT_ST(0), p);
                for (int k = 0; k < 4; k++)
                    factor = xfe_mul(factor, ext_ram_factor(a, clk, read, bfe_add(base, EXT_MONT(k)), EM(EXT_ST(k + 1), r)));
                for (int k = 0; k < 6; k++)
                    factor = xfe_mul(factor, ext_ram_factor(a, clk, read, bfe_add(base, EXT_MONT(4 + k)), EM(EXT_HV(k), p)));
            } else if (pi.op == OP_MERKLE_STEP_MEM) {
                for (int k = 0; k < 5; k++)
                    factor = xfe_mul(factor, ext_ram_factor(a, clk, read, bfe_add(EM(EXT_ST(7), p), EXT_MONT(k)), EM(EXT_HV(k), p)));
            } else if (pi.op == OP_BHORNER_STEP) {
                factor = ext_ram_factor(a, clk, read, EM(EXT_ST(5), p), EM(EXT_HV(0), p));
            } else if (pi.op == OP_XHORNER_STEP) {
                for (int k = 0; k < 3; k++)
                    factor = xfe_mul(factor, ext_ram_factor(a, clk, read, bfe_sub(EM(EXT_ST(5), p), EXT_MONT(2 - k)), EM(EXT_HV(k), p)));
            }
        }
        ext_put(a, AX_PROCESSOR_RAM_TABLE_PERM_ARG, r, factor);
    }
    // jump-stack permutation argument: every row (processor.rs:243-262)
    {
        xfe cr = xfe_mul_bfe(ext_ch(a, CH_JUMP_STACK_CLK_WEIGHT), EM(MC_PROCESSOR_CLK, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_CI_WEIGHT, EM(MC_PROCESSOR_CI, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSP_WEIGHT, EM(MC_PROCESSOR_JSP, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSO_WEIGHT, EM(MC_PROCESSOR_JSO, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSD_WEIGHT, EM(MC_PROCESSOR_JSD, r));
        ext_put(a, AX_PROCESSOR_JUMP_STACK_TABLE_PERM_ARG, r, xfe_sub(ext_ch(a, CH_JUMP_STACK_INDETERMINATE), cr));
    }
    // hash input: acts on the CURRENT row (processor.rs:266-343)
    {
        const int ci = (int)ext_value(EM(MC_PROCESSOR_CI, r));
        xfe fa = xfe_one(), fb = xfe_zero();
        if (ci == OP_HASH || ci == OP_MERKLE_STEP || ci == OP_MERKLE_STEP_MEM) {
            fa = ext_ch(a, CH_HASH_INPUT_INDETERMINATE);
            if (ci == OP_HASH) {
                fb = ext_weighted(a, 10, [&](int k) { return EM(EXT_ST(k), r); });
            } else if ((ext_value(EM(EXT_ST(5), r)) & 1) == 0) {
                fb = ext_weighted(a, 10, [&](int k) { return k < 5 ? EM(EXT_ST(k), r) : EM(EXT_HV(k - 5), r); });
            } else {
                fb = ext_weighted(a, 10, [&](int k) { return k < 5 ? EM(EXT_HV(k), r) : EM(EXT_ST(k - 5), r); });
            }
        }
        ext_affine(a, AX_PROCESSOR_HASH_INPUT_EVAL_ARG, SLOT_PROC_HASH_INPUT, r, fa, fb);
    }
    // hash digest (processor.rs:346-379)
    {
        xfe fa = xfe_one(), fb = xfe_zero();
        if (pci == OP_HASH || pci == OP_MERKLE_STEP || pci == OP_MERKLE_STEP_MEM) {
            fa = ext_ch(a, CH_HASH_DIGEST_INDETERMINATE);
            fb = ext_weighted(a, 5, [&](int k) { return EM(EXT_ST(k), r); });
        }
        ext_affine(a, AX_PROCESSOR_HASH_DIGEST_EVAL_ARG, SLOT_PROC_HASH_DIGEST, r, fa, fb);
    }
    // sponge (processor.rs:383-464)
    {
        xfe fa = xfe_one(), fb = xfe_zero();
        const xfe ciw = ext_ch(a, CH_HASH_CIWEIGHT);
        if (pci == OP_SPONGE_INIT) {
            fa = ext_ch(a, CH_SPONGE_INDETERMINATE);
            fb = xfe_mul_bfe(ciw, EXT_MONT(OP_SPONGE_INIT));
        } else if (pci == OP_SPONGE_ABSORB) {
            fa = ext_ch(a, CH_SPONGE_INDETERMINATE);
            fb = xfe_add(xfe_mul_bfe(ciw, EXT_MONT(OP_SPONGE_ABSORB)), ext_weighted(a, 10, [&](int k) { return EM(EXT_ST(k), p); }));
        } else if (pci == OP_SPONGE_ABSORB_MEM) {
            fa = ext_ch(a, CH_SPONGE_INDETERMINATE);
            fb = xfe_add(xfe_mul_bfe(ciw, EXT_MONT(OP_SPONGE_ABSORB)),
                         ext_weighted(a, 10, [&](int k) { return k < 4 ? EM(EXT_ST(k + 1), r) : EM(EXT_HV(k - 4), p); }));
        } else if (pci == OP_SPONGE_SQUEEZE) {
            fa = ext_ch(a, CH_SPONGE_INDETERMINATE);
            fb = xfe_add(xfe_mul_bfe(ciw, EXT_MONT(OP_SPONGE_SQUEEZE)), ext_weighted(a, 10, [&](int k) { return EM(EXT_ST(k), r); }));
        }
        ext_affine(a, AX_PROCESSOR_SPONGE_EVAL_ARG, SLOT_PROC_SPONGE, r, fa, fb);
    }
    // u32 lookup client (processor.rs:466-534)
    {
        xfe term = xfe_zero();
        const u64 ci_word = has_prev ? EM(MC_PROCESSOR_CI, p) : 0;
        if (pci == OP_SPLIT) {
            term = ext_u32_term(a, EM(EXT_ST(0), r), EM(EXT_ST(1), r), ci_word, false, 0);
        } else if (pci == OP_LT || pci == OP_AND || pci == OP_POW) {
            term = ext_u32_term(a, EM(EXT_ST(0), p), EM(EXT_ST(1), p), ci_word, true, EM(EXT_ST(0), r));
        } else if (pci == OP_XOR) {   // a & b = (a + b - a ^ b) / 2
            const u64 half = 0x8000000000000000ull;   // Montgomery word of 1/2
            const u64 and_result = bfe_mul(bfe_sub(bfe_add(EM(EXT_ST(0), p), EM(EXT_ST(1), p)), EM(EXT_ST(0), r)), half);
            term = ext_u32_term(a, EM(EXT_ST(0), p), EM(EXT_ST(1), p), EXT_MONT(OP_AND), true, and_result);
        } else if (pci == OP_LOG2_FLOOR || pci == OP_POP_COUNT) {
            term = ext_u32_term(a, EM(EXT_ST(0), p), 0, ci_word, true, EM(EXT_ST(0), r));
        } else if (pci == OP_DIV_MOD) {
            term = xfe_add(ext_u32_term(a, EM(EXT_ST(0), r), EM(EXT_ST(1), p), EXT_MONT(OP_LT), true, EXT_MONT(1)),
                           ext_u32_term(a, EM(EXT_ST(0), p), EM(EXT_ST(1), r), EXT_MONT(OP_SPLIT), false, 0));
        } else if (pci == OP_MERKLE_STEP || pci == OP_MERKLE_STEP_MEM) {
            term = ext_u32_term(a, EM(EXT_ST(5), p), EM(EXT_ST(5), r), EXT_MONT(OP_SPLIT), false, 0);
        }
        ext_put(a, AX_PROCESSOR_U32_LOOKUP_CLIENT_LOG_DERIVATIVE, r, term);
    }
    // clock-jump-difference lookup server (processor.rs:536-561)
    {
        xfe term = xfe_zero();
        const u64 mult = EM(MC_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_MULTIPLICITY, r);
        if (mult != 0)
            term = xfe_mul_bfe(xfe_inv(xfe_sub_bfe(ext_ch(a, CH_CLOCK_JUMP_DIFFERENCE_LOOKUP_INDETERMINATE), EM(MC_PROCESSOR_CLK, r))), mult);
        ext_put(a, AX_PROCESSOR_CLOCK_JUMP_DIFFERENCE_LOOKUP_SERVER_LOG_DERIVATIVE, r, term);
    }
}

// ------------------------------------------------------------------------------------------------ memory-like tables
TVM_D xfe ext_cjd_term(const ExtendArgs& a, u64 clk, u64 prev_clk) {
    return xfe_inv(xfe_sub_bfe(ext_ch(a, CH_CLOCK_JUMP_DIFFERENCE_LOOKUP_INDETERMINATE), bfe_sub(clk, prev_clk)));
}
__global__ void k_ext_terms_memory(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const u64 p = r ? r - 1 : 0;
    // ---- OpStack (op_stack.rs:108-174)
    {
        const bool padding = EM(MC_OPSTACK_IB1_SHRINK_STACK, r) == EXT_MONT(2);
        xfe factor = xfe_one(), term = xfe_zero();
        if (!padding) {
            xfe cr = xfe_mul_bfe(ext_ch(a, CH_OP_STACK_CLK_WEIGHT), EM(MC_OPSTACK_CLK, r));
            cr = ext_acc(a, cr, CH_OP_STACK_IB1_WEIGHT, EM(MC_OPSTACK_IB1_SHRINK_STACK, r));
            cr = ext_acc(a, cr, CH_OP_STACK_POINTER_WEIGHT, EM(MC_OPSTACK_STACK_POINTER, r));
            cr = ext_acc(a, cr, CH_OP_STACK_FIRST_UNDERFLOW_ELEMENT_WEIGHT, EM(MC_OPSTACK_FIRST_UNDERFLOW_ELEMENT, r));
            factor = xfe_sub(ext_ch(a, CH_OP_STACK_INDETERMINATE), cr);
            if (r && EM(MC_OPSTACK_STACK_POINTER, p) == EM(MC_OPSTACK_STACK_POINTER, r))
                term = ext_cjd_term(a, EM(MC_OPSTACK_CLK, r), EM(MC_OPSTACK_CLK, p));
        }
        ext_put(a, AX_OPSTACK_RUNNING_PRODUCT_PERM_ARG, r, factor);
        ext_put(a, AX_OPSTACK_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, r, term);
    }
    // ---- Ram (ram.rs:264-399)
    {
        const bool padding = EM(MC_RAM_INSTRUCTION_TYPE, r) == EXT_MONT(2);
        const xfe bez = ext_ch(a, CH_RAM_TABLE_BEZOUT_RELATION_INDETERMINATE);
        const u64 ramp = EM(MC_RAM_RAM_POINTER, r);
        const bool changed = r && !padding && EM(MC_RAM_RAM_POINTER, p) != ramp;
        // running product of (bezout indeterminate - ram pointer) over the distinct pointers
        ext_put(a, AX_RAM_RUNNING_PRODUCT_OF_RAMP, r, (r == 0 || changed) ? xfe_sub_bfe(bez, ramp) : xfe_one());
        // Bezout coefficients: Horner over the distinct pointers; row 0 holds the leading coefficient
        const u64 c0 = EM(MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT0, r), c1 = EM(MC_RAM_BEZOUT_COEFFICIENT_POLYNOMIAL_COEFFICIENT1, r);
        if (r == 0) {
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT0, SLOT_RAM_BC0, r, xfe_zero(), xfe_lift(c0));
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT1, SLOT_RAM_BC1, r, xfe_zero(), xfe_lift(c1));
        } else if (changed) {
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT0, SLOT_RAM_BC0, r, bez, xfe_lift(c0));
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT1, SLOT_RAM_BC1, r, bez, xfe_lift(c1));
        } else {
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT0, SLOT_RAM_BC0, r, xfe_one(), xfe_zero());
            ext_affine(a, AX_RAM_BEZOUT_COEFFICIENT1, SLOT_RAM_BC1, r, xfe_one(), xfe_zero());
        }
        xfe factor = xfe_one(), term = xfe_zero();
        if (!padding) {
            factor = ext_ram_factor(a, EM(MC_RAM_CLK, r), EM(MC_RAM_INSTRUCTION_TYPE, r), ramp, EM(MC_RAM_RAM_VALUE, r));
            if (r && !changed) term = ext_cjd_term(a, EM(MC_RAM_CLK, r), EM(MC_RAM_CLK, p));
        }
        ext_put(a, AX_RAM_RUNNING_PRODUCT_PERM_ARG, r, factor);
        ext_put(a, AX_RAM_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, r, term);
    }
    // ---- JumpStack (jump_stack.rs:31-91): no padding indicator, every row counts
    {
        xfe cr = xfe_mul_bfe(ext_ch(a, CH_JUMP_STACK_CLK_WEIGHT), EM(MC_JUMPSTACK_CLK, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_CI_WEIGHT, EM(MC_JUMPSTACK_CI, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSP_WEIGHT, EM(MC_JUMPSTACK_JSP, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSO_WEIGHT, EM(MC_JUMPSTACK_JSO, r));
        cr = ext_acc(a, cr, CH_JUMP_STACK_JSD_WEIGHT, EM(MC_JUMPSTACK_JSD, r));
        ext_put(a, AX_JUMPSTACK_RUNNING_PRODUCT_PERM_ARG, r, xfe_sub(ext_ch(a, CH_JUMP_STACK_INDETERMINATE), cr));
        xfe term = xfe_zero();
        if (r && EM(MC_JUMPSTACK_JSP, p) == EM(MC_JUMPSTACK_JSP, r)) term = ext_cjd_term(a, EM(MC_JUMPSTACK_CLK, r), EM(MC_JUMPSTACK_CLK, p));
        ext_put(a, AX_JUMPSTACK_CLOCK_JUMP_DIFFERENCE_LOOKUP_CLIENT_LOG_DERIVATIVE, r, term);
    }
}
// second round: the formal derivative of the running product of RAM pointers reads that product's scanned column
// (ram.rs:264-300): fd' = (bez - ramp) * fd + rp_previous when the pointer changes
__global__ void k_ext_terms_ram_fd(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    if (r == 0) {
        ext_affine(a, AX_RAM_FORMAL_DERIVATIVE, SLOT_RAM_FD, r, xfe_zero(), xfe_one());
        return;
    }
    const bool padding = EM(MC_RAM_INSTRUCTION_TYPE, r) == EXT_MONT(2);
    const u64 ramp = EM(MC_RAM_RAM_POINTER, r);
    if (!padding && EM(MC_RAM_RAM_POINTER, r - 1) != ramp)
        ext_affine(a, AX_RAM_FORMAL_DERIVATIVE, SLOT_RAM_FD, r, xfe_sub_bfe(ext_ch(a, CH_RAM_TABLE_BEZOUT_RELATION_INDETERMINATE), ramp),
                   ext_get(a, AX_RAM_RUNNING_PRODUCT_OF_RAMP, r - 1));
    else
        ext_affine(a, AX_RAM_FORMAL_DERIVATIVE, SLOT_RAM_FD, r, xfe_one(), xfe_zero());
}

// ------------------------------------------------------------------------------------------------ Hash table
// state element k (< 4) re-composed from its four 16-bit limbs, out of Montgomery representation (hash.rs:335-349)
TVM_D u64 ext_hash_state(const ExtendArgs& a, int k, u64 r) {
    const int base = MC_HASH_STATE0_HIGHEST_LK_IN + 4 * k;
    u64 v = bfe_mul(EM(base, r), EXT_MONT(1ull << 16));                 // highest * 2^16 ...
    v = bfe_add(v, EM(base + 1, r));
    v = bfe_add(bfe_mul(v, EXT_MONT(1ull << 16)), EM(base + 2, r));
    v = bfe_add(bfe_mul(v, EXT_MONT(1ull << 16)), EM(base + 3, r));
    // ... = highest*2^48 + mid_high*2^32 + mid_low*2^16 + lowest; times the inverse of the Montgomery modulus 2^64:
    // as a Montgomery word that inverse is 1, and a Montgomery product with the raw word 1 divides by 2^64
    return bfe_mul(v, 1);
}
TVM_D u64 ext_hash_rate(const ExtendArgs& a, int k, u64 r) { return k < 4 ? ext_hash_state(a, k, r) : EM(MC_HASH_STATE4 + (k - 4), r); }
__global__ void k_ext_terms_hash(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    const u64 mode = EM(MC_HASH_MODE, r), round = EM(MC_HASH_ROUND_NUMBER, r), ci = EM(MC_HASH_CI, r);
    const bool round0 = round == 0, last = round == EXT_MONT(5), is_init = ci == EXT_MONT(OP_SPONGE_INIT);
    xfe fa = xfe_one(), fb = xfe_zero();
    if (mode == EXT_MONT(1) && round0) {   // program hashing: receive a chunk
        const xfe x = ext_ch(a, CH_PROGRAM_ATTESTATION_PREPARE_CHUNK_INDETERMINATE);
        xfe chunk = xfe_one();
        for (int k = 0; k < 10; k++) chunk = xfe_add_bfe(xfe_mul(chunk, x), ext_hash_rate(a, k, r));
        fa = ext_ch(a, CH_PROGRAM_ATTESTATION_SEND_CHUNK_INDETERMINATE), fb = chunk;
    }
    ext_affine(a, AX_HASH_RECEIVE_CHUNK_RUNNING_EVALUATION, SLOT_HASH_RECEIVE, r, fa, fb);
    fa = xfe_one(), fb = xfe_zero();
    if (mode == EXT_MONT(3) && round0) {
        fa = ext_ch(a, CH_HASH_INPUT_INDETERMINATE);
        fb = ext_weighted(a, 10, [&](int k) { return ext_hash_rate(a, k, r); });
    }
    ext_affine(a, AX_HASH_HASH_INPUT_RUNNING_EVALUATION, SLOT_HASH_INPUT, r, fa, fb);
    fa = xfe_one(), fb = xfe_zero();
    if (mode == EXT_MONT(3) && last) {
        fa = ext_ch(a, CH_HASH_DIGEST_INDETERMINATE);
        fb = ext_weighted(a, 5, [&](int k) { return ext_hash_rate(a, k, r); });
    }
    ext_affine(a, AX_HASH_HASH_DIGEST_RUNNING_EVALUATION, SLOT_HASH_DIGEST, r, fa, fb);
    fa = xfe_one(), fb = xfe_zero();
    if (mode == EXT_MONT(2) && round0) {
        fa = ext_ch(a, CH_SPONGE_INDETERMINATE);
        fb = xfe_mul_bfe(ext_ch(a, CH_HASH_CIWEIGHT), ci);
        if (!is_init) fb = xfe_add(fb, ext_weighted(a, 10, [&](int k) { return ext_hash_rate(a, k, r); }));
    }
    ext_affine(a, AX_HASH_SPONGE_RUNNING_EVALUATION, SLOT_HASH_SPONGE, r, fa, fb);
    // sixteen cascade lookups (hash.rs:414-425, 480-563)
    const bool lookups = mode != 0 && !last && !is_init;
    for (int k = 0; k < 16; k++) {
        xfe term = xfe_zero();
        if (lookups) {
            xfe ce = xfe_sub(ext_ch(a, CH_HASH_CASCADE_LOOKUP_INDETERMINATE),
                             xfe_mul_bfe(ext_ch(a, CH_HASH_CASCADE_LOOK_IN_WEIGHT), EM(MC_HASH_STATE0_HIGHEST_LK_IN + k, r)));
            ce = xfe_sub(ce, xfe_mul_bfe(ext_ch(a, CH_HASH_CASCADE_LOOK_OUT_WEIGHT), EM(MC_HASH_STATE0_HIGHEST_LK_OUT + k, r)));
            term = xfe_inv(ce);
        }
        ext_put(a, AX_HASH_CASCADE_STATE0_HIGHEST_CLIENT_LOG_DERIVATIVE + k, r, term);
    }
}

// ------------------------------------------------------------------------------------------------ Cascade, Lookup, U32
__global__ void k_ext_terms_lookups(ExtendArgs a) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    // ---- Cascade (cascade.rs:66-124)
    {
        xfe hash_term = xfe_zero(), lookup_term = xfe_zero();
        if (EM(MC_CASCADE_IS_PADDING, r) != EXT_MONT(1)) {
            const u64 two8 = EXT_MONT(256);
            const u64 in_lo = EM(MC_CASCADE_LOOK_IN_LO, r), in_hi = EM(MC_CASCADE_LOOK_IN_HI, r);
            const u64 out_lo = EM(MC_CASCADE_LOOK_OUT_LO, r), out_hi = EM(MC_CASCADE_LOOK_OUT_HI, r);
            xfe cr = xfe_mul_bfe(ext_ch(a, CH_HASH_CASCADE_LOOK_IN_WEIGHT), bfe_add(bfe_mul(two8, in_hi), in_lo));
            cr = ext_acc(a, cr, CH_HASH_CASCADE_LOOK_OUT_WEIGHT, bfe_add(bfe_mul(two8, out_hi), out_lo));
            hash_term = xfe_mul_bfe(ext_inv_of(ext_ch(a, CH_HASH_CASCADE_LOOKUP_INDETERMINATE), cr), EM(MC_CASCADE_LOOKUP_MULTIPLICITY, r));
            const xfe ind = ext_ch(a, CH_CASCADE_LOOKUP_INDETERMINATE);
            xfe lo = xfe_mul_bfe(ext_ch(a, CH_LOOKUP_TABLE_INPUT_WEIGHT), in_lo);
            lo = ext_acc(a, lo, CH_LOOKUP_TABLE_OUTPUT_WEIGHT, out_lo);
            xfe hi = xfe_mul_bfe(ext_ch(a, CH_LOOKUP_TABLE_INPUT_WEIGHT), in_hi);
            hi = ext_acc(a, hi, CH_LOOKUP_TABLE_OUTPUT_WEIGHT, out_hi);
            lookup_term = xfe_add(ext_inv_of(ind, lo), ext_inv_of(ind, hi));
        }
        ext_put(a, AX_CASCADE_HASH_TABLE_SERVER_LOG_DERIVATIVE, r, hash_term);
        ext_put(a, AX_CASCADE_LOOKUP_TABLE_CLIENT_LOG_DERIVATIVE, r, lookup_term);
    }
    // ---- Lookup (lookup.rs:24-76)
    {
        xfe term = xfe_zero(), fa = xfe_one(), fb = xfe_zero();
        if (EM(MC_LOOKUP_IS_PADDING, r) != EXT_MONT(1)) {
            xfe cr = xfe_mul_bfe(ext_ch(a, CH_LOOKUP_TABLE_INPUT_WEIGHT), EM(MC_LOOKUP_LOOK_IN, r));
            cr = ext_acc(a, cr, CH_LOOKUP_TABLE_OUTPUT_WEIGHT, EM(MC_LOOKUP_LOOK_OUT, r));
            term = xfe_mul_bfe(ext_inv_of(ext_ch(a, CH_CASCADE_LOOKUP_INDETERMINATE), cr), EM(MC_LOOKUP_LOOKUP_MULTIPLICITY, r));
            fa = ext_ch(a, CH_LOOKUP_TABLE_PUBLIC_INDETERMINATE), fb = xfe_lift(EM(MC_LOOKUP_LOOK_OUT, r));
        }
        ext_put(a, AX_LOOKUP_CASCADE_TABLE_SERVER_LOG_DERIVATIVE, r, term);
        ext_affine(a, AX_LOOKUP_PUBLIC_EVALUATION_ARGUMENT, SLOT_LOOKUP_PUBLIC, r, fa, fb);
    }
    // ---- U32 (u32.rs:154-193)
    {
        xfe term = xfe_zero();
        if (EM(MC_U32_COPY_FLAG, r) == EXT_MONT(1)) {
            xfe cr = xfe_mul_bfe(ext_ch(a, CH_U32_CI_WEIGHT), EM(MC_U32_CI, r));
            cr = ext_acc(a, cr, CH_U32_LHS_WEIGHT, EM(MC_U32_LHS, r));
            cr = ext_acc(a, cr, CH_U32_RHS_WEIGHT, EM(MC_U32_RHS, r));
            cr = ext_acc(a, cr, CH_U32_RESULT_WEIGHT, EM(MC_U32_RESULT, r));
            term = xfe_mul_bfe(ext_inv_of(ext_ch(a, CH_U32_INDETERMINATE), cr), EM(MC_U32_LOOKUP_MULTIPLICITY, r));
        }
        ext_put(a, AX_U32_LOOKUP_SERVER_LOG_DERIVATIVE, r, term);
    }
}

// ------------------------------------------------------------------------------------------------ the scan
// An element is the affine map y -> a*y + b; (second o first) = (a2*a1, a2*b1 + b2).  Products carry only a, sums only b.
struct ExtMap {
    xfe a, b;
};
TVM_D ExtMap ext_identity() { ExtMap m; m.a = xfe_one(); m.b = xfe_zero(); return m; }
TVM_D ExtMap ext_compose(int kind, ExtMap second, ExtMap first) {
    ExtMap m = first;
    if (kind == EXT_KIND_PROD) {
        m.a = xfe_mul(second.a, first.a);
    } else if (kind == EXT_KIND_SUM) {
        m.b = xfe_add(second.b, first.b);
    } else {
        m.a = xfe_mul(second.a, first.a);
        m.b = xfe_add(xfe_mul(second.a, first.b), second.b);
    }
    return m;
}
TVM_D ExtMap ext_load(const ExtendArgs& a, int kind, int col, int slot, u64 r) {
    ExtMap m = ext_identity();
    const xfe t = ext_get(a, col, r);
    if (kind == EXT_KIND_PROD) {
        m.a = t;
    } else {
        m.b = t;
        if (kind == EXT_KIND_AFFINE) {
            const u64* p = a.a_buf + ((u64)slot * a.n + r) * 3;
            m.a = xfe_make(p[0], p[1], p[2]);
        }
    }
    return m;
}
// y for the initial value 1 of running products and evaluations, 0 of sums (cross_table_argument.rs:40-90)
TVM_D xfe ext_apply(int kind, ExtMap m) {
    if (kind == EXT_KIND_PROD) return m.a;
    if (kind == EXT_KIND_SUM) return m.b;
    return xfe_add(m.a, m.b);
}
#define EXT_SCAN_THREADS 256
// rows per work-item.  (Round 4: 16 rows with the terms re-read in the second pass -- 4.06 compositions per row instead of
// 7.25 -- measured SLOWER, extend 4.95 -> 5.83 ms at 2^20 rows: a work-item's rows are 24 x K contiguous bytes, and at K = 16 a
// wavefront's load touches 64 lines 384 bytes apart.)
#define EXT_SCAN_K 4
#define EXT_SCAN_TILE (EXT_SCAN_THREADS * EXT_SCAN_K)
TVM_D void ext_lds_put(u64* lds, int i, ExtMap m) {
    u64* p = lds + 6 * i;
    p[0] = m.a.c0, p[1] = m.a.c1, p[2] = m.a.c2, p[3] = m.b.c0, p[4] = m.b.c1, p[5] = m.b.c2;
}
TVM_D ExtMap ext_lds_get(const u64* lds, int i) {
    const u64* p = lds + 6 * i;
    ExtMap m;
    m.a = xfe_make(p[0], p[1], p[2]);
    m.b = xfe_make(p[3], p[4], p[5]);
    return m;
}
// inclusive scan of one map per work-item across the workgroup (Hillis-Steele in LDS); returns the inclusive prefix
TVM_D ExtMap ext_block_scan(int kind, ExtMap mine, u64* lds, int tid, int nt) {
    ext_lds_put(lds, tid, mine);
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {
        ExtMap other = ext_identity();
        const bool take = tid >= d;
        if (take) other = ext_lds_get(lds, tid - d);
        __syncthreads();
        if (take) {
            mine = ext_compose(kind, mine, other);
            ext_lds_put(lds, tid, mine);
        }
        __syncthreads();
    }
    return mine;
}
// phase 1: aggregate of each tile of EXT_SCAN_TILE rows -> agg[col][tile]
__global__ void __launch_bounds__(EXT_SCAN_THREADS) k_ext_scan_reduce(ExtendArgs a, ExtendCols cols, u64 n_tiles, u64* agg) {
    __shared__ u64 lds[6 * EXT_SCAN_THREADS];
    const int col = blockIdx.y, kind = cols.kind[col], slot = cols.slot[col];
    if (kind == EXT_KIND_SKIP) return;
    const int tid = threadIdx.x;
    const u64 r0 = ((u64)blockIdx.x * EXT_SCAN_THREADS + tid) * EXT_SCAN_K;
    ExtMap m = ext_identity();
    for (int k = 0; k < EXT_SCAN_K; k++)
        if (r0 + k < a.n) m = ext_compose(kind, ext_load(a, kind, col, slot, r0 + k), m);
    m = ext_block_scan(kind, m, lds, tid, EXT_SCAN_THREADS);
    if (tid == EXT_SCAN_THREADS - 1) ext_lds_put(agg + 6 * ((u64)col * n_tiles + blockIdx.x), 0, m);
}
// phase 2: exclusive scan of the tile aggregates of each column, in place (one workgroup per column)
__global__ void __launch_bounds__(EXT_SCAN_THREADS) k_ext_scan_tiles(ExtendCols cols, u64 n_tiles, u64* agg) {
    __shared__ u64 lds[6 * EXT_SCAN_THREADS];
    const int col = blockIdx.x, kind = cols.kind[col];
    if (kind == EXT_KIND_SKIP) return;
    const int tid = threadIdx.x;
    u64* mine = agg + 6 * (u64)col * n_tiles;
    ExtMap carry = ext_identity();
    for (u64 base = 0; base < n_tiles; base += EXT_SCAN_THREADS) {
        const u64 i = base + tid;
        ExtMap m = i < n_tiles ? ext_lds_get(mine, (int)i) : ext_identity();
        ExtMap incl = ext_block_scan(kind, m, lds, tid, EXT_SCAN_THREADS);
        // exclusive prefix of tile i = carry, then everything before i in this batch
        ExtMap excl = carry;
        if (tid > 0) excl = ext_compose(kind, ext_lds_get(lds, tid - 1), carry);
        const ExtMap total = ext_compose(kind, ext_lds_get(lds, EXT_SCAN_THREADS - 1), carry);
        __syncthreads();
        if (i < n_tiles) ext_lds_put(mine, (int)i, excl);
        carry = total;
        (void)incl;
    }
}
// phase 3: the values
__global__ void __launch_bounds__(EXT_SCAN_THREADS) k_ext_scan_apply(ExtendArgs a, ExtendCols cols, u64 n_tiles, const u64* agg) {
    __shared__ u64 lds[6 * EXT_SCAN_THREADS];
    const int col = blockIdx.y, kind = cols.kind[col], slot = cols.slot[col];
    if (kind == EXT_KIND_SKIP) return;
    const int tid = threadIdx.x;
    const u64 r0 = ((u64)blockIdx.x * EXT_SCAN_THREADS + tid) * EXT_SCAN_K;
    ExtMap el[EXT_SCAN_K];   // (both loops fully unrolled, no early exit: indexed dynamically the array lived in scratch, 208 B per lane)
    ExtMap m = ext_identity();
#pragma unroll
    for (int k = 0; k < EXT_SCAN_K; k++) {
        el[k] = r0 + k < a.n ? ext_load(a, kind, col, slot, r0 + k) : ext_identity();
        m = ext_compose(kind, el[k], m);
    }
    ext_block_scan(kind, m, lds, tid, EXT_SCAN_THREADS);
    ExtMap carry = ext_lds_get(agg + 6 * ((u64)col * n_tiles + blockIdx.x), 0);
    if (tid > 0) carry = ext_compose(kind, ext_lds_get(lds, tid - 1), carry);
#pragma unroll
    for (int k = 0; k < EXT_SCAN_K; k++) {
        if (r0 + k < a.n) {
            carry = ext_compose(kind, el[k], carry);
            ext_put(a, col, r0 + k, ext_apply(kind, carry));
        }
    }
}

static int extend_scan(tvm_ctx* c, const ExtendArgs& a, int round, u64* agg, u64 n_tiles) {
    const ExtendCols cols = extend_cols(round);
    TVM_LAUNCH(k_ext_scan_reduce, dim3((unsigned)n_tiles, EXT_NUM_COLS), dim3(EXT_SCAN_THREADS), 0, c->stream, a, cols, n_tiles, agg);
    TVM_LAUNCH(k_ext_scan_tiles, dim3(EXT_NUM_COLS), dim3(EXT_SCAN_THREADS), 0, c->stream, cols, n_tiles, agg);
    TVM_LAUNCH(k_ext_scan_apply, dim3((unsigned)n_tiles, EXT_NUM_COLS), dim3(EXT_SCAN_THREADS), 0, c->stream, a, cols, n_tiles, agg);
    return TVM_OK;
}

int extend_aux_table(tvm_ctx* c, const u64* d_main, u64* d_aux, const u64* d_challenges, u64 n) {
    ExtendArgs a;
    a.main = d_main, a.aux = d_aux, a.ch = d_challenges, a.n = n;
    const u64 n_tiles = (n + EXT_SCAN_TILE - 1) / EXT_SCAN_TILE;
    a.a_buf = (u64*)pool_alloc(c, (size_t)EXT_NUM_A_SLOTS * n * 3 * sizeof(u64));
    u64* agg = (u64*)pool_alloc(c, (6 * (size_t)EXT_NUM_COLS * n_tiles) * sizeof(u64));
    if (!a.a_buf || !agg) {
        pool_release(c, a.a_buf);
        pool_release(c, agg);
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "extend scratch");
    }
    const int bs = 256;
    const dim3 grid((unsigned)((n + bs - 1) / bs)), block(bs);
    TVM_LAUNCH(k_ext_terms_program, grid, block, 0, c->stream, a);
    TVM_LAUNCH(k_ext_terms_processor, grid, block, 0, c->stream, a);
    TVM_LAUNCH(k_ext_terms_memory, grid, block, 0, c->stream, a);
    TVM_LAUNCH(k_ext_terms_hash, grid, block, 0, c->stream, a);
    TVM_LAUNCH(k_ext_terms_lookups, grid, block, 0, c->stream, a);
    extend_scan(c, a, 0, agg, n_tiles);
    TVM_LAUNCH(k_ext_terms_program_send, grid, block, 0, c->stream, a);
    TVM_LAUNCH(k_ext_terms_ram_fd, grid, block, 0, c->stream, a);
    extend_scan(c, a, 1, agg, n_tiles);
    pool_release(c, a.a_buf);   // stream-ordered reuse: later requests on this context's stream run after the kernels above
    pool_release(c, agg);
    return TVM_OK;
}

}  // namespace tvm
