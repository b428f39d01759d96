"""Processor Table AIR -- restated from /root/reference/triton-air/src/table/processor.rs:38-3279
(prose: specification/src/processor-table.md, instruction-specific-transition-constraints.md,
instruction-groups.md).  Statement and iterator-evaluation order mirror the reference because node
creation order fixes circuit ids, which degree lowering depends on."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain, circuit_sum
from .defs import (AUX, DIGEST_LEN, EVAL_ARG_INITIAL, EXTENSION_DEGREE, LOOKUP_ARG_INITIAL, MAIN, PERM_ARG_INITIAL,
                   TIP5_RATE, Ch)
from .isa import ALL_INSTRUCTIONS, NUM_INSTRUCTION_BITS, NUM_OP_STACK_REGISTERS, OPCODE, ib

M, A = MAIN["Processor"], AUX["Processor"]
NUM_HELPER_VARIABLE_REGISTERS = 6
NUMBER_OF_WORDS_LEGAL_VALUES = [1, 2, 3, 4, 5]       # NumberOfWords::legal_values (triton-isa/src/op_stack.rs)
NUMBER_OF_WORDS_ILLEGAL_VALUES = [0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]


def ST(i):
    """ProcessorTable::op_stack_column_by_index (processor.rs:52-74)"""
    return getattr(M, f"ST{i}")


def HV(i):
    return getattr(M, f"HV{i}")


def stack_weight_by_index(i):
    return getattr(Ch, f"StackWeight{i}")


def _cm(b):
    return lambda col: b.input(CurrentMain(col))


def _nm(b):
    return lambda col: b.input(NextMain(col))


def _ca(b):
    return lambda col: b.input(CurrentAux(col))


def _na(b):
    return lambda col: b.input(NextAux(col))


# ------------------------------------------------------------------------------------------------
def initial_constraints(b):  # processor.rs:81-216
    constant, x_constant, challenge = b.b_constant, b.x_constant, b.challenge
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))

    clk_is_0 = main_row(M.CLK)
    ip_is_0 = main_row(M.IP)
    jsp_is_0 = main_row(M.JSP)
    jso_is_0 = main_row(M.JSO)
    jsd_is_0 = main_row(M.JSD)
    st_is_0 = [main_row(ST(i)) for i in range(11)]
    op_stack_pointer_is_16 = main_row(M.OpStackPointer) - constant(16)

    program_digest = [main_row(ST(i)) for i in range(11, 16)]
    compressed_program_digest = b.x_constant(EVAL_ARG_INITIAL)
    for digest_element in program_digest:
        compressed_program_digest = (compressed_program_digest * challenge(Ch.CompressProgramDigestIndeterminate)
                                     + digest_element)
    compressed_program_digest_is_expected_program_digest = compressed_program_digest - challenge(Ch.CompressedProgramDigest)

    input_eval_initialized = aux_row(A.InputTableEvalArg) - x_constant(EVAL_ARG_INITIAL)

    instruction_lookup_indeterminate = challenge(Ch.InstructionLookupIndeterminate)
    instruction_ci_weight = challenge(Ch.ProgramInstructionWeight)
    instruction_nia_weight = challenge(Ch.ProgramNextInstructionWeight)
    compressed_row_for_instruction_lookup = (instruction_ci_weight * main_row(M.CI)
                                             + instruction_nia_weight * main_row(M.NIA))
    instruction_lookup_log_derivative_initialized = (
        (aux_row(A.InstructionLookupClientLogDerivative) - x_constant(LOOKUP_ARG_INITIAL))
        * (instruction_lookup_indeterminate - compressed_row_for_instruction_lookup)
        - constant(1))

    output_eval_initialized = aux_row(A.OutputTableEvalArg) - x_constant(EVAL_ARG_INITIAL)
    op_stack_perm_initialized = aux_row(A.OpStackTablePermArg) - x_constant(PERM_ARG_INITIAL)
    ram_perm_initialized = aux_row(A.RamTablePermArg) - x_constant(PERM_ARG_INITIAL)

    jump_stack_indeterminate = challenge(Ch.JumpStackIndeterminate)
    jump_stack_ci_weight = challenge(Ch.JumpStackCiWeight)
    compressed_row_for_jump_stack_table = jump_stack_ci_weight * main_row(M.CI)
    jump_stack_perm_initialized = (aux_row(A.JumpStackTablePermArg)
                                   - x_constant(PERM_ARG_INITIAL) * (jump_stack_indeterminate - compressed_row_for_jump_stack_table))

    clock_jump_diff_lookup_initialized = (
        aux_row(A.ClockJumpDifferenceLookupServerLogDerivative) * challenge(Ch.ClockJumpDifferenceLookupIndeterminate)
        - main_row(M.ClockJumpDifferenceLookupMultiplicity))

    hash_selector = main_row(M.CI) - constant(OPCODE["Hash"])
    hash_deselector = instruction_deselector_single_row(b, "Hash")
    hash_input_indeterminate = challenge(Ch.HashInputIndeterminate)
    compressed_row = constant(0)
    hash_input_has_absorbed_first_row = (aux_row(A.HashInputEvalArg)
                                         - hash_input_indeterminate * x_constant(EVAL_ARG_INITIAL)
                                         - compressed_row)
    hash_input_is_default_initial = aux_row(A.HashInputEvalArg) - x_constant(EVAL_ARG_INITIAL)
    hash_input_initialized = (hash_selector * hash_input_is_default_initial
                              + hash_deselector * hash_input_has_absorbed_first_row)

    hash_digest_initialized = aux_row(A.HashDigestEvalArg) - x_constant(EVAL_ARG_INITIAL)
    sponge_initialized = aux_row(A.SpongeEvalArg) - x_constant(EVAL_ARG_INITIAL)
    u32_log_derivative_initialized = aux_row(A.U32LookupClientLogDerivative) - x_constant(LOOKUP_ARG_INITIAL)

    return ([clk_is_0, ip_is_0, jsp_is_0, jso_is_0, jsd_is_0] + st_is_0
            + [compressed_program_digest_is_expected_program_digest, op_stack_pointer_is_16, input_eval_initialized,
               instruction_lookup_log_derivative_initialized, output_eval_initialized, op_stack_perm_initialized,
               ram_perm_initialized, jump_stack_perm_initialized, clock_jump_diff_lookup_initialized,
               hash_input_initialized, hash_digest_initialized, sponge_initialized, u32_log_derivative_initialized])


def consistency_constraints(b):  # processor.rs:218-262
    constant = b.b_constant
    main_row = lambda col: b.input(Main(col))

    ib_composition = (main_row(M.IB0)
                      + constant(1 << 1) * main_row(M.IB1)
                      + constant(1 << 2) * main_row(M.IB2)
                      + constant(1 << 3) * main_row(M.IB3)
                      + constant(1 << 4) * main_row(M.IB4)
                      + constant(1 << 5) * main_row(M.IB5)
                      + constant(1 << 6) * main_row(M.IB6))
    ci_corresponds_to_ib0_thru_ib6 = main_row(M.CI) - ib_composition

    ib_is_bit = [main_row(getattr(M, f"IB{i}")) * (main_row(getattr(M, f"IB{i}")) - constant(1)) for i in range(7)]
    is_padding_is_bit = main_row(M.IsPadding) * (main_row(M.IsPadding) - constant(1))
    cjd_multiplicity_is_0_in_padding_rows = (main_row(M.IsPadding)
                                             * (main_row(M.CLK) - constant(1))
                                             * main_row(M.ClockJumpDifferenceLookupMultiplicity))
    return ib_is_bit + [is_padding_is_bit, ci_corresponds_to_ib0_thru_ib6, cjd_multiplicity_is_0_in_padding_rows]


def transition_constraints(b):  # processor.rs:264-325
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)

    clk_increases_by_1 = next_main_row(M.CLK) - curr_main_row(M.CLK) - constant(1)
    is_padding_is_0_or_does_not_change = curr_main_row(M.IsPadding) * (next_main_row(M.IsPadding) - curr_main_row(M.IsPadding))

    all_instruction_deselectors = [instruction_deselector_current_row(b, name) for name, _ in ALL_INSTRUCTIONS]
    acc = constant(0)
    for d in all_instruction_deselectors:
        acc = acc + d
    exactly_one_instruction_deselector_must_be_inactive = acc - constant(1)

    instruction_independent_constraints = [clk_increases_by_1, is_padding_is_0_or_does_not_change,
                                           exactly_one_instruction_deselector_must_be_inactive]

    all_instruction_transition_constraints = [transition_constraints_for_instruction(b, name) for name, _ in ALL_INSTRUCTIONS]
    deselected = combine_instruction_constraints_with_deselectors(b, all_instruction_deselectors,
                                                                  all_instruction_transition_constraints)
    doubly_deselected = combine_transition_constraints_with_padding_constraints(b, deselected)

    table_linking_constraints = [
        log_derivative_accumulates_clk_next(b),
        log_derivative_for_instruction_lookup_updates_correctly(b),
        running_product_for_jump_stack_table_updates_correctly(b),
        running_evaluation_hash_input_updates_correctly(b),
        running_evaluation_hash_digest_updates_correctly(b),
        running_evaluation_sponge_updates_correctly(b),
        log_derivative_with_u32_table_updates_correctly(b),
    ]
    return instruction_independent_constraints + doubly_deselected + table_linking_constraints


def terminal_constraints(b):  # processor.rs:327-337
    return [b.input(Main(M.CI)) - b.b_constant(OPCODE["Halt"])]


def combine_instruction_constraints_with_deselectors(b, deselectors, all_tc_polys):  # processor.rs:340-373
    max_number_of_constraints = max(len(t) for t in all_tc_polys)
    zero_poly = b.b_constant(0)
    transposed = [[t[idx] if idx < len(t) else zero_poly for t in all_tc_polys] for idx in range(max_number_of_constraints)]
    return [circuit_sum(d * tc for d, tc in zip(deselectors, row)) for row in transposed]


def combine_transition_constraints_with_padding_constraints(b, instruction_transition_constraints):  # :375-419
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)

    padding_row_transition_constraints = (
        [next_main_row(M.IP) - curr_main_row(M.IP),
         next_main_row(M.CI) - curr_main_row(M.CI),
         next_main_row(M.NIA) - curr_main_row(M.NIA)]
        + instruction_group_keep_jump_stack(b)
        + instruction_group_keep_op_stack(b)
        + instruction_group_no_ram(b)
        + instruction_group_no_io(b))

    padding_row_deselector = constant(1) - next_main_row(M.IsPadding)
    padding_row_selector = next_main_row(M.IsPadding)

    n = max(len(instruction_transition_constraints), len(padding_row_transition_constraints))
    out = []
    for idx in range(n):
        zero_a = constant(0)   # `unwrap_or(&constant(0))` evaluates its argument eagerly
        instruction_constraint = instruction_transition_constraints[idx] if idx < len(instruction_transition_constraints) else zero_a
        zero_b = constant(0)
        padding_constraint = padding_row_transition_constraints[idx] if idx < len(padding_row_transition_constraints) else zero_b
        out.append(instruction_constraint * padding_row_deselector + padding_constraint * padding_row_selector)
    return out


# ------------------------------------------------------------------------------------------------
# instruction groups (processor.rs:421-719)
def instruction_group_decompose_arg(b):
    constant = b.b_constant
    curr_main_row = _cm(b)
    hv_is_a_bit = [curr_main_row(HV(i)) * (curr_main_row(HV(i)) - constant(1)) for i in range(4)]
    helper_variables_are_binary_decomposition_of_nia = (curr_main_row(M.NIA)
                                                        - constant(8) * curr_main_row(M.HV3)
                                                        - constant(4) * curr_main_row(M.HV2)
                                                        - constant(2) * curr_main_row(M.HV1)
                                                        - curr_main_row(M.HV0))
    return hv_is_a_bit + [helper_variables_are_binary_decomposition_of_nia]


def instruction_group_no_ram(b):
    return [_na(b)(A.RamTablePermArg) - _ca(b)(A.RamTablePermArg)]


def instruction_group_no_io(b):
    return [running_evaluation_for_standard_input_remains_unchanged(b),
            running_evaluation_for_standard_output_remains_unchanged(b)]


def instruction_group_op_stack_remains_except_top_n(b, n):
    assert n <= NUM_OP_STACK_REGISTERS
    curr_row, next_row = _cm(b), _nm(b)
    stack = [ST(i) for i in range(NUM_OP_STACK_REGISTERS)]
    next_stack = [next_row(st) for st in stack]
    curr_stack = [curr_row(st) for st in stack]

    def compress_stack_except_top_n(stk):
        weight = lambda i: b.challenge(stack_weight_by_index(i))
        return circuit_sum(weight(i) * st for i, st in list(enumerate(stk))[n:])

    all_but_n_top_elements_remain = compress_stack_except_top_n(next_stack) - compress_stack_except_top_n(curr_stack)
    constraints = instruction_group_keep_op_stack_height(b)
    constraints.append(all_but_n_top_elements_remain)
    return constraints


def instruction_group_keep_op_stack(b):
    return instruction_group_op_stack_remains_except_top_n(b, 0)


def instruction_group_keep_op_stack_height(b):
    op_stack_pointer_curr = b.input(CurrentMain(M.OpStackPointer))
    op_stack_pointer_next = b.input(NextMain(M.OpStackPointer))
    osp_remains_unchanged = op_stack_pointer_next - op_stack_pointer_curr
    perm_curr = b.input(CurrentAux(A.OpStackTablePermArg))
    perm_next = b.input(NextAux(A.OpStackTablePermArg))
    perm_arg_remains_unchanged = perm_next - perm_curr
    return [osp_remains_unchanged, perm_arg_remains_unchanged]


def instruction_group_grow_op_stack_and_top_two_elements_unconstrained(b):
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)
    out = [next_main_row(ST(i + 1)) - curr_main_row(ST(i)) for i in range(1, 15)]
    out.append(next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) - constant(1))
    out.append(running_product_op_stack_accounts_for_growing_stack_by(b, 1))
    return out


def instruction_group_grow_op_stack(b):
    specific = [_nm(b)(M.ST1) - _cm(b)(M.ST0)]
    return specific + instruction_group_grow_op_stack_and_top_two_elements_unconstrained(b)


def instruction_group_op_stack_shrinks_and_top_three_elements_unconstrained(b):
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)
    out = [next_main_row(ST(i)) - curr_main_row(ST(i + 1)) for i in range(3, 15)]
    out.append(next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) + constant(1))
    out.append(running_product_op_stack_accounts_for_shrinking_stack_by(b, 1))
    return out


def instruction_group_binop(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    specific = [next_main_row(M.ST1) - curr_main_row(M.ST2), next_main_row(M.ST2) - curr_main_row(M.ST3)]
    return specific + instruction_group_op_stack_shrinks_and_top_three_elements_unconstrained(b)


def instruction_group_shrink_op_stack(b):
    specific = [_nm(b)(M.ST0) - _cm(b)(M.ST1)]
    return specific + instruction_group_binop(b)


def instruction_group_keep_jump_stack(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    return [next_main_row(M.JSP) - curr_main_row(M.JSP),
            next_main_row(M.JSO) - curr_main_row(M.JSO),
            next_main_row(M.JSD) - curr_main_row(M.JSD)]


def instruction_group_step_1(b):
    ip_increases_by_one = _nm(b)(M.IP) - _cm(b)(M.IP) - b.b_constant(1)
    return instruction_group_keep_jump_stack(b) + [ip_increases_by_one]


def instruction_group_step_2(b):
    ip_increases_by_two = _nm(b)(M.IP) - _cm(b)(M.IP) - b.b_constant(2)
    return instruction_group_keep_jump_stack(b) + [ip_increases_by_two]


# ------------------------------------------------------------------------------------------------
# instruction deselectors (processor.rs:721-803)
def instruction_deselector_common_functionality(b, instruction, instruction_bit_polynomials):
    constant = b.b_constant
    one = lambda: constant(1)
    selector_bits = [ib(instruction, i) for i in range(NUM_INSTRUCTION_BITS)]
    acc = one()
    for x_ib, ib_of_instr in zip(instruction_bit_polynomials, selector_bits):
        acc = acc * (x_ib * constant(ib_of_instr) + (one() - x_ib) * constant(1 - ib_of_instr))
    return acc


def instruction_deselector_current_row(b, instruction):
    polys = [b.input(CurrentMain(getattr(M, f"IB{i}"))) for i in range(NUM_INSTRUCTION_BITS)]
    return instruction_deselector_common_functionality(b, instruction, polys)


def instruction_deselector_next_row(b, instruction):
    polys = [b.input(NextMain(getattr(M, f"IB{i}"))) for i in range(NUM_INSTRUCTION_BITS)]
    return instruction_deselector_common_functionality(b, instruction, polys)


def instruction_deselector_single_row(b, instruction):
    polys = [b.input(Main(getattr(M, f"IB{i}"))) for i in range(NUM_INSTRUCTION_BITS)]
    return instruction_deselector_common_functionality(b, instruction, polys)


# ------------------------------------------------------------------------------------------------
# helpers shared by several instructions
def helper_variable(b, index):  # processor.rs:3237-3250
    return b.input(CurrentMain(HV(index)))


def indicator_polynomial(b, index):  # processor.rs:3207-3235
    one = lambda: b.b_constant(1)
    hv = lambda idx: helper_variable(b, idx)
    acc = None
    for bit in (3, 2, 1, 0):
        factor = hv(bit) if (index >> bit) & 1 else one() - hv(bit)
        acc = factor if acc is None else acc * factor
    return acc


def map_then_drop_back(f, items, n):
    """`iter.map(f).dropping_back(n).collect()`: itertools' dropping_back eagerly pulls the last n
    elements through the map (in reverse order) before the rest is collected."""
    items = list(items)
    for x in reversed(items[len(items) - n:] if n else []):
        f(x)
    return [f(x) for x in items[:len(items) - n]]


def combine_mutually_exclusive_constraint_groups(b, all_constraint_groups):  # processor.rs:2211-2229
    num_constraints = max((len(g) for g in all_constraint_groups), default=0)
    combined = []
    for i in range(num_constraints):
        acc = b.b_constant(0)
        for group in all_constraint_groups:
            if i < len(group):
                acc = acc + group[i]
        combined.append(acc)
    return combined


def single_factor_for_permutation_argument_with_op_stack_table(b, row_with_shorter_stack_indicator, offset_):
    """processor.rs:2388-2417"""
    constant, challenge = b.b_constant, b.challenge
    curr_main_row = _cm(b)
    row_with_shorter_stack = lambda col: b.input(row_with_shorter_stack_indicator(col))

    stack_element_index = NUM_OP_STACK_REGISTERS - 1 - offset_
    underflow_element = row_with_shorter_stack(ST(stack_element_index))
    op_stack_pointer = row_with_shorter_stack(M.OpStackPointer)
    offset = constant(offset_)
    offset_op_stack_pointer = op_stack_pointer + offset

    compressed_row = (challenge(Ch.OpStackClkWeight) * curr_main_row(M.CLK)
                      + challenge(Ch.OpStackIb1Weight) * curr_main_row(M.IB1)
                      + challenge(Ch.OpStackPointerWeight) * offset_op_stack_pointer
                      + challenge(Ch.OpStackFirstUnderflowElementWeight) * underflow_element)
    return challenge(Ch.OpStackIndeterminate) - compressed_row


def running_product_op_stack_accounts_for_growing_stack_by(b, n):  # processor.rs:2340-2362
    factor = b.b_constant(1)
    for off in range(n):
        factor = factor * single_factor_for_permutation_argument_with_op_stack_table(b, CurrentMain, off)
    return _na(b)(A.OpStackTablePermArg) - _ca(b)(A.OpStackTablePermArg) * factor


def running_product_op_stack_accounts_for_shrinking_stack_by(b, n):  # processor.rs:2364-2386
    factor = b.b_constant(1)
    for off in range(n):
        factor = factor * single_factor_for_permutation_argument_with_op_stack_table(b, NextMain, off)
    return _na(b)(A.OpStackTablePermArg) - _ca(b)(A.OpStackTablePermArg) * factor


def constraints_for_shrinking_stack_by(b, n):  # processor.rs:2231-2263
    constant = b.b_constant
    curr_row, next_row = _cm(b), _nm(b)
    stack = [ST(i) for i in range(NUM_OP_STACK_REGISTERS)]
    new_stack = [next_row(st) for st in stack[:NUM_OP_STACK_REGISTERS - n]]
    old_stack_with_top_n_removed = [curr_row(st) for st in stack[n:]]

    def compress(stk):
        weight = lambda i: b.challenge(stack_weight_by_index(i))
        return circuit_sum(weight(i) * st for i, st in enumerate(stk))
    compressed_new_stack = compress(new_stack)
    compressed_old_stack = compress(old_stack_with_top_n_removed)

    op_stack_pointer_shrinks_by_n = next_row(M.OpStackPointer) - curr_row(M.OpStackPointer) + constant(n)
    new_stack_is_old_stack_with_top_n_removed = compressed_new_stack - compressed_old_stack
    return [op_stack_pointer_shrinks_by_n, new_stack_is_old_stack_with_top_n_removed,
            running_product_op_stack_accounts_for_shrinking_stack_by(b, n)]


def constraints_for_growing_stack_by(b, n):  # processor.rs:2265-2297
    constant = b.b_constant
    curr_row, next_row = _cm(b), _nm(b)
    stack = [ST(i) for i in range(NUM_OP_STACK_REGISTERS)]
    new_stack = [next_row(st) for st in stack[n:]]
    old_stack_with_top_n_added = map_then_drop_back(curr_row, stack, n)

    def compress(stk):
        weight = lambda i: b.challenge(stack_weight_by_index(i))
        return circuit_sum(weight(i) * st for i, st in enumerate(stk))
    compressed_new_stack = compress(new_stack)
    compressed_old_stack = compress(old_stack_with_top_n_added)

    op_stack_pointer_grows_by_n = next_row(M.OpStackPointer) - curr_row(M.OpStackPointer) - constant(n)
    new_stack_is_old_stack_with_top_n_added = compressed_new_stack - compressed_old_stack
    return [op_stack_pointer_grows_by_n, new_stack_is_old_stack_with_top_n_added,
            running_product_op_stack_accounts_for_growing_stack_by(b, n)]


def conditional_constraints_for_shrinking_stack_by(b, n):
    return [indicator_polynomial(b, n) * c for c in constraints_for_shrinking_stack_by(b, n)]


def conditional_constraints_for_growing_stack_by(b, n):
    return [indicator_polynomial(b, n) * c for c in constraints_for_growing_stack_by(b, n)]


def stack_shrinks_by_any_of(b, shrinkages):
    groups = [conditional_constraints_for_shrinking_stack_by(b, n) for n in shrinkages]
    return combine_mutually_exclusive_constraint_groups(b, groups)


def stack_grows_by_any_of(b, growths):
    groups = [conditional_constraints_for_growing_stack_by(b, n) for n in growths]
    return combine_mutually_exclusive_constraint_groups(b, groups)


def prohibit_any_illegal_number_of_words(b):  # processor.rs:2075-2084 (array map is eager)
    polys = [indicator_polynomial(b, n) for n in NUMBER_OF_WORDS_ILLEGAL_VALUES]
    return [circuit_sum(polys)]


def constraints_for_shrinking_stack_by_3_and_top_3_unconstrained(b):  # processor.rs:2159-2181
    curr_main_row, next_main_row = _cm(b), _nm(b)
    out = [next_main_row(ST(i)) - curr_main_row(ST(i + 3)) for i in range(3, 13)]
    out.append(next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) + b.b_constant(3))
    out.append(running_product_op_stack_accounts_for_shrinking_stack_by(b, 3))
    return out


def running_evaluation_for_standard_input_remains_unchanged(b):
    return _na(b)(A.InputTableEvalArg) - _ca(b)(A.InputTableEvalArg)


def running_evaluation_for_standard_output_remains_unchanged(b):
    return _na(b)(A.OutputTableEvalArg) - _ca(b)(A.OutputTableEvalArg)


def grow_stack_by_n_and_read_n_symbols_from_input(b, n):  # processor.rs:2119-2139
    indeterminate = lambda: b.challenge(Ch.StandardInputIndeterminate)
    next_main_row = _nm(b)
    running_evaluation = _ca(b)(A.InputTableEvalArg)
    for i in reversed(range(n)):
        running_evaluation = indeterminate() * running_evaluation + next_main_row(ST(i))
    running_evaluation_update = _na(b)(A.InputTableEvalArg) - running_evaluation
    conditional_update = indicator_polynomial(b, n) * running_evaluation_update
    constraints = conditional_constraints_for_growing_stack_by(b, n)
    constraints.append(conditional_update)
    return constraints


def shrink_stack_by_n_and_write_n_symbols_to_output(b, n):  # processor.rs:2141-2163
    indeterminate = lambda: b.challenge(Ch.StandardOutputIndeterminate)
    curr_main_row = _cm(b)
    running_evaluation = _ca(b)(A.OutputTableEvalArg)
    for i in range(n):
        running_evaluation = indeterminate() * running_evaluation + curr_main_row(ST(i))
    running_evaluation_update = _na(b)(A.OutputTableEvalArg) - running_evaluation
    conditional_update = indicator_polynomial(b, n) * running_evaluation_update
    constraints = conditional_constraints_for_shrinking_stack_by(b, n)
    constraints.append(conditional_update)
    return constraints


# ---- RAM helpers (processor.rs:1884-1917, 2419-2573) ----------------------------------------------
INSTRUCTION_TYPE_WRITE, INSTRUCTION_TYPE_READ = 0, 1


def read_from_ram_to(b, ram_pointers, destinations):
    curr_main_row, challenge, constant = _cm(b), b.challenge, b.b_constant

    def compress_row(ram_pointer, destination):
        return (curr_main_row(M.CLK) * challenge(Ch.RamClkWeight)
                + constant(INSTRUCTION_TYPE_READ) * challenge(Ch.RamInstructionTypeWeight)
                + ram_pointer * challenge(Ch.RamPointerWeight)
                + destination * challenge(Ch.RamValueWeight))

    factor = None
    for ram_pointer, destination in zip(ram_pointers, destinations):
        item = challenge(Ch.RamIndeterminate) - compress_row(ram_pointer, destination)
        factor = item if factor is None else factor * item
    if factor is None:
        factor = constant(1)
    return _ca(b)(A.RamTablePermArg) * factor - _na(b)(A.RamTablePermArg)


def single_factor_for_permutation_argument_with_ram_table(b, row_with_longer_stack_indicator, instruction_type, offset_):
    constant, challenge = b.b_constant, b.challenge
    curr_main_row = _cm(b)
    row_with_longer_stack = lambda col: b.input(row_with_longer_stack_indicator(col))

    ram_value = row_with_longer_stack(ST(offset_ + 1))
    additional_offset = 1 if instruction_type == INSTRUCTION_TYPE_READ else 0
    ram_pointer = row_with_longer_stack(M.ST0)
    offset = constant(additional_offset + offset_)
    offset_ram_pointer = ram_pointer + offset

    compressed_row = (curr_main_row(M.CLK) * challenge(Ch.RamClkWeight)
                      + constant(instruction_type) * challenge(Ch.RamInstructionTypeWeight)
                      + offset_ram_pointer * challenge(Ch.RamPointerWeight)
                      + ram_value * challenge(Ch.RamValueWeight))
    return challenge(Ch.RamIndeterminate) - compressed_row


def running_product_ram_accounts_for_writing_n_elements(b, n):
    factor = b.b_constant(1)
    for off in range(n):
        factor = factor * single_factor_for_permutation_argument_with_ram_table(b, CurrentMain, INSTRUCTION_TYPE_WRITE, off)
    return _na(b)(A.RamTablePermArg) - _ca(b)(A.RamTablePermArg) * factor


def running_product_ram_accounts_for_reading_n_elements(b, n):
    factor = b.b_constant(1)
    for off in range(n):
        factor = factor * single_factor_for_permutation_argument_with_ram_table(b, NextMain, INSTRUCTION_TYPE_READ, off)
    return _na(b)(A.RamTablePermArg) - _ca(b)(A.RamTablePermArg) * factor


def shrink_stack_by_n_and_write_n_elements_to_ram(b, n):
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)
    op_stack_pointer_shrinks_by_n = next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) + constant(n)
    ram_pointer_grows_by_n = next_main_row(M.ST0) - curr_main_row(M.ST0) - constant(n)
    constraints = [op_stack_pointer_shrinks_by_n, ram_pointer_grows_by_n,
                   running_product_op_stack_accounts_for_shrinking_stack_by(b, n),
                   running_product_ram_accounts_for_writing_n_elements(b, n)]
    for i in range(n + 1, NUM_OP_STACK_REGISTERS):
        constraints.append(next_main_row(ST(i - n)) - curr_main_row(ST(i)))
    return constraints


def grow_stack_by_n_and_read_n_elements_from_ram(b, n):
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)
    op_stack_pointer_grows_by_n = next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) - constant(n)
    ram_pointer_shrinks_by_n = next_main_row(M.ST0) - curr_main_row(M.ST0) + constant(n)
    constraints = [op_stack_pointer_grows_by_n, ram_pointer_shrinks_by_n,
                   running_product_op_stack_accounts_for_growing_stack_by(b, n),
                   running_product_ram_accounts_for_reading_n_elements(b, n)]
    for i in range(1, NUM_OP_STACK_REGISTERS - n):
        constraints.append(next_main_row(ST(i + n)) - curr_main_row(ST(i)))
    return constraints


def write_to_ram_any_of(b, number_of_words):
    groups = [[indicator_polynomial(b, n) * c for c in shrink_stack_by_n_and_write_n_elements_to_ram(b, n)]
              for n in number_of_words]
    return combine_mutually_exclusive_constraint_groups(b, groups)


def read_from_ram_any_of(b, number_of_words):
    groups = [[indicator_polynomial(b, n) * c for c in grow_stack_by_n_and_read_n_elements_from_ram(b, n)]
              for n in number_of_words]
    return combine_mutually_exclusive_constraint_groups(b, groups)


def xx_product(x, y):  # processor.rs:1919-1930
    x_0, x_1, x_2 = x
    y_0, y_1, y_2 = y
    z0 = x_0 * y_0
    z1 = x_1 * y_0 + x_0 * y_1
    z2 = x_2 * y_0 + x_1 * y_1 + x_0 * y_2
    z3 = x_2 * y_1 + x_1 * y_2
    z4 = x_2 * y_2
    return [z0 - z3, z1 - z4 + z3, z2 + z4]


def xb_product(x, y):  # processor.rs:1932-1941
    return [x[0] * y, x[1] * y, x[2] * y]


# ------------------------------------------------------------------------------------------------
# per-instruction transition constraints (processor.rs:805-2020)
def instruction_pop(b):
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b)
            + stack_shrinks_by_any_of(b, NUMBER_OF_WORDS_LEGAL_VALUES) + prohibit_any_illegal_number_of_words(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_push(b):
    specific = [_nm(b)(M.ST0) - _cm(b)(M.NIA)]
    return (specific + instruction_group_grow_op_stack(b) + instruction_group_step_2(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_divine(b):
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b)
            + stack_grows_by_any_of(b, NUMBER_OF_WORDS_LEGAL_VALUES) + prohibit_any_illegal_number_of_words(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def _rearranged_stack_instruction(b, rearrange):
    curr_row, next_row = _cm(b), _nm(b)
    stack = [ST(i) for i in range(NUM_OP_STACK_REGISTERS)]
    next_stack = [next_row(st) for st in stack]

    def compress(stk):
        weight = lambda i: b.challenge(stack_weight_by_index(i))
        return circuit_sum(weight(i) * st for i, st in enumerate(stk))

    def item(i):
        indicator = indicator_polynomial(b, i)
        compressed_next = compress(next_stack)
        curr_stack = [curr_row(st) for st in rearrange(list(stack), i)]
        return indicator * (compressed_next - compress(curr_stack))

    combined = circuit_sum(item(i) for i in range(NUM_OP_STACK_REGISTERS))
    return ([combined] + instruction_group_decompose_arg(b) + instruction_group_step_2(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b) + instruction_group_keep_op_stack_height(b))


def instruction_pick(b):
    def rearrange(stack, i):
        new_top = stack.pop(i)
        stack.insert(0, new_top)
        return stack
    return _rearranged_stack_instruction(b, rearrange)


def instruction_place(b):
    def rearrange(stack, i):
        old_top = stack.pop(0)
        stack.insert(i, old_top)
        return stack
    return _rearranged_stack_instruction(b, rearrange)


def instruction_swap(b):
    def rearrange(stack, i):
        stack[0], stack[i] = stack[i], stack[0]
        return stack
    return _rearranged_stack_instruction(b, rearrange)


def instruction_dup(b):
    curr_row, next_row = _cm(b), _nm(b)
    duplicate_element = lambda i: indicator_polynomial(b, i) * (next_row(M.ST0) - curr_row(ST(i)))
    duplicate_indicated_element = circuit_sum(duplicate_element(i) for i in range(NUM_OP_STACK_REGISTERS))
    return ([duplicate_indicated_element] + instruction_group_decompose_arg(b) + instruction_group_step_2(b)
            + instruction_group_grow_op_stack(b) + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_nop(b):
    return (instruction_group_step_1(b) + instruction_group_keep_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def next_instruction_range_check_constraints_for_instruction_skiz(b):
    constant = b.b_constant
    curr_main_row = _cm(b)
    is_0_or_1 = lambda var: curr_main_row(var) * (curr_main_row(var) - constant(1))
    is_0_or_1_or_2_or_3 = lambda var: (curr_main_row(var)
                                       * (curr_main_row(var) - constant(1))
                                       * (curr_main_row(var) - constant(2))
                                       * (curr_main_row(var) - constant(3)))
    return [is_0_or_1(M.HV1), is_0_or_1_or_2_or_3(M.HV2), is_0_or_1_or_2_or_3(M.HV3),
            is_0_or_1_or_2_or_3(M.HV4), is_0_or_1_or_2_or_3(M.HV5)]


def instruction_skiz(b):
    constant = b.b_constant
    one = lambda: constant(1)
    curr_main_row, next_main_row = _cm(b), _nm(b)

    hv0_is_inverse_of_st0 = curr_main_row(M.HV0) * curr_main_row(M.ST0) - one()
    hv0_is_inverse_of_st0_or_hv0_is_0 = hv0_is_inverse_of_st0 * curr_main_row(M.HV0)
    hv0_is_inverse_of_st0_or_st0_is_0 = hv0_is_inverse_of_st0 * curr_main_row(M.ST0)

    nia_decomposes_to_hvs = (curr_main_row(M.NIA)
                             - curr_main_row(M.HV1)
                             - constant(1 << 1) * curr_main_row(M.HV2)
                             - constant(1 << 3) * curr_main_row(M.HV3)
                             - constant(1 << 5) * curr_main_row(M.HV4)
                             - constant(1 << 7) * curr_main_row(M.HV5))

    ip_case_1 = (next_main_row(M.IP) - curr_main_row(M.IP) - constant(1)) * curr_main_row(M.ST0)
    ip_case_2 = ((next_main_row(M.IP) - curr_main_row(M.IP) - constant(2))
                 * (curr_main_row(M.ST0) * curr_main_row(M.HV0) - one())
                 * (curr_main_row(M.HV1) - one()))
    ip_case_3 = ((next_main_row(M.IP) - curr_main_row(M.IP) - constant(3))
                 * (curr_main_row(M.ST0) * curr_main_row(M.HV0) - one())
                 * curr_main_row(M.HV1))
    ip_incr_by_1_or_2_or_3 = ip_case_1 + ip_case_2 + ip_case_3

    specific = [hv0_is_inverse_of_st0_or_hv0_is_0, hv0_is_inverse_of_st0_or_st0_is_0, nia_decomposes_to_hvs,
                ip_incr_by_1_or_2_or_3]
    return (specific + next_instruction_range_check_constraints_for_instruction_skiz(b)
            + instruction_group_keep_jump_stack(b) + instruction_group_shrink_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_call(b):
    constant = b.b_constant
    curr_main_row, next_main_row = _cm(b), _nm(b)
    jsp_incr_1 = next_main_row(M.JSP) - curr_main_row(M.JSP) - constant(1)
    jso_becomes_ip_plus_2 = next_main_row(M.JSO) - curr_main_row(M.IP) - constant(2)
    jsd_becomes_nia = next_main_row(M.JSD) - curr_main_row(M.NIA)
    ip_becomes_nia = next_main_row(M.IP) - curr_main_row(M.NIA)
    specific = [jsp_incr_1, jso_becomes_ip_plus_2, jsd_becomes_nia, ip_becomes_nia]
    return specific + instruction_group_keep_op_stack(b) + instruction_group_no_ram(b) + instruction_group_no_io(b)


def instruction_return(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    jsp_decrements_by_1 = next_main_row(M.JSP) - curr_main_row(M.JSP) + b.b_constant(1)
    ip_is_set_to_jso = next_main_row(M.IP) - curr_main_row(M.JSO)
    specific = [jsp_decrements_by_1, ip_is_set_to_jso]
    return specific + instruction_group_keep_op_stack(b) + instruction_group_no_ram(b) + instruction_group_no_io(b)


def instruction_recurse(b):
    specific = [_nm(b)(M.IP) - _cm(b)(M.JSD)]
    return (specific + instruction_group_keep_jump_stack(b) + instruction_group_keep_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_recurse_or_return(b):
    one = lambda: b.b_constant(1)
    curr_row, next_row = _cm(b), _nm(b)
    st5_eq_st6 = lambda: curr_row(M.HV0) * (curr_row(M.ST6) - curr_row(M.ST5))
    st5_neq_st6 = lambda: one() - st5_eq_st6()

    c0 = st5_neq_st6() * curr_row(M.HV0)
    c1 = st5_neq_st6() * (curr_row(M.ST6) - curr_row(M.ST5))
    specific = [c0, c1]

    maybe_return = [st5_neq_st6() * (next_row(M.IP) - curr_row(M.JSO)),
                    st5_neq_st6() * (next_row(M.JSP) - curr_row(M.JSP) + one())]
    maybe_recurse = [st5_eq_st6() * (next_row(M.IP) - curr_row(M.JSD)),
                     st5_eq_st6() * (next_row(M.JSP) - curr_row(M.JSP)),
                     st5_eq_st6() * (next_row(M.JSO) - curr_row(M.JSO)),
                     st5_eq_st6() * (next_row(M.JSD) - curr_row(M.JSD))]
    specific.extend(combine_mutually_exclusive_constraint_groups(b, [maybe_return, maybe_recurse]))
    return specific + instruction_group_keep_op_stack(b) + instruction_group_no_ram(b) + instruction_group_no_io(b)


def instruction_assert(b):
    specific = [_cm(b)(M.ST0) - b.b_constant(1)]
    return (specific + instruction_group_step_1(b) + instruction_group_shrink_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_halt(b):
    specific = [_nm(b)(M.CI) - _cm(b)(M.CI)]
    return (specific + instruction_group_step_1(b) + instruction_group_keep_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_read_mem(b):
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b)
            + read_from_ram_any_of(b, NUMBER_OF_WORDS_LEGAL_VALUES) + prohibit_any_illegal_number_of_words(b)
            + instruction_group_no_io(b))


def instruction_write_mem(b):
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b)
            + write_to_ram_any_of(b, NUMBER_OF_WORDS_LEGAL_VALUES) + prohibit_any_illegal_number_of_words(b)
            + instruction_group_no_io(b))


def instruction_hash(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    shrinks = [next_main_row(ST(i)) - curr_main_row(ST(i + 5)) for i in range(5, 11)]
    shrinks.append(next_main_row(M.OpStackPointer) - curr_main_row(M.OpStackPointer) + b.b_constant(5))
    shrinks.append(running_product_op_stack_accounts_for_shrinking_stack_by(b, 5))
    return instruction_group_step_1(b) + shrinks + instruction_group_no_ram(b) + instruction_group_no_io(b)


def instruction_merkle_step_shared_constraints(b):
    constant = b.b_constant
    one = lambda: constant(1)
    curr, nxt = _cm(b), _nm(b)
    hv5_is_0_or_1 = curr(M.HV5) * (curr(M.HV5) - one())
    new_st5_is_previous_st5_div_2 = constant(2) * nxt(M.ST5) + curr(M.HV5) - curr(M.ST5)
    return [hv5_is_0_or_1, new_st5_is_previous_st5_div_2] + instruction_group_step_1(b) + instruction_group_no_io(b)


def instruction_merkle_step(b):
    return (instruction_merkle_step_shared_constraints(b) + instruction_group_op_stack_remains_except_top_n(b, 6)
            + instruction_group_no_ram(b))


def instruction_merkle_step_mem(b):
    constant = b.b_constant
    stack_weight = lambda i: b.challenge(stack_weight_by_index(i))
    curr, nxt = _cm(b), _nm(b)
    ram_pointers = [curr(M.ST7) + constant(i) for i in range(5)]
    ram_read_destinations = [curr(HV(i)) for i in range(5)]
    read_from_ram_to_hvs = read_from_ram_to(b, ram_pointers, ram_read_destinations)
    st6_does_not_change = nxt(M.ST6) - curr(M.ST6)
    st7_increments_by_5 = nxt(M.ST7) - curr(M.ST7) - constant(5)
    st6_and_st7_update_correctly = stack_weight(6) * st6_does_not_change + stack_weight(7) * st7_increments_by_5
    return ([st6_and_st7_update_correctly, read_from_ram_to_hvs] + instruction_merkle_step_shared_constraints(b)
            + instruction_group_op_stack_remains_except_top_n(b, 8))


def instruction_assert_vector(b):
    curr_main_row = _cm(b)
    specific = [curr_main_row(ST(i + 5)) - curr_main_row(ST(i)) for i in range(5)]
    return (specific + instruction_group_step_1(b) + constraints_for_shrinking_stack_by(b, 5)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_sponge_init(b):
    return (instruction_group_step_1(b) + instruction_group_keep_op_stack(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_sponge_absorb(b):
    return (instruction_group_step_1(b) + constraints_for_shrinking_stack_by(b, 10)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_sponge_absorb_mem(b):
    curr, nxt = _cm(b), _nm(b)
    constant = b.b_constant
    increment_ram_pointer = nxt(M.ST0) - curr(M.ST0) - constant(TIP5_RATE)
    ram_pointers = [curr(M.ST0) + constant(i) for i in range(10)]
    ram_read_destinations = [nxt(M.ST1), nxt(M.ST2), nxt(M.ST3), nxt(M.ST4),
                             curr(M.HV0), curr(M.HV1), curr(M.HV2), curr(M.HV3), curr(M.HV4), curr(M.HV5)]
    read_from_ram = read_from_ram_to(b, ram_pointers, ram_read_destinations)
    return ([increment_ram_pointer, read_from_ram] + instruction_group_step_1(b)
            + instruction_group_op_stack_remains_except_top_n(b, 5) + instruction_group_no_io(b))


def instruction_sponge_squeeze(b):
    return (instruction_group_step_1(b) + constraints_for_growing_stack_by(b, 10)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_add(b):
    specific = [_nm(b)(M.ST0) - _cm(b)(M.ST0) - _cm(b)(M.ST1)]
    return (specific + instruction_group_step_1(b) + instruction_group_binop(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_addi(b):
    specific = [_nm(b)(M.ST0) - _cm(b)(M.ST0) - _cm(b)(M.NIA)]
    return (specific + instruction_group_step_2(b) + instruction_group_op_stack_remains_except_top_n(b, 1)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_mul(b):
    specific = [_nm(b)(M.ST0) - _cm(b)(M.ST0) * _cm(b)(M.ST1)]
    return (specific + instruction_group_step_1(b) + instruction_group_binop(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_invert(b):
    specific = [_nm(b)(M.ST0) * _cm(b)(M.ST0) - b.b_constant(1)]
    return (specific + instruction_group_step_1(b) + instruction_group_op_stack_remains_except_top_n(b, 1)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_eq(b):
    one = lambda: b.b_constant(1)
    curr_main_row, next_main_row = _cm(b), _nm(b)
    st0_eq_st1 = lambda: one() - curr_main_row(M.HV0) * (curr_main_row(M.ST1) - curr_main_row(M.ST0))
    hv0_is_inverse_of_diff_or_hv0_is_0 = curr_main_row(M.HV0) * st0_eq_st1()
    hv0_is_inverse_of_diff_or_diff_is_0 = (curr_main_row(M.ST1) - curr_main_row(M.ST0)) * st0_eq_st1()
    st0_becomes_1_if_diff_is_not_invertible = next_main_row(M.ST0) - st0_eq_st1()
    specific = [hv0_is_inverse_of_diff_or_hv0_is_0, hv0_is_inverse_of_diff_or_diff_is_0,
                st0_becomes_1_if_diff_is_not_invertible]
    return (specific + instruction_group_step_1(b) + instruction_group_binop(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_split(b):
    constant = b.b_constant
    one = lambda: constant(1)
    curr_main_row, next_main_row = _cm(b), _nm(b)
    st0_decomposes = curr_main_row(M.ST0) - (constant(1 << 32) * next_main_row(M.ST1) + next_main_row(M.ST0))
    hv0 = curr_main_row(M.HV0)
    hi = next_main_row(M.ST1)
    lo = next_main_row(M.ST0)
    ffff_ffff = constant(0xFFFF_FFFF)
    hv0_holds_inverse_or_low_bits_are_0 = lo * (hv0 * (hi - ffff_ffff) - one())
    specific = [st0_decomposes, hv0_holds_inverse_or_low_bits_are_0]
    return (specific + instruction_group_grow_op_stack_and_top_two_elements_unconstrained(b)
            + instruction_group_step_1(b) + instruction_group_no_ram(b) + instruction_group_no_io(b))


def _binop_u32(b):
    return (instruction_group_step_1(b) + instruction_group_binop(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


instruction_lt = instruction_and = instruction_xor = instruction_pow = _binop_u32


def _unop_u32(b):
    return (instruction_group_step_1(b) + instruction_group_op_stack_remains_except_top_n(b, 1)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


instruction_log_2_floor = instruction_pop_count = _unop_u32


def instruction_div_mod(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    specific = [curr_main_row(M.ST0) - curr_main_row(M.ST1) * next_main_row(M.ST1) - next_main_row(M.ST0)]
    return (specific + instruction_group_step_1(b) + instruction_group_op_stack_remains_except_top_n(b, 2)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_xx_add(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    specific = [next_main_row(ST(i)) - curr_main_row(ST(i)) - curr_main_row(ST(i + 3)) for i in range(3)]
    return (specific + constraints_for_shrinking_stack_by_3_and_top_3_unconstrained(b) + instruction_group_step_1(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_xx_mul(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    x0, x1, x2, y0, y1, y2 = [curr_main_row(ST(i)) for i in range(6)]
    c0, c1, c2 = xx_product([x0, x1, x2], [y0, y1, y2])
    specific = [next_main_row(M.ST0) - c0, next_main_row(M.ST1) - c1, next_main_row(M.ST2) - c2]
    return (specific + constraints_for_shrinking_stack_by_3_and_top_3_unconstrained(b) + instruction_group_step_1(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_xinv(b):
    constant = b.b_constant
    c, n = _cm(b), _nm(b)
    first = c(M.ST0) * n(M.ST0) - c(M.ST2) * n(M.ST1) - c(M.ST1) * n(M.ST2) - constant(1)
    second = (c(M.ST1) * n(M.ST0) + c(M.ST0) * n(M.ST1) - c(M.ST2) * n(M.ST2)
              + c(M.ST2) * n(M.ST1) + c(M.ST1) * n(M.ST2))
    third = c(M.ST2) * n(M.ST0) + c(M.ST1) * n(M.ST1) + c(M.ST0) * n(M.ST2) + c(M.ST2) * n(M.ST2)
    return ([first, second, third] + instruction_group_op_stack_remains_except_top_n(b, 3) + instruction_group_step_1(b)
            + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_xb_mul(b):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    x, y0, y1, y2 = [curr_main_row(ST(i)) for i in range(4)]
    c0, c1, c2 = xb_product([y0, y1, y2], x)
    specific = [next_main_row(M.ST0) - c0, next_main_row(M.ST1) - c1, next_main_row(M.ST2) - c2]
    return (specific + instruction_group_op_stack_shrinks_and_top_three_elements_unconstrained(b)
            + instruction_group_step_1(b) + instruction_group_no_ram(b) + instruction_group_no_io(b))


def instruction_read_io(b):
    groups = [grow_stack_by_n_and_read_n_symbols_from_input(b, n) for n in NUMBER_OF_WORDS_LEGAL_VALUES]
    read_any_legal_number_of_words = combine_mutually_exclusive_constraint_groups(b, groups)
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b) + read_any_legal_number_of_words
            + prohibit_any_illegal_number_of_words(b) + instruction_group_no_ram(b)
            + [running_evaluation_for_standard_output_remains_unchanged(b)])


def instruction_write_io(b):
    groups = [shrink_stack_by_n_and_write_n_symbols_to_output(b, n) for n in NUMBER_OF_WORDS_LEGAL_VALUES]
    write_any = combine_mutually_exclusive_constraint_groups(b, groups)
    return (instruction_group_step_2(b) + instruction_group_decompose_arg(b) + write_any
            + prohibit_any_illegal_number_of_words(b) + instruction_group_no_ram(b)
            + [running_evaluation_for_standard_input_remains_unchanged(b)])


def _horner_step(b, ram_pointers, destinations, st5_decrement, add_hv):
    curr_main_row, next_main_row = _cm(b), _nm(b)
    constant = b.b_constant
    read_from_ram = read_from_ram_to(b, ram_pointers, destinations)
    indeterminate = [curr_main_row(M.ST0), curr_main_row(M.ST1), curr_main_row(M.ST2)]
    evaluation = [curr_main_row(M.ST7), curr_main_row(M.ST8), curr_main_row(M.ST9)]
    product_0, product_1, product_2 = xx_product(indeterminate, evaluation)
    stack_weight = lambda i: b.challenge(stack_weight_by_index(i))
    curr_stack_compressed = (stack_weight(0) * curr_main_row(M.ST0)
                             + stack_weight(1) * curr_main_row(M.ST1)
                             + stack_weight(2) * curr_main_row(M.ST2)
                             + stack_weight(3) * curr_main_row(M.ST3)
                             + stack_weight(4) * curr_main_row(M.ST4)
                             + stack_weight(5) * (curr_main_row(M.ST5) - constant(st5_decrement))
                             + stack_weight(6) * curr_main_row(M.ST6)
                             + stack_weight(7) * (product_0 + curr_main_row(M.HV0))
                             + stack_weight(8) * (product_1 + curr_main_row(M.HV1) if add_hv else product_1)
                             + stack_weight(9) * (product_2 + curr_main_row(M.HV2) if add_hv else product_2))
    next_stack_compressed = circuit_sum(stack_weight(i) * next_main_row(ST(i)) for i in range(10))
    stack_changes_correctly = next_stack_compressed - curr_stack_compressed
    return ([stack_changes_correctly, read_from_ram] + instruction_group_no_io(b) + instruction_group_step_1(b)
            + instruction_group_op_stack_remains_except_top_n(b, 10))


def instruction_b_horner_step(b):
    curr_main_row = _cm(b)
    ram_pointers = [curr_main_row(M.ST5)]
    destinations = [curr_main_row(M.HV0)]
    return _horner_step(b, ram_pointers, destinations, 1, False)


def instruction_x_horner_step(b):
    curr_main_row = _cm(b)
    ram_pointers = [curr_main_row(M.ST5) - b.b_constant(i) for i in range(3)]
    destinations = [curr_main_row(M.HV2), curr_main_row(M.HV1), curr_main_row(M.HV0)]
    return _horner_step(b, ram_pointers, destinations, 3, True)


_INSTRUCTION_FUNCTIONS = {
    "Pop": instruction_pop, "Push": instruction_push, "Divine": instruction_divine, "Pick": instruction_pick,
    "Place": instruction_place, "Dup": instruction_dup, "Swap": instruction_swap, "Halt": instruction_halt,
    "Nop": instruction_nop, "Skiz": instruction_skiz, "Call": instruction_call, "Return": instruction_return,
    "Recurse": instruction_recurse, "RecurseOrReturn": instruction_recurse_or_return, "Assert": instruction_assert,
    "ReadMem": instruction_read_mem, "WriteMem": instruction_write_mem, "Hash": instruction_hash,
    "AssertVector": instruction_assert_vector, "SpongeInit": instruction_sponge_init,
    "SpongeAbsorb": instruction_sponge_absorb, "SpongeAbsorbMem": instruction_sponge_absorb_mem,
    "SpongeSqueeze": instruction_sponge_squeeze, "Add": instruction_add, "AddI": instruction_addi,
    "Mul": instruction_mul, "Invert": instruction_invert, "Eq": instruction_eq, "Split": instruction_split,
    "Lt": instruction_lt, "And": instruction_and, "Xor": instruction_xor, "Log2Floor": instruction_log_2_floor,
    "Pow": instruction_pow, "DivMod": instruction_div_mod, "PopCount": instruction_pop_count,
    "XxAdd": instruction_xx_add, "XxMul": instruction_xx_mul, "XInvert": instruction_xinv, "XbMul": instruction_xb_mul,
    "ReadIo": instruction_read_io, "WriteIo": instruction_write_io, "MerkleStep": instruction_merkle_step,
    "MerkleStepMem": instruction_merkle_step_mem, "BHornerStep": instruction_b_horner_step,
    "XHornerStep": instruction_x_horner_step,
}


def transition_constraints_for_instruction(b, instruction):  # processor.rs:2022-2073
    return _INSTRUCTION_FUNCTIONS[instruction](b)


# ------------------------------------------------------------------------------------------------
# table-linking transition constraints (processor.rs:2086-2117, 2165-2191, 2575-3180)
def log_derivative_accumulates_clk_next(b):
    challenge = b.challenge
    next_main_row, curr_aux_row, next_aux_row = _nm(b), _ca(b), _na(b)
    return ((next_aux_row(A.ClockJumpDifferenceLookupServerLogDerivative)
             - curr_aux_row(A.ClockJumpDifferenceLookupServerLogDerivative))
            * (challenge(Ch.ClockJumpDifferenceLookupIndeterminate) - next_main_row(M.CLK))
            - next_main_row(M.ClockJumpDifferenceLookupMultiplicity))


def log_derivative_for_instruction_lookup_updates_correctly(b):
    one = lambda: b.b_constant(1)
    challenge = b.challenge
    next_main_row, curr_aux_row, next_aux_row = _nm(b), _ca(b), _na(b)
    compressed_row = (challenge(Ch.ProgramAddressWeight) * next_main_row(M.IP)
                      + challenge(Ch.ProgramInstructionWeight) * next_main_row(M.CI)
                      + challenge(Ch.ProgramNextInstructionWeight) * next_main_row(M.NIA))
    log_derivative_updates = ((next_aux_row(A.InstructionLookupClientLogDerivative)
                               - curr_aux_row(A.InstructionLookupClientLogDerivative))
                              * (challenge(Ch.InstructionLookupIndeterminate) - compressed_row)
                              - one())
    log_derivative_remains = (next_aux_row(A.InstructionLookupClientLogDerivative)
                              - curr_aux_row(A.InstructionLookupClientLogDerivative))
    return ((one() - next_main_row(M.IsPadding)) * log_derivative_updates
            + next_main_row(M.IsPadding) * log_derivative_remains)


def running_product_for_jump_stack_table_updates_correctly(b):
    challenge = b.challenge
    next_main_row, curr_aux_row, next_aux_row = _nm(b), _ca(b), _na(b)
    compressed_row = (challenge(Ch.JumpStackClkWeight) * next_main_row(M.CLK)
                      + challenge(Ch.JumpStackCiWeight) * next_main_row(M.CI)
                      + challenge(Ch.JumpStackJspWeight) * next_main_row(M.JSP)
                      + challenge(Ch.JumpStackJsoWeight) * next_main_row(M.JSO)
                      + challenge(Ch.JumpStackJsdWeight) * next_main_row(M.JSD))
    return (next_aux_row(A.JumpStackTablePermArg)
            - curr_aux_row(A.JumpStackTablePermArg) * (challenge(Ch.JumpStackIndeterminate) - compressed_row))


def running_evaluation_hash_input_updates_correctly(b):
    constant = b.b_constant
    one = lambda: constant(1)
    challenge = b.challenge
    next_main_row, curr_aux_row, next_aux_row = _nm(b), _ca(b), _na(b)

    hash_deselector = instruction_deselector_next_row(b, "Hash")
    merkle_step_deselector = instruction_deselector_next_row(b, "MerkleStep")
    merkle_step_mem_deselector = instruction_deselector_next_row(b, "MerkleStepMem")
    hash_and_merkle_step_selector = ((next_main_row(M.CI) - constant(OPCODE["Hash"]))
                                     * (next_main_row(M.CI) - constant(OPCODE["MerkleStep"]))
                                     * (next_main_row(M.CI) - constant(OPCODE["MerkleStepMem"])))

    weights = [challenge(stack_weight_by_index(i)) for i in range(10)]
    state_for_hash = [next_main_row(ST(i)) for i in range(10)]
    compressed_hash_row = circuit_sum(w * s for w, s in zip(weights, state_for_hash))

    is_left_sibling = lambda: next_main_row(M.HV5)
    is_right_sibling = lambda: one() - next_main_row(M.HV5)
    merkle_step_state_element = lambda l, r: is_right_sibling() * next_main_row(l) + is_left_sibling() * next_main_row(r)
    state_for_merkle_step = ([merkle_step_state_element(ST(i), HV(i)) for i in range(5)]
                             + [merkle_step_state_element(HV(i), ST(i)) for i in range(5)])
    compressed_merkle_step_row = circuit_sum(w * s for w, s in zip(weights, state_for_merkle_step))

    running_evaluation_updates_with = lambda compressed_row: (
        next_aux_row(A.HashInputEvalArg)
        - challenge(Ch.HashInputIndeterminate) * curr_aux_row(A.HashInputEvalArg)
        - compressed_row)
    running_evaluation_remains = next_aux_row(A.HashInputEvalArg) - curr_aux_row(A.HashInputEvalArg)

    return (hash_and_merkle_step_selector * running_evaluation_remains
            + hash_deselector * running_evaluation_updates_with(compressed_hash_row)
            + merkle_step_deselector * running_evaluation_updates_with(compressed_merkle_step_row)
            + merkle_step_mem_deselector * running_evaluation_updates_with(compressed_merkle_step_row))


def running_evaluation_hash_digest_updates_correctly(b):
    constant, challenge = b.b_constant, b.challenge
    curr_main_row, next_main_row, curr_aux_row, next_aux_row = _cm(b), _nm(b), _ca(b), _na(b)

    hash_deselector = instruction_deselector_current_row(b, "Hash")
    merkle_step_deselector = instruction_deselector_current_row(b, "MerkleStep")
    merkle_step_mem_deselector = instruction_deselector_current_row(b, "MerkleStepMem")
    hash_and_merkle_step_selector = ((curr_main_row(M.CI) - constant(OPCODE["Hash"]))
                                     * (curr_main_row(M.CI) - constant(OPCODE["MerkleStep"]))
                                     * (curr_main_row(M.CI) - constant(OPCODE["MerkleStepMem"])))

    weights = [challenge(stack_weight_by_index(i)) for i in range(5)]
    state = [next_main_row(ST(i)) for i in range(5)]
    compressed_row = circuit_sum(w * s for w, s in zip(weights, state))

    running_evaluation_updates = (next_aux_row(A.HashDigestEvalArg)
                                  - challenge(Ch.HashDigestIndeterminate) * curr_aux_row(A.HashDigestEvalArg)
                                  - compressed_row)
    running_evaluation_remains = next_aux_row(A.HashDigestEvalArg) - curr_aux_row(A.HashDigestEvalArg)

    return (hash_and_merkle_step_selector * running_evaluation_remains
            + (hash_deselector + merkle_step_deselector + merkle_step_mem_deselector) * running_evaluation_updates)


def running_evaluation_sponge_updates_correctly(b):
    constant, challenge = b.b_constant, b.challenge
    curr_main_row, next_main_row, curr_aux_row, next_aux_row = _cm(b), _nm(b), _ca(b), _na(b)

    sponge_init_deselector = instruction_deselector_current_row(b, "SpongeInit")
    sponge_absorb_deselector = instruction_deselector_current_row(b, "SpongeAbsorb")
    sponge_absorb_mem_deselector = instruction_deselector_current_row(b, "SpongeAbsorbMem")
    sponge_squeeze_deselector = instruction_deselector_current_row(b, "SpongeSqueeze")

    sponge_instruction_selector = ((curr_main_row(M.CI) - constant(OPCODE["SpongeInit"]))
                                   * (curr_main_row(M.CI) - constant(OPCODE["SpongeAbsorb"]))
                                   * (curr_main_row(M.CI) - constant(OPCODE["SpongeAbsorbMem"]))
                                   * (curr_main_row(M.CI) - constant(OPCODE["SpongeSqueeze"])))

    def weighted_sum(state):
        weights = [challenge(stack_weight_by_index(i)) for i in range(10)]
        return circuit_sum(w * st for w, st in zip(weights, state))

    compressed_row_current = weighted_sum([curr_main_row(ST(i)) for i in range(10)])
    compressed_row_next = weighted_sum([next_main_row(ST(i)) for i in range(10)])

    updates_for_sponge_init = (next_aux_row(A.SpongeEvalArg)
                               - challenge(Ch.SpongeIndeterminate) * curr_aux_row(A.SpongeEvalArg)
                               - challenge(Ch.HashCIWeight) * curr_main_row(M.CI))
    updates_for_absorb = updates_for_sponge_init - compressed_row_current
    updates_for_squeeze = updates_for_sponge_init - compressed_row_next
    running_evaluation_remains = next_aux_row(A.SpongeEvalArg) - curr_aux_row(A.SpongeEvalArg)

    stack_elements = [next_main_row(ST(i)) for i in range(1, 5)]
    hv_elements = [curr_main_row(HV(i)) for i in range(6)]
    compressed_row_absorb_mem = weighted_sum(stack_elements + hv_elements)
    updates_for_absorb_mem = (next_aux_row(A.SpongeEvalArg)
                              - challenge(Ch.SpongeIndeterminate) * curr_aux_row(A.SpongeEvalArg)
                              - challenge(Ch.HashCIWeight) * constant(OPCODE["SpongeAbsorb"])
                              - compressed_row_absorb_mem)

    return (sponge_instruction_selector * running_evaluation_remains
            + sponge_init_deselector * updates_for_sponge_init
            + sponge_absorb_deselector * updates_for_absorb
            + sponge_absorb_mem_deselector * updates_for_absorb_mem
            + sponge_squeeze_deselector * updates_for_squeeze)


def log_derivative_with_u32_table_updates_correctly(b):
    from .circuit import P

    constant, challenge = b.b_constant, b.challenge
    one = lambda: constant(1)
    two_inverse = b.b_constant(pow(2, -1, P))
    curr_main_row, next_main_row, curr_aux_row, next_aux_row = _cm(b), _nm(b), _ca(b), _na(b)

    split_deselector = instruction_deselector_current_row(b, "Split")
    lt_deselector = instruction_deselector_current_row(b, "Lt")
    and_deselector = instruction_deselector_current_row(b, "And")
    xor_deselector = instruction_deselector_current_row(b, "Xor")
    pow_deselector = instruction_deselector_current_row(b, "Pow")
    log_2_floor_deselector = instruction_deselector_current_row(b, "Log2Floor")
    div_mod_deselector = instruction_deselector_current_row(b, "DivMod")
    pop_count_deselector = instruction_deselector_current_row(b, "PopCount")
    merkle_step_deselector = instruction_deselector_current_row(b, "MerkleStep")
    merkle_step_mem_deselector = instruction_deselector_current_row(b, "MerkleStepMem")

    running_sum = curr_aux_row(A.U32LookupClientLogDerivative)
    running_sum_next = next_aux_row(A.U32LookupClientLogDerivative)

    split_factor = (challenge(Ch.U32Indeterminate)
                    - challenge(Ch.U32LhsWeight) * next_main_row(M.ST0)
                    - challenge(Ch.U32RhsWeight) * next_main_row(M.ST1)
                    - challenge(Ch.U32CiWeight) * curr_main_row(M.CI))
    binop_factor = (challenge(Ch.U32Indeterminate)
                    - challenge(Ch.U32LhsWeight) * curr_main_row(M.ST0)
                    - challenge(Ch.U32RhsWeight) * curr_main_row(M.ST1)
                    - challenge(Ch.U32CiWeight) * curr_main_row(M.CI)
                    - challenge(Ch.U32ResultWeight) * next_main_row(M.ST0))
    xor_factor = (challenge(Ch.U32Indeterminate)
                  - challenge(Ch.U32LhsWeight) * curr_main_row(M.ST0)
                  - challenge(Ch.U32RhsWeight) * curr_main_row(M.ST1)
                  - challenge(Ch.U32CiWeight) * constant(OPCODE["And"])
                  - challenge(Ch.U32ResultWeight)
                  * (curr_main_row(M.ST0) + curr_main_row(M.ST1) - next_main_row(M.ST0))
                  * two_inverse)
    unop_factor = (challenge(Ch.U32Indeterminate)
                   - challenge(Ch.U32LhsWeight) * curr_main_row(M.ST0)
                   - challenge(Ch.U32CiWeight) * curr_main_row(M.CI)
                   - challenge(Ch.U32ResultWeight) * next_main_row(M.ST0))
    div_mod_factor_for_lt = (challenge(Ch.U32Indeterminate)
                             - challenge(Ch.U32LhsWeight) * next_main_row(M.ST0)
                             - challenge(Ch.U32RhsWeight) * curr_main_row(M.ST1)
                             - challenge(Ch.U32CiWeight) * constant(OPCODE["Lt"])
                             - challenge(Ch.U32ResultWeight))
    div_mod_factor_for_range_check = (challenge(Ch.U32Indeterminate)
                                      - challenge(Ch.U32LhsWeight) * curr_main_row(M.ST0)
                                      - challenge(Ch.U32RhsWeight) * next_main_row(M.ST1)
                                      - challenge(Ch.U32CiWeight) * constant(OPCODE["Split"]))
    merkle_step_range_check_factor = (challenge(Ch.U32Indeterminate)
                                      - challenge(Ch.U32LhsWeight) * curr_main_row(M.ST5)
                                      - challenge(Ch.U32RhsWeight) * next_main_row(M.ST5)
                                      - challenge(Ch.U32CiWeight) * constant(OPCODE["Split"]))

    absorbs_split = (running_sum_next - running_sum) * split_factor - one()
    absorbs_binop = (running_sum_next - running_sum) * binop_factor - one()
    absorbs_xor = (running_sum_next - running_sum) * xor_factor - one()
    absorbs_unop = (running_sum_next - running_sum) * unop_factor - one()
    absorbs_merkle_step = (running_sum_next - running_sum) * merkle_step_range_check_factor - one()

    split_summand = split_deselector * absorbs_split
    lt_summand = lt_deselector * absorbs_binop
    and_summand = and_deselector * absorbs_binop
    xor_summand = xor_deselector * absorbs_xor
    pow_summand = pow_deselector * absorbs_binop
    log_2_floor_summand = log_2_floor_deselector * absorbs_unop
    div_mod_summand = div_mod_deselector * (
        (running_sum_next - running_sum) * div_mod_factor_for_lt * div_mod_factor_for_range_check
        - div_mod_factor_for_lt
        - div_mod_factor_for_range_check)
    pop_count_summand = pop_count_deselector * absorbs_unop
    merkle_step_summand = merkle_step_deselector * absorbs_merkle_step
    merkle_step_mem_summand = merkle_step_mem_deselector * absorbs_merkle_step
    no_update_summand = (one() - curr_main_row(M.IB2)) * (running_sum_next - running_sum)

    return (split_summand + lt_summand + and_summand + xor_summand + pow_summand + log_2_floor_summand
            + div_mod_summand + pop_count_summand + merkle_step_summand + merkle_step_mem_summand + no_update_summand)
