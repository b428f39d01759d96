"""U32 Table AIR -- restated from /root/reference/triton-air/src/table/u32.rs:27-390
(prose: specification/src/u32-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, LOOKUP_ARG_INITIAL, MAIN, Ch
from .isa import OPCODE

M, A = MAIN["U32"], AUX["U32"]


def instruction_deselector(instruction_to_select, b, current_instruction):
    """u32.rs:372-390"""
    acc = b.b_constant(1)
    for instr in ("Split", "Lt", "And", "Log2Floor", "Pow", "PopCount"):
        if instr == instruction_to_select:
            continue
        acc = acc * (current_instruction - b.b_constant(OPCODE[instr]))
    return acc


def initial_constraints(b):
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))
    challenge = b.challenge
    one = b.b_constant(1)

    copy_flag = main_row(M.CopyFlag)
    lhs = main_row(M.LHS)
    rhs = main_row(M.RHS)
    ci = main_row(M.CI)
    result = main_row(M.Result)
    lookup_multiplicity = main_row(M.LookupMultiplicity)
    running_sum_log_derivative = aux_row(A.LookupServerLogDerivative)

    compressed_row = (challenge(Ch.U32LhsWeight) * lhs
                      + challenge(Ch.U32RhsWeight) * rhs
                      + challenge(Ch.U32CiWeight) * ci
                      + challenge(Ch.U32ResultWeight) * result)
    if_copy_flag_1_then_accumulated = copy_flag * (
        running_sum_log_derivative * (challenge(Ch.U32Indeterminate) - compressed_row) - lookup_multiplicity)
    default_initial = b.x_constant(LOOKUP_ARG_INITIAL)
    if_copy_flag_0_then_default = (copy_flag - one) * (running_sum_log_derivative - default_initial)
    return [if_copy_flag_0_then_default + if_copy_flag_1_then_accumulated]


def consistency_constraints(b):
    main_row = lambda col: b.input(Main(col))
    one = lambda: b.b_constant(1)
    two = lambda: b.b_constant(2)

    copy_flag = main_row(M.CopyFlag)
    bits = main_row(M.Bits)
    bits_minus_33_inv = main_row(M.BitsMinus33Inv)
    ci = main_row(M.CI)
    lhs = main_row(M.LHS)
    lhs_inv = main_row(M.LhsInv)
    rhs = main_row(M.RHS)
    rhs_inv = main_row(M.RhsInv)
    result = main_row(M.Result)
    lookup_multiplicity = main_row(M.LookupMultiplicity)

    desel = lambda instr: instruction_deselector(instr, b, ci)

    copy_flag_is_bit = copy_flag * (one() - copy_flag)
    copy_flag_is_0_or_bits_is_0 = copy_flag * bits
    bits_minus_33_inv_is_inverse = one() - bits_minus_33_inv * (bits - b.b_constant(33))
    lhs_inv_is_0_or_inverse = lhs_inv * (one() - lhs * lhs_inv)
    lhs_is_0_or_inverse = lhs * (one() - lhs * lhs_inv)
    rhs_inv_is_0_or_inverse = rhs_inv * (one() - rhs * rhs_inv)
    rhs_is_0_or_inverse = rhs * (one() - rhs * rhs_inv)
    lt_copy_flag_0 = (desel("Lt") * (copy_flag - one()) * (one() - lhs * lhs_inv) * (one() - rhs * rhs_inv)
                      * (result - two()))
    lt_copy_flag_1 = desel("Lt") * copy_flag * (one() - lhs * lhs_inv) * (one() - rhs * rhs_inv) * result
    and_init = desel("And") * (one() - lhs * lhs_inv) * (one() - rhs * rhs_inv) * result
    pow_init = desel("Pow") * (one() - rhs * rhs_inv) * (result - one())
    log_2_floor_init = desel("Log2Floor") * (copy_flag - one()) * (one() - lhs * lhs_inv) * (result + one())
    pop_count_init = desel("PopCount") * (one() - lhs * lhs_inv) * result
    if_log_2_floor_on_0_then_vm_crashes = desel("Log2Floor") * copy_flag * (one() - lhs * lhs_inv)
    if_copy_flag_is_0_then_lookup_multiplicity_is_0 = (copy_flag - one()) * lookup_multiplicity

    return [copy_flag_is_bit, copy_flag_is_0_or_bits_is_0, bits_minus_33_inv_is_inverse, lhs_inv_is_0_or_inverse,
            lhs_is_0_or_inverse, rhs_inv_is_0_or_inverse, rhs_is_0_or_inverse, lt_copy_flag_0, lt_copy_flag_1, and_init,
            pow_init, log_2_floor_init, pop_count_init, if_log_2_floor_on_0_then_vm_crashes,
            if_copy_flag_is_0_then_lookup_multiplicity_is_0]


def transition_constraints(b):
    curr_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))
    curr_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))
    challenge = b.challenge
    one = lambda: b.b_constant(1)
    two = lambda: b.b_constant(2)

    copy_flag = curr_main_row(M.CopyFlag)
    bits = curr_main_row(M.Bits)
    ci = curr_main_row(M.CI)
    lhs = curr_main_row(M.LHS)
    rhs = curr_main_row(M.RHS)
    result = curr_main_row(M.Result)
    running_sum_log_derivative = curr_aux_row(A.LookupServerLogDerivative)

    copy_flag_next = next_main_row(M.CopyFlag)
    bits_next = next_main_row(M.Bits)
    ci_next = next_main_row(M.CI)
    lhs_next = next_main_row(M.LHS)
    rhs_next = next_main_row(M.RHS)
    result_next = next_main_row(M.Result)
    lhs_inv_next = next_main_row(M.LhsInv)
    lookup_multiplicity_next = next_main_row(M.LookupMultiplicity)
    running_sum_log_derivative_next = next_aux_row(A.LookupServerLogDerivative)

    desel = lambda instr: instruction_deselector(instr, b, ci_next)

    ci_is_pow = ci - b.b_constant(OPCODE["Pow"])
    lhs_lsb = lhs - two() * lhs_next
    rhs_lsb = rhs - two() * rhs_next

    c0 = copy_flag_next * lhs * ci_is_pow
    c1 = copy_flag_next * rhs
    c2 = (copy_flag_next - one()) * (ci_next - ci)
    c3 = (copy_flag_next - one()) * lhs * ci_is_pow * (bits_next - bits - one())
    c4 = (copy_flag_next - one()) * rhs * (bits_next - bits - one())
    c5 = (copy_flag_next - one()) * ci_is_pow * lhs_lsb * (lhs_lsb - one())
    c6 = (copy_flag_next - one()) * rhs_lsb * (rhs_lsb - one())

    c7 = (copy_flag_next - one()) * desel("Lt") * (result_next - one()) * (result_next - two()) * result
    c8 = (copy_flag_next - one()) * desel("Lt") * result_next * (result_next - two()) * (result - one())
    c9 = ((copy_flag_next - one()) * desel("Lt") * result_next * (result_next - one())
          * (lhs_lsb - one()) * rhs_lsb * (result - one()))
    c10 = ((copy_flag_next - one()) * desel("Lt") * result_next * (result_next - one())
           * lhs_lsb * (rhs_lsb - one()) * result)
    c11 = ((copy_flag_next - one()) * desel("Lt") * result_next * (result_next - one())
           * (one() - lhs_lsb - rhs_lsb + two() * lhs_lsb * rhs_lsb)
           * (copy_flag - one()) * (result - two()))
    c12 = ((copy_flag_next - one()) * desel("Lt") * result_next * (result_next - one())
           * (one() - lhs_lsb - rhs_lsb + two() * lhs_lsb * rhs_lsb)
           * copy_flag * result)

    c13 = (copy_flag_next - one()) * desel("And") * (result - two() * result_next - lhs_lsb * rhs_lsb)

    c14 = ((copy_flag_next - one()) * desel("Log2Floor") * (one() - lhs_next * lhs_inv_next) * lhs * (result - bits))
    c15 = (copy_flag_next - one()) * desel("Log2Floor") * lhs_next * (result_next - result)

    c16 = (copy_flag_next - one()) * desel("Pow") * (lhs_next - lhs)
    c17 = (copy_flag_next - one()) * desel("Pow") * (rhs_lsb - one()) * (result - result_next * result_next)
    c18 = (copy_flag_next - one()) * desel("Pow") * rhs_lsb * (result - result_next * result_next * lhs)

    c19 = (copy_flag_next - one()) * desel("PopCount") * (result - result_next - lhs_lsb)

    c20 = (copy_flag_next - one()) * (running_sum_log_derivative_next - running_sum_log_derivative)

    compressed_row_next = (challenge(Ch.U32CiWeight) * ci_next
                           + challenge(Ch.U32LhsWeight) * lhs_next
                           + challenge(Ch.U32RhsWeight) * rhs_next
                           + challenge(Ch.U32ResultWeight) * result_next)
    c21 = copy_flag_next * (
        (running_sum_log_derivative_next - running_sum_log_derivative)
        * (challenge(Ch.U32Indeterminate) - compressed_row_next)
        - lookup_multiplicity_next)

    return [c0, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15, c16, c17, c18, c19, c20, c21]


def terminal_constraints(b):
    main_row = lambda col: b.input(Main(col))
    ci = main_row(M.CI)
    lhs = main_row(M.LHS)
    rhs = main_row(M.RHS)
    lhs_is_0_or_ci_is_pow = lhs * (ci - b.b_constant(OPCODE["Pow"]))
    return [lhs_is_0_or_ci_is_pow, rhs]
