"""RAM Table AIR -- restated from /root/reference/triton-air/src/table/ram.rs:21-283
(prose: specification/src/random-access-memory-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, LOOKUP_ARG_INITIAL, MAIN, PERM_ARG_INITIAL, Ch

M, A = MAIN["Ram"], AUX["Ram"]
INSTRUCTION_TYPE_WRITE, INSTRUCTION_TYPE_READ, PADDING_INDICATOR = 0, 1, 2


def initial_constraints(b):
    challenge, constant, x_constant = b.challenge, b.b_constant, b.x_constant
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))

    first_row_is_padding_row = main_row(M.InstructionType) - constant(PADDING_INDICATOR)
    first_row_is_not_padding_row = ((main_row(M.InstructionType) - constant(INSTRUCTION_TYPE_READ))
                                    * (main_row(M.InstructionType) - constant(INSTRUCTION_TYPE_WRITE)))

    bcpc0_is_0 = main_row(M.BezoutCoefficientPolynomialCoefficient0)
    bc0_is_0 = aux_row(A.BezoutCoefficient0)
    bc1_is_bcpc1 = aux_row(A.BezoutCoefficient1) - main_row(M.BezoutCoefficientPolynomialCoefficient1)
    formal_derivative_is_1 = aux_row(A.FormalDerivative) - constant(1)
    running_product_polynomial_is_initialized_correctly = (
        aux_row(A.RunningProductOfRAMP)
        - challenge(Ch.RamTableBezoutRelationIndeterminate)
        + main_row(M.RamPointer))

    clock_jump_diff_log_derivative_is_default_initial = (
        aux_row(A.ClockJumpDifferenceLookupClientLogDerivative) - x_constant(LOOKUP_ARG_INITIAL))

    compressed_row = (main_row(M.CLK) * challenge(Ch.RamClkWeight)
                      + main_row(M.InstructionType) * challenge(Ch.RamInstructionTypeWeight)
                      + main_row(M.RamPointer) * challenge(Ch.RamPointerWeight)
                      + main_row(M.RamValue) * challenge(Ch.RamValueWeight))
    rppa_has_accumulated_first_row = (aux_row(A.RunningProductPermArg)
                                      - challenge(Ch.RamIndeterminate)
                                      + compressed_row)
    rppa_is_default_initial = aux_row(A.RunningProductPermArg) - x_constant(PERM_ARG_INITIAL)
    rppa_starts_correctly = (rppa_has_accumulated_first_row * first_row_is_padding_row
                             + rppa_is_default_initial * first_row_is_not_padding_row)
    return [bcpc0_is_0, bc0_is_0, bc1_is_bcpc1, running_product_polynomial_is_initialized_correctly,
            formal_derivative_is_1, rppa_starts_correctly, clock_jump_diff_log_derivative_is_default_initial]


def consistency_constraints(b):
    constant = b.b_constant
    instruction_type = lambda: b.input(Main(M.InstructionType))
    instruction_type_is_legal = ((instruction_type() - constant(INSTRUCTION_TYPE_WRITE))
                                 * (instruction_type() - constant(INSTRUCTION_TYPE_READ))
                                 * (instruction_type() - constant(PADDING_INDICATOR)))
    return [instruction_type_is_legal]


def transition_constraints(b):
    constant, challenge = b.b_constant, b.challenge
    curr_main_row = lambda col: b.input(CurrentMain(col))
    curr_aux_row = lambda col: b.input(CurrentAux(col))
    next_main_row = lambda col: b.input(NextMain(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    one = constant(1)
    bezout_challenge = challenge(Ch.RamTableBezoutRelationIndeterminate)

    clock = curr_main_row(M.CLK)
    ram_pointer = curr_main_row(M.RamPointer)
    ram_value = curr_main_row(M.RamValue)
    instruction_type = curr_main_row(M.InstructionType)
    inverse_of_ram_pointer_difference = curr_main_row(M.InverseOfRampDifference)
    bcpc0 = curr_main_row(M.BezoutCoefficientPolynomialCoefficient0)
    bcpc1 = curr_main_row(M.BezoutCoefficientPolynomialCoefficient1)

    running_product_ram_pointer = curr_aux_row(A.RunningProductOfRAMP)
    fd = curr_aux_row(A.FormalDerivative)
    bc0 = curr_aux_row(A.BezoutCoefficient0)
    bc1 = curr_aux_row(A.BezoutCoefficient1)
    rppa = curr_aux_row(A.RunningProductPermArg)
    clock_jump_diff_log_derivative = curr_aux_row(A.ClockJumpDifferenceLookupClientLogDerivative)

    clock_next = next_main_row(M.CLK)
    ram_pointer_next = next_main_row(M.RamPointer)
    ram_value_next = next_main_row(M.RamValue)
    instruction_type_next = next_main_row(M.InstructionType)
    bcpc0_next = next_main_row(M.BezoutCoefficientPolynomialCoefficient0)
    bcpc1_next = next_main_row(M.BezoutCoefficientPolynomialCoefficient1)

    running_product_ram_pointer_next = next_aux_row(A.RunningProductOfRAMP)
    fd_next = next_aux_row(A.FormalDerivative)
    bc0_next = next_aux_row(A.BezoutCoefficient0)
    bc1_next = next_aux_row(A.BezoutCoefficient1)
    rppa_next = next_aux_row(A.RunningProductPermArg)
    clock_jump_diff_log_derivative_next = next_aux_row(A.ClockJumpDifferenceLookupClientLogDerivative)

    next_row_is_padding_row = instruction_type_next - constant(PADDING_INDICATOR)
    if_current_padding_then_next_padding = ((instruction_type - constant(INSTRUCTION_TYPE_READ))
                                            * (instruction_type - constant(INSTRUCTION_TYPE_WRITE))
                                            * next_row_is_padding_row)

    ram_pointer_difference = ram_pointer_next - ram_pointer
    ram_pointer_changes = one - ram_pointer_difference * inverse_of_ram_pointer_difference

    iord_is_0_or_inverse = inverse_of_ram_pointer_difference * ram_pointer_changes
    diff_is_0_or_iord_is_inverse = ram_pointer_difference * ram_pointer_changes

    ram_pointer_changes_or_write_mem_or_ram_value_stays = (
        ram_pointer_changes * (constant(INSTRUCTION_TYPE_WRITE) - instruction_type_next) * (ram_value_next - ram_value))

    bcbp0_only_changes_if_ram_pointer_changes = ram_pointer_changes * (bcpc0_next - bcpc0)
    bcbp1_only_changes_if_ram_pointer_changes = ram_pointer_changes * (bcpc1_next - bcpc1)

    running_product_ram_pointer_updates_correctly = (
        ram_pointer_difference
        * (running_product_ram_pointer_next - running_product_ram_pointer * (bezout_challenge - ram_pointer_next))
        + ram_pointer_changes * (running_product_ram_pointer_next - running_product_ram_pointer))

    formal_derivative_updates_correctly = (
        ram_pointer_difference
        * (fd_next - running_product_ram_pointer - (bezout_challenge - ram_pointer_next) * fd)
        + ram_pointer_changes * (fd_next - fd))

    bezout_coefficient_0_is_constructed_correctly = (
        ram_pointer_difference * (bc0_next - bezout_challenge * bc0 - bcpc0_next)
        + ram_pointer_changes * (bc0_next - bc0))
    bezout_coefficient_1_is_constructed_correctly = (
        ram_pointer_difference * (bc1_next - bezout_challenge * bc1 - bcpc1_next)
        + ram_pointer_changes * (bc1_next - bc1))

    compressed_row = (clock_next * challenge(Ch.RamClkWeight)
                      + ram_pointer_next * challenge(Ch.RamPointerWeight)
                      + ram_value_next * challenge(Ch.RamValueWeight)
                      + instruction_type_next * challenge(Ch.RamInstructionTypeWeight))
    rppa_accumulates_next_row = rppa_next - rppa * (challenge(Ch.RamIndeterminate) - compressed_row)

    next_row_is_not_padding_row = ((instruction_type_next - constant(INSTRUCTION_TYPE_READ))
                                   * (instruction_type_next - constant(INSTRUCTION_TYPE_WRITE)))
    rppa_remains_unchanged = rppa_next - rppa
    rppa_updates_correctly = (rppa_accumulates_next_row * next_row_is_padding_row
                              + rppa_remains_unchanged * next_row_is_not_padding_row)

    clock_difference = clock_next - clock
    log_derivative_accumulates = (
        (clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative)
        * (challenge(Ch.ClockJumpDifferenceLookupIndeterminate) - clock_difference)
        - one)
    log_derivative_remains = clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative

    acc_or_ptr_changes_or_padding = log_derivative_accumulates * ram_pointer_changes * next_row_is_padding_row
    remains_or_ptr_same_or_padding = log_derivative_remains * ram_pointer_difference * next_row_is_padding_row
    remains_or_next_not_padding = log_derivative_remains * next_row_is_not_padding_row
    log_derivative_updates_correctly = (acc_or_ptr_changes_or_padding + remains_or_ptr_same_or_padding
                                        + remains_or_next_not_padding)

    return [if_current_padding_then_next_padding, iord_is_0_or_inverse, diff_is_0_or_iord_is_inverse,
            ram_pointer_changes_or_write_mem_or_ram_value_stays, bcbp0_only_changes_if_ram_pointer_changes,
            bcbp1_only_changes_if_ram_pointer_changes, running_product_ram_pointer_updates_correctly,
            formal_derivative_updates_correctly, bezout_coefficient_0_is_constructed_correctly,
            bezout_coefficient_1_is_constructed_correctly, rppa_updates_correctly, log_derivative_updates_correctly]


def terminal_constraints(b):
    aux_row = lambda col: b.input(Aux(col))
    bezout_relation_holds = (aux_row(A.BezoutCoefficient0) * aux_row(A.RunningProductOfRAMP)
                             + aux_row(A.BezoutCoefficient1) * aux_row(A.FormalDerivative)
                             - b.b_constant(1))
    return [bezout_relation_holds]
