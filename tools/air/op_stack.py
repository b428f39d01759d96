"""Op Stack Table AIR -- restated from /root/reference/triton-air/src/table/op_stack.rs:25-206
(prose: specification/src/operational-stack-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, LOOKUP_ARG_INITIAL, MAIN, PERM_ARG_INITIAL, Ch
from .isa import NUM_OP_STACK_REGISTERS

M, A = MAIN["OpStack"], AUX["OpStack"]
PADDING_VALUE = 2


def initial_constraints(b):
    challenge, constant, x_constant = b.challenge, b.b_constant, b.x_constant
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))

    initial_stack_length = constant(NUM_OP_STACK_REGISTERS)
    padding_indicator = constant(PADDING_VALUE)
    stack_pointer_is_16 = main_row(M.StackPointer) - initial_stack_length

    compressed_row = (challenge(Ch.OpStackClkWeight) * main_row(M.CLK)
                      + challenge(Ch.OpStackIb1Weight) * main_row(M.IB1ShrinkStack)
                      + challenge(Ch.OpStackPointerWeight) * initial_stack_length
                      + challenge(Ch.OpStackFirstUnderflowElementWeight) * main_row(M.FirstUnderflowElement))
    rppa_initial = challenge(Ch.OpStackIndeterminate) - compressed_row
    rppa_has_accumulated_first_row = aux_row(A.RunningProductPermArg) - rppa_initial
    rppa_is_default_initial = aux_row(A.RunningProductPermArg) - x_constant(PERM_ARG_INITIAL)

    first_row_is_padding_row = main_row(M.IB1ShrinkStack) - padding_indicator
    first_row_is_not_padding_row = main_row(M.IB1ShrinkStack) * (main_row(M.IB1ShrinkStack) - constant(1))
    rppa_starts_correctly = (rppa_has_accumulated_first_row * first_row_is_padding_row
                             + rppa_is_default_initial * first_row_is_not_padding_row)

    lookup_argument_initial = x_constant(LOOKUP_ARG_INITIAL)
    clock_jump_diff_log_derivative_is_initialized_correctly = (
        aux_row(A.ClockJumpDifferenceLookupClientLogDerivative) - lookup_argument_initial)
    return [stack_pointer_is_16, rppa_starts_correctly, clock_jump_diff_log_derivative_is_initialized_correctly]


def consistency_constraints(b):
    constant = b.b_constant
    ib1 = lambda: b.input(Main(M.IB1ShrinkStack))
    ib1_is_legal = ib1() * (ib1() - constant(1)) * (ib1() - constant(PADDING_VALUE))
    return [ib1_is_legal]


def transition_constraints(b):
    constant, challenge = b.b_constant, b.challenge
    current_main_row = lambda col: b.input(CurrentMain(col))
    current_aux_row = lambda col: b.input(CurrentAux(col))
    next_main_row = lambda col: b.input(NextMain(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    one = constant(1)
    padding_indicator = constant(PADDING_VALUE)

    clk = current_main_row(M.CLK)
    ib1_shrink_stack = current_main_row(M.IB1ShrinkStack)
    stack_pointer = current_main_row(M.StackPointer)
    first_underflow_element = current_main_row(M.FirstUnderflowElement)
    rppa = current_aux_row(A.RunningProductPermArg)
    clock_jump_diff_log_derivative = current_aux_row(A.ClockJumpDifferenceLookupClientLogDerivative)

    clk_next = next_main_row(M.CLK)
    ib1_shrink_stack_next = next_main_row(M.IB1ShrinkStack)
    stack_pointer_next = next_main_row(M.StackPointer)
    first_underflow_element_next = next_main_row(M.FirstUnderflowElement)
    rppa_next = next_aux_row(A.RunningProductPermArg)
    clock_jump_diff_log_derivative_next = next_aux_row(A.ClockJumpDifferenceLookupClientLogDerivative)

    stack_pointer_increases_by_1_or_does_not_change = (
        (stack_pointer_next - stack_pointer - one) * (stack_pointer_next - stack_pointer))
    sp_inc_or_underflow_unchanged_or_next_grows = (
        (stack_pointer_next - stack_pointer - one)
        * (first_underflow_element_next - first_underflow_element)
        * ib1_shrink_stack_next)

    next_row_is_padding_row = ib1_shrink_stack_next - padding_indicator
    if_current_row_is_padding_row_then_next_row_is_padding_row = (
        ib1_shrink_stack * (ib1_shrink_stack - one) * next_row_is_padding_row)

    compressed_row = (b.challenge(Ch.OpStackClkWeight) * clk_next
                      + b.challenge(Ch.OpStackIb1Weight) * ib1_shrink_stack_next
                      + b.challenge(Ch.OpStackPointerWeight) * stack_pointer_next
                      + b.challenge(Ch.OpStackFirstUnderflowElementWeight) * first_underflow_element_next)
    rppa_updates = rppa_next - rppa * (challenge(Ch.OpStackIndeterminate) - compressed_row)

    next_row_is_not_padding_row = ib1_shrink_stack_next * (ib1_shrink_stack_next - one)
    rppa_remains = rppa_next - rppa
    rppa_updates_correctly = rppa_updates * next_row_is_padding_row + rppa_remains * next_row_is_not_padding_row

    clk_diff = clk_next - clk
    log_derivative_accumulates = (
        (clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative)
        * (challenge(Ch.ClockJumpDifferenceLookupIndeterminate) - clk_diff)
        - one)
    log_derivative_remains = clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative

    acc_or_sp_changes_or_padding = (log_derivative_accumulates
                                    * (stack_pointer_next - stack_pointer - one)
                                    * next_row_is_padding_row)
    remains_or_sp_doesnt_change = log_derivative_remains * (stack_pointer_next - stack_pointer)
    remains_or_next_not_padding = log_derivative_remains * next_row_is_not_padding_row
    log_derivative_updates_correctly = acc_or_sp_changes_or_padding + remains_or_sp_doesnt_change + remains_or_next_not_padding

    return [stack_pointer_increases_by_1_or_does_not_change, sp_inc_or_underflow_unchanged_or_next_grows,
            if_current_row_is_padding_row_then_next_row_is_padding_row, rppa_updates_correctly,
            log_derivative_updates_correctly]


def terminal_constraints(b):
    return []
