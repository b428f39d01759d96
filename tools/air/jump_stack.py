"""Jump Stack Table AIR -- restated from /root/reference/triton-air/src/table/jump_stack.rs:38-165
(prose: specification/src/jump-stack-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, LOOKUP_ARG_INITIAL, MAIN, Ch
from .isa import OPCODE

M, A = MAIN["JumpStack"], AUX["JumpStack"]


def initial_constraints(b):
    clk = b.input(Main(M.CLK))
    jsp = b.input(Main(M.JSP))
    jso = b.input(Main(M.JSO))
    jsd = b.input(Main(M.JSD))
    ci = b.input(Main(M.CI))
    rppa = b.input(Aux(A.RunningProductPermArg))
    clock_jump_diff_log_derivative = b.input(Aux(A.ClockJumpDifferenceLookupClientLogDerivative))

    processor_perm_indeterminate = b.challenge(Ch.JumpStackIndeterminate)
    compressed_row = b.challenge(Ch.JumpStackCiWeight) * ci
    rppa_starts_correctly = rppa - (processor_perm_indeterminate - compressed_row)
    clock_jump_diff_log_derivative_starts_correctly = clock_jump_diff_log_derivative - b.x_constant(LOOKUP_ARG_INITIAL)
    return [clk, jsp, jso, jsd, rppa_starts_correctly, clock_jump_diff_log_derivative_starts_correctly]


def consistency_constraints(b):
    return []


def transition_constraints(b):
    one = lambda: b.b_constant(1)
    call_opcode = b.b_constant(OPCODE["Call"])
    return_opcode = b.b_constant(OPCODE["Return"])
    recurse_or_return_opcode = b.b_constant(OPCODE["RecurseOrReturn"])

    clk = b.input(CurrentMain(M.CLK))
    ci = b.input(CurrentMain(M.CI))
    jsp = b.input(CurrentMain(M.JSP))
    jso = b.input(CurrentMain(M.JSO))
    jsd = b.input(CurrentMain(M.JSD))
    rppa = b.input(CurrentAux(A.RunningProductPermArg))
    clock_jump_diff_log_derivative = b.input(CurrentAux(A.ClockJumpDifferenceLookupClientLogDerivative))

    clk_next = b.input(NextMain(M.CLK))
    ci_next = b.input(NextMain(M.CI))
    jsp_next = b.input(NextMain(M.JSP))
    jso_next = b.input(NextMain(M.JSO))
    jsd_next = b.input(NextMain(M.JSD))
    rppa_next = b.input(NextAux(A.RunningProductPermArg))
    clock_jump_diff_log_derivative_next = b.input(NextAux(A.ClockJumpDifferenceLookupClientLogDerivative))

    jsp_inc_or_stays = (jsp_next - jsp - one()) * (jsp_next - jsp)
    jsp_inc_by_one_or_ci_can_return = (jsp_next - jsp - one()) * (ci - return_opcode) * (ci - recurse_or_return_opcode)
    jsp_inc_or_jso_stays_or_ci_can_ret = jsp_inc_by_one_or_ci_can_return * (jso_next - jso)
    jsp_inc_or_jsd_stays_or_ci_can_ret = jsp_inc_by_one_or_ci_can_return * (jsd_next - jsd)
    jsp_inc_or_clk_inc_or_ci_call_or_ci_can_ret = (jsp_inc_by_one_or_ci_can_return
                                                   * (clk_next - clk - one())
                                                   * (ci - call_opcode))

    compressed_row = (b.challenge(Ch.JumpStackClkWeight) * clk_next
                      + b.challenge(Ch.JumpStackCiWeight) * ci_next
                      + b.challenge(Ch.JumpStackJspWeight) * jsp_next
                      + b.challenge(Ch.JumpStackJsoWeight) * jso_next
                      + b.challenge(Ch.JumpStackJsdWeight) * jsd_next)
    rppa_updates_correctly = rppa_next - rppa * (b.challenge(Ch.JumpStackIndeterminate) - compressed_row)

    log_derivative_remains = clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative
    clk_diff = clk_next - clk
    log_derivative_accumulates = (
        (clock_jump_diff_log_derivative_next - clock_jump_diff_log_derivative)
        * (b.challenge(Ch.ClockJumpDifferenceLookupIndeterminate) - clk_diff)
        - one())
    log_derivative_updates_correctly = ((jsp_next - jsp - one()) * log_derivative_accumulates
                                        + (jsp_next - jsp) * log_derivative_remains)
    return [jsp_inc_or_stays, jsp_inc_or_jso_stays_or_ci_can_ret, jsp_inc_or_jsd_stays_or_ci_can_ret,
            jsp_inc_or_clk_inc_or_ci_call_or_ci_can_ret, rppa_updates_correctly, log_derivative_updates_correctly]


def terminal_constraints(b):
    return []
