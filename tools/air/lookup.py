"""Lookup Table AIR -- restated from /root/reference/triton-air/src/table/lookup.rs:38-186
(prose: specification/src/lookup-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, EVAL_ARG_INITIAL, LOOKUP_ARG_INITIAL, MAIN, Ch

M, A = MAIN["Lookup"], AUX["Lookup"]


def initial_constraints(b):
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))
    challenge = b.challenge

    lookup_input = main_row(M.LookIn)
    lookup_output = main_row(M.LookOut)
    lookup_multiplicity = main_row(M.LookupMultiplicity)
    cascade_table_server_log_derivative = aux_row(A.CascadeTableServerLogDerivative)
    public_evaluation_argument = aux_row(A.PublicEvaluationArgument)

    lookup_argument_default_initial = b.x_constant(LOOKUP_ARG_INITIAL)
    cascade_table_indeterminate = challenge(Ch.CascadeLookupIndeterminate)
    compressed_row = lookup_output * challenge(Ch.LookupTableOutputWeight)
    cascade_log_derivative_initialized = (
        (cascade_table_server_log_derivative - lookup_argument_default_initial)
        * (cascade_table_indeterminate - compressed_row)
        - lookup_multiplicity)

    eval_argument_default_initial = b.x_constant(EVAL_ARG_INITIAL)
    public_indeterminate = challenge(Ch.LookupTablePublicIndeterminate)
    public_eval_initialized = (public_evaluation_argument
                               - eval_argument_default_initial * public_indeterminate
                               - lookup_output)
    return [lookup_input, cascade_log_derivative_initialized, public_eval_initialized]


def consistency_constraints(b):
    main_row = lambda col: b.input(Main(col))
    padding_is_0_or_1 = main_row(M.IsPadding) * (b.b_constant(1) - main_row(M.IsPadding))
    return [padding_is_0_or_1]


def transition_constraints(b):
    one = lambda: b.b_constant(1)
    current_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))
    current_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))
    challenge = b.challenge

    lookup_input = current_main_row(M.LookIn)
    is_padding = current_main_row(M.IsPadding)
    cascade_log = current_aux_row(A.CascadeTableServerLogDerivative)
    public_eval = current_aux_row(A.PublicEvaluationArgument)

    lookup_input_next = next_main_row(M.LookIn)
    lookup_output_next = next_main_row(M.LookOut)
    lookup_multiplicity_next = next_main_row(M.LookupMultiplicity)
    is_padding_next = next_main_row(M.IsPadding)
    cascade_log_next = next_aux_row(A.CascadeTableServerLogDerivative)
    public_eval_next = next_aux_row(A.PublicEvaluationArgument)

    if_current_padding_then_next_padding = is_padding * (one() - is_padding_next)

    if_next_padding_then_input_0 = is_padding_next * lookup_input_next
    if_next_not_padding_then_input_increments = (one() - is_padding_next) * (lookup_input_next - lookup_input - one())
    lookup_input_increments_iff = if_next_padding_then_input_0 + if_next_not_padding_then_input_increments

    cascade_table_indeterminate = challenge(Ch.CascadeLookupIndeterminate)
    compressed_row = (lookup_input_next * challenge(Ch.LookupTableInputWeight)
                      + lookup_output_next * challenge(Ch.LookupTableOutputWeight))
    cascade_log_remains = cascade_log_next - cascade_log
    cascade_log_updates = ((cascade_log_next - cascade_log) * (cascade_table_indeterminate - compressed_row)
                           - lookup_multiplicity_next)
    cascade_log_updates_iff = (one() - is_padding_next) * cascade_log_updates + is_padding_next * cascade_log_remains

    public_indeterminate = challenge(Ch.LookupTablePublicIndeterminate)
    public_eval_remains = public_eval_next - public_eval
    public_eval_updates = public_eval_next - public_eval * public_indeterminate - lookup_output_next
    public_eval_updates_iff = (one() - is_padding_next) * public_eval_updates + is_padding_next * public_eval_remains

    return [if_current_padding_then_next_padding, lookup_input_increments_iff, cascade_log_updates_iff,
            public_eval_updates_iff]


def terminal_constraints(b):
    return [b.input(Aux(A.PublicEvaluationArgument)) - b.challenge(Ch.LookupTablePublicTerminal)]
