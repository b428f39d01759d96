"""Cascade Table AIR -- restated from /root/reference/triton-air/src/table/cascade.rs:30-216
(prose: specification/src/cascade-table.md)."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, LOOKUP_ARG_INITIAL, MAIN, Ch

M, A = MAIN["Cascade"], AUX["Cascade"]


def initial_constraints(b):
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))
    challenge = b.challenge

    one = lambda: b.b_constant(1)
    two = lambda: b.b_constant(2)
    two_pow_8 = b.b_constant(1 << 8)
    lookup_arg_default_initial = b.x_constant(LOOKUP_ARG_INITIAL)

    is_padding = main_row(M.IsPadding)
    look_in_hi = main_row(M.LookInHi)
    look_in_lo = main_row(M.LookInLo)
    look_out_hi = main_row(M.LookOutHi)
    look_out_lo = main_row(M.LookOutLo)
    lookup_multiplicity = main_row(M.LookupMultiplicity)
    hash_log = aux_row(A.HashTableServerLogDerivative)
    lookup_log = aux_row(A.LookupTableClientLogDerivative)

    hash_indeterminate = challenge(Ch.HashCascadeLookupIndeterminate)
    hash_input_weight = challenge(Ch.HashCascadeLookInWeight)
    hash_output_weight = challenge(Ch.HashCascadeLookOutWeight)
    lookup_indeterminate = challenge(Ch.CascadeLookupIndeterminate)
    lookup_input_weight = challenge(Ch.LookupTableInputWeight)
    lookup_output_weight = challenge(Ch.LookupTableOutputWeight)

    compressed_row_hash = (hash_input_weight * (two_pow_8 * look_in_hi + look_in_lo)
                           + hash_output_weight * (two_pow_8 * look_out_hi + look_out_lo))
    hash_log_is_default_initial = hash_log - lookup_arg_default_initial
    hash_log_accumulated_first_row = ((hash_log - lookup_arg_default_initial)
                                      * (hash_indeterminate - compressed_row_hash)
                                      - lookup_multiplicity)
    hash_log_initialized = ((one() - is_padding) * hash_log_accumulated_first_row
                            + is_padding * hash_log_is_default_initial)

    compressed_row_lo = lookup_input_weight * look_in_lo + lookup_output_weight * look_out_lo
    compressed_row_hi = lookup_input_weight * look_in_hi + lookup_output_weight * look_out_hi
    lookup_log_is_default_initial = lookup_log - lookup_arg_default_initial
    lookup_log_accumulated_first_row = ((lookup_log - lookup_arg_default_initial)
                                        * (lookup_indeterminate - compressed_row_lo)
                                        * (lookup_indeterminate - compressed_row_hi)
                                        - two() * lookup_indeterminate
                                        + compressed_row_lo
                                        + compressed_row_hi)
    lookup_log_initialized = ((one() - is_padding) * lookup_log_accumulated_first_row
                              + is_padding * lookup_log_is_default_initial)
    return [hash_log_initialized, lookup_log_initialized]


def consistency_constraints(b):
    one = b.b_constant(1)
    is_padding = b.input(Main(M.IsPadding))
    return [is_padding * (one - is_padding)]


def transition_constraints(b):
    challenge, constant = b.challenge, b.b_constant
    curr_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))
    curr_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    one = constant(1)
    two = constant(2)
    two_pow_8 = constant(1 << 8)

    is_padding = curr_main_row(M.IsPadding)
    hash_log = curr_aux_row(A.HashTableServerLogDerivative)
    lookup_log = curr_aux_row(A.LookupTableClientLogDerivative)

    is_padding_next = next_main_row(M.IsPadding)
    look_in_hi_next = next_main_row(M.LookInHi)
    look_in_lo_next = next_main_row(M.LookInLo)
    look_out_hi_next = next_main_row(M.LookOutHi)
    look_out_lo_next = next_main_row(M.LookOutLo)
    lookup_multiplicity_next = next_main_row(M.LookupMultiplicity)
    hash_log_next = next_aux_row(A.HashTableServerLogDerivative)
    lookup_log_next = next_aux_row(A.LookupTableClientLogDerivative)

    hash_indeterminate = challenge(Ch.HashCascadeLookupIndeterminate)
    hash_input_weight = challenge(Ch.HashCascadeLookInWeight)
    hash_output_weight = challenge(Ch.HashCascadeLookOutWeight)
    lookup_indeterminate = challenge(Ch.CascadeLookupIndeterminate)
    lookup_input_weight = challenge(Ch.LookupTableInputWeight)
    lookup_output_weight = challenge(Ch.LookupTableOutputWeight)

    if_current_padding_then_next_padding = is_padding * (one - is_padding_next)

    compressed_next_row_hash = (hash_input_weight * (two_pow_8 * look_in_hi_next + look_in_lo_next)
                                + hash_output_weight * (two_pow_8 * look_out_hi_next + look_out_lo_next))
    hash_log_remains = hash_log_next - hash_log
    hash_log_accumulates_next_row = ((hash_log_next - hash_log) * (hash_indeterminate - compressed_next_row_hash)
                                     - lookup_multiplicity_next)
    hash_log_updates_correctly = ((one - is_padding_next) * hash_log_accumulates_next_row
                                  + is_padding_next * hash_log_remains)

    compressed_row_lo_next = lookup_input_weight * look_in_lo_next + lookup_output_weight * look_out_lo_next
    compressed_row_hi_next = lookup_input_weight * look_in_hi_next + lookup_output_weight * look_out_hi_next
    lookup_log_remains = lookup_log_next - lookup_log
    lookup_log_accumulates_next_row = ((lookup_log_next - lookup_log)
                                       * (lookup_indeterminate - compressed_row_lo_next)
                                       * (lookup_indeterminate - compressed_row_hi_next)
                                       - two * lookup_indeterminate
                                       + compressed_row_lo_next
                                       + compressed_row_hi_next)
    lookup_log_updates_correctly = ((one - is_padding_next) * lookup_log_accumulates_next_row
                                    + is_padding_next * lookup_log_remains)
    return [if_current_padding_then_next_padding, hash_log_updates_correctly, lookup_log_updates_correctly]


def terminal_constraints(b):
    return []
