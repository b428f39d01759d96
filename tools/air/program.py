"""Program Table AIR -- restated from /root/reference/triton-air/src/table/program.rs:30-278
(prose: specification/src/program-table.md).  Expression order mirrors the reference because node
creation order determines circuit ids."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain
from .defs import AUX, EVAL_ARG_INITIAL, LOOKUP_ARG_INITIAL, MAIN, TIP5_RATE, Ch

M, A = MAIN["Program"], AUX["Program"]


def initial_constraints(b):
    challenge, x_constant = b.challenge, b.x_constant
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))

    address = main_row(M.Address)
    instruction = main_row(M.Instruction)
    index_in_chunk = main_row(M.IndexInChunk)
    is_hash_input_padding = main_row(M.IsHashInputPadding)
    instruction_lookup_log_derivative = aux_row(A.InstructionLookupServerLogDerivative)
    prepare_chunk_running_evaluation = aux_row(A.PrepareChunkRunningEvaluation)
    send_chunk_running_evaluation = aux_row(A.SendChunkRunningEvaluation)

    lookup_arg_initial = x_constant(LOOKUP_ARG_INITIAL)
    eval_arg_initial = x_constant(EVAL_ARG_INITIAL)
    prepare_chunk_indeterminate = challenge(Ch.ProgramAttestationPrepareChunkIndeterminate)

    log_derivative_initialized = instruction_lookup_log_derivative - lookup_arg_initial
    prepare_chunk_absorbed_first = (prepare_chunk_running_evaluation
                                    - eval_arg_initial * prepare_chunk_indeterminate
                                    - instruction)
    send_chunk_is_default_initial = send_chunk_running_evaluation - eval_arg_initial
    return [address, index_in_chunk, is_hash_input_padding, log_derivative_initialized,
            prepare_chunk_absorbed_first, send_chunk_is_default_initial]


def consistency_constraints(b):
    constant = b.b_constant
    main_row = lambda col: b.input(Main(col))
    one = constant(1)
    max_index_in_chunk = constant(TIP5_RATE - 1)

    index_in_chunk = main_row(M.IndexInChunk)
    max_minus_index_in_chunk_inv = main_row(M.MaxMinusIndexInChunkInv)
    is_hash_input_padding = main_row(M.IsHashInputPadding)
    is_table_padding = main_row(M.IsTablePadding)

    max_minus_index_in_chunk = max_index_in_chunk - index_in_chunk
    inv_is_zero_or_inverse = (one - max_minus_index_in_chunk * max_minus_index_in_chunk_inv) * max_minus_index_in_chunk_inv
    val_is_zero_or_inverse = (one - max_minus_index_in_chunk * max_minus_index_in_chunk_inv) * max_minus_index_in_chunk
    is_hash_input_padding_is_bit = is_hash_input_padding * (is_hash_input_padding - one)
    is_table_padding_is_bit = is_table_padding * (is_table_padding - one)
    table_padding_implies_hash_input_padding = is_table_padding * (one - is_hash_input_padding)
    return [inv_is_zero_or_inverse, val_is_zero_or_inverse, is_hash_input_padding_is_bit, is_table_padding_is_bit,
            table_padding_implies_hash_input_padding]


def transition_constraints(b):
    challenge, constant = b.challenge, b.b_constant
    current_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))
    current_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    one = constant(1)
    rate_minus_one = constant(TIP5_RATE - 1)
    prepare_chunk_indeterminate = challenge(Ch.ProgramAttestationPrepareChunkIndeterminate)
    send_chunk_indeterminate = challenge(Ch.ProgramAttestationSendChunkIndeterminate)

    address = current_main_row(M.Address)
    instruction = current_main_row(M.Instruction)
    lookup_multiplicity = current_main_row(M.LookupMultiplicity)
    index_in_chunk = current_main_row(M.IndexInChunk)
    max_minus_index_in_chunk_inv = current_main_row(M.MaxMinusIndexInChunkInv)
    is_hash_input_padding = current_main_row(M.IsHashInputPadding)
    is_table_padding = current_main_row(M.IsTablePadding)
    log_derivative = current_aux_row(A.InstructionLookupServerLogDerivative)
    prepare_chunk_running_evaluation = current_aux_row(A.PrepareChunkRunningEvaluation)
    send_chunk_running_evaluation = current_aux_row(A.SendChunkRunningEvaluation)

    address_next = next_main_row(M.Address)
    instruction_next = next_main_row(M.Instruction)
    index_in_chunk_next = next_main_row(M.IndexInChunk)
    max_minus_index_in_chunk_inv_next = next_main_row(M.MaxMinusIndexInChunkInv)
    is_hash_input_padding_next = next_main_row(M.IsHashInputPadding)
    is_table_padding_next = next_main_row(M.IsTablePadding)
    log_derivative_next = next_aux_row(A.InstructionLookupServerLogDerivative)
    prepare_chunk_running_evaluation_next = next_aux_row(A.PrepareChunkRunningEvaluation)
    send_chunk_running_evaluation_next = next_aux_row(A.SendChunkRunningEvaluation)

    address_increases_by_one = address_next - (address + one)
    is_table_padding_is_0_or_remains_unchanged = is_table_padding * (is_table_padding_next - is_table_padding)

    index_in_chunk_cycles_correctly = (
        (one - max_minus_index_in_chunk_inv * (rate_minus_one - index_in_chunk)) * index_in_chunk_next
        + max_minus_index_in_chunk_inv * (index_in_chunk_next - index_in_chunk - one))

    hash_input_indicator_is_0_or_remains_unchanged = is_hash_input_padding * (is_hash_input_padding_next - one)
    first_hash_input_padding_is_1 = (is_hash_input_padding - one) * is_hash_input_padding_next * (instruction_next - one)
    hash_input_padding_is_0_after_the_first_1 = is_hash_input_padding * instruction_next

    next_row_is_table_padding_row = is_table_padding_next - one
    table_padding_starts = (is_hash_input_padding
                            * (one - max_minus_index_in_chunk_inv * (rate_minus_one - index_in_chunk))
                            * next_row_is_table_padding_row)

    log_derivative_remains = log_derivative_next - log_derivative
    compressed_row = (challenge(Ch.ProgramAddressWeight) * address
                      + challenge(Ch.ProgramInstructionWeight) * instruction
                      + challenge(Ch.ProgramNextInstructionWeight) * instruction_next)
    indeterminate = challenge(Ch.InstructionLookupIndeterminate)
    log_derivative_updates = (log_derivative_next - log_derivative) * (indeterminate - compressed_row) - lookup_multiplicity
    log_derivative_updates_iff_not_padding = ((one - is_hash_input_padding) * log_derivative_updates
                                              + is_hash_input_padding * log_derivative_remains)

    prepare_absorbs = (prepare_chunk_running_evaluation_next
                       - prepare_chunk_indeterminate * prepare_chunk_running_evaluation
                       - instruction_next)
    prepare_resets_and_absorbs = prepare_chunk_running_evaluation_next - prepare_chunk_indeterminate - instruction_next
    index_in_chunk_is_max = rate_minus_one - index_in_chunk
    index_in_chunk_is_not_max = one - max_minus_index_in_chunk_inv * (rate_minus_one - index_in_chunk)
    prepare_resets_every_rate_rows = (index_in_chunk_is_max * prepare_absorbs
                                      + index_in_chunk_is_not_max * prepare_resets_and_absorbs)

    send_absorbs_next_chunk = (send_chunk_running_evaluation_next
                               - send_chunk_indeterminate * send_chunk_running_evaluation
                               - prepare_chunk_running_evaluation_next)
    send_does_not_change = send_chunk_running_evaluation_next - send_chunk_running_evaluation
    index_in_chunk_next_is_max = rate_minus_one - index_in_chunk_next
    index_in_chunk_next_is_not_max = one - max_minus_index_in_chunk_inv_next * index_in_chunk_next_is_max

    send_absorbs_iff = (send_absorbs_next_chunk * next_row_is_table_padding_row * index_in_chunk_next_is_not_max
                        + send_does_not_change * is_table_padding_next
                        + send_does_not_change * index_in_chunk_next_is_max)
    return [address_increases_by_one, is_table_padding_is_0_or_remains_unchanged, index_in_chunk_cycles_correctly,
            hash_input_indicator_is_0_or_remains_unchanged, first_hash_input_padding_is_1,
            hash_input_padding_is_0_after_the_first_1, table_padding_starts, log_derivative_updates_iff_not_padding,
            prepare_resets_every_rate_rows, send_absorbs_iff]


def terminal_constraints(b):
    constant = b.b_constant
    main_row = lambda col: b.input(Main(col))
    index_in_chunk = main_row(M.IndexInChunk)
    is_hash_input_padding = main_row(M.IsHashInputPadding)
    is_table_padding = main_row(M.IsTablePadding)
    hash_input_padding_is_one = is_hash_input_padding - constant(1)
    index_in_chunk_is_max_or_row_is_padding_row = (index_in_chunk - constant(TIP5_RATE - 1)) * (is_table_padding - constant(1))
    return [hash_input_padding_is_one, index_in_chunk_is_max_or_row_is_padding_row]
