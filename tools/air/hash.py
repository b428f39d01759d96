"""Hash Table AIR -- restated from /root/reference/triton-air/src/table/hash.rs:29-1350
(prose: specification/src/hash-table.md, tips/tip-0005/tip-0005.md).  Statement order mirrors the
reference (including where it builds constraints lazily) because node creation order fixes ids."""
from .circuit import Aux, CurrentAux, CurrentMain, Main, NextAux, NextMain, P, circuit_sum
from .defs import (AUX, DIGEST_LEN, EVAL_ARG_INITIAL, LOOKUP_ARG_INITIAL, MAIN, TIP5_NUM_ROUNDS, TIP5_RATE,
                   TIP5_STATE_SIZE, Ch)
from .isa import OPCODE

M, A = MAIN["Hash"], AUX["Hash"]
NUM_ROUNDS = TIP5_NUM_ROUNDS
MONTGOMERY_MODULUS = (1 << 64) % P          # hash.rs:29-30
POWER_MAP_EXPONENT = 7
MODE = {"ProgramHashing": 1, "Sponge": 2, "Hash": 3, "Pad": 0}   # hash.rs:1373-1382
MODE_ITER = ["ProgramHashing", "Sponge", "Hash", "Pad"]           # enum order, hash.rs:1354-1371

LIMBS = ("Highest", "MidHigh", "MidLow", "Lowest")


def _tip5_tables():
    from tools.gen_tip5_constants import MDS_FIRST_COLUMN, round_constants_montgomery_raw

    rinv = pow(1 << 64, -1, P)
    return MDS_FIRST_COLUMN, [c * rinv % P for c in round_constants_montgomery_raw()]


MDS_FIRST_COLUMN, ROUND_CONSTANTS = _tip5_tables()


def mds_matrix_entry(row_idx, col_idx):  # hash.rs:48-55
    return MDS_FIRST_COLUMN[(TIP5_STATE_SIZE + row_idx - col_idx) % TIP5_STATE_SIZE]


def re_compose_16_bit_limbs(b, highest, mid_high, mid_low, lowest):  # hash.rs:69-84
    constant = b.b_constant
    montgomery_modulus_inv = b.b_constant(pow(MONTGOMERY_MODULUS, -1, P))
    sum_of_shifted_limbs = highest * constant(1 << 48) + mid_high * constant(1 << 32) + mid_low * constant(1 << 16) + lowest
    return sum_of_shifted_limbs * montgomery_modulus_inv


def round_number_deselector(b, round_number_node, round_number_to_deselect):  # hash.rs:86-106
    assert round_number_to_deselect <= NUM_ROUNDS
    constant = b.b_constant
    acc = constant(1) if round_number_to_deselect == 0 else round_number_node
    for r in range(1, NUM_ROUNDS + 1):
        if r == round_number_to_deselect:
            continue
        acc = acc * (round_number_node - constant(r))
    return acc


def select_mode(b, mode_node, mode_to_select):  # hash.rs:108-114
    return mode_node - b.b_constant(MODE[mode_to_select])


def mode_deselector(b, mode_node, mode_to_deselect):  # hash.rs:116-126
    constant = b.b_constant
    acc = constant(1)
    for mode in MODE_ITER:
        if mode == mode_to_deselect:
            continue
        acc = acc * (mode_node - constant(MODE[mode]))
    return acc


def instruction_deselector(b, current_instruction_node, instruction_to_deselect):  # hash.rs:128-148
    relevant = ["Hash", "SpongeInit", "SpongeAbsorb", "SpongeSqueeze"]
    assert instruction_to_deselect in relevant
    acc = b.b_constant(1)
    for instr in relevant:
        if instr == instruction_to_deselect:
            continue
        acc = acc * (current_instruction_node - b.b_constant(OPCODE[instr]))
    return acc


def re_compose_states_0_through_3_before_lookup(b, indicator):  # hash.rs:204-240
    states = []
    for i in range(4):
        limbs = [b.input(indicator(getattr(M, f"State{i}{limb}LkIn"))) for limb in LIMBS]
        states.append(re_compose_16_bit_limbs(b, *limbs))
    return states


def tip5_constraints_as_circuits(b):  # hash.rs:242-388
    constant = b.b_constant
    current_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))

    after_lookup = []
    for i in range(4):
        limbs = [current_main_row(getattr(M, f"State{i}{limb}LkOut")) for limb in LIMBS]
        after_lookup.append(re_compose_16_bit_limbs(b, *limbs))

    before_power_map = [current_main_row(getattr(M, f"State{i}")) for i in range(4, 16)]
    acc = list(before_power_map)
    for _ in range(1, POWER_MAP_EXPONENT):
        for i in range(len(acc)):
            acc[i] = acc[i] * before_power_map[i]
    state_after_s_box_application = after_lookup + acc

    zero = constant(0)
    state_after_matrix_multiplication = [zero] * TIP5_STATE_SIZE
    for row_idx in range(TIP5_STATE_SIZE):
        for col_idx, state in enumerate(state_after_s_box_application):
            matrix_entry = constant(mds_matrix_entry(row_idx, col_idx))
            state_after_matrix_multiplication[row_idx] = state_after_matrix_multiplication[row_idx] + matrix_entry * state

    round_constants = [current_main_row(getattr(M, f"Constant{i}")) for i in range(TIP5_STATE_SIZE)]
    state_after_round_constant_addition = [st + rndc for st, rndc in zip(state_after_matrix_multiplication, round_constants)]

    state_next = re_compose_states_0_through_3_before_lookup(b, NextMain)
    state_next = state_next + [next_main_row(getattr(M, f"State{i}")) for i in range(4, 16)]

    round_number_next = next_main_row(M.RoundNumber)
    updates = [round_number_next * (se - sen) for se, sen in zip(state_after_round_constant_addition, state_next)]
    return state_next, updates


def cascade_log_derivative_update_circuit(b, look_in_column, look_out_column, cascade_log_derivative_column):
    """hash.rs:390-441"""
    challenge, constant = b.challenge, b.b_constant
    opcode = lambda name: b.b_constant(OPCODE[name])
    next_main_row = lambda col: b.input(NextMain(col))
    current_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    cascade_indeterminate = challenge(Ch.HashCascadeLookupIndeterminate)
    look_in_weight = challenge(Ch.HashCascadeLookInWeight)
    look_out_weight = challenge(Ch.HashCascadeLookOutWeight)

    ci_next = next_main_row(M.CI)
    mode_next = next_main_row(M.Mode)
    round_number_next = next_main_row(M.RoundNumber)
    cascade_log_derivative = current_aux_row(cascade_log_derivative_column)
    cascade_log_derivative_next = next_aux_row(cascade_log_derivative_column)

    compressed_row = look_in_weight * next_main_row(look_in_column) + look_out_weight * next_main_row(look_out_column)

    remains = cascade_log_derivative_next - cascade_log_derivative
    updates = (cascade_log_derivative_next - cascade_log_derivative) * (cascade_indeterminate - compressed_row) - constant(1)

    pad_or_max_round_or_sponge_init = (select_mode(b, mode_next, "Pad")
                                       * (round_number_next - constant(NUM_ROUNDS))
                                       * (ci_next - opcode("SpongeInit")))
    round_number_next_is_not_num_rounds = round_number_deselector(b, round_number_next, NUM_ROUNDS)
    ci_next_is_not_sponge_init = instruction_deselector(b, ci_next, "SpongeInit")
    next_row_is_padding_row = mode_deselector(b, mode_next, "Pad")

    return (pad_or_max_round_or_sponge_init * updates
            + round_number_next_is_not_num_rounds * remains
            + ci_next_is_not_sponge_init * remains
            + next_row_is_padding_row * remains)


def _cascade_columns():
    for i in range(4):
        for limb in LIMBS:
            yield (getattr(M, f"State{i}{limb}LkIn"), getattr(M, f"State{i}{limb}LkOut"),
                   getattr(A, f"CascadeState{i}{limb}ClientLogDerivative"))


def initial_constraints(b):  # hash.rs:447-598
    challenge, constant = b.challenge, b.b_constant
    main_row = lambda col: b.input(Main(col))
    aux_row = lambda col: b.input(Aux(col))

    running_evaluation_initial = b.x_constant(EVAL_ARG_INITIAL)
    lookup_arg_default_initial = b.x_constant(LOOKUP_ARG_INITIAL)

    mode = main_row(M.Mode)
    running_evaluation_hash_input = aux_row(A.HashInputRunningEvaluation)
    running_evaluation_hash_digest = aux_row(A.HashDigestRunningEvaluation)
    running_evaluation_sponge = aux_row(A.SpongeRunningEvaluation)
    running_evaluation_receive_chunk = aux_row(A.ReceiveChunkRunningEvaluation)

    cascade_indeterminate = challenge(Ch.HashCascadeLookupIndeterminate)
    look_in_weight = challenge(Ch.HashCascadeLookInWeight)
    look_out_weight = challenge(Ch.HashCascadeLookOutWeight)
    prepare_chunk_indeterminate = challenge(Ch.ProgramAttestationPrepareChunkIndeterminate)
    receive_chunk_indeterminate = challenge(Ch.ProgramAttestationSendChunkIndeterminate)

    state_0_3 = re_compose_states_0_through_3_before_lookup(b, Main)
    state_rate_part = state_0_3 + [main_row(getattr(M, f"State{i}")) for i in range(4, 10)]
    compressed_chunk = running_evaluation_initial
    for state_element in state_rate_part:
        compressed_chunk = compressed_chunk * prepare_chunk_indeterminate + state_element
    receive_chunk_initialized = (running_evaluation_receive_chunk
                                 - receive_chunk_indeterminate * running_evaluation_initial
                                 - compressed_chunk)

    def cascade_log_derivative_init_circuit(look_in_column, look_out_column, cascade_log_derivative_column):
        look_in = main_row(look_in_column)
        look_out = main_row(look_out_column)
        compressed_row = look_in_weight * look_in + look_out_weight * look_out
        cascade_log_derivative = aux_row(cascade_log_derivative_column)
        return ((cascade_log_derivative - lookup_arg_default_initial)
                * (cascade_indeterminate - compressed_row)
                - constant(1))

    mode_is_program_hashing = select_mode(b, mode, "ProgramHashing")
    round_number_is_0 = main_row(M.RoundNumber)
    hash_input_is_default_initial = running_evaluation_hash_input - running_evaluation_initial
    hash_digest_is_default_initial = running_evaluation_hash_digest - running_evaluation_initial
    sponge_is_default_initial = running_evaluation_sponge - running_evaluation_initial

    out = [mode_is_program_hashing, round_number_is_0, hash_input_is_default_initial, hash_digest_is_default_initial,
           sponge_is_default_initial, receive_chunk_initialized]
    for cols in _cascade_columns():
        out.append(cascade_log_derivative_init_circuit(*cols))
    return out


def consistency_constraints(b):  # hash.rs:600-803
    opcode = lambda name: b.b_constant(OPCODE[name])
    constant = b.b_constant
    main_row = lambda col: b.input(Main(col))

    mode = main_row(M.Mode)
    ci = main_row(M.CI)
    round_number = main_row(M.RoundNumber)

    ci_is_hash = ci - opcode("Hash")
    ci_is_sponge_init = ci - opcode("SpongeInit")
    ci_is_sponge_absorb = ci - opcode("SpongeAbsorb")
    ci_is_sponge_squeeze = ci - opcode("SpongeSqueeze")

    mode_is_not_hash = mode_deselector(b, mode, "Hash")
    round_number_is_not_0 = round_number_deselector(b, round_number, 0)

    mode_is_a_valid_mode = mode_deselector(b, mode, "Pad") * select_mode(b, mode, "Pad")
    if_mode_is_not_sponge_then_ci_is_hash = select_mode(b, mode, "Sponge") * ci_is_hash
    if_mode_is_sponge_then_ci_is_a_sponge_instruction = (mode_deselector(b, mode, "Sponge")
                                                         * ci_is_sponge_init * ci_is_sponge_absorb * ci_is_sponge_squeeze)
    if_padding_mode_then_round_number_is_0 = mode_deselector(b, mode, "Pad") * round_number

    if_ci_is_sponge_init_then_ = ci_is_hash * ci_is_sponge_absorb * ci_is_sponge_squeeze
    if_ci_is_sponge_init_then_round_number_is_0 = if_ci_is_sponge_init_then_ * round_number

    # lazy in the reference ((10..=15).map(..), consumed by `extend` below)
    if_ci_is_sponge_init_then_rate_is_0 = (
        if_ci_is_sponge_init_then_ * main_row(getattr(M, f"State{i}")) for i in range(10, 16))

    if_mode_is_hash_and_round_no_is_0_then_ = round_number_is_not_0 * mode_is_not_hash

    def _states_are_1():
        for i in range(10, 16):
            state_element = main_row(getattr(M, f"State{i}"))
            yield if_mode_is_hash_and_round_no_is_0_then_ * (state_element - constant(1))
    if_mode_is_hash_and_round_no_is_0_then_states_10_through_15_are_1 = _states_are_1()

    one = constant(1)
    two_pow_16 = constant(1 << 16)
    two_pow_32 = constant(1 << 32)
    hi_minus = [two_pow_32 - one
                - main_row(getattr(M, f"State{i}HighestLkIn")) * two_pow_16
                - main_row(getattr(M, f"State{i}MidHighLkIn")) for i in range(4)]
    hi_inv = [main_row(getattr(M, f"State{i}Inv")) for i in range(4)]
    not_all_1s = [hi_minus[i] * hi_inv[i] - one for i in range(4)]
    inv_is_inv_or_is_zero = [not_all_1s[i] * hi_inv[i] for i in range(4)]
    inv_is_inv_or_hi_is_zero = [not_all_1s[i] * hi_minus[i] for i in range(4)]
    lo_limbs = [main_row(getattr(M, f"State{i}MidLowLkIn")) * two_pow_16
                + main_row(getattr(M, f"State{i}LowestLkIn")) for i in range(4)]
    if_hi_all_1_then_lo_all_0 = [not_all_1s[i] * lo_limbs[i] for i in range(4)]

    constraints = [mode_is_a_valid_mode, if_mode_is_not_sponge_then_ci_is_hash,
                   if_mode_is_sponge_then_ci_is_a_sponge_instruction, if_padding_mode_then_round_number_is_0,
                   if_ci_is_sponge_init_then_round_number_is_0]
    constraints += inv_is_inv_or_is_zero + inv_is_inv_or_hi_is_zero + if_hi_all_1_then_lo_all_0
    constraints.extend(if_ci_is_sponge_init_then_rate_is_0)
    constraints.extend(if_mode_is_hash_and_round_no_is_0_then_states_10_through_15_are_1)

    for idx in range(TIP5_STATE_SIZE):
        round_constant_column_circuit = main_row(getattr(M, f"Constant{idx}"))
        circuit = constant(0)
        for round_idx in range(NUM_ROUNDS):
            round_constant = b.b_constant(ROUND_CONSTANTS[TIP5_STATE_SIZE * round_idx + idx])
            round_deselector_circuit = round_number_deselector(b, round_number, round_idx)
            circuit = circuit + round_deselector_circuit * (round_constant_column_circuit - round_constant)
        constraints.append(circuit)
    return constraints


def transition_constraints(b):  # hash.rs:805-1250
    challenge, constant = b.challenge, b.b_constant
    opcode = lambda name: b.b_constant(OPCODE[name])

    opcode_hash = opcode("Hash")
    opcode_sponge_init = opcode("SpongeInit")
    opcode_sponge_absorb = opcode("SpongeAbsorb")
    opcode_sponge_squeeze = opcode("SpongeSqueeze")

    current_main_row = lambda col: b.input(CurrentMain(col))
    next_main_row = lambda col: b.input(NextMain(col))
    current_aux_row = lambda col: b.input(CurrentAux(col))
    next_aux_row = lambda col: b.input(NextAux(col))

    running_evaluation_initial = b.x_constant(EVAL_ARG_INITIAL)

    prepare_chunk_indeterminate = challenge(Ch.ProgramAttestationPrepareChunkIndeterminate)
    receive_chunk_indeterminate = challenge(Ch.ProgramAttestationSendChunkIndeterminate)
    compress_program_digest_indeterminate = challenge(Ch.CompressProgramDigestIndeterminate)
    expected_program_digest = challenge(Ch.CompressedProgramDigest)
    hash_input_eval_indeterminate = challenge(Ch.HashInputIndeterminate)
    hash_digest_eval_indeterminate = challenge(Ch.HashDigestIndeterminate)
    sponge_indeterminate = challenge(Ch.SpongeIndeterminate)

    mode = current_main_row(M.Mode)
    ci = current_main_row(M.CI)
    round_number = current_main_row(M.RoundNumber)
    re_receive_chunk = current_aux_row(A.ReceiveChunkRunningEvaluation)
    re_hash_input = current_aux_row(A.HashInputRunningEvaluation)
    re_hash_digest = current_aux_row(A.HashDigestRunningEvaluation)
    re_sponge = current_aux_row(A.SpongeRunningEvaluation)

    mode_next = next_main_row(M.Mode)
    ci_next = next_main_row(M.CI)
    round_number_next = next_main_row(M.RoundNumber)
    re_receive_chunk_next = next_aux_row(A.ReceiveChunkRunningEvaluation)
    re_hash_input_next = next_aux_row(A.HashInputRunningEvaluation)
    re_hash_digest_next = next_aux_row(A.HashDigestRunningEvaluation)
    re_sponge_next = next_aux_row(A.SpongeRunningEvaluation)

    state_current = re_compose_states_0_through_3_before_lookup(b, CurrentMain)
    state_current = state_current + [current_main_row(getattr(M, f"State{i}")) for i in range(4, 16)]

    state_next, hash_function_round_correctly_performs_update = tip5_constraints_as_circuits(b)

    state_weights = [challenge(getattr(Ch, f"StackWeight{i}")) for i in range(16)]

    round_number_is_not_num_rounds = round_number_deselector(b, round_number, NUM_ROUNDS)
    round_number_is_0_through_4_or_round_number_next_is_0 = round_number_is_not_num_rounds * round_number_next

    next_mode_is_padding_mode_or_round_number_is_num_rounds_or_increments_by_one = (
        select_mode(b, mode_next, "Pad")
        * (ci - opcode_sponge_init)
        * (round_number - constant(NUM_ROUNDS))
        * (round_number_next - round_number - constant(1)))

    if_ci_is_sponge_init_then_round_number_next_is_0 = instruction_deselector(b, ci, "SpongeInit") * round_number_next

    compressed_digest = running_evaluation_initial
    for digest_element in state_current[:DIGEST_LEN]:
        compressed_digest = compressed_digest * compress_program_digest_indeterminate + digest_element
    if_mode_changes_from_program_hashing_then_current_digest_is_expected_program_digest = (
        mode_deselector(b, mode, "ProgramHashing")
        * select_mode(b, mode_next, "ProgramHashing")
        * (compressed_digest - expected_program_digest))

    if_mode_is_program_hashing_and_next_mode_is_sponge_then_ci_next_is_sponge_init = (
        mode_deselector(b, mode, "ProgramHashing")
        * mode_deselector(b, mode_next, "Sponge")
        * (ci_next - opcode_sponge_init))

    if_round_number_is_not_max_and_ci_is_not_sponge_init_then_ci_doesnt_change = (
        (round_number - constant(NUM_ROUNDS)) * (ci - opcode_sponge_init) * (ci_next - ci))
    if_round_number_is_not_max_and_ci_is_not_sponge_init_then_mode_doesnt_change = (
        (round_number - constant(NUM_ROUNDS)) * (ci - opcode_sponge_init) * (mode_next - mode))

    if_mode_is_sponge_then_mode_next_is_sponge_or_hash_or_pad = (
        mode_deselector(b, mode, "Sponge")
        * select_mode(b, mode_next, "Sponge")
        * select_mode(b, mode_next, "Hash")
        * select_mode(b, mode_next, "Pad"))
    if_mode_is_hash_then_mode_next_is_hash_or_pad = (
        mode_deselector(b, mode, "Hash") * select_mode(b, mode_next, "Hash") * select_mode(b, mode_next, "Pad"))
    if_mode_is_pad_then_mode_next_is_pad = mode_deselector(b, mode, "Pad") * select_mode(b, mode_next, "Pad")

    difference_of_capacity_registers = [nxt - cur for cur, nxt in zip(state_current[TIP5_RATE:], state_next[TIP5_RATE:])]
    randomized_sum_of_capacity_differences = circuit_sum(
        w * d for w, d in zip(state_weights[TIP5_RATE:], difference_of_capacity_registers))

    capacity_doesnt_change_at_section_start_when_program_hashing_or_absorbing = (
        round_number_deselector(b, round_number_next, 0)
        * select_mode(b, mode_next, "Hash")
        * select_mode(b, mode_next, "Pad")
        * (ci_next - opcode_sponge_init)
        * randomized_sum_of_capacity_differences)

    difference_of_state_registers = [nxt - cur for cur, nxt in zip(state_current, state_next)]
    randomized_sum_of_state_differences = circuit_sum(w * d for w, d in zip(state_weights, difference_of_state_registers))
    if_round_number_next_is_0_and_ci_next_is_squeeze_then_state_doesnt_change = (
        round_number_deselector(b, round_number_next, 0)
        * instruction_deselector(b, ci_next, "SpongeSqueeze")
        * randomized_sum_of_state_differences)

    re_hash_input_remains = re_hash_input_next - re_hash_input
    tip5_input = state_next[:TIP5_RATE]
    compressed_row_from_processor = circuit_sum(w * s for s, w in zip(tip5_input, state_weights[:TIP5_RATE]))
    re_hash_input_updates = re_hash_input_next - hash_input_eval_indeterminate * re_hash_input - compressed_row_from_processor
    running_evaluation_hash_input_is_updated_correctly = (
        round_number_deselector(b, round_number_next, 0) * mode_deselector(b, mode_next, "Hash") * re_hash_input_updates
        + round_number_next * re_hash_input_remains
        + (b.b_constant(MODE["Hash"]) - mode_next) * re_hash_input_remains)

    round_number_next_is_num_rounds = round_number_next - constant(NUM_ROUNDS)
    re_hash_digest_remains = re_hash_digest_next - re_hash_digest
    hash_digest = state_next[:DIGEST_LEN]
    compressed_row_hash_digest = circuit_sum(w * s for s, w in zip(hash_digest, state_weights[:DIGEST_LEN]))
    re_hash_digest_updates = re_hash_digest_next - hash_digest_eval_indeterminate * re_hash_digest - compressed_row_hash_digest
    running_evaluation_hash_digest_is_updated_correctly = (
        round_number_deselector(b, round_number_next, NUM_ROUNDS) * mode_deselector(b, mode_next, "Hash") * re_hash_digest_updates
        + round_number_next_is_num_rounds * re_hash_digest_remains
        + select_mode(b, mode_next, "Hash") * re_hash_digest_remains)

    compressed_row_next = circuit_sum(w * s for w, s in zip(state_weights[:TIP5_RATE], state_next[:TIP5_RATE]))
    re_sponge_has_accumulated_ci = re_sponge_next - sponge_indeterminate * re_sponge - challenge(Ch.HashCIWeight) * ci_next
    re_sponge_has_accumulated_next_row = re_sponge_has_accumulated_ci - compressed_row_next
    if_round_no_next_0_and_ci_next_is_spongy_then_running_evaluation_sponge_updates = (
        round_number_deselector(b, round_number_next, 0) * (ci_next - opcode_hash) * re_sponge_has_accumulated_next_row)

    re_sponge_remains = re_sponge_next - re_sponge
    if_round_no_next_is_not_0_then_running_evaluation_sponge_remains = round_number_next * re_sponge_remains
    if_ci_next_is_not_spongy_then_running_evaluation_sponge_remains = (
        (ci_next - opcode_sponge_init) * (ci_next - opcode_sponge_absorb) * (ci_next - opcode_sponge_squeeze) * re_sponge_remains)
    running_evaluation_sponge_is_updated_correctly = (
        if_round_no_next_0_and_ci_next_is_spongy_then_running_evaluation_sponge_updates
        + if_round_no_next_is_not_0_then_running_evaluation_sponge_remains
        + if_ci_next_is_not_spongy_then_running_evaluation_sponge_remains)

    compressed_chunk = running_evaluation_initial
    for rate_element in state_next[:TIP5_RATE]:
        compressed_chunk = compressed_chunk * prepare_chunk_indeterminate + rate_element
    receive_chunk_absorbs = re_receive_chunk_next - receive_chunk_indeterminate * re_receive_chunk - compressed_chunk
    receive_chunk_remains = re_receive_chunk_next - re_receive_chunk
    receive_chunk_of_instructions_iff_next_mode_is_prog_hashing_and_next_round_number_is_0 = (
        round_number_deselector(b, round_number_next, 0) * mode_deselector(b, mode_next, "ProgramHashing") * receive_chunk_absorbs
        + round_number_next * receive_chunk_remains
        + select_mode(b, mode_next, "ProgramHashing") * receive_chunk_remains)

    constraints = [
        round_number_is_0_through_4_or_round_number_next_is_0,
        next_mode_is_padding_mode_or_round_number_is_num_rounds_or_increments_by_one,
        if_ci_is_sponge_init_then_round_number_next_is_0,
        receive_chunk_of_instructions_iff_next_mode_is_prog_hashing_and_next_round_number_is_0,
        if_mode_changes_from_program_hashing_then_current_digest_is_expected_program_digest,
        if_mode_is_program_hashing_and_next_mode_is_sponge_then_ci_next_is_sponge_init,
        if_round_number_is_not_max_and_ci_is_not_sponge_init_then_ci_doesnt_change,
        if_round_number_is_not_max_and_ci_is_not_sponge_init_then_mode_doesnt_change,
        if_mode_is_sponge_then_mode_next_is_sponge_or_hash_or_pad,
        if_mode_is_hash_then_mode_next_is_hash_or_pad,
        if_mode_is_pad_then_mode_next_is_pad,
        capacity_doesnt_change_at_section_start_when_program_hashing_or_absorbing,
        if_round_number_next_is_0_and_ci_next_is_squeeze_then_state_doesnt_change,
        running_evaluation_hash_input_is_updated_correctly,
        running_evaluation_hash_digest_is_updated_correctly,
        running_evaluation_sponge_is_updated_correctly,
    ]
    for cols in _cascade_columns():
        constraints.append(cascade_log_derivative_update_circuit(b, *cols))
    return constraints + list(hash_function_round_correctly_performs_update)


def terminal_constraints(b):  # hash.rs:1252-1300
    challenge, constant = b.challenge, b.b_constant
    opcode = lambda name: b.b_constant(OPCODE[name])
    main_row = lambda col: b.input(Main(col))

    mode = main_row(M.Mode)
    round_number = main_row(M.RoundNumber)
    compress_program_digest_indeterminate = challenge(Ch.CompressProgramDigestIndeterminate)
    expected_program_digest = challenge(Ch.CompressedProgramDigest)
    max_round_number = constant(NUM_ROUNDS)

    state_0_3 = re_compose_states_0_through_3_before_lookup(b, Main)
    state_4 = main_row(M.State4)
    program_digest = state_0_3 + [state_4]
    compressed_digest = b.x_constant(EVAL_ARG_INITIAL)
    for digest_element in program_digest:
        compressed_digest = compressed_digest * compress_program_digest_indeterminate + digest_element
    if_mode_is_program_hashing_then_current_digest_is_expected_program_digest = (
        mode_deselector(b, mode, "ProgramHashing") * (compressed_digest - expected_program_digest))
    if_mode_is_not_pad_and_ci_is_not_sponge_init_then_round_number_is_max_round_number = (
        select_mode(b, mode, "Pad") * (main_row(M.CI) - opcode("SpongeInit")) * (round_number - max_round_number))
    return [if_mode_is_program_hashing_then_current_digest_is_expected_program_digest,
            if_mode_is_not_pad_and_ci_is_not_sponge_init_then_round_number_is_max_round_number]
