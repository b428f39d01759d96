"""Grand Cross-Table Argument -- restated from /root/reference/triton-air/src/cross_table_argument.rs:92-214
(prose: specification/src/table-linking.md)."""
from .circuit import Aux
from .defs import AUX, Ch

LIMBS = ("Highest", "MidHigh", "MidLow", "Lowest")


def initial_constraints(b):
    return []


def consistency_constraints(b):
    return []


def transition_constraints(b):
    return []


def terminal_constraints(b):
    challenge = b.challenge
    aux = lambda table, name: b.input(Aux(getattr(AUX[table], name)))

    program_attestation = aux("Program", "SendChunkRunningEvaluation") - aux("Hash", "ReceiveChunkRunningEvaluation")
    input_to_processor = challenge(Ch.StandardInputTerminal) - aux("Processor", "InputTableEvalArg")
    processor_to_output = aux("Processor", "OutputTableEvalArg") - challenge(Ch.StandardOutputTerminal)
    instruction_lookup = (aux("Processor", "InstructionLookupClientLogDerivative")
                          - aux("Program", "InstructionLookupServerLogDerivative"))
    processor_to_op_stack = aux("Processor", "OpStackTablePermArg") - aux("OpStack", "RunningProductPermArg")
    processor_to_ram = aux("Processor", "RamTablePermArg") - aux("Ram", "RunningProductPermArg")
    processor_to_jump_stack = aux("Processor", "JumpStackTablePermArg") - aux("JumpStack", "RunningProductPermArg")
    hash_input = aux("Processor", "HashInputEvalArg") - aux("Hash", "HashInputRunningEvaluation")
    hash_digest = aux("Hash", "HashDigestRunningEvaluation") - aux("Processor", "HashDigestEvalArg")
    sponge = aux("Processor", "SpongeEvalArg") - aux("Hash", "SpongeRunningEvaluation")
    hash_to_cascade = aux("Cascade", "HashTableServerLogDerivative")
    for i in range(4):
        for limb in LIMBS:
            hash_to_cascade = hash_to_cascade - aux("Hash", f"CascadeState{i}{limb}ClientLogDerivative")
    cascade_to_lookup = aux("Cascade", "LookupTableClientLogDerivative") - aux("Lookup", "CascadeTableServerLogDerivative")
    processor_to_u32 = aux("Processor", "U32LookupClientLogDerivative") - aux("U32", "LookupServerLogDerivative")
    clock_jump_difference_lookup = (aux("Processor", "ClockJumpDifferenceLookupServerLogDerivative")
                                    - aux("OpStack", "ClockJumpDifferenceLookupClientLogDerivative")
                                    - aux("Ram", "ClockJumpDifferenceLookupClientLogDerivative")
                                    - aux("JumpStack", "ClockJumpDifferenceLookupClientLogDerivative"))
    return [program_attestation, input_to_processor, processor_to_output, instruction_lookup, processor_to_op_stack,
            processor_to_ram, processor_to_jump_stack, hash_input, hash_digest, sponge, hash_to_cascade,
            cascade_to_lookup, processor_to_u32, clock_jump_difference_lookup]
