"""Opcodes of the Triton ISA (/root/reference/triton-isa/src/instruction.rs:30-76 order, :315-363 opcodes;
prose: specification/src/instructions.md).  Only what the AIR needs."""
# (name, opcode) in ALL_INSTRUCTIONS order
ALL_INSTRUCTIONS = [
    ("Pop", 3), ("Push", 1), ("Divine", 9), ("Pick", 17), ("Place", 25), ("Dup", 33), ("Swap", 41), ("Halt", 0),
    ("Nop", 8), ("Skiz", 2), ("Call", 49), ("Return", 16), ("Recurse", 24), ("RecurseOrReturn", 32), ("Assert", 10),
    ("ReadMem", 57), ("WriteMem", 11), ("Hash", 18), ("AssertVector", 26), ("SpongeInit", 40), ("SpongeAbsorb", 34),
    ("SpongeAbsorbMem", 48), ("SpongeSqueeze", 56), ("Add", 42), ("AddI", 65), ("Mul", 50), ("Invert", 64), ("Eq", 58),
    ("Split", 4), ("Lt", 6), ("And", 14), ("Xor", 22), ("Log2Floor", 12), ("Pow", 30), ("DivMod", 20), ("PopCount", 28),
    ("XxAdd", 66), ("XxMul", 74), ("XInvert", 72), ("XbMul", 82), ("ReadIo", 73), ("WriteIo", 19), ("MerkleStep", 36),
    ("MerkleStepMem", 44), ("BHornerStep", 80), ("XHornerStep", 88),
]
OPCODE = dict(ALL_INSTRUCTIONS)
NUM_INSTRUCTION_BITS = 7           # InstructionBit::COUNT (instruction.rs:678-...)
NUM_OP_STACK_REGISTERS = 16


def ib(name, bit):
    """Instruction::ib (instruction.rs:437-439)"""
    return (OPCODE[name] >> bit) & 1
