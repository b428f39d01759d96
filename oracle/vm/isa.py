"""Triton ISA: instruction table, a parser for the assembly subset the reference's test programs use, and the program
encodings.  Follows /root/reference/triton-isa/src/instruction.rs:30-76 (order), :315-363 (opcodes), :417-429 (sizes),
/root/reference/triton-isa/src/program.rs:367-382 (`to_bwords`), :399-402 (`hash`)."""
P = 2**64 - 2**32 + 1

# name -> (opcode, has_argument)
INSTRUCTIONS = {
    "pop": (3, True), "push": (1, True), "divine": (9, True), "pick": (17, True), "place": (25, True),
    "dup": (33, True), "swap": (41, True), "halt": (0, False), "nop": (8, False), "skiz": (2, False),
    "call": (49, True), "return": (16, False), "recurse": (24, False), "recurse_or_return": (32, False),
    "assert": (10, False), "read_mem": (57, True), "write_mem": (11, True), "hash": (18, False),
    "assert_vector": (26, False), "sponge_init": (40, False), "sponge_absorb": (34, False),
    "sponge_absorb_mem": (48, False), "sponge_squeeze": (56, False), "add": (42, False), "addi": (65, True),
    "mul": (50, False), "invert": (64, False), "eq": (58, False), "split": (4, False), "lt": (6, False),
    "and": (14, False), "xor": (22, False), "log_2_floor": (12, False), "pow": (30, False), "div_mod": (20, False),
    "pop_count": (28, False), "xx_add": (66, False), "xx_mul": (74, False), "x_invert": (72, False),
    "xb_mul": (82, False), "read_io": (73, True), "write_io": (19, True), "merkle_step": (36, False),
    "merkle_step_mem": (44, False), "b_horner_step": (80, False), "x_horner_step": (88, False),
}
OPCODE_TO_NAME = {op: name for name, (op, _) in INSTRUCTIONS.items()}


def size(name):
    return 2 if INSTRUCTIONS[name][1] else 1


class Program:
    """instructions: list of (name, arg or None), `call` arguments resolved to addresses.
    `words`: the instruction list as the VM indexes it (instruction.rs / program.rs: a double-word instruction occupies
    two consecutive addresses)."""

    def __init__(self, instructions, labels=None):
        self.instructions = instructions
        self.labels = labels or {}

    def to_bwords(self):
        out = []
        for name, arg in self.instructions:
            out.append(INSTRUCTIONS[name][0])
            if arg is not None:
                out.append(arg % P)
        return out

    def __len__(self):
        return sum(size(n) for n, _ in self.instructions)


def parse(source):
    """Assembly text -> Program.  Supports `//` comments, `label:` definitions, `call label`, negative immediates."""
    tokens = []
    for line in source.splitlines():
        line = line.split("//")[0]
        tokens += line.split()
    raw, labels, address, i = [], {}, 0, 0
    while i < len(tokens):
        t = tokens[i]
        if t.endswith(":"):
            assert t[:-1] not in labels, f"duplicate label {t}"
            labels[t[:-1]] = address
            i += 1
            continue
        assert t in INSTRUCTIONS, f"unknown instruction {t!r}"
        if INSTRUCTIONS[t][1]:
            raw.append((t, tokens[i + 1]))
            i += 2
        else:
            raw.append((t, None))
            i += 1
        address += size(t)
    instructions = []
    for name, arg in raw:
        if arg is None:
            instructions.append((name, None))
        elif name == "call":
            instructions.append((name, labels[arg]))
        else:
            instructions.append((name, int(arg) % P))
    return Program(instructions, labels)
