"""TEST INFRASTRUCTURE (oracle): Triton VM's trace execution, restated from
/root/reference/triton-vm/src/vm.rs (state machine :244-1260), /root/reference/triton-isa/src/op_stack.rs (op stack with
underflow-IO recording :56-300) and /root/reference/triton-vm/src/aet.rs (what is recorded :186-368).
All field values are canonical python integers."""
from collections import OrderedDict, deque

import numpy as np

from .. import oracle as orc
from . import isa
from .isa import INSTRUCTIONS, P

NUM_OP_STACK_REGISTERS = 16
RATE, DIGEST_LEN, NUM_ROUNDS = 10, 5, 5
NUM_HELPER_VARIABLES = 6

STACK_DELTA = {
    "push": 1, "pick": 0, "place": 0, "dup": 1, "swap": 0, "halt": 0, "nop": 0, "skiz": -1, "call": 0, "return": 0,
    "recurse": 0, "recurse_or_return": 0, "assert": -1, "hash": -5, "assert_vector": -5, "sponge_init": 0,
    "sponge_absorb": -10, "sponge_absorb_mem": 0, "sponge_squeeze": 10, "add": -1, "addi": 0, "mul": -1, "invert": 0,
    "eq": -1, "split": 1, "lt": -1, "and": -1, "xor": -1, "log_2_floor": 0, "pow": -1, "div_mod": 0, "pop_count": 0,
    "xx_add": -3, "xx_mul": -3, "x_invert": 0, "xb_mul": -1, "merkle_step": 0, "merkle_step_mem": 0,
    "b_horner_step": 0, "x_horner_step": 0,
}
U32_INSTRUCTIONS = {"split", "lt", "and", "xor", "log_2_floor", "pow", "div_mod", "pop_count", "merkle_step",
                    "merkle_step_mem"}


def op_stack_size_influence(name, arg):
    """instruction.rs:496-545"""
    if name in ("pop", "write_mem", "write_io"):
        return -arg
    if name in ("divine", "read_mem", "read_io"):
        return arg
    return STACK_DELTA[name]


def inv(x):
    return pow(x % P, -1, P)


def inverse_or_zero(x):
    x %= P
    return pow(x, -1, P) if x else 0


def xfe_mul(a, b):
    d0, d1 = a[0] * b[0], a[0] * b[1] + a[1] * b[0]
    d2, d3, d4 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0], a[1] * b[2] + a[2] * b[1], a[2] * b[2]
    return [(d0 - d3) % P, (d1 + d3 - d4) % P, (d2 + d4) % P]


def xfe_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def xfe_inv(a):
    return [int(v) for v in orc.from_mont(orc.xfe_inv(orc.to_mont(list(a))))]


def tip5_trace(state):
    """Tip5::trace: [6][16] canonical values."""
    t = orc.from_mont(orc.tip5_trace(orc.to_mont([s % P for s in state])))
    return [[int(v) for v in row] for row in t]


def hash_varlen(words):
    return [int(v) for v in orc.from_mont(orc.hash_varlen(orc.to_mont([w % P for w in words])))]


class VMError(Exception):
    pass


class OpStack:
    """op_stack.rs:38-176; `stack[-1]` is the top (ST0)."""

    def __init__(self, program_digest):
        self.stack = [0] * NUM_OP_STACK_REGISTERS
        self.stack[:DIGEST_LEN] = list(reversed(program_digest))
        self.io = []

    def __len__(self):
        return len(self.stack)

    def __getitem__(self, i):
        return self.stack[len(self.stack) - 1 - i]

    def __setitem__(self, i, v):
        self.stack[len(self.stack) - 1 - i] = v % P

    def first_underflow_element(self):
        top = len(self.stack) - 1
        if top - NUM_OP_STACK_REGISTERS < 0:
            return 0
        return self.stack[top - NUM_OP_STACK_REGISTERS]

    def push(self, v):
        self.stack.append(v % P)
        self.io.append(("w", self.first_underflow_element()))

    def pop(self):
        self.io.append(("r", self.first_underflow_element()))
        return self.stack.pop()

    def insert(self, index, v):
        self.stack.insert(len(self.stack) - index, v % P)
        self.io.append(("w", self.first_underflow_element()))

    def remove(self, index):
        self.io.append(("r", self.first_underflow_element()))
        return self.stack.pop(len(self.stack) - 1 - index)

    def pop_n(self, n):
        return [self.pop() for _ in range(n)]

    def get_u32(self, i):
        v = self[i]
        if v >> 32:
            raise VMError(f"failed to convert {v} into u32")
        return v

    def pop_u32(self):
        v = self.pop()
        if v >> 32:
            raise VMError(f"failed to convert {v} into u32")
        return v

    def push_xfe(self, x):
        for c in reversed(x):
            self.push(c)

    def pop_xfe(self):
        return self.pop_n(3)

    def peek_xfe(self, i):
        return [self[i], self[i + 1], self[i + 2]]


def canonicalize_io(seq):
    """UnderflowIO::canonicalize_sequence (op_stack.rs:234-257): drop adjacent read/write pairs with the same payload."""
    seq = list(seq)
    while True:
        for k in range(len(seq) - 1):
            (t0, p0), (t1, p1) = seq[k], seq[k + 1]
            if t0 != t1 and p0 == p1:
                del seq[k:k + 2]
                break
        else:
            return seq


class AET:
    """AlgebraicExecutionTrace (aet.rs:41-96)."""

    def __init__(self, program):
        self.program = program
        self.instruction_multiplicities = [0] * len(program)
        self.processor_trace = []
        self.op_stack_underflow_trace = []     # rows (clk, shrink, pointer, first underflow element)
        self.ram_trace = []                    # rows (clk, instruction type, pointer, value)
        self.program_hash_trace = []           # rows: (ci, round, state[16])
        self.hash_trace = []
        self.sponge_trace = []
        self.u32_entries = OrderedDict()       # (instruction name, lhs, rhs) -> multiplicity
        self.cascade_multiplicities = OrderedDict()
        self.lookup_multiplicities = [0] * 256
        self._fill_program_hash_trace()

    # aet.rs:174-228
    def padded_program_length(self):
        n = len(self.program) + 1
        return (n + RATE - 1) // RATE * RATE

    def _fill_program_hash_trace(self):
        words = self.program.to_bwords()
        padded = (words + [1] + [0] * RATE)[:self.padded_program_length()]
        state = [0] * 16
        for k in range(0, len(padded), RATE):
            state[:RATE] = padded[k:k + RATE]
            trace = tip5_trace(state)
            self._increase_lookup_multiplicities(trace)
            self.program_hash_trace += [(INSTRUCTIONS["hash"][0], r, row) for r, row in enumerate(trace)]
            state = list(trace[-1])
        assert state[:DIGEST_LEN] == hash_varlen(words)

    # aet.rs:288-343
    def _increase_lookup_multiplicities(self, trace):
        for row in trace[:-1]:
            for element in row[:4]:
                for limb in sixteen_bit_limbs(element):
                    if limb in self.cascade_multiplicities:
                        self.cascade_multiplicities[limb] += 1
                    else:
                        self.cascade_multiplicities[limb] = 1
                        self.lookup_multiplicities[limb & 0xFF] += 1
                        self.lookup_multiplicities[limb >> 8] += 1

    def append_hash_trace(self, trace):
        self._increase_lookup_multiplicities(trace)
        self.hash_trace += [(INSTRUCTIONS["hash"][0], r, row) for r, row in enumerate(trace)]

    def append_initial_sponge_state(self):
        self.sponge_trace.append((INSTRUCTIONS["sponge_init"][0], 0, [0] * 16))

    def append_sponge_trace(self, name, trace):
        self._increase_lookup_multiplicities(trace)
        self.sponge_trace += [(INSTRUCTIONS[name][0], r, row) for r, row in enumerate(trace)]

    def record_u32(self, name, lhs, rhs):
        key = (name, lhs % P, rhs % P)
        self.u32_entries[key] = self.u32_entries.get(key, 0) + 1

    # aet.rs:140-172
    def u32_table_height(self):
        return sum(u32_height_contribution(*k) for k in self.u32_entries)

    def height_of_table(self, table):
        return {
            "Program": self.padded_program_length(), "Processor": len(self.processor_trace),
            "OpStack": len(self.op_stack_underflow_trace), "Ram": len(self.ram_trace),
            "JumpStack": len(self.processor_trace),
            "Hash": len(self.sponge_trace) + len(self.hash_trace) + len(self.program_hash_trace),
            "Cascade": len(self.cascade_multiplicities), "Lookup": 256, "U32": self.u32_table_height(),
        }[table]

    def height(self):
        return max(self.height_of_table(t) for t in ("Program", "Processor", "OpStack", "Ram", "JumpStack", "Hash",
                                                     "Cascade", "Lookup", "U32"))

    def padded_height(self):
        return 1 << (self.height() - 1).bit_length()


def sixteen_bit_limbs(x):
    """table/hash.rs:30-33: the limbs of the Montgomery representation R*x."""
    r = (x % P) * (1 << 64) % P
    return [(r >> s) & 0xFFFF for s in (0, 16, 32, 48)]


def u32_height_contribution(name, lhs, rhs):
    """table/u32.rs:53-64"""
    dominant = rhs if name == "pow" else max(lhs, rhs)
    return 1 if dominant == 0 else 2 + (dominant.bit_length() - 1)


class VM:
    """VMState + VM::trace_execution (vm.rs:159-206, 244-1260)."""

    def __init__(self, program, public_input=(), secret_input=(), secret_digests=(), ram=None):
        self.program = program
        self.by_address = []
        for name, arg in program.instructions:
            self.by_address += [(name, arg)] * isa.size(name)
        self.public_input = deque(v % P for v in public_input)
        self.public_output = []
        self.secret = deque(v % P for v in secret_input)
        self.secret_digests = deque([list(d) for d in secret_digests])
        self.ram = {k % P: v % P for k, v in (ram or {}).items()}
        self.ram_calls = []
        self.program_digest = hash_varlen(program.to_bwords())
        self.op_stack = OpStack(self.program_digest)
        self.jump_stack = []
        self.cycle_count = 0
        self.ip = 0
        self.sponge = None
        self.halting = False

    # ---- helpers ------------------------------------------------------------------------------------------------
    def current_instruction(self):
        return self.by_address[self.ip] if self.ip < len(self.by_address) else None

    def next_instruction(self):
        cur = self.current_instruction()
        if cur is None:
            return None
        nip = self.ip + isa.size(cur[0])
        return self.by_address[nip] if nip < len(self.by_address) else None

    def next_instruction_or_argument(self):
        cur = self.current_instruction()
        if cur is None:
            return 0
        if cur[1] is not None:
            return cur[1]
        nxt = self.next_instruction()
        return INSTRUCTIONS[nxt[0]][0] if nxt else 1

    def ram_read(self, pointer):
        pointer %= P
        value = self.ram.get(pointer, 0)
        self.ram_calls.append((self.cycle_count, 1, pointer, value))   # INSTRUCTION_TYPE_READ = 1
        return value

    def ram_write(self, pointer, value):
        pointer %= P
        self.ram_calls.append((self.cycle_count, 0, pointer, value % P))
        self.ram[pointer] = value % P

    # vm.rs:270-349
    def helper_variables(self):
        hv = [0] * NUM_HELPER_VARIABLES
        cur = self.current_instruction()
        if cur is None:
            return hv
        name, arg = cur
        st = self.op_stack
        rd = lambda a: self.ram.get(a % P, 0)
        if name in ("pop", "divine", "pick", "place", "dup", "swap", "read_mem", "write_mem", "read_io", "write_io"):
            hv[:4] = [(arg >> i) & 1 for i in range(4)]
        elif name == "skiz":
            hv[0] = inverse_or_zero(st[0])
            op = self.next_instruction_or_argument()
            hv[1:6] = [op % 2, (op >> 1) % 4, (op >> 3) % 4, (op >> 5) % 4, op >> 7]
        elif name == "recurse_or_return":
            hv[0] = inverse_or_zero(st[6] - st[5])
        elif name == "sponge_absorb_mem":
            hv[:6] = [rd(st[0] + 4 + i) for i in range(6)]
        elif name == "merkle_step":
            hv[:5] = self.secret_digests[0] if self.secret_digests else [0] * 5
            hv[5] = st[5] % 2
        elif name == "merkle_step_mem":
            hv[:5] = [rd(st[7] + i) for i in range(5)]
            hv[5] = st[5] % 2
        elif name == "split":
            lo, hi = st[0] & 0xFFFFFFFF, st[0] >> 32
            if lo:
                hv[0] = inverse_or_zero(hi - 0xFFFFFFFF)
        elif name == "eq":
            hv[0] = inverse_or_zero(st[1] - st[0])
        elif name == "b_horner_step":
            hv[0] = rd(st[5])
        elif name == "x_horner_step":
            hv[2], hv[1], hv[0] = rd(st[5]), rd(st[5] - 1), rd(st[5] - 2)
        return hv

    # vm.rs:1113-1176
    def to_processor_row(self):
        cur = self.current_instruction() or ("nop", None)
        opcode = INSTRUCTIONS[cur[0]][0]
        js = self.jump_stack[-1] if self.jump_stack else (0, 0)
        row = [self.cycle_count, 0, self.ip, opcode, self.next_instruction_or_argument()]
        row += [(opcode >> b) & 1 for b in range(7)]
        row += [len(self.jump_stack), js[0], js[1]]
        row += [self.op_stack[i] for i in range(16)]
        row += [len(self.op_stack)]
        row += self.helper_variables()
        row += [0]                                      # ClockJumpDifferenceLookupMultiplicity (filled later)
        return row

    # ---- one step (vm.rs:362-428) -----------------------------------------------------------------------------------
    def step(self, aet=None):
        if self.halting:
            raise VMError("machine halted")
        cur = self.current_instruction()
        if cur is None:
            raise VMError("instruction pointer overflow")
        name, arg = cur
        if len(self.op_stack) + op_stack_size_influence(name, arg) < NUM_OP_STACK_REGISTERS:
            raise VMError("op stack too shallow")
        self.op_stack.io = []
        self.ram_calls = []
        calls = getattr(self, "i_" + name)(arg) or []
        # op-stack table entries (table/op_stack.rs:61-89)
        seq = canonicalize_io(self.op_stack.io)
        assert len({t for t, _ in seq}) <= 1
        pointer = len(self.op_stack)
        writing = all(t == "w" for t, _ in seq)
        pointer = pointer - len(seq) if writing else pointer + len(seq)
        for t, payload in seq:
            if t == "r":
                pointer -= 1
            calls.append(("op_stack", (self.cycle_count, 1 if t == "r" else 0, pointer, payload)))
            if t == "w":
                pointer += 1
        self.cycle_count += 1
        return calls

    def _ram(self):
        return [("ram", c) for c in self.ram_calls]

    def i_pop(self, n):
        self.op_stack.pop_n(n)
        self.ip += 2

    def i_push(self, v):
        self.op_stack.push(v)
        self.ip += 2

    def i_divine(self, n):
        if len(self.secret) < n:
            raise VMError("secret input empty")
        for _ in range(n):
            self.op_stack.push(self.secret.popleft())
        self.ip += 2

    def i_pick(self, i):
        self.op_stack.push(self.op_stack.remove(i))
        self.ip += 2

    def i_place(self, i):
        self.op_stack.insert(i, self.op_stack.pop())
        self.ip += 2

    def i_dup(self, i):
        self.op_stack.push(self.op_stack[i])
        self.ip += 2

    def i_swap(self, i):
        a, b = self.op_stack[0], self.op_stack[i]
        self.op_stack[0], self.op_stack[i] = b, a
        self.ip += 2

    def i_nop(self, _):
        self.ip += 1

    def i_skiz(self, _):
        top = self.op_stack.pop()
        if top == 0:
            nxt = self.next_instruction()
            if nxt is None:
                raise VMError("instruction pointer overflow")
            self.ip += 1 + isa.size(nxt[0])
        else:
            self.ip += 1

    def i_call(self, dest):
        self.jump_stack.append((self.ip + 2, dest))
        self.ip = dest

    def i_return(self, _):
        if not self.jump_stack:
            raise VMError("jump stack is empty")
        origin, _d = self.jump_stack.pop()
        self.ip = origin

    def i_recurse(self, _):
        if not self.jump_stack:
            raise VMError("jump stack is empty")
        self.ip = self.jump_stack[-1][1]

    def i_recurse_or_return(self, _):
        if not self.jump_stack:
            raise VMError("jump stack is empty")
        if self.op_stack[5] == self.op_stack[6]:
            self.ip = self.jump_stack.pop()[0]
        else:
            self.ip = self.jump_stack[-1][1]

    def i_assert(self, _):
        if self.op_stack[0] != 1:
            raise VMError("assertion failed")
        self.op_stack.pop()
        self.ip += 1

    def i_halt(self, _):
        self.halting = True
        self.ip += 1

    def i_read_mem(self, n):
        pointer = self.op_stack.pop()
        for _ in range(n):
            self.op_stack.push(self.ram_read(pointer))
            pointer = (pointer - 1) % P
        self.op_stack.push(pointer)
        self.ip += 2
        return self._ram()

    def i_write_mem(self, n):
        pointer = self.op_stack.pop()
        for _ in range(n):
            self.ram_write(pointer, self.op_stack.pop())
            pointer = (pointer + 1) % P
        self.op_stack.push(pointer)
        self.ip += 2
        return self._ram()

    def i_hash(self, _):
        to_hash = self.op_stack.pop_n(RATE)
        trace = tip5_trace(to_hash + [1] * 6)            # sponge::Domain::FixedLength: capacity of ones
        for v in reversed(trace[-1][:DIGEST_LEN]):
            self.op_stack.push(v)
        self.ip += 1
        return [("hash", trace)]

    def i_sponge_init(self, _):
        self.sponge = [0] * 16
        self.ip += 1
        return [("sponge_reset", None)]

    def i_sponge_absorb(self, _):
        if self.sponge is None:
            raise VMError("sponge not initialized")
        self.sponge[:RATE] = self.op_stack.pop_n(RATE)
        trace = tip5_trace(self.sponge)
        self.sponge = list(trace[-1])
        self.ip += 1
        return [("sponge", ("sponge_absorb", trace))]

    def i_sponge_absorb_mem(self, _):
        if self.sponge is None:
            raise VMError("sponge not initialized")
        pointer = self.op_stack.pop()
        for i in range(RATE):
            element = self.ram_read(pointer)
            pointer = (pointer + 1) % P
            self.sponge[i] = element
            if i < RATE - NUM_HELPER_VARIABLES:
                self.op_stack[i] = element
        self.op_stack.push(pointer)
        trace = tip5_trace(self.sponge)
        self.sponge = list(trace[-1])
        self.ip += 1
        return self._ram() + [("sponge", ("sponge_absorb", trace))]

    def i_sponge_squeeze(self, _):
        if self.sponge is None:
            raise VMError("sponge not initialized")
        for i in reversed(range(RATE)):
            self.op_stack.push(self.sponge[i])
        trace = tip5_trace(self.sponge)
        self.sponge = list(trace[-1])
        self.ip += 1
        return [("sponge", ("sponge_squeeze", trace))]

    def i_assert_vector(self, _):
        for i in range(DIGEST_LEN):
            if self.op_stack[i] != self.op_stack[i + DIGEST_LEN]:
                raise VMError("vector assertion failed")
        self.op_stack.pop_n(DIGEST_LEN)
        self.ip += 1

    def i_add(self, _):
        a, b = self.op_stack.pop(), self.op_stack.pop()
        self.op_stack.push(a + b)
        self.ip += 1

    def i_addi(self, v):
        self.op_stack[0] = self.op_stack[0] + v
        self.ip += 2

    def i_mul(self, _):
        a, b = self.op_stack.pop(), self.op_stack.pop()
        self.op_stack.push(a * b)
        self.ip += 1

    def i_invert(self, _):
        if self.op_stack[0] == 0:
            raise VMError("inverse of zero")
        self.op_stack.push(inv(self.op_stack.pop()))
        self.ip += 1

    def i_eq(self, _):
        a, b = self.op_stack.pop(), self.op_stack.pop()
        self.op_stack.push(int(a == b))
        self.ip += 1

    def i_split(self, _):
        top = self.op_stack.pop()
        lo, hi = top & 0xFFFFFFFF, top >> 32
        self.op_stack.push(hi)
        self.op_stack.push(lo)
        self.ip += 1
        return [("u32", ("split", lo, hi))]

    def _u32_pair(self):
        self.op_stack.get_u32(0)
        self.op_stack.get_u32(1)
        return self.op_stack.pop_u32(), self.op_stack.pop_u32()

    def i_lt(self, _):
        lhs, rhs = self._u32_pair()
        self.op_stack.push(int(lhs < rhs))
        self.ip += 1
        return [("u32", ("lt", lhs, rhs))]

    def i_and(self, _):
        lhs, rhs = self._u32_pair()
        self.op_stack.push(lhs & rhs)
        self.ip += 1
        return [("u32", ("and", lhs, rhs))]

    def i_xor(self, _):
        lhs, rhs = self._u32_pair()
        self.op_stack.push(lhs ^ rhs)
        self.ip += 1
        return [("u32", ("and", lhs, rhs))]

    def i_log_2_floor(self, _):
        self.op_stack.get_u32(0)
        if self.op_stack[0] == 0:
            raise VMError("logarithm of zero")
        top = self.op_stack.pop_u32()
        self.op_stack.push(top.bit_length() - 1)
        self.ip += 1
        return [("u32", ("log_2_floor", top, 0))]

    def i_pow(self, _):
        self.op_stack.get_u32(1)
        base = self.op_stack.pop()
        exponent = self.op_stack.pop_u32()
        self.op_stack.push(pow(base, exponent, P))
        self.ip += 1
        return [("u32", ("pow", base, exponent))]

    def i_div_mod(self, _):
        self.op_stack.get_u32(0)
        self.op_stack.get_u32(1)
        if self.op_stack[1] == 0:
            raise VMError("division by zero")
        numerator, denominator = self.op_stack.pop_u32(), self.op_stack.pop_u32()
        quotient, remainder = numerator // denominator, numerator % denominator
        self.op_stack.push(quotient)
        self.op_stack.push(remainder)
        self.ip += 1
        return [("u32", ("lt", remainder, denominator)), ("u32", ("split", numerator, quotient))]

    def i_pop_count(self, _):
        self.op_stack.get_u32(0)
        top = self.op_stack.pop_u32()
        self.op_stack.push(bin(top).count("1"))
        self.ip += 1
        return [("u32", ("pop_count", top, 0))]

    def i_xx_add(self, _):
        a, b = self.op_stack.pop_xfe(), self.op_stack.pop_xfe()
        self.op_stack.push_xfe(xfe_add(a, b))
        self.ip += 1

    def i_xx_mul(self, _):
        a, b = self.op_stack.pop_xfe(), self.op_stack.pop_xfe()
        self.op_stack.push_xfe(xfe_mul(a, b))
        self.ip += 1

    def i_x_invert(self, _):
        top = self.op_stack.peek_xfe(0)
        if top == [0, 0, 0]:
            raise VMError("inverse of zero")
        self.op_stack.pop_xfe()
        self.op_stack.push_xfe(xfe_inv(top))
        self.ip += 1

    def i_xb_mul(self, _):
        lhs = self.op_stack.pop()
        rhs = self.op_stack.pop_xfe()
        self.op_stack.push_xfe([lhs * c % P for c in rhs])
        self.ip += 1

    def i_write_io(self, n):
        for _ in range(n):
            self.public_output.append(self.op_stack.pop())
        self.ip += 2

    def i_read_io(self, n):
        if len(self.public_input) < n:
            raise VMError("public input empty")
        for _ in range(n):
            self.op_stack.push(self.public_input.popleft())
        self.ip += 2

    def i_merkle_step(self, _):
        self.op_stack.get_u32(5)
        if not self.secret_digests:
            raise VMError("secret digest input empty")
        return self._merkle_step(self.secret_digests.popleft())

    def i_merkle_step_mem(self, _):
        self.op_stack.get_u32(5)
        pointer = self.op_stack[7]
        sibling = []
        for _k in range(DIGEST_LEN):
            sibling.append(self.ram_read(pointer))
            pointer = (pointer + 1) % P
        self.op_stack[7] = pointer
        return self._merkle_step(sibling) + self._ram()

    def _merkle_step(self, sibling):
        node_index = self.op_stack.get_u32(5)
        parent = node_index // 2
        acc = self.op_stack.pop_n(DIGEST_LEN)
        left, right = (acc, sibling) if node_index % 2 == 0 else (sibling, acc)
        trace = tip5_trace(list(left) + list(right) + [1] * 6)
        for v in reversed(trace[-1][:DIGEST_LEN]):
            self.op_stack.push(v)
        self.op_stack[5] = parent
        self.ip += 1
        return [("hash", trace), ("u32", ("split", node_index, parent))]

    def _horner(self, coefficient):
        x = self.op_stack.peek_xfe(0)
        acc = xfe_add(xfe_mul(self.op_stack.peek_xfe(7), x), coefficient)
        self.op_stack[7], self.op_stack[8], self.op_stack[9] = acc
        self.ip += 1
        return self._ram()

    def i_b_horner_step(self, _):
        pointer = self.op_stack[5]
        c = self.ram_read(pointer)
        self.op_stack[5] = (pointer - 1) % P
        return self._horner([c, 0, 0])

    def i_x_horner_step(self, _):
        pointer = self.op_stack[5]
        coefficient = [0, 0, 0]
        for k in (2, 1, 0):
            coefficient[k] = self.ram_read(pointer)
            pointer = (pointer - 1) % P
        self.op_stack[5] = pointer
        return self._horner(coefficient)


def trace_execution(program, public_input=(), secret_input=(), secret_digests=(), ram=None):
    """VM::trace_execution (vm.rs:159-206) -> (AET, public output)."""
    vm = VM(program, public_input, secret_input, secret_digests, ram)
    aet = AET(program)
    while not vm.halting:
        if vm.ip >= len(aet.instruction_multiplicities):
            raise VMError("instruction pointer overflow")
        aet.instruction_multiplicities[vm.ip] += 1
        aet.processor_trace.append(vm.to_processor_row())
        for kind, payload in vm.step():
            if kind == "hash":
                aet.append_hash_trace(payload)
            elif kind == "sponge_reset":
                aet.append_initial_sponge_state()
            elif kind == "sponge":
                aet.append_sponge_trace(*payload)
            elif kind == "u32":
                aet.record_u32(*payload)
            elif kind == "op_stack":
                aet.op_stack_underflow_trace.append(payload)
            elif kind == "ram":
                aet.ram_trace.append(payload)
    return aet, vm.public_output
