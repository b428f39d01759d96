"""TEST INFRASTRUCTURE (oracle): the master tables' `fill`, `pad` and `extend`, restated from
/root/reference/triton-vm/src/table/master_table.rs:881-1075 and the per-table files
/root/reference/triton-vm/src/table/{program,processor,op_stack,ram,jump_stack,hash,cascade,lookup,u32}.rs.
Canonical python integers, one row at a time (small traces only).  Tables are lists of rows; `master_main` /
`master_aux` assemble the reference's column layout (triton-air/src/table.rs:27-103)."""
from tools.air import names

from . import isa
from .isa import INSTRUCTIONS, P
from .vm import (DIGEST_LEN, NUM_ROUNDS, RATE, inv, inverse_or_zero, op_stack_size_influence, sixteen_bit_limbs,
                 xfe_add, xfe_inv, xfe_mul)

TABLES = names.TABLES
MAIN_WIDTH = {t: len(names.MAIN_COLUMNS[t]) for t in TABLES}
AUX_WIDTH = {t: len(names.AUX_COLUMNS[t]) for t in TABLES}
M = {t: {n: i for i, n in enumerate(names.MAIN_COLUMNS[t])} for t in TABLES}     # per-table column indices
A = {t: {n: i for i, n in enumerate(names.AUX_COLUMNS[t])} for t in TABLES}
CH = {n: i for i, n in enumerate(names.CHALLENGES)}
OP = {name: op for name, (op, _) in INSTRUCTIONS.items()}
NUM_MAIN, NUM_AUX = sum(MAIN_WIDTH.values()), sum(AUX_WIDTH.values())           # 149, 49

TIP5_LOOKUP = None
TIP5_ROUND_CONSTANTS = None


def _tip5_constants():
    """Lookup table and round constants (canonical values), parsed from the generated oracle header."""
    global TIP5_LOOKUP, TIP5_ROUND_CONSTANTS
    if TIP5_LOOKUP is None:
        import os
        import re

        text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tip5_constants.h")).read()

        def array(name):
            body = re.search(name + r"\[\d+\] = \{(.*?)\};", text, re.S).group(1)
            return [int(t.rstrip("ul")) for t in re.findall(r"\d+u?l*", body)]

        r_inv = pow(1 << 64, -1, P)
        TIP5_ROUND_CONSTANTS = [v * r_inv % P for v in array("ORACLE_TIP5_ROUND_CONSTANTS")]
        TIP5_LOOKUP = array("ORACLE_TIP5_LOOKUP")
    return TIP5_LOOKUP, TIP5_ROUND_CONSTANTS


# ---- extension-field helpers -----------------------------------------------------------------------------------------
def X(b):
    return [b % P, 0, 0]


def xsub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def xscale(a, b):
    return [x * b % P for x in a]


def xsub_b(a, b):
    return [(a[0] - b) % P, a[1], a[2]]


def xadd_b(a, b):
    return [(a[0] + b) % P, a[1], a[2]]


def compress(ch, pairs):
    """sum of challenge * base-field value"""
    acc = [0, 0, 0]
    for cid, v in pairs:
        acc = xfe_add(acc, xscale(ch[CH[cid]], v))
    return acc


# ======================================================================================================= main table fill
def fill_op_stack(aet):
    """table/op_stack.rs:186-203, 251-281"""
    rows = sorted(aet.op_stack_underflow_trace, key=lambda r: (r[2], r[0]))
    rows = [list(r) for r in rows]
    cjd = [(b[0] - a[0]) % P for a, b in zip(rows, rows[1:]) if a[2] == b[2]]
    return rows, cjd


def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % P
    return out


def poly_eval(a, x):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % P
    return acc


def bezout_coefficient_polynomials_coefficients(roots):
    """table/ram.rs:152-207: a * rp + b * fd = 1 for rp = prod (X - r), fd = rp'."""
    n = len(roots)
    if n == 0:
        return [], []
    rp = [1]
    for r in roots:
        rp = poly_mul(rp, [(-r) % P, 1])
    fd = [(i * c) % P for i, c in enumerate(rp)][1:]
    # b: degree < n with b(r) = 1 / fd(r); Lagrange with denominators prod_{s != r} (r - s) = fd(r)
    b = [0] * n
    for r in roots:
        # numerator polynomial rp / (X - r) by synthetic division
        q, carry = [0] * n, 0
        for i in range(n, 0, -1):
            carry = (rp[i] + carry * r) % P
            q[i - 1] = carry
        w = inv(poly_eval(fd, r))
        scale = w * w % P                         # value 1/fd(r), divided by the Lagrange denominator fd(r)
        for i in range(n):
            b[i] = (b[i] + q[i] * scale) % P
    one_minus = [(-c) % P for c in poly_mul(fd, b)]
    one_minus[0] = (one_minus[0] + 1) % P
    # clean division by the monic rp
    num = list(one_minus)
    a = [0] * max(len(num) - n, 1)
    for i in range(len(num) - 1, n - 1, -1):
        c = num[i]
        a[i - n] = c
        if c:
            for j in range(n + 1):
                num[i - n + j] = (num[i - n + j] - c * rp[j]) % P
    assert not any(num[:n]), "division must be clean"
    while len(a) > 1 and a[-1] == 0:
        a.pop()
    while len(b) > 1 and b[-1] == 0:
        b.pop()
    a = (a + [0] * n)[:n]
    b = (b + [0] * n)[:n]
    return a, b


def fill_ram(aet):
    """table/ram.rs:64-84, 214-262"""
    m = M["Ram"]
    rows = []
    for clk, typ, ptr, val in sorted(aet.ram_trace, key=lambda r: (r[2], r[0])):
        row = [0] * MAIN_WIDTH["Ram"]
        row[m["CLK"]], row[m["InstructionType"]], row[m["RamPointer"]], row[m["RamValue"]] = clk, typ, ptr, val
        rows.append(row)
    if not rows:
        return rows, []
    unique = list(dict.fromkeys(r[m["RamPointer"]] for r in rows))
    b0, b1 = bezout_coefficient_polynomials_coefficients(unique)
    c0, c1 = b0.pop(), b1.pop()
    rows[0][m["BezoutCoefficientPolynomialCoefficient0"]] = c0
    rows[0][m["BezoutCoefficientPolynomialCoefficient1"]] = c1
    cjd = []
    for cur, nxt in zip(rows, rows[1:]):
        ramp_diff = (nxt[m["RamPointer"]] - cur[m["RamPointer"]]) % P
        if ramp_diff == 0:
            cjd.append((nxt[m["CLK"]] - cur[m["CLK"]]) % P)
        else:
            c0, c1 = b0.pop(), b1.pop()
        cur[m["InverseOfRampDifference"]] = inverse_or_zero(ramp_diff)
        nxt[m["BezoutCoefficientPolynomialCoefficient0"]] = c0
        nxt[m["BezoutCoefficientPolynomialCoefficient1"]] = c1
    assert not b0 and not b1
    return rows, cjd


def fill_jump_stack(aet):
    """table/jump_stack.rs:93-142"""
    pm = M["Processor"]
    by_jsp = []
    for r in aet.processor_trace:
        jsp = r[pm["JSP"]]
        entry = (r[pm["CLK"]], r[pm["CI"]], r[pm["JSO"]], r[pm["JSD"]])
        if jsp < len(by_jsp):
            by_jsp[jsp].append(entry)
        else:
            assert jsp == len(by_jsp)
            by_jsp.append([entry])
    rows = [[clk, ci, jsp, jso, jsd] for jsp, group in enumerate(by_jsp) for clk, ci, jso, jsd in group]
    cjd = [(b[0] - a[0]) % P for a, b in zip(rows, rows[1:]) if a[2] == b[2]]
    return rows, cjd


def fill_processor(aet, cjds):
    """table/processor.rs:44-68"""
    rows = [list(r) for r in aet.processor_trace]
    col = M["Processor"]["ClockJumpDifferenceLookupMultiplicity"]
    for d in cjds:
        rows[d][col] += 1
    return rows


def fill_program(aet):
    """table/program.rs:33-75"""
    m = M["Program"]
    words = aet.program.to_bwords()
    n, padded_len = len(words), aet.padded_program_length()
    padded = (words + [1] + [0] * padded_len)[:padded_len]
    rows = []
    for i, w in enumerate(padded):
        row = [0] * MAIN_WIDTH["Program"]
        row[m["Address"]], row[m["Instruction"]] = i, w
        row[m["LookupMultiplicity"]] = aet.instruction_multiplicities[i] if i < n else 0
        row[m["IndexInChunk"]] = i % RATE
        row[m["MaxMinusIndexInChunkInv"]] = inverse_or_zero(RATE - 1 - i % RATE)
        row[m["IsHashInputPadding"]] = 0 if i < n else 1
        rows.append(row)
    return rows


def lookup_16_bit_limb(limb):
    lut, _ = _tip5_constants()
    return (lut[limb >> 8] << 8) + lut[limb & 0xFF]


def inverse_or_zero_of_highest_2_limbs(x):
    limbs = sixteen_bit_limbs(x)
    return inverse_or_zero(0xFFFFFFFF - ((limbs[3] << 16) + limbs[2]))


def hash_table_row(mode, ci, round_number, state):
    """table/hash.rs:35-243"""
    m = M["Hash"]
    _, rc = _tip5_constants()
    row = [0] * MAIN_WIDTH["Hash"]
    row[m["Mode"]], row[m["CI"]], row[m["RoundNumber"]] = mode, ci, round_number
    for k in range(4):
        limbs = sixteen_bit_limbs(state[k])
        for name, limb in zip(("Lowest", "MidLow", "MidHigh", "Highest"), limbs):
            row[m[f"State{k}{name}LkIn"]] = limb
            row[m[f"State{k}{name}LkOut"]] = lookup_16_bit_limb(limb)
        row[m[f"State{k}Inv"]] = inverse_or_zero_of_highest_2_limbs(state[k])
    for k in range(4, 16):
        row[m[f"State{k}"]] = state[k]
    constants = rc[16 * round_number:16 * round_number + 16] if round_number < NUM_ROUNDS else [0] * 16
    for k in range(16):
        row[m[f"Constant{k}"]] = constants[k]
    return row


def fill_hash(aet):
    """table/hash.rs:249-278: program hashing (mode 1), sponge (2), hash (3)"""
    rows = []
    for mode, trace in ((1, aet.program_hash_trace), (2, aet.sponge_trace), (3, aet.hash_trace)):
        rows += [hash_table_row(mode, ci, r, state) for ci, r, state in trace]
    return rows


def fill_cascade(aet):
    """table/cascade.rs:42-58"""
    m = M["Cascade"]
    lut, _ = _tip5_constants()
    rows = []
    for limb, mult in aet.cascade_multiplicities.items():
        row = [0] * MAIN_WIDTH["Cascade"]
        lo, hi = limb & 0xFF, limb >> 8
        row[m["LookInLo"]], row[m["LookInHi"]], row[m["LookOutLo"]], row[m["LookOutHi"]] = lo, hi, lut[lo], lut[hi]
        row[m["LookupMultiplicity"]] = mult
        rows.append(row)
    return rows


def fill_lookup(aet):
    """table/lookup.rs:84-112"""
    m = M["Lookup"]
    lut, _ = _tip5_constants()
    rows = []
    for i in range(256):
        row = [0] * MAIN_WIDTH["Lookup"]
        row[m["LookIn"]], row[m["LookOut"]], row[m["LookupMultiplicity"]] = i, lut[i], aet.lookup_multiplicities[i]
        rows.append(row)
    return rows


def u32_section(name, lhs, rhs, multiplicity):
    """table/u32.rs:101-125, 196-291 (the recursion unrolled: rows top-down, results bottom-up)"""
    m = M["U32"]
    rows = []
    row = [0] * MAIN_WIDTH["U32"]
    row[m["CopyFlag"]], row[m["Bits"]], row[m["BitsMinus33Inv"]] = 1, 0, inv(-33)
    row[m["CI"]], row[m["LHS"]], row[m["RHS"]], row[m["LookupMultiplicity"]] = OP[name], lhs, rhs, multiplicity
    rows.append(row)
    while not ((rows[-1][m["LHS"]] == 0 or name == "pow") and rows[-1][m["RHS"]] == 0):
        cur = rows[-1]
        nxt = list(cur)
        nxt[m["CopyFlag"]] = 0
        nxt[m["Bits"]] = cur[m["Bits"]] + 1
        nxt[m["BitsMinus33Inv"]] = inv(nxt[m["Bits"]] - 33)
        nxt[m["LHS"]] = cur[m["LHS"]] if name == "pow" else cur[m["LHS"]] >> 1
        nxt[m["RHS"]] = cur[m["RHS"]] >> 1
        nxt[m["LookupMultiplicity"]] = 0
        nxt[m["LhsInv"]] = nxt[m["RhsInv"]] = nxt[m["Result"]] = 0
        rows.append(nxt)
    last = rows[-1]
    last[m["Result"]] = {"split": 0, "lt": 2, "and": 0, "log_2_floor": P - 1, "pow": 1, "pop_count": 0}[name]
    if name == "lt" and last[m["Bits"]] == 0:
        last[m["Result"]] = 0
    last[m["LhsInv"]] = inverse_or_zero(last[m["LHS"]])
    for k in range(len(rows) - 2, -1, -1):
        row, nxt = rows[k], rows[k + 1]
        lhs_lsb, rhs_lsb = row[m["LHS"]] % 2, row[m["RHS"]] % 2
        row[m["LhsInv"]], row[m["RhsInv"]] = inverse_or_zero(row[m["LHS"]]), inverse_or_zero(row[m["RHS"]])
        nr = nxt[m["Result"]]
        if name == "split":
            res = nr
        elif name == "lt":
            if nr in (0, 1):
                res = nr
            elif (lhs_lsb, rhs_lsb) == (0, 1):
                res = 1
            elif (lhs_lsb, rhs_lsb) == (1, 0):
                res = 0
            else:
                res = 0 if row[m["CopyFlag"]] == 1 else 2
        elif name == "and":
            res = (2 * nr + lhs_lsb * rhs_lsb) % P
        elif name == "log_2_floor":
            res = P - 1 if row[m["LHS"]] == 0 else (nr if nxt[m["LHS"]] != 0 else row[m["Bits"]])
        elif name == "pow":
            res = nr * nr % P if rhs_lsb == 0 else nr * nr % P * row[m["LHS"]] % P
        else:
            res = (nr + lhs_lsb) % P
        row[m["Result"]] = res
    return rows


def fill_u32(aet):
    rows = []
    for (name, lhs, rhs), mult in aet.u32_entries.items():
        rows += u32_section(name, lhs, rhs, mult)
    return rows


# ================================================================================================================== pad
def pad_program(rows, n):
    m = M["Program"]
    for i in range(len(rows), n):
        row = [0] * MAIN_WIDTH["Program"]
        row[m["Address"]], row[m["IndexInChunk"]] = i, i % RATE
        row[m["MaxMinusIndexInChunkInv"]] = inverse_or_zero(RATE - 1 - i % RATE)
        row[m["IsHashInputPadding"]] = row[m["IsTablePadding"]] = 1
        rows.append(row)


def pad_processor(rows, n):
    """table/processor.rs:70-96"""
    m = M["Processor"]
    table_len = len(rows)
    template = list(rows[-1])
    template[m["IsPadding"]], template[m["ClockJumpDifferenceLookupMultiplicity"]] = 1, 0
    for clk in range(table_len, n):
        row = list(template)
        row[m["CLK"]] = clk
        rows.append(row)
    rows[1][m["ClockJumpDifferenceLookupMultiplicity"]] += n - table_len


def pad_op_stack(rows, n):
    m = M["OpStack"]
    template = list(rows[-1]) if rows else [0] * MAIN_WIDTH["OpStack"]
    template[m["IB1ShrinkStack"]] = 2
    if not rows:
        template[m["StackPointer"]] = 16
    rows += [list(template) for _ in range(n - len(rows))]


def pad_ram(rows, n):
    m = M["Ram"]
    template = list(rows[-1]) if rows else [0] * MAIN_WIDTH["Ram"]
    template[m["InstructionType"]] = 2
    if not rows:
        template[m["BezoutCoefficientPolynomialCoefficient1"]] = 1
    rows += [list(template) for _ in range(n - len(rows))]


def pad_jump_stack(rows, n):
    """table/jump_stack.rs:144-199: the padding rows are inserted after the row with the largest clock"""
    table_len = len(rows)
    k = next(i for i, r in enumerate(rows) if r[0] == table_len - 1)
    tail = rows[k + 1:]
    del rows[k + 1:]
    template = rows[k]
    for clk in range(table_len, n):
        row = list(template)
        row[0] = clk
        rows.append(row)
    rows += tail


def pad_hash(rows, n):
    m = M["Hash"]
    _, rc = _tip5_constants()
    for _ in range(n - len(rows)):
        row = [0] * MAIN_WIDTH["Hash"]
        for k in range(4):
            row[m[f"State{k}Inv"]] = inverse_or_zero_of_highest_2_limbs(0)
        for k in range(16):
            row[m[f"Constant{k}"]] = rc[k]
        row[m["Mode"]], row[m["CI"]] = 0, OP["hash"]
        rows.append(row)


def pad_flag(table):
    def pad(rows, n):
        for _ in range(n - len(rows)):
            row = [0] * MAIN_WIDTH[table]
            row[M[table]["IsPadding"]] = 1
            rows.append(row)
    return pad


def pad_u32(rows, n):
    m = M["U32"]
    template = [0] * MAIN_WIDTH["U32"]
    template[m["CI"]], template[m["BitsMinus33Inv"]] = OP["split"], inv(-33)
    if rows:
        last = rows[-1]
        for c in ("CI", "LHS", "LhsInv", "Result"):
            template[m[c]] = last[m[c]]
        if template[m["CI"]] == OP["lt"]:
            template[m["Result"]] = 2
    rows += [list(template) for _ in range(n - len(rows))]


class MasterMainTable:
    """MasterMainTable::new + pad (master_table.rs:881-983), without the degree-lowering columns (those are
    oracle/degree_lowering.py's job)."""

    def __init__(self, aet, padded_height=None):
        self.aet = aet
        op_stack, cjd_os = fill_op_stack(aet)
        ram, cjd_ram = fill_ram(aet)
        jump_stack, cjd_js = fill_jump_stack(aet)
        self.tables = {
            "Program": fill_program(aet), "Processor": fill_processor(aet, cjd_os + cjd_ram + cjd_js),
            "OpStack": op_stack, "Ram": ram, "JumpStack": jump_stack, "Hash": fill_hash(aet),
            "Cascade": fill_cascade(aet), "Lookup": fill_lookup(aet), "U32": fill_u32(aet),
        }
        self.lengths = {t: len(rows) for t, rows in self.tables.items()}
        for t in TABLES:
            assert self.lengths[t] == aet.height_of_table(t), t
        self.padded_height = padded_height or aet.padded_height()

    def pad(self):
        n = self.padded_height
        pads = {"Program": pad_program, "Processor": pad_processor, "OpStack": pad_op_stack, "Ram": pad_ram,
                "JumpStack": pad_jump_stack, "Hash": pad_hash, "Cascade": pad_flag("Cascade"),
                "Lookup": pad_flag("Lookup"), "U32": pad_u32}
        for t in TABLES:
            pads[t](self.tables[t], n)
            assert len(self.tables[t]) == n
        return self

    def columns(self):
        """column-major [149][n] canonical values"""
        cols = []
        for t in TABLES:
            rows = self.tables[t]
            cols += [[r[c] % P for r in rows] for c in range(MAIN_WIDTH[t])]
        return cols


# =============================================================================================================== extend
def instruction_from_row(row):
    """table/processor.rs:760-770 -> (name, arg) or None"""
    pm = M["Processor"]
    name = isa.OPCODE_TO_NAME.get(row[pm["CI"]])
    if name is None:
        return None
    if INSTRUCTIONS[name][1]:
        arg = row[pm["NIA"]]
        if name in ("pop", "divine", "read_mem", "write_mem", "read_io", "write_io") and not 1 <= arg <= 5:
            return None
        if name in ("pick", "place", "dup", "swap") and not 0 <= arg <= 15:
            return None
        return name, arg
    return name, None


def extend_processor(rows, ch):
    pm = M["Processor"]
    n = len(rows)
    ST = [pm[f"ST{i}"] for i in range(16)]
    HV = [pm[f"HV{i}"] for i in range(6)]
    C = lambda name: ch[CH[name]]
    out = {name: [] for name in names.AUX_COLUMNS["Processor"]}
    weights = [ch[CH["StackWeight0"] + i] for i in range(10)]
    inp, outp = X(1), X(1)
    os_perm, ram_perm, js_perm = X(1), X(1), X(1)
    hash_in, hash_digest, sponge = X(1), X(1), X(1)
    u32_ld, cjd_ld, instr_ld = X(0), X(0), X(0)

    def weighted(values):
        acc = [0, 0, 0]
        for w, v in zip(weights, values):
            acc = xfe_add(acc, xscale(w, v))
        return acc

    for i, cur in enumerate(rows):
        prev = rows[i - 1] if i else None
        pi = instruction_from_row(prev) if prev is not None else None
        pci = prev[pm["CI"]] if prev is not None else None
        # input / output evaluation arguments (processor.rs:133-175)
        if pi and pi[0] == "read_io":
            for k in reversed(range(pi[1])):
                inp = xadd_b(xfe_mul(inp, C("StandardInputIndeterminate")), cur[ST[k]])
        if pi and pi[0] == "write_io":
            for k in range(pi[1]):
                outp = xadd_b(xfe_mul(outp, C("StandardOutputIndeterminate")), prev[ST[k]])
        out["InputTableEvalArg"].append(inp)
        out["OutputTableEvalArg"].append(outp)
        # instruction lookup (processor.rs:177-208)
        if cur[pm["IsPadding"]] != 1:
            cr = compress(ch, [("ProgramAddressWeight", cur[pm["IP"]]), ("ProgramInstructionWeight", cur[pm["CI"]]),
                               ("ProgramNextInstructionWeight", cur[pm["NIA"]])])
            instr_ld = xfe_add(instr_ld, xfe_inv(xsub(C("InstructionLookupIndeterminate"), cr)))
        out["InstructionLookupClientLogDerivative"].append(instr_ld)
        # op stack permutation argument (processor.rs:210-224, 563-611)
        if prev is not None and cur[pm["IsPadding"]] != 1 and pi is not None:
            delta = op_stack_size_influence(*pi)
            shorter = prev if delta > 0 else cur
            for off in range(abs(delta)):
                cr = compress(ch, [("OpStackClkWeight", prev[pm["CLK"]]), ("OpStackIb1Weight", prev[pm["IB1"]]),
                                   ("OpStackPointerWeight", (shorter[pm["OpStackPointer"]] + off) % P),
                                   ("OpStackFirstUnderflowElementWeight", shorter[ST[15 - off]])])
                os_perm = xfe_mul(os_perm, xsub(C("OpStackIndeterminate"), cr))
        out["OpStackTablePermArg"].append(os_perm)
        # ram permutation argument (processor.rs:226-241, 613-735)
        if prev is not None and cur[pm["IsPadding"]] != 1 and pi is not None:
            name = pi[0]
            accesses, typ = [], 1
            if name in ("read_mem", "write_mem"):
                typ = 1 if name == "read_mem" else 0
                longer = cur if name == "read_mem" else prev
                for off in range(pi[1]):
                    pointer = longer[ST[0]] + off + (1 if name == "read_mem" else 0)
                    accesses.append((pointer % P, longer[ST[off + 1]]))
            elif name == "sponge_absorb_mem":
                p0 = prev[ST[0]]
                accesses = [((p0 + k) % P, cur[ST[k + 1]]) for k in range(4)] + \
                           [((p0 + 4 + k) % P, prev[HV[k]]) for k in range(6)]
            elif name == "merkle_step_mem":
                accesses = [((prev[ST[7]] + k) % P, prev[HV[k]]) for k in range(5)]
            elif name == "b_horner_step":
                accesses = [(prev[ST[5]], prev[HV[0]])]
            elif name == "x_horner_step":
                accesses = [((prev[ST[5]] - 2 + k) % P, prev[HV[k]]) for k in range(3)]
            for pointer, value in accesses:
                cr = compress(ch, [("RamClkWeight", prev[pm["CLK"]]), ("RamInstructionTypeWeight", typ),
                                   ("RamPointerWeight", pointer), ("RamValueWeight", value)])
                ram_perm = xfe_mul(ram_perm, xsub(C("RamIndeterminate"), cr))
        out["RamTablePermArg"].append(ram_perm)
        # jump stack permutation argument (processor.rs:243-262)
        cr = compress(ch, [("JumpStackClkWeight", cur[pm["CLK"]]), ("JumpStackCiWeight", cur[pm["CI"]]),
                           ("JumpStackJspWeight", cur[pm["JSP"]]), ("JumpStackJsoWeight", cur[pm["JSO"]]),
                           ("JumpStackJsdWeight", cur[pm["JSD"]])])
        js_perm = xfe_mul(js_perm, xsub(C("JumpStackIndeterminate"), cr))
        out["JumpStackTablePermArg"].append(js_perm)
        # hash input (processor.rs:266-343): acts on the CURRENT row
        ci = cur[pm["CI"]]
        if ci in (OP["hash"], OP["merkle_step"], OP["merkle_step_mem"]):
            if ci == OP["hash"]:
                values = [cur[ST[k]] for k in range(10)]
            elif cur[ST[5]] % 2 == 0:
                values = [cur[ST[k]] for k in range(5)] + [cur[HV[k]] for k in range(5)]
            else:
                values = [cur[HV[k]] for k in range(5)] + [cur[ST[k]] for k in range(5)]
            hash_in = xfe_add(xfe_mul(hash_in, C("HashInputIndeterminate")), weighted(values))
        out["HashInputEvalArg"].append(hash_in)
        # hash digest (processor.rs:346-379)
        if pci in (OP["hash"], OP["merkle_step"], OP["merkle_step_mem"]):
            hash_digest = xfe_add(xfe_mul(hash_digest, C("HashDigestIndeterminate")),
                                  weighted([cur[ST[k]] for k in range(5)]))
        out["HashDigestEvalArg"].append(hash_digest)
        # sponge (processor.rs:383-464)
        if pci is not None:
            hci = C("HashCIWeight")
            if pci == OP["sponge_init"]:
                sponge = xfe_add(xfe_mul(sponge, C("SpongeIndeterminate")), xscale(hci, OP["sponge_init"]))
            elif pci == OP["sponge_absorb"]:
                sponge = xfe_add(xfe_add(xfe_mul(sponge, C("SpongeIndeterminate")), xscale(hci, OP["sponge_absorb"])),
                                 weighted([prev[ST[k]] for k in range(10)]))
            elif pci == OP["sponge_absorb_mem"]:
                values = [cur[ST[k]] for k in range(1, 5)] + [prev[HV[k]] for k in range(6)]
                sponge = xfe_add(xfe_add(xfe_mul(sponge, C("SpongeIndeterminate")), xscale(hci, OP["sponge_absorb"])),
                                 weighted(values))
            elif pci == OP["sponge_squeeze"]:
                sponge = xfe_add(xfe_add(xfe_mul(sponge, C("SpongeIndeterminate")), xscale(hci, OP["sponge_squeeze"])),
                                 weighted([cur[ST[k]] for k in range(10)]))
        out["SpongeEvalArg"].append(sponge)
        # u32 lookup (processor.rs:466-557)
        if pci is not None:
            U = lambda lhs, rhs, ci_, res=None: xfe_inv(xsub(C("U32Indeterminate"), compress(
                ch, [("U32LhsWeight", lhs), ("U32RhsWeight", rhs), ("U32CiWeight", ci_)] +
                ([("U32ResultWeight", res)] if res is not None else []))))
            if pci == OP["split"]:
                u32_ld = xfe_add(u32_ld, U(cur[ST[0]], cur[ST[1]], pci))
            elif pci in (OP["lt"], OP["and"], OP["pow"]):
                u32_ld = xfe_add(u32_ld, U(prev[ST[0]], prev[ST[1]], pci, cur[ST[0]]))
            elif pci == OP["xor"]:
                and_result = (prev[ST[0]] + prev[ST[1]] - cur[ST[0]]) * inv(2) % P
                u32_ld = xfe_add(u32_ld, U(prev[ST[0]], prev[ST[1]], OP["and"], and_result))
            elif pci in (OP["log_2_floor"], OP["pop_count"]):
                u32_ld = xfe_add(u32_ld, U(prev[ST[0]], 0, pci, cur[ST[0]]))
            elif pci == OP["div_mod"]:
                u32_ld = xfe_add(u32_ld, U(cur[ST[0]], prev[ST[1]], OP["lt"], 1))
                u32_ld = xfe_add(u32_ld, U(prev[ST[0]], cur[ST[1]], OP["split"]))
            elif pci in (OP["merkle_step"], OP["merkle_step_mem"]):
                u32_ld = xfe_add(u32_ld, U(prev[ST[5]], cur[ST[5]], OP["split"]))
        out["U32LookupClientLogDerivative"].append(u32_ld)
        # clock jump difference lookup server (processor.rs:536-561)
        mult = cur[pm["ClockJumpDifferenceLookupMultiplicity"]]
        if mult:
            term = xfe_inv(xsub_b(C("ClockJumpDifferenceLookupIndeterminate"), cur[pm["CLK"]]))
            cjd_ld = xfe_add(cjd_ld, xscale(term, mult))
        out["ClockJumpDifferenceLookupServerLogDerivative"].append(cjd_ld)
    assert all(len(v) == n for v in out.values())
    return [out[name] for name in names.AUX_COLUMNS["Processor"]]


def extend_program(rows, ch):
    """table/program.rs:129-269"""
    m = M["Program"]
    C = lambda name: ch[CH[name]]
    n = len(rows)
    ld, prep, send = X(0), X(1), X(1)
    cols = [[], [], []]
    for i, row in enumerate(rows):
        cols[0].append(ld)
        if i + 1 < n and row[m["IsHashInputPadding"]] != 1:
            nxt = rows[i + 1]
            cr = compress(ch, [("ProgramAddressWeight", row[m["Address"]]), ("ProgramInstructionWeight", row[m["Instruction"]]),
                               ("ProgramNextInstructionWeight", nxt[m["Instruction"]])])
            ld = xfe_add(ld, xscale(xfe_inv(xsub(C("InstructionLookupIndeterminate"), cr)), row[m["LookupMultiplicity"]]))
        if row[m["IndexInChunk"]] == 0:
            prep = X(1)
        prep = xadd_b(xfe_mul(prep, C("ProgramAttestationPrepareChunkIndeterminate")), row[m["Instruction"]])
        if row[m["IsTablePadding"]] != 1 and row[m["IndexInChunk"]] == RATE - 1:
            send = xfe_add(xfe_mul(send, C("ProgramAttestationSendChunkIndeterminate")), prep)
        cols[1].append(prep)
        cols[2].append(send)
    return cols


def extend_op_stack(rows, ch):
    """table/op_stack.rs:108-174"""
    m = M["OpStack"]
    C = lambda name: ch[CH[name]]
    rp, ld = X(1), X(0)
    cols = [[], []]
    padding_reached = False
    for i, row in enumerate(rows):
        if row[m["IB1ShrinkStack"]] != 2:
            cr = compress(ch, [("OpStackClkWeight", row[m["CLK"]]), ("OpStackIb1Weight", row[m["IB1ShrinkStack"]]),
                               ("OpStackPointerWeight", row[m["StackPointer"]]),
                               ("OpStackFirstUnderflowElementWeight", row[m["FirstUnderflowElement"]])])
            rp = xfe_mul(rp, xsub(C("OpStackIndeterminate"), cr))
        cols[0].append(rp)
        if i and not padding_reached:
            prev = rows[i - 1]
            if row[m["IB1ShrinkStack"]] == 2:
                padding_reached = True
            elif prev[m["StackPointer"]] == row[m["StackPointer"]]:
                diff = (row[m["CLK"]] - prev[m["CLK"]]) % P
                ld = xfe_add(ld, xfe_inv(xsub_b(C("ClockJumpDifferenceLookupIndeterminate"), diff)))
        cols[1].append(ld)
    return cols


def extend_ram(rows, ch):
    """table/ram.rs:264-399"""
    m = M["Ram"]
    C = lambda name: ch[CH[name]]
    bez = C("RamTableBezoutRelationIndeterminate")
    rp = xsub_b(bez, rows[0][m["RamPointer"]])
    fd = X(1)
    bc0, bc1 = X(rows[0][m["BezoutCoefficientPolynomialCoefficient0"]]), X(rows[0][m["BezoutCoefficientPolynomialCoefficient1"]])
    perm, ld = X(1), X(0)
    cols = [[] for _ in range(6)]
    pad_seen = False
    for i, row in enumerate(rows):
        is_pad = row[m["InstructionType"]] == 2
        pad_seen = pad_seen or is_pad
        if i and not is_pad:
            prev = rows[i - 1]
            if prev[m["RamPointer"]] != row[m["RamPointer"]]:
                factor = xsub_b(bez, row[m["RamPointer"]])
                fd = xfe_add(xfe_mul(factor, fd), rp)
                rp = xfe_mul(rp, factor)
        if i and not pad_seen:
            prev = rows[i - 1]
            if prev[m["RamPointer"]] != row[m["RamPointer"]]:
                bc0 = xadd_b(xfe_mul(bc0, bez), row[m["BezoutCoefficientPolynomialCoefficient0"]])
                bc1 = xadd_b(xfe_mul(bc1, bez), row[m["BezoutCoefficientPolynomialCoefficient1"]])
            else:
                diff = (row[m["CLK"]] - prev[m["CLK"]]) % P
                ld = xfe_add(ld, xfe_inv(xsub_b(C("ClockJumpDifferenceLookupIndeterminate"), diff)))
        if not pad_seen:
            cr = compress(ch, [("RamClkWeight", row[m["CLK"]]), ("RamInstructionTypeWeight", row[m["InstructionType"]]),
                               ("RamPointerWeight", row[m["RamPointer"]]), ("RamValueWeight", row[m["RamValue"]])])
            perm = xfe_mul(perm, xsub(C("RamIndeterminate"), cr))
        for c, v in zip(cols, (rp, fd, bc0, bc1, perm, ld)):
            c.append(v)
    return cols


def extend_jump_stack(rows, ch):
    """table/jump_stack.rs:31-91"""
    C = lambda name: ch[CH[name]]
    rp, ld = X(1), X(0)
    cols = [[], []]
    for i, row in enumerate(rows):
        cr = compress(ch, [("JumpStackClkWeight", row[0]), ("JumpStackCiWeight", row[1]), ("JumpStackJspWeight", row[2]),
                           ("JumpStackJsoWeight", row[3]), ("JumpStackJsdWeight", row[4])])
        rp = xfe_mul(rp, xsub(C("JumpStackIndeterminate"), cr))
        cols[0].append(rp)
        if i and rows[i - 1][2] == row[2]:
            diff = (row[0] - rows[i - 1][0]) % P
            ld = xfe_add(ld, xfe_inv(xsub_b(C("ClockJumpDifferenceLookupIndeterminate"), diff)))
        cols[1].append(ld)
    return cols


def extend_hash(rows, ch):
    """table/hash.rs:311-600"""
    m = M["Hash"]
    C = lambda name: ch[CH[name]]
    weights = [ch[CH["StackWeight0"] + i] for i in range(10)]
    r_inv = inv(1 << 64)
    recv, hin, hdig, sponge = X(1), X(1), X(1), X(1)
    lds = [X(0) for _ in range(16)]
    cols = [[] for _ in range(20)]
    limb_names = ("Highest", "MidHigh", "MidLow", "Lowest")

    def rate_registers(row):
        regs = []
        for k in range(4):
            v = (row[m[f"State{k}HighestLkIn"]] << 48) + (row[m[f"State{k}MidHighLkIn"]] << 32) + \
                (row[m[f"State{k}MidLowLkIn"]] << 16) + row[m[f"State{k}LowestLkIn"]]
            regs.append(v % P * r_inv % P)
        return regs + [row[m[f"State{k}"]] for k in range(4, 10)]

    def weighted(values):
        acc = [0, 0, 0]
        for w, v in zip(weights, values):
            acc = xfe_add(acc, xscale(w, v))
        return acc

    for row in rows:
        mode, rnd, ci = row[m["Mode"]], row[m["RoundNumber"]], row[m["CI"]]
        is_init = ci == OP["sponge_init"]
        if mode == 1 and rnd == 0:
            chunk = X(1)
            for v in rate_registers(row):
                chunk = xadd_b(xfe_mul(chunk, C("ProgramAttestationPrepareChunkIndeterminate")), v)
            recv = xfe_add(xfe_mul(recv, C("ProgramAttestationSendChunkIndeterminate")), chunk)
        if mode == 2 and rnd == 0:
            sponge = xfe_add(xfe_mul(sponge, C("SpongeIndeterminate")), xscale(C("HashCIWeight"), ci))
            if not is_init:
                sponge = xfe_add(sponge, weighted(rate_registers(row)))
        if mode == 3 and rnd == 0:
            hin = xfe_add(xfe_mul(hin, C("HashInputIndeterminate")), weighted(rate_registers(row)))
        if mode == 3 and rnd == NUM_ROUNDS:
            hdig = xfe_add(xfe_mul(hdig, C("HashDigestIndeterminate")), weighted(rate_registers(row)[:DIGEST_LEN]))
        if mode != 0 and rnd != NUM_ROUNDS and not is_init:
            for k in range(4):
                for j, ln in enumerate(limb_names):
                    ce = xsub(xsub(C("HashCascadeLookupIndeterminate"),
                                   xscale(C("HashCascadeLookInWeight"), row[m[f"State{k}{ln}LkIn"]])),
                              xscale(C("HashCascadeLookOutWeight"), row[m[f"State{k}{ln}LkOut"]]))
                    lds[4 * k + j] = xfe_add(lds[4 * k + j], xfe_inv(ce))
        for c, v in zip(cols, [recv, hin, hdig, sponge] + lds):
            c.append(v)
    return cols


def extend_cascade(rows, ch):
    m = M["Cascade"]
    C = lambda name: ch[CH[name]]
    hash_ld, lookup_ld = X(0), X(0)
    cols = [[], []]
    for row in rows:
        if row[m["IsPadding"]] != 1:
            look_in = (row[m["LookInHi"]] << 8) + row[m["LookInLo"]]
            look_out = (row[m["LookOutHi"]] << 8) + row[m["LookOutLo"]]
            cr = compress(ch, [("HashCascadeLookInWeight", look_in), ("HashCascadeLookOutWeight", look_out)])
            hash_ld = xfe_add(hash_ld, xscale(xfe_inv(xsub(C("HashCascadeLookupIndeterminate"), cr)), row[m["LookupMultiplicity"]]))
            for half in ("Lo", "Hi"):
                cr = compress(ch, [("LookupTableInputWeight", row[m["LookIn" + half]]),
                                   ("LookupTableOutputWeight", row[m["LookOut" + half]])])
                lookup_ld = xfe_add(lookup_ld, xfe_inv(xsub(C("CascadeLookupIndeterminate"), cr)))
        cols[0].append(hash_ld)
        cols[1].append(lookup_ld)
    return cols


def extend_lookup(rows, ch):
    m = M["Lookup"]
    C = lambda name: ch[CH[name]]
    ld, ev = X(0), X(1)
    cols = [[], []]
    pad_seen = False
    for row in rows:
        pad_seen = pad_seen or row[m["IsPadding"]] == 1
        if not pad_seen:
            cr = compress(ch, [("LookupTableInputWeight", row[m["LookIn"]]), ("LookupTableOutputWeight", row[m["LookOut"]])])
            ld = xfe_add(ld, xscale(xfe_inv(xsub(C("CascadeLookupIndeterminate"), cr)), row[m["LookupMultiplicity"]]))
            ev = xadd_b(xfe_mul(ev, C("LookupTablePublicIndeterminate")), row[m["LookOut"]])
        cols[0].append(ld)
        cols[1].append(ev)
    return cols


def extend_u32(rows, ch):
    m = M["U32"]
    C = lambda name: ch[CH[name]]
    ld = X(0)
    col = []
    for row in rows:
        if row[m["CopyFlag"]] == 1:
            cr = compress(ch, [("U32CiWeight", row[m["CI"]]), ("U32LhsWeight", row[m["LHS"]]),
                               ("U32RhsWeight", row[m["RHS"]]), ("U32ResultWeight", row[m["Result"]])])
            ld = xfe_add(ld, xscale(xfe_inv(xsub(C("U32Indeterminate"), cr)), row[m["LookupMultiplicity"]]))
        col.append(ld)
    return [col]


EXTEND = {"Program": extend_program, "Processor": extend_processor, "OpStack": extend_op_stack, "Ram": extend_ram,
          "JumpStack": extend_jump_stack, "Hash": extend_hash, "Cascade": extend_cascade, "Lookup": extend_lookup,
          "U32": extend_u32}


def extend(main_tables, challenges):
    """MasterMainTable::extend (master_table.rs:1006-1075) without the degree-lowering and randomizer columns:
    [49][n] XFE (lists of 3 canonical values).  main_tables: {table: padded rows}; challenges: [63] XFE."""
    cols = []
    for t in TABLES:
        got = EXTEND[t](main_tables[t], challenges)
        assert len(got) == AUX_WIDTH[t], t
        cols += got
    return cols


def derive_challenges(sampled, program_digest, public_input, public_output):
    """Challenges::new (challenges.rs:85-121): 59 sampled + 4 derived."""
    lut, _ = _tip5_constants()

    def terminal(symbols, x):
        acc = X(1)
        for s in symbols:
            acc = xadd_b(xfe_mul(acc, x), s)
        return acc

    ch = [list(c) for c in sampled]
    assert len(ch) == 59
    ch.append(terminal(public_input, ch[CH["StandardInputIndeterminate"]]))
    ch.append(terminal(public_output, ch[CH["StandardOutputIndeterminate"]]))
    ch.append(terminal(lut, ch[CH["LookupTablePublicIndeterminate"]]))
    ch.append(terminal(program_digest, ch[CH["CompressProgramDigestIndeterminate"]]))
    return ch
