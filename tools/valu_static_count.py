"""Static VALU instruction count of the row-hashing kernel (k_hash_rows_mfma) from the gfx950 ISA hipcc emits for
csrc/hash.hip with the product build's flags: the round loop (the innermost loop with the v_mfma instructions) and the
per-permutation code around it.  Writes profiles/valu_counts.json, which bench.py's `roofline_valu` object reads:
   wave-level VALU instructions per (extended row x permutation) = (5 * round + absorb) / 16 rows per wavefront.
Usage: python tools/valu_static_count.py"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_body(asm, name):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3tvm\d+" + name + r"E\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end]


def loops(body):
    """[(first line, last line)] of the natural loops: a label and the last backward branch to it"""
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    out = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i))
    return out


# Issue cost of a wave64 VALU instruction in cycles of its SIMD, from profiles/r02_valu_rates_microbench.txt
# (tools/ubench/valu_rates.hip, asm-volatile kernels: cycles = 256 / lane-ops per clock per CU):
#   plain 32-bit VOP1/VOP2 (v_mov_b32, v_xor_b32, v_add_u32 ...)   ~100 lane-ops/clk/CU -> 2.56
#   carry-out / VOP3 forms (v_add_co_u32, v_alignbyte_b32, v_lshl_add_u32 ...)  ~59   -> 4.3
#   v_mad_u64_u32                                                              ~49   -> 5.2
# The model uses the nominal pipe rates behind those measurements -- 2 cycles (SIMD-32 pass x 2, MI355X_MICROARCH.md
# "Wave scheduling"), 4 cycles, and the measured 5.2 for the 64-bit multiply-add -- so that `frac` is an upper bound on
# how far the kernel is from pure VALU issue (with the measured 2.56 / 4.3 / 5.2 the modelled cycles exceed the launch
# time by 9 %: the microbenchmark's eight dependent chains per lane do not reach the pipes' peak).  The model is good to
# a few per cent: a `frac` near 1 means "VALU-issue bound", not a measured utilisation.
CYCLES = {"plain": 2.0, "other": 4.0, "mad64": 5.0}
PLAIN = ("v_mov_b32_e32", "v_xor_b32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_lshlrev_b32_e32",
         "v_lshrrev_b32_e32", "v_not_b32_e32", "v_subrev_u32_e32")


def count(lines):
    c = {"valu": 0, "mfma": 0, "salu": 0, "lds": 0, "vmem": 0, "s_nop": 0, "valu_plain": 0, "valu_mad64": 0, "valu_other": 0}
    for l in lines:
        op = l.split()[0] if l.strip() and not l.strip().startswith((";", ".")) else ""
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["valu_mad64" if op == "v_mad_u64_u32" else "valu_plain" if op in PLAIN else "valu_other"] += 1
        elif op == "s_nop":
            c["s_nop"] += 1
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier")):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
    return c


def main():
    from triton_vm_amd.build import CSRC, ARCH, _hipcc

    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "hash.s")
        subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                               "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "hash.hip")],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    body = kernel_body(asm, "k_hash_rows_mfma")
    # The round loop = the innermost loop (depth 2) with the matrix instructions, the permutation loop the depth-1 loop around it:
    # by the loop annotations LLVM writes next to every block (since round 6 the round loop has a uniform branch inside -- the lean
    # last round of tip5.h -- and is laid out rotated: "a label and the last backward branch to it" no longer finds it).
    blocks, cur = [], None   # [(label, annotation, lines)]
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)$", l) or re.match(r"^; %bb\.(\d+):\s*;?(.*)$", l)
        if m:
            cur = [m.group(1), m.group(2), []]
            blocks.append(cur)
        elif cur is not None:
            if "Loop Header" in l or "Inner Loop Header" in l or "Parent Loop" in l:
                cur[1] += " " + l
            cur[2].append(l)
    def depth2_header(b):
        txt = b[1] + " ".join(x for x in b[2][:2] if x.lstrip().startswith(";"))
        m = re.search(r"Header=(BB\d+_\d+) Depth=2", txt)
        if m:
            return m.group(1)
        return b[0].lstrip(".L") if ("Inner Loop Header: Depth=2" in txt or "Depth=2" in txt and "Loop Header" in txt) else None
    inner = {}
    for b in blocks:
        h = depth2_header(b)
        if h:
            inner.setdefault(h, []).append(b)
    rnd_blocks = next(v for v in inner.values() if any("v_mfma" in x for b in v for x in b[2]))
    in_round = {id(b) for b in rnd_blocks}
    parent = next(m.group(1) for b in rnd_blocks for x in [b[1]] + b[2][:3] if (m := re.search(r"Parent Loop (BB\d+_\d+) Depth=1", x)))
    perm_blocks = [b for b in blocks if id(b) not in in_round and
                   (b[0].lstrip(".L") == parent or re.search(r"Header=" + parent + r" Depth=1", b[1] + " ".join(b[2][:2])))]
    c_round = count([x for b in rnd_blocks for x in b[2]])
    c_outside = count([x for b in perm_blocks for x in b[2]])
    # the lean last round skips the block with the rate words' multiply-add chains (the block of the round loop that holds
    # v_mad_u64_u32 but no matrix instruction and is not the largest): executed in 4 of a non-final permutation's 5 rounds
    mad_blocks = sorted((b for b in rnd_blocks if any(x.split()[:1] == ["v_mad_u64_u32"] for x in b[2]) and not any("v_mfma" in x for x in b[2])),
                        key=lambda b: len(b[2]))
    skipped = count(mad_blocks[0][2])["valu"] if len(mad_blocks) > 1 else 0
    per_wave_perm = 5 * c_round["valu"] - skipped + c_outside["valu"]
    cyc = lambda c: c["valu_plain"] * CYCLES["plain"] + c["valu_other"] * CYCLES["other"] + c["valu_mad64"] * CYCLES["mad64"]
    issue_cycles = 5 * cyc(c_round) + cyc(c_outside)
    rec = {"k_hash_rows_mfma": {
        "source": "static count over the gfx950 ISA of csrc/hash.hip (tools/valu_static_count.py); one wavefront = 16 rows.  Since round 6 "
                  "the round loop holds a uniform branch (the lean last round, tip5.h) and the static round count includes BOTH of its sides: "
                  "an upper bound by ~1 %; the measured figure is profiles/kernel_counters.json's (SQ_INSTS_VALU)",
        "round_loop": c_round, "per_permutation_outside_the_round_loop": c_outside,
        "valu_skipped_in_the_lean_last_round_of_a_non_final_permutation": skipped,
        "wave_valu_instructions_per_wave_permutation": per_wave_perm,
        "wave_valu_instructions_per_row_permutation": per_wave_perm / 16.0,
        "issue_cost_model_cycles": CYCLES,
        "modelled_valu_issue_cycles_per_wave_permutation": round(issue_cycles, 1),
        "modelled_valu_issue_cycles_per_row_permutation": round(issue_cycles / 16.0, 2),
        "x7_share_of_round_valu": round(12 * 16 / c_round["valu"], 3)}}
    path = os.path.join(ROOT, "profiles", "valu_counts.json")
    with open(path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
