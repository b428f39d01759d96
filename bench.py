#!/usr/bin/env python3
"""bench.py -- `Prover::prove(claim, aet)` on MI355X, BASELINE.json's metric (trace-cells/s in prove()).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-rows 20] [--data real|synthetic] [--ldt fri|stir|auto]

Default (`--data real`): BASELINE.json configs[1] for real -- the reference's prove_fib program
(triton-dev-util/src/example_programs.rs:6-38) is run for 2^20 - 17 cycles, and ONE STEP is the whole of
`Prover::prove(claim, aet)` (/root/reference/triton-vm/src/stark.rs:331-719) on that algebraic execution trace:
fill + pad of the master main table, the seeded trace randomizers, main LDE, row hashing + Merkle tree, Fiat-Shamir,
extend (aux table), aux LDE, hashing + Merkle, AIR / quotient codeword, quotient segments, out-of-domain rows, linear
combination, DEEP, the low-degree test, the openings, the proof's encoding -- through the C++ host
(triton_vm_amd/host/triton_host.cpp, `triton_vm::prove_execution`) over the C ABI.  Stark::default() parameters with
LdtChoice::Fri (what BASELINE.json's configurations name; the STIR that Stark::default() picks by itself at this size is
measured beside it).  The proof of the last timed step is put through Verifier::verify before the line is printed.

What is OUTSIDE the timed region: running the program (the VM is host work in the reference too and not part of
`prove`; here the oracle-side VM, oracle/vm, stands in for the Rust VM -- workload generation, not a product path) and
the one-time upload of the execution trace: the timed steps read a DEVICE-RESIDENT trace (inputs resident in HBM when the
timed region starts); the same step from host arrays (327 MB over PCIe inside the step) is reported as
`pcie_inclusive`.  `--data synthetic` times the hot path alone on random tables (rounds 1-2's headline; kept as the
`synthetic_hot_path` key of the default run).

metric = padded_rows * 652 / seconds per step (652 = 379 main + 3 * 91 aux base-field words per row).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).  Default: ONE proof split over the N
GPUs by cosets of the trace domain (triton_vm_amd/sharded.py; N must divide 8): all-to-all of leaf digests, all-gather
of the quotient codeword -- total work fixed, "scaling": "strong".  --replicas: every rank proves its own instance
(no data-path collective, "scaling": "weak").  Barrier + max-over-ranks timing either way.
"""
import argparse
import json
import os
import sys
import time

# this pool's host driver shares device memory between processes through dmabuf only: RCCL's setup of a multi-rank communicator
# needs this before the HIP runtime starts, whoever launched the rank (the driver's torchrun or spawn_ranks below)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MASTER_WORDS = 652
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_LANE_OPS_PER_CLK_PER_CU = 128   # MI355X_MICROARCH.md: 4 SIMDs x 32 lanes per clock = 2 cycles per wave64 instruction
N_CUS, CLOCK_GHZ = 256, 2.4
LDE_ALGORITHMIC_BYTES_PER_CELL = 72   # read 8 B, write 8 * (L/N = 8) B per base-field trace cell (default expansion)
PROVER_SEED = bytes(range(32))        # set_randomness_seed_which_may_break_zero_knowledge (stark.rs:322-328): reproducible proofs


# ---- process set-up: overridable, so that tests/bench_on_emulation.py can run this script's control flow on CPU ------
def make_context(local_rank):
    from triton_vm_amd import Context

    return Context(device=local_rank)          # raises without a GPU: there is no fallback


def visible_devices():
    import torch

    return torch.cuda.device_count()


def init_distributed(local_rank):
    """-> (torch.distributed, device) for a multi-rank launch"""
    import torch
    import torch.distributed as dist

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(local_rank)  # torch first, then the Context (triton_vm_amd/sharded.py)
    dist.init_process_group(backend="nccl", device_id=device)
    return dist, device


USE_CPP_HOST = True


def load_host_library():
    """the C++ host above the C ABI (triton_vm_amd/host/), linked against the product library"""
    from triton_vm_amd import native_host

    return native_host.load_host_library()


def make_comm(dist, device, rank, world, local_rank):
    """this rank's communicator for the sharded C++ host: RCCL on the context's stream (triton_vm_amd/host/rccl_comm.cpp).
    Rank 0 draws the ncclUniqueId; torch.distributed (already initialised for the barrier / max-over-ranks timing contract)
    carries its 128 bytes to the other ranks."""
    import numpy as np
    import torch

    from triton_vm_amd import native_host

    uid = torch.from_numpy(native_host.RcclComm.unique_id() if rank == 0 else np.zeros(128, np.uint8)).to(device)
    dist.broadcast(uid, 0)
    return native_host.RcclComm(uid.cpu().numpy(), rank, world, local_rank)


# ---- the CPU baseline ----------------------------------------------------------------------------------------------
def available_cores():
    """CPUs this process may actually run on: the affinity mask, capped by the cgroup's CPU quota (a container on a 256-thread
    host is often allowed a few cores' worth of time: os.cpu_count() says 256 there and is not the number to parallelise by)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, round(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, round(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(log2_rows, log2_expansion=2):
    """The CPU port (`kind: "port"`; the reference is Rust and cannot be built here): oracle/tvm_oracle_fast.c -- the hot
    stages restated in the forms a CPU implementation uses (Tip5 with the integer-halves MDS, table-driven transforms, a typed
    walk of the AIR circuit, batched inversions; held bit-equal to the textbook oracle by tests/test_oracle_fast.py), OpenMP over
    rows / (column, coset) pairs -- on a bounded sample of the same workload, stage by stage, each stage scaled to prove():
      LDE            C main-table columns at the full height onto the expansion-times-2 cosets                  x 652 / C
      row hashing    Tip5 hash_varlen over 2^18 rows of that C-column table                                     x permutations(full) / permutations(sample)
      Merkle         one tree over those leaf digests                                                           x (3 table trees + the FRI round trees ~ 1) x L / 2^18
      AIR            all 604 constraints + zerofiers on 2^13 full-width quotient-domain rows                    x |quotient domain| / 2^13
      DEEP, FRI      4 DEEP components and one fold on 2^18-point codewords                                     x |LDT domain| / 2^18 (folds: x 2)
    `cores` = the OpenMP threads used = the CPUs this process may run on (affinity and cgroup quota, not os.cpu_count());
    `parallel_speedup` = the measured ratio of the all-thread and the one-thread hashing rate."""
    import ctypes

    import numpy as np

    from oracle import oracle as orc

    fast = orc.fast
    orc.lib()
    cores = available_cores()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        set_threads = gomp.omp_set_num_threads
    except OSError:
        set_threads = lambda n: None   # noqa: E731
    n = 1 << log2_rows
    X = 2 << log2_expansion
    L = X * n
    rng = np.random.default_rng(1)
    g = orc.lib().orc_bfe_generator()
    perms = lambda w: w // 10 + 1   # noqa: E731
    # calibration: one thread against all of them on the same hashing job
    cal = orc.random_elements(rng, (1 << 14, 64))
    set_threads(1)
    t0 = time.perf_counter()
    fast.hash_rows(cal)
    us_per_perm_1 = (time.perf_counter() - t0) / (cal.shape[0] * perms(64)) * 1e6
    t0 = time.perf_counter()
    fast.ntt(orc.random_elements(rng, 1 << 18), orc.lib().orc_bfe_primitive_root(1 << 18))
    ns_per_butterfly_1 = (time.perf_counter() - t0) / ((1 << 18) * 9) * 1e9
    set_threads(cores)
    tiled = np.tile(cal, (max(1, min(cores, 16)), 1))
    fast.hash_rows(tiled[:1 << 12])   # (the thread team starts here, not inside the timed call)
    t0 = time.perf_counter()
    fast.hash_rows(tiled)
    us_per_perm_all = (time.perf_counter() - t0) / (tiled.shape[0] * perms(64)) * 1e6
    speedup = us_per_perm_1 / us_per_perm_all
    del cal, tiled
    big = cores >= 8                      # sample sizes: about 10-30 s of CPU work either way
    cols, h = (64 if big else 16), 198
    t, scaled = {}, {}

    def timed(name, scale, fn):
        t0 = time.perf_counter()
        out = fn()
        t[name] = time.perf_counter() - t0
        scaled[name] = t[name] * scale
        return out

    trace = orc.random_elements(rng, (cols, n))
    rnd = orc.random_elements(rng, (cols, h))
    ev = orc.domain_of_length(L, offset=g)
    table = timed("lde", MASTER_WORDS / cols, lambda: fast.lde_table(trace, rnd, ev))
    hs = min(L, 1 << (20 if big else 18))         # rows hashed / leaves of the sampled tree
    digests = timed("hash_rows", (perms(379) + perms(273) + perms(15)) / perms(cols) * L / hs, lambda: fast.hash_rows(table[:hs]))
    timed("merkle", 4.0 * L / hs, lambda: fast.merkle_tree(digests))
    del table, digests, trace
    q_s = 1 << (15 if big else 13)
    main_rows = orc.random_elements(rng, (q_s, 379))
    aux_rows = orc.random_elements(rng, (q_s, 91, 3))
    ch, w = orc.random_elements(rng, (63, 3)), orc.random_elements(rng, (604, 3))
    timed("air", L / q_s, lambda: fast.quotients_combined(main_rows, aux_rows, orc.domain_of_length(q_s // 8), orc.domain_of_length(q_s, offset=g), ch, w))
    d_s = orc.domain_of_length(1 << 18, offset=g)
    cw = orc.random_elements(rng, (d_s.length, 3))
    pt, val = orc.random_elements(rng, 3), orc.random_elements(rng, 3)
    timed("deep", 4 * L / d_s.length, lambda: fast.deep_codeword(cw, d_s, pt, val))
    timed("fri_fold", 2 * L / d_s.length, lambda: orc.fri_split_and_fold(cw, d_s, pt))
    est = sum(scaled.values())
    butterflies = cols * (X + 1) * n * log2_rows / 2
    return {"value": round(n * MASTER_WORDS / est, 1), "unit": "trace-cells/s", "cores": cores, "kind": "port",
            "host_threads_reported_by_os": os.cpu_count(), "parallel_speedup": round(speedup, 1),
            "estimated_prove_seconds": round(est, 1), "sample_seconds": {k: round(v, 3) for k, v in t.items()},
            "scaled_seconds": {k: round(v, 1) for k, v in scaled.items()},
            "rates": {"tip5_us_per_permutation_one_thread": round(us_per_perm_1, 2),
                      "tip5_us_per_permutation_all_threads": round(us_per_perm_all, 3),
                      "row_hashing_us_per_permutation": round(t["hash_rows"] / (hs * perms(cols)) * 1e6, 3),
                      "ntt_ns_per_butterfly_one_thread": round(ns_per_butterfly_1, 1),
                      "lde_ns_per_butterfly": round(t["lde"] / butterflies * 1e9, 2),
                      "air_us_per_row": round(t["air"] / q_s * 1e6, 1)},
            "sample": f"oracle/tvm_oracle_fast.c (C, OpenMP, {cores} threads = the CPUs this process may use; os.cpu_count() = {os.cpu_count()}): LDE of {cols} "
                      f"main columns at 2^{log2_rows} rows onto the {X}x domain, Tip5 hashing of {hs} of that table's rows, one Merkle tree over them, the "
                      f"AIR on {q_s} full-width quotient rows, DEEP (4 components) and one FRI fold on 2^18-point codewords; every stage "
                      f"scaled to the full prove() ({sum(t.values()):.1f} s measured -> {est:.0f} s estimated).  An optimised restatement, "
                      "NOT the Rust prover (no cargo in this image): a reported baseline, not a speed-up claim"}


def cpu_baseline_full(log2_rows, log2_expansion=2):
    """`--cpu-baseline full`: the same port (oracle/tvm_oracle_fast.c, OpenMP) over the WHOLE work of one prove() at this height -- every
    column of both tables extended, every row of the three tables hashed, the three table trees and the FRI round trees built, the AIR on
    every quotient-domain row, DEEP and every FRI fold -- timed stage by stage, nothing scaled.  To stay inside a host's memory the
    tables are produced and consumed in blocks (64 columns for the extension, 2^20 rows for the hashing, 2^15 rows for the AIR), and a
    block's CONTENT is synthetic and reused (the cost of these stages does not depend on the values): it is a timing of all of the
    work, not a proof.  About 70 s on 16 cores at 2^20 rows; the result is committed under profiles/ and cited by the default line."""
    import ctypes

    import numpy as np

    from oracle import oracle as orc

    fast = orc.fast
    orc.lib()
    cores = available_cores()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    n = 1 << log2_rows
    X = 2 << log2_expansion
    L = X * n
    rng = np.random.default_rng(1)
    g = orc.lib().orc_bfe_generator()
    t = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        out = fn()
        t[name] = t.get(name, 0.0) + time.perf_counter() - t0
        return out

    # low-degree extension: 652 base-field columns, 64 at a time (the block is discarded: the hashing below reads a synthetic one)
    ev = orc.domain_of_length(L, offset=g)
    cols_done = 0
    while cols_done < MASTER_WORDS:
        c = min(64, MASTER_WORDS - cols_done)
        trace, rnd = orc.random_elements(rng, (c, n)), orc.random_elements(rng, (c, 198))
        timed("lde", lambda: fast.lde_table(trace, rnd, ev))
        cols_done += c
    del trace, rnd
    # row hashing: every row of the main (379 words), auxiliary (273) and quotient-segment (15) tables, 2^20 rows at a time
    rows_blk = min(L, 1 << 20)
    digests = None
    for width in (379, 273, 15):
        block = orc.random_elements(rng, (rows_blk, width))
        for _ in range(L // rows_blk):
            digests = timed("hash_rows", lambda: fast.hash_rows(block))
        del block
    # Merkle trees: three table trees of L leaves and the FRI rounds' trees (L, L/2, ... leaves: another 2 L leaves' worth)
    leaves = np.tile(digests, (L // digests.shape[0], 1)) if digests.shape[0] < L else digests
    for _ in range(3):
        timed("merkle", lambda: fast.merkle_tree(leaves))
    m = L
    while m >= 1024:
        timed("merkle", lambda: fast.merkle_tree(leaves[:m]))
        m //= 2
    del leaves, digests
    # AIR: all constraints and zerofiers on every row of the quotient domain, 2^15 rows at a time
    q_s = min(L, 1 << 15)
    main_rows, aux_rows = orc.random_elements(rng, (q_s, 379)), orc.random_elements(rng, (q_s, 91, 3))
    ch, w = orc.random_elements(rng, (63, 3)), orc.random_elements(rng, (604, 3))
    tr_s, qd_s = orc.domain_of_length(q_s // 8), orc.domain_of_length(q_s, offset=g)
    for _ in range(L // q_s):
        timed("air", lambda: fast.quotients_combined(main_rows, aux_rows, tr_s, qd_s, ch, w))
    # DEEP (4 components on the LDT domain) and the FRI folds (L, L/2, ... points)
    cw = orc.random_elements(rng, (L, 3))
    pt, val = orc.random_elements(rng, 3), orc.random_elements(rng, 3)
    for _ in range(4):
        timed("deep", lambda: fast.deep_codeword(cw, ev, pt, val))
    m = L
    while m >= 1024:
        d_m = orc.domain_of_length(m, offset=g)
        cw_m = cw[:m]
        timed("fri_fold", lambda: orc.fri_split_and_fold(cw_m, d_m, pt))
        m //= 2
    total = sum(t.values())
    return {"value": round(n * MASTER_WORDS / total, 1), "unit": "trace-cells/s", "cores": cores, "kind": "port", "mode": "full",
            "host_threads_reported_by_os": os.cpu_count(), "prove_seconds": round(total, 1), "stage_seconds": {k: round(v, 2) for k, v in t.items()},
            "sample": f"oracle/tvm_oracle_fast.c (C, OpenMP, {cores} threads) over ALL the work of one prove() at 2^{log2_rows} rows, nothing scaled: "
                      f"{MASTER_WORDS} columns extended onto the {X}x domain, {L} rows of the 379- / 273- / 15-word tables hashed, the table and FRI "
                      f"trees, the AIR on {L} quotient rows, DEEP and the folds ({total:.0f} s; blocks of synthetic content, reused).  An optimised "
                      "restatement, NOT the Rust prover (no cargo in this image): a reported baseline, not a speed-up claim"}


def cpu_baseline_full_on_record(log2_rows):
    """the committed result of `--cpu-baseline full` at this height (profiles/r*_cpu_baseline_full_2p<rows>.json: the newest), or None"""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_cpu_baseline_full_2p{log2_rows}.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
        return {"file": os.path.relpath(files[-1], ROOT), "value": rec["value"], "unit": rec["unit"], "cores": rec["cores"],
                "prove_seconds": rec["prove_seconds"], "stage_seconds": rec["stage_seconds"]}
    except (OSError, ValueError, KeyError):
        return None


def kernel_counters():
    """profiles/kernel_counters.json: per-dispatch hardware counters of the SHIPPED hot kernels (rocprofv3 --pmc in separate passes,
    tools/pmc.sh -> tools/kernel_counters.py; the file names its source profile and command)"""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_counters.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


VALU_PEAK_T = VALU_PEAK_LANE_OPS_PER_CLK_PER_CU * N_CUS * CLOCK_GHZ * 1e9 / 1e12   # T lane-ops/s, plain VALU (MI355X_MICROARCH.md)


def lde_roofline(lde_ms, n_rows, n_cols, expansion, share, counters, shape_matches):
    """`roofline` of the dominant kernel FAMILY by bytes, the main-table LDE.  The contract's figure -- algorithmic bytes (SURVEY
    8(d): 8 B read + 8 B x expansion written per trace cell) / launch time against the 8 TB/s HBM peak -- is `achieved` / `frac`.
    What bounds the kernels is stated next to it from MEASURED counters: `traffic` (fabric bytes, PMC) and the VALU rate from
    SQ_INSTS_VALU (wave instructions x 64 lanes / time against the plain-VALU peak); `bound` is the larger of the two fractions'
    resource.  `work` is the figure that cannot be raised by executing more instructions: butterflies per second."""
    cells = n_rows * n_cols
    bytes_per_cell = 8 + 8 * expansion / share
    secs = lde_ms * 1e-3
    achieved = cells * bytes_per_cell / secs / 1e9
    out = {"kernel": "main-table LDE: tvm_lde_table of 379 columns (k_lde_pass1_rows + k_lde_pass2_fused + k_lde_pass3_rows, column chunks of 96)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
           "launch_ms": round(lde_ms, 3), "algorithmic_bytes_per_launch": int(cells * bytes_per_cell), "traffic": None, "bound": "valu",
           "launch_ms_measured": "live in this run with HIP events on the context's stream (tvm_timer_*), over tvm_lde_table on a synthetic table of the "
                                 "same shape AFTER the timed proofs (hot_kernel_timings), not inside them; stage_ms_cpp_host['main LDE'] is the same "
                                 "call inside a proof"}
    log_n = n_rows.bit_length() - 1
    butterflies = cells * (1 + expansion / share) * log_n / 2          # one inverse + expansion/share forward transforms per column
    out["work"] = {"butterflies_per_s": round(butterflies / secs, 1), "unit": "radix-2 butterflies/s (log2(n)/2 per point and transform)"}
    fam = (counters or {}).get("lde") if shape_matches else None
    if fam:
        traffic = fam["hbm_bytes_per_trace_cell"] * cells
        insts = fam["wave_valu_instructions_per_trace_cell"] * cells
        hbm_counter_frac = traffic / secs / 1e9 / HBM_PEAK_GBPS
        valu_frac = insts * 64 / secs / 1e12 / VALU_PEAK_T
        out.update(traffic=int(traffic), traffic_source=fam.get("source"), traffic_source_commit=(counters or {}).get("commit"),
                   hbm={"algorithmic_gbps": round(achieved, 1), "algorithmic_frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "counter_gbps": round(traffic / secs / 1e9, 1), "counter_frac": round(hbm_counter_frac, 4),
                        "traffic_over_algorithmic": round(traffic / (cells * bytes_per_cell), 2)},
                   valu={"wave_instructions_per_launch": int(insts), "lane_ops_per_s_T": round(insts * 64 / secs / 1e12, 2), "peak_T": round(VALU_PEAK_T, 2),
                         "frac": round(valu_frac, 4), "wave_instructions_per_butterfly": round(insts * 64 / butterflies / 64, 3),
                         "lane_ops_per_layer_point": round(insts * 64 / (cells * (1 + expansion / share) * log_n), 2),
                         "source": "SQ_INSTS_VALU, " + str(fam.get("source"))},
                   bound="valu" if valu_frac >= hbm_counter_frac else "hbm",
                   bound_note="both fractions are of hardware peaks; the kernels issue carry-chain and v_mad_u64_u32 instructions at about half "
                              "the plain-VALU rate (DESIGN.md section 3), so the VALU fraction understates how close they are to their issue limit")
    return out


def hash_roofline(hash_ms, rows, n_words, counters):
    """second roofline object: main-table row hashing (k_hash_rows_mfma), the largest stage of a proof.  VALU-bound: the fraction is
    MEASURED wave VALU instructions (SQ_INSTS_VALU per row and permutation, profiles/kernel_counters.json) x 64 lanes / launch time
    against the plain-VALU peak; `work` = Tip5 permutations per second."""
    perms = n_words // 10 + 1            # absorb blocks of 10 words incl. the padding block (master_table.rs:667-716)
    secs = hash_ms * 1e-3
    out = {"kernel": "k_hash_rows_mfma (main-table row hashing)", "bound": "valu", "launch_ms": round(hash_ms, 3), "permutations_per_row": perms,
           "work": {"tip5_permutations_per_s": round(rows * perms / secs, 1)},
           "hbm": {"algorithmic_gbps": round(rows * n_words * 8 / secs / 1e9, 1), "algorithmic_frac": round(rows * n_words * 8 / secs / 1e9 / HBM_PEAK_GBPS, 4)}}
    k = (counters or {}).get("hash_rows")
    if k:
        insts = k["wave_valu_instructions_per_row_permutation"] * rows * perms
        out.update(achieved=round(insts * 64 / secs / 1e12, 2), peak=round(VALU_PEAK_T, 2), unit="T VALU lane-ops/s",
                   frac=round(insts * 64 / secs / 1e12 / VALU_PEAK_T, 4),
                   wave_valu_instructions_per_row_permutation=k["wave_valu_instructions_per_row_permutation"],
                   lds_bank_conflict_share=k.get("lds_bank_conflict_share"), source="SQ_INSTS_VALU, " + str(k.get("source")))
    return out


XGMI_EGRESS_GBPS, COLLECTIVE_LATENCY_US = 300.0, 25.0   # assumptions of the projection below: 7 xGMI links x ~153 GB/s per GPU on
#                                                          paper; 300 GB/s is what an all-to-all / all-gather is assumed to sustain


def simulate_ranks(ctx, host_lib, n_ranks, aet, padded_height, claim, kw, column_chunks=0):
    """One proof as `n_ranks` ranks of the sharded C++ prover on ONE GPU, in lockstep: every rank has its own context and
    stream, the communicators are the in-process ones (collectives = rendezvous + device-to-device copies), and between two
    collectives only one rank computes at a time -- so each rank's stage times are measured without contention, which is what
    that rank would spend on its own GPU.  Returns per-stage per-rank compute ms, the exchanged bytes, and the projection
    critical path = sum over stages of the slowest rank + bytes / assumed xGMI bandwidth.  A single-GPU measurement of the
    multi-GPU code path, NOT a multi-GPU measurement."""
    import threading

    from triton_vm_amd import native_host

    comms = native_host.LocalComms(host_lib, n_ranks, lockstep=True)
    ctxs = [type(ctx)(device=0, lib=ctx.lib) for _ in range(n_ranks)]
    out, errors = {}, []
    # ONE copy of the replicated trace-side tables for the ranks of this process (triton_host.hpp, tvmh_comm::share): rank 0 fills,
    # pads and extends, the others read its arrays -- 21.8 GB once instead of eight times at 2^22 rows, which is what lets BASELINE
    # configs[2] (2^22 rows over 8 GPUs) run through this code path on one 288 GB device.  The replicated stages then show their
    # time on rank 0 only; the slowest rank per stage -- what the projection sums -- is unchanged.
    host_lib.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 1)
    host_lib.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, column_chunks)   # > 0: the inverse transforms split by columns
    if padded_height >= 1 << 22:
        # n_ranks working sets on ONE device leave no room for 96-column chunks of intermediates per rank (6.4 GB each at 2^22 rows): with
        # them the ranks give their cached blocks back and fetch them again in every proof, and a rank's 12.7 GB table allocation took
        # 190 ms in one run (profiles/r05_q_*).  32-column chunks (TVM_OPTION_LDE_CHUNK_COLUMNS; 2 % slower extension) keep the lockstep run
        # inside the device; a real rank has a device of its own and the default chunk width.
        for c in ctxs:
            c._check(c.lib.tvm_ctx_set_option(c.handle, 2, 32), "tvm_ctx_set_option")

    def run(r, phase):
        try:
            out[(r, phase)] = native_host.prove_execution_sharded(ctxs[r], host_lib, comms.ptrs[r], aet, padded_height, claim, PROVER_SEED,
                                                                  jit_passes=1, **kw)
        except BaseException as e:   # noqa: BLE001
            errors.append((r, e))
            comms.abort()            # the other ranks leave their collectives instead of waiting for this one

    reports = []
    repeats = 3
    try:
        for phase in range(1 + repeats):   # phase 0 warms the pools of the contexts up, then `repeats` measured proofs
            threads = [threading.Thread(target=run, args=(r, phase)) for r in range(n_ranks)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise RuntimeError(f"simulated rank failed: {errors[0]}")
            reports.append(comms.report())
    finally:
        host_lib.tvmh_set_option(native_host.OPTION_SHARE_REPLICATED_TABLES, 0)
        host_lib.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, 0)
        comms.close()
        for c in ctxs:
            c.close()
    # per stage and rank: the MINIMUM over the measured proofs (the report accumulates: differences of consecutive reports).  What
    # is estimated is the rank's compute time on a GPU of its own; what disturbs it here -- the other ranks' host threads, an
    # allocation, a clock dip -- only ever adds (one box showed rank 0's first stage at 21 ms in two proofs of three, 2.5 ms
    # everywhere else).  The sum with the MEDIAN instead is reported next to it.
    per_proof = [{k: [b - a for a, b in zip(reports[i].get(k, [0.0] * n_ranks), v)] for k, v in reports[i + 1].items()} for i in range(repeats)]
    stages = {k: [round(min(pp[k][r] for pp in per_proof), 3) for r in range(n_ranks)] for k in per_proof[0]}
    median_sum = sum(max(sorted(pp[k][r] for pp in per_proof)[repeats // 2] for r in range(n_ranks)) for k in per_proof[0])
    proofs = [out[(r, 1)][0] for r in range(n_ranks)]
    exchanges = out[(0, 1)][1]["exchanges"]
    sent = sum(e["bytes_sent"] for e in exchanges.values())
    calls = sum(e["calls"] for e in exchanges.values())
    compute = sum(max(v) for v in stages.values())
    replicated = sum(max(v) for k, v in stages.items() if k in ("trace tables (fill, pad, randomizers)", "extend"))
    exchange_ms = sent / (XGMI_EGRESS_GBPS * 1e6) + calls * COLLECTIVE_LATENCY_US * 1e-3
    split = {}
    if column_chunks:
        # the column split's coefficient exchange runs on the communicator's own stream, chunk by chunk, under the extension of the
        # chunks already there: how much of it hides is what a multi-GPU box would measure -- both ends are stated
        coeff = sum(e["bytes_sent"] for k, e in exchanges.items() if k.endswith("coefficients"))
        coeff_ms = coeff / (XGMI_EGRESS_GBPS * 1e6)
        split = {"column_split": {"chunks_per_table": column_chunks, "coefficient_bytes_sent_per_rank": coeff,
                                  "coefficient_exchange_ms_at_assumed_bandwidth": round(coeff_ms, 3),
                                  "projected_ms_per_proof_exchange_fully_exposed": round(compute + exchange_ms, 3),
                                  "projected_ms_per_proof_exchange_fully_hidden": round(compute + exchange_ms - coeff_ms, 3),
                                  "note": "TVMH_OPTION_COLUMN_SPLIT: every rank interpolates its own columns, the coefficient forms are "
                                          "all-gathered (MasterTable::low_degree_extend_over); `projected_ms_per_proof` below is the fully "
                                          "exposed end"}}
    return {"ranks": n_ranks, **split, "measured_proofs": repeats, "stage_ms_per_rank": stages, "slowest_rank_sum_ms": round(compute, 3),
            "slowest_rank_sum_ms_with_median_stage_times": round(median_sum, 3),
            "replicated_stages_ms": round(replicated, 3), "exchanges_of_rank_0": exchanges, "bytes_sent_per_rank": sent, "collective_calls": calls,
            "projected_exchange_ms": round(exchange_ms, 3), "projected_ms_per_proof": round(compute + exchange_ms, 3),
            "assumptions": f"{XGMI_EGRESS_GBPS:.0f} GB/s sustained per-rank egress over xGMI, {COLLECTIVE_LATENCY_US:.0f} us per collective call",
            "all_ranks_same_proof": all((p.size == proofs[0].size and (p == proofs[0]).all()) for p in proofs),
            "replicated_tables": "one copy shared by the simulated ranks (rank 0 builds them; TVMH_OPTION_SHARE_REPLICATED_TABLES)",
            "proof": proofs[0],
            "note": "single-GPU LOCKSTEP run of the N-rank code path (one rank computes at a time, own context and stream per rank): per-rank "
                    "compute is measured, the exchanges are projected; not a multi-GPU measurement"}


def column_split_bracket(sim, single_gpu_stage_ms, n_rows):
    """north_star's OTHER sharding, priced from this run's measurements: the trace columns split over the ranks for the inverse
    transforms (rank r interpolates W/R columns, the coefficients are all-gathered, every rank evaluates all columns on ITS cosets).
    The coset sharding replicates the inverse transform of every column on every rank; its cost per rank is isolated from two
    measured LDE times of the same table -- one coset (a rank of R) and X cosets (the single GPU): T_1 = I + F, T_X = I + X F.
    The split removes (R - 1)/R of I and adds an all-gather of every coefficient: 652 words x N x 8 B x (R - 1)/R received per rank.
    Both ends of the overlap assumption are stated: the exchange fully hidden behind the forward transforms and the hashing, and
    fully exposed.  (Nothing here is built: the figures say whether it would pay -- DESIGN.md section 6.)"""
    R = sim["ranks"]
    stages = sim["stage_ms_per_rank"]
    out = {}
    try:
        inv = {}
        for name in ("main LDE", "aux LDE"):
            t1, tx = max(stages[name]), single_gpu_stage_ms[name]
            inv[name] = max(0.0, (R * t1 - tx) / (R - 1))
        saved = sum(inv.values()) * (R - 1) / R
        received = MASTER_WORDS * n_rows * 8 * (R - 1) // R
        exchange_ms = received / (XGMI_EGRESS_GBPS * 1e6)
        now = sim["slowest_rank_sum_ms"] + sim["projected_exchange_ms"]
        hidden, exposed = now - saved, now - saved + exchange_ms
        out = {"replicated_inverse_transform_ms_per_rank": {k: round(v, 3) for k, v in inv.items()}, "compute_saved_ms_per_rank": round(saved, 3),
               "coefficient_bytes_received_per_rank": int(received), "exchange_ms_at_assumed_bandwidth": round(exchange_ms, 3),
               "projected_ms_per_proof_now": round(now, 3), "projected_ms_exchange_fully_hidden": round(hidden, 3),
               "projected_ms_exchange_fully_exposed": round(exposed, 3),
               "fraction_of_exchange_that_must_be_hidden_to_break_even": round(max(0.0, min(1.0, 1 - saved / exchange_ms)), 3),
               "note": "coset sharding (the default, measured above) against the column split of the inverse transforms -- BUILT since round 5 "
                       "(TVMH_OPTION_COLUMN_SPLIT, --column-split; measured in the lockstep harness, unverified over RCCL): the split pays "
                       "only for the part of the all-gather that overlaps with compute"}
    except (KeyError, ZeroDivisionError) as e:   # noqa: BLE001 (an extra)
        out = {"error": str(e)}
    return out


def spawn_ranks(n):
    """Re-launch this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1)."""
    import socket
    import subprocess

    have = visible_devices()
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def timed_steps(step, steps, warmup, device_sync, dist=None, device="cuda"):
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    (device sync + barrier) on both sides; returns the MAX over ranks of the elapsed seconds.
    `dist` is torch.distributed (initialised) or None (tests/test_bench_distributed.py, gloo)."""
    def barrier():
        device_sync()                            # the context's stream (the kernels of the step run there)
        if dist is not None:
            if device == "cuda":
                import torch

                torch.cuda.synchronize()         # ... and torch's streams (the collectives of the sharded proof)
            dist.barrier()
            device_sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def hot_kernel_timings(ctx, params):
    """live HIP-event timing (on the context's stream) of the main-table LDE -- the `roofline` kernel family -- and of the
    main-table row hashing, on a table of the workload's shape (the work of neither depends on the contents)"""
    from triton_vm_amd.master_table import MasterTable

    n, h = params.trace.length, params.h
    mt = MasterTable.from_device(ctx, ctx.synthetic(379 * n, 1000), ctx.synthetic(379 * h, 1001), 379, n, h, params.trace, params.quotient,
                                 params.ldt, 1)
    lde_ms, hash_ms = [], []
    mt.maybe_low_degree_extend_all_columns()      # untimed: this table's shape allocates its scratch and power tables once
    for _ in range(3):
        ctx.timer_start()
        mt.maybe_low_degree_extend_all_columns()  # over the domain the prover extends in one go
        lde_ms.append(ctx.timer_stop())
    rows = mt.ldt_domain.length
    d_digests = ctx.alloc(5 * rows)
    for _ in range(3):
        ctx.timer_start()
        ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, mt._need_table(), rows, d_digests.ptr), "tvm_hash_rows")
        hash_ms.append(ctx.timer_stop())
    del d_digests
    mt.clear_cache()
    mt.d_trace.free()
    mt.d_randomizers.free()
    return sum(lde_ms) / 3, sum(hash_ms) / 3, rows


def smi_snapshot():
    """clocks / power of the GPU as rocm-smi reports them (diagnostics for a run that lands on a slow box)"""
    import subprocess

    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(out.stdout) if out.returncode == 0 else {"error": out.stderr[-300:]}
    except Exception as e:  # noqa: BLE001 (diagnostics only)
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-rows", type=int, default=20)
    ap.add_argument("--data", choices=["real", "synthetic"], default="real",
                    help="real: Prover::prove(claim, aet) on the execution trace of --program; synthetic: the hot path on random tables")
    ap.add_argument("--program", choices=["fib", "u32", "sponge", "ram"], default="fib",
                    help="real data: prove_fib (BASELINE configs[1], the default), the u32 loop (configs[3]), the sponge loop (configs[4] "
                         "with --log2-expansion 4), the RAM loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["sample", "full"], default="sample",
                    help="sample (default): 10-30 s of sampled stage work scaled to prove(); full: the port over ALL the work of one prove(), "
                         "~70 s on 16 cores at 2^20 rows (the committed result, profiles/*cpu_baseline_full*.json, is cited by the default line)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (PCIe-inclusive, synthetic hot path, STIR)")
    ap.add_argument("--trace-randomizers", type=int, default=198, help="synthetic data: Stark::default() with FRI has 198 (stark.rs:2083-2089)")
    ap.add_argument("--queries", type=int, default=173, help="synthetic data: FRI collinearity checks at 160 bits, expansion 4: 173")
    ap.add_argument("--log2-expansion", type=int, default=2, help="log2 of the LDT expansion factor: 2 (Stark::default()); 4 is "
                    "BASELINE config 5's FRI log-blowup (the quotient domain is then the short domain)")
    ap.add_argument("--ldt", choices=["fri", "stir", "auto"], default="fri",
                    help="low-degree test: fri (what BASELINE.json names), stir, or auto = Stark::ldt's rule (STIR from 2^16 rows on)")
    ap.add_argument("--memory-policy", action="store_true",
                    help="real data, single GPU: the C++ host's sharded entry with jit_passes = 0 -- the reference's policy (master_table.rs:268-271): "
                         "the cached extension, and if the device cannot hold it, coset-wise with as few passes as fit (2^23 rows)")
    ap.add_argument("--jit-passes", type=int, default=0, help="synthetic data, single GPU: evaluate the extended tables coset-wise in this "
                    "many passes (triton_vm_amd/jit.py, the reference's JIT path) instead of caching them")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent proofs per GPU instead of one sharded proof")
    ap.add_argument("--sharded", action="store_true", help="run the sharded prover (process group, collectives) even with ONE rank: the "
                    "code path of the N > 1 runs on a single-GPU box (plumbing check; the collectives are identities)")
    ap.add_argument("--split-all-trees", action="store_true", help="sharded proof: build EVERY Merkle tree split over the ranks, not only those of "
                    "at least 2^21 leaves")
    ap.add_argument("--simulate-gpus", type=int, default=0, help="single GPU: run one proof as this many ranks of the sharded C++ prover in "
                    "LOCKSTEP (one rank computes at a time, communicators between the contexts of this process) and report, per stage, what "
                    "each rank computes -- the measured critical path of an N-GPU run, without N GPUs (DESIGN.md section 6)")
    ap.add_argument("--column-split", type=int, default=0, help="sharded proof (--gpus N > 1, --simulate-gpus N): split the inverse transforms of the "
                    "table extensions by columns over the ranks and exchange the coefficients in this many chunks per table "
                    "(TVMH_OPTION_COLUMN_SPLIT; north_star's column sharding where it applies) instead of replicating them")
    ap.add_argument("--air-fork", type=int, default=-1, help="TVM_OPTION_AIR_FORK_MAX_WORKGROUPS for the run (A/B; -1: the library's default, 256; "
                    "0: the parts of the AIR never run side by side)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="K=V", help="tvm_ctx_set_option(K, V) on the context before the run (A/B of "
                    "tuning options, include/triton_hip.h: e.g. 6=0 builds the narrow Merkle levels one launch per level)")
    ap.add_argument("--host-trace", type=int, default=0, help="TVMH_OPTION_TRACE for the run: the C++ host's wall time per step of prove_execution on "
                    "stderr (1: the stream drained at every step; 2: not drained -- the host's own time)")
    ap.add_argument("--host", choices=["cpp", "python"], default="cpp",
                    help="host side that sequences the C-ABI calls of the timed step: the C++ mirror of Prover::prove "
                         "(triton_vm_amd/host/, the default where it applies: cached tables, one proof per GPU) or the "
                         "Python mirror (always used for --jit-passes and the sharded proof)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same
        # command the driver's torch.distributed.run form uses) -- and fail loudly when the node has fewer GPUs.
        sys.exit(spawn_ranks(args.gpus))
    if args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist, device = None, None
    if world > 1 or args.sharded:
        if "RANK" not in os.environ:   # --sharded without a launcher: a one-rank group on this process
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist, device = init_distributed(local_rank)
    dev_kind = "cuda" if device is None or device.type == "cuda" else "cpu"

    from triton_vm_amd.prover import Prover, StarkParameters, stark_parameters

    ctx = make_context(local_rank)
    if args.air_fork >= 0:
        ctx.air_fork_max_workgroups(args.air_fork)
    for kv in args.ctx_option:
        k, v = kv.split("=")
        ctx._check(ctx.lib.tvm_ctx_set_option(ctx.handle, int(k), int(v)), "tvm_ctx_set_option")
    sharded = (world > 1 or args.sharded) and not args.replicas
    coset_wise = bool(args.jit_passes or args.memory_policy)   # the C++ host's sharded entry with no communicator
    ldt = None if args.ldt == "auto" else args.ldt
    effective_ldt = ldt or ("fri" if args.log2_rows < 16 else "stir")
    host_lib, comm = None, None
    if args.host == "cpp" and USE_CPP_HOST:
        from triton_vm_amd import native_host

        try:
            host_lib = load_host_library()
        except Exception as e:  # no g++ on this machine: the Python mirror sequences the same C-ABI calls
            print(f"bench.py: C++ host library unavailable ({e}); timing the Python host", file=sys.stderr)
    if host_lib is not None and args.host_trace:
        host_lib.tvmh_set_option(native_host.OPTION_TRACE, args.host_trace)
    if sharded and host_lib is not None:
        host_lib.tvmh_set_option(native_host.OPTION_COLUMN_SPLIT, args.column_split)
        try:
            comm = make_comm(dist, device, rank, world, local_rank)
            ok = 1
        except Exception as e:   # noqa: BLE001 -- e.g. no RCCL headers to build libtriton_rccl.so against
            print(f"bench.py: RCCL communicator unavailable on rank {rank} ({e}); the Python sharded host (torch.distributed collectives) takes over", file=sys.stderr)
            ok = 0
        if world > 1:   # every rank must take the same path
            import torch

            flag = torch.tensor([ok], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if not ok:
            comm, host_lib = None, None
    # trees of at least 2^21 leaves are built split over the ranks (the three table trees and the first FRI rounds at 2^20 rows);
    # --split-all-trees forces every tree through the split path (plumbing check: ~14 ms of extra copies and host round trips)
    split_min = 0 if args.split_all_trees else 1 << 21
    out_extra, last = {}, {}

    if args.data == "real":
        # ---- workload generation (outside every timed region): run the program, upload the trace once ----------------------
        from oracle.vm import workload      # the oracle-side VM stands in for the reference's Rust VM

        from triton_vm_amd.master_table import aet_to_device
        from triton_vm_amd.proof_stream import Claim

        t0 = time.perf_counter()
        e = workload.execution(args.program, args.log2_rows)
        vm_s = time.perf_counter() - t0
        claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
        padded_height = e["padded_height"]
        resident = aet_to_device(ctx, e["aet"])
        kw = dict(log2_expansion=args.log2_expansion, ldt=ldt)

        def prove_from(aet, profile=False):
            if host_lib is not None and (sharded or coset_wise):
                last["proof"], last["stats"] = native_host.prove_execution_sharded(
                    ctx, host_lib, comm.ptr if comm is not None else None, aet, padded_height, claim, PROVER_SEED,
                    jit_passes=args.jit_passes or (1 if sharded else 0), split_tree_min_leaves=split_min, profile=profile, **kw)
            elif sharded:
                from triton_vm_amd.sharded import ShardedProver

                prover = ShardedProver.from_execution(ctx, dist, device, aet, padded_height, claim, PROVER_SEED, **kw)
                if args.sharded:
                    prover.split_tree_min_leaves = split_min
                last["proof"] = prover.prove().proof().words
                prover.release()
            elif host_lib is not None:
                last["proof"] = native_host.prove_execution(ctx, host_lib, aet, padded_height, claim, PROVER_SEED, **kw)
            else:
                prover = Prover.from_execution(ctx, aet, padded_height, claim, PROVER_SEED, **kw)
                last["proof"] = prover.prove().proof().words
                prover.release()

        step = lambda: prove_from(resident)
        params = stark_parameters(args.log2_rows, 160, args.log2_expansion, ldt)
        cells_per_step = padded_height * MASTER_WORDS * (1 if sharded or world == 1 else world)
        host = ("cpp (sharded)" if host_lib is not None else "python (sharded)") if sharded else ("cpp" if host_lib is not None else "python")
    else:
        ldt_s = effective_ldt
        params = StarkParameters(args.log2_rows, num_trace_randomizers=args.trace_randomizers,
                                 num_collinearity_checks=args.queries, ldt=ldt_s, log2_expansion=args.log2_expansion)
        if sharded and host_lib is not None:
            prover = Prover(ctx, params, seed=1000)          # the same tables on every rank; the C++ sharded host proves them
        elif sharded:
            from triton_vm_amd.sharded import ShardedProver

            prover = ShardedProver(ctx, params, dist, device, seed=1000)
        elif args.jit_passes and host_lib is None:
            from triton_vm_amd.jit import JitProver

            prover = JitProver(ctx, params, args.jit_passes, seed=1000 + rank)
        else:
            prover = Prover(ctx, params, seed=1000 + rank)
        cells_per_step = params.padded_height * MASTER_WORDS * (1 if sharded else world)
        step, host = prover.prove, "python"
        if host_lib is not None and (sharded or coset_wise):
            step = lambda: native_host.prove_sharded(ctx, host_lib, comm.ptr if comm is not None else None, params, prover.main.d_trace,   # noqa: E731
                                                     prover.main.d_randomizers, prover.aux.d_trace, prover.aux.d_randomizers,
                                                     prover.quotient_randomizer, jit_passes=args.jit_passes or 1, split_tree_min_leaves=split_min)
            host = "cpp (sharded)" if sharded else "cpp"
        elif host_lib is not None:
            native = native_host.NativeProver(ctx, host_lib, params, prover.main.d_trace, prover.main.d_randomizers,
                                              prover.aux.d_trace, prover.aux.d_randomizers, prover.quotient_randomizer)
            step, host = (lambda: native.prove(parse=False)), "cpp"

    elapsed = timed_steps(step, args.steps, args.warmup, ctx.sync, dist, device=dev_kind)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- after the timed region: acceptance of the last proof, the per-stage profile, the roofline kernels, the extras -------
    verified = None
    if args.data == "real" and rank == 0:
        from triton_vm_amd.verifier import Verifier

        t0 = time.perf_counter()
        accepted_at = Verifier(ctx, log2_expansion=args.log2_expansion, ldt=ldt).verify(claim, last["proof"])   # raises on rejection
        verified = {"verifier": "triton_vm_amd.verifier.Verifier (Verifier::verify, stark.rs:1388-1763)", "accepted": True,
                    "revealed_rows": len(accepted_at), "proof_words": int(last["proof"].size), "seconds": round(time.perf_counter() - t0, 2)}
    barrier()
    stage_ms, stage_wall, t_prof, rank_stats = {}, {}, 0.0, None
    cpp_stats = host_lib is not None and (sharded or coset_wise) and args.data == "real"
    if cpp_stats:   # the sharded / coset-wise C++ host times its own stages (stream drained at every stage boundary)
        t_prof = time.perf_counter()
        prove_from(resident, profile=True)
        t_prof = 1e3 * (time.perf_counter() - t_prof)
        rank_stats = [last["stats"]]
        if dist is not None and world > 1:
            rank_stats = [None] * world
            dist.all_gather_object(rank_stats, last["stats"])
        stage_ms = dict(rank_stats[0]["stage_ms"])
    elif (rank == 0 or sharded) and not (host_lib is not None and (sharded or coset_wise)):  # a sharded prove() contains collectives: every rank has to take part
        if args.data == "real":
            if sharded:
                from triton_vm_amd.sharded import ShardedProver

                prof = ShardedProver.from_execution(ctx, dist, device, resident, padded_height, claim, PROVER_SEED, **kw)
            else:
                prof = Prover.from_execution(ctx, resident, padded_height, claim, PROVER_SEED, **kw)
        else:
            prof = prover
            prof.timings, prof.wall = {}, {}
        t_prof = time.perf_counter()
        prof.prove(profile=True)
        t_prof = 1e3 * (time.perf_counter() - t_prof)
        stage_ms, stage_wall = dict(prof.timings), dict(prof.wall)
        if args.data == "real":
            prof.release()
            del prof
    barrier()
    # `stage_ms` above is the Python mirror's profile (its host work between launches sits inside the stages).  The production host's
    # own stage times of one more proof, through its sharded entry with no communicator and one pass (the same kernels in the same
    # order; the stream drained at every stage boundary): what the C++ host spends per stage.
    stage_ms_cpp = None
    if rank == 0 and world == 1 and host_lib is not None and args.data == "real" and not cpp_stats:
        try:
            _, st = native_host.prove_execution_sharded(ctx, host_lib, None, resident, padded_height, claim, PROVER_SEED, jit_passes=1,
                                                        profile=True, **kw)   # first call: allocations
            _, st = native_host.prove_execution_sharded(ctx, host_lib, None, resident, padded_height, claim, PROVER_SEED, jit_passes=1,
                                                        profile=True, **kw)
            stage_ms_cpp = {k: round(v, 3) for k, v in dict(st["stage_ms"]).items()}
        except Exception as e:  # noqa: BLE001  (reported, never fatal for the line)
            stage_ms_cpp = {"error": str(e)[:200]}
    lde_avg_ms = hash_avg_ms = None
    if rank == 0:
        passes_used = int((rank_stats or [{}])[0].get("passes", 0) or 0) if not sharded else 0   # what the memory policy settled on
        share = world if sharded else (passes_used or args.jit_passes or 1)
        kp = params
        if share > 1:   # a rank (or a coset-wise pass) extends onto its share of the rows
            from triton_vm_amd.sharded import local_domain

            kp = StarkParameters(args.log2_rows, num_trace_randomizers=params.h, log2_expansion=args.log2_expansion)
            kp.ldt = kp.quotient = local_domain(params.ldt, 0, share)
        lde_avg_ms, hash_avg_ms, hash_rows = hot_kernel_timings(ctx, kp)
    barrier()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        share = world if sharded else (passes_used or args.jit_passes or 1)
        counters = kernel_counters()
        expansion = params.ldt.length // params.trace.length
        shape_key = f"2p{args.log2_rows}_x{expansion}"
        if counters and share == 1 and shape_key in counters.get("shapes", {}):   # counters taken on this shape (tools/kernel_counters.py)
            counters = counters["shapes"][shape_key]
            shape_ok = True
        else:
            shape_ok = share == 1 and args.log2_expansion == 2 and args.log2_rows == 20   # (the default counters were taken on this shape)
        roofline = lde_roofline(lde_avg_ms, params.trace.length, 379, expansion, share, counters, shape_ok)
        if args.data == "real":
            workload_text = (f"Prover::prove(claim, aet) for real: {args.program} program run for {e['cycles']} cycles (public input "
                             f"{e['index']}), padded height 2^{args.log2_rows}, 379 main + 91 aux columns (652 words/row); Stark::default() parameters with "
                             + ("LdtChoice::Fri" if effective_ldt == "fri" else "STIR (the automatic choice at this height)")
                             + f" (expansion {2 << args.log2_expansion >> 1}, {params.h} trace randomizers); every step = fill + pad + randomizers + main LDE + "
                             "Merkle + extend + aux LDE + Merkle + AIR quotients + segments + out-of-domain rows + combination + DEEP + low-degree test + openings + "
                             "proof encoding, from a device-resident execution trace; AIR in valid-trace mode (exact on the valid trace of an "
                             "execution: the proof is word for word the exact mode's, tests/test_gpu_baseline_configs.py); the VM run and the "
                             "one-time trace upload are outside the step")
        else:
            workload_text = (f"prove() hot path on SYNTHETIC prove_fib-shaped tables: 2^{args.log2_rows} padded rows, 379 main + 91 aux columns "
                             f"(652 words/row), {effective_ldt.upper()}, expansion {2 << args.log2_expansion >> 1}, {params.h} trace randomizers; traces resident in "
                             "HBM; exact (row-by-row) AIR; fill / pad / extend not part of the step")
        out = {
            "metric": "trace-cells/sec (padded_rows x master_cols) in prove()",
            "value": round(cells_per_step * args.steps / elapsed, 1),
            "unit": "trace-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak" if (world > 1 and args.replicas) else "strong", "vs_baseline": None,
            "dtype": "u64 (F_p, p = 2^64 - 2^32 + 1, Montgomery) and its cubic extension",
            "data": "real" if args.data == "real" else "synthetic",
            "config": {"workload": workload_text,
                       "host": {"cpp": "C++ mirror of Prover::prove over the C ABI (triton_vm_amd/host/triton_host.cpp)",
                                "python": "Python mirror of Prover::prove over the C ABI (triton_vm_amd/prover.py)",
                                "python (sharded)": "Python mirror, one proof over the ranks (triton_vm_amd/sharded.py)",
                                "cpp (sharded)": "C++ host, one proof over the ranks: ShardedProver (triton_vm_amd/host/sharded_host.cpp), collectives by "
                                                 "RCCL on the context's stream (triton_vm_amd/host/rccl_comm.cpp)"}[host],
                       "padded_rows": params.padded_height, "master_words": MASTER_WORDS, "ldt": effective_ldt,
                       "ldt_domain": params.ldt.length, "parallelism": (f"one proof over {world} GPUs: coset sharding of the extended tables, all-to-all of leaf "
                                       "digests, all-gather of the quotient codeword" if sharded else f"{world} independent proofs, one per GPU" if world > 1
                                       else f"single GPU, tables evaluated coset-wise in {share} passes (nothing cached)"
                                       + (" -- chosen by the host's memory policy (jit_passes = 0: the cached path first)" if args.memory_policy and not args.jit_passes else "")
                                       if coset_wise and share > 1 else "single GPU")},
            "roofline": roofline,
            "roofline_valu": hash_roofline(hash_avg_ms, hash_rows, 379, counters),
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "stage_wall_ms": {k: round(v, 3) for k, v in stage_wall.items()},
            **({"stage_ms_cpp_host": stage_ms_cpp} if stage_ms_cpp is not None else {}),
            "profiled_prove_wall_ms": round(t_prof, 3),
        }
        if verified is not None:
            out["verified"] = verified
        if args.memory_policy and not sharded:
            out["memory_policy"] = {"entry": "tvmh_prove_execution_sharded(comm = NULL, jit_passes = 0)", "passes": share,
                                    "note": "every proof starts with the cached path and, where the device cannot hold it, starts over coset-wise "
                                            "(master_table.rs:268-271, stark.rs:730-768); the failed attempts are inside ms_per_step"}
        if rank_stats is not None:   # per rank: stage times of one profiled proof and the bytes each collective sent
            out["ranks"] = rank_stats
        if args.data == "real":
            out["workload_generation"] = {"vm_seconds": round(vm_s, 2), "cycles": e["cycles"], "table_heights": e["table_heights"],
                                          "note": "oracle-side VM (oracle/vm), outside the timed region; the product path starts at Prover::prove(claim, aet)"}
        if stage_ms.get("AIR quotients", 0.0) > (80.0 if args.log2_rows == 20 and world == 1 else 1e9):
            # the unexplained slow mode of the AIR kernels seen on 2 of ~40 boxes in round 2 (HISTORY.md 5.1): leave evidence
            out["air_slow_mode"] = {"stage_ms": stage_ms["AIR quotients"], "smi": smi_snapshot()}
        extras = world == 1 and not sharded and not coset_wise and not args.no_extras
        if extras and args.data == "real":
            # (1) the same step with the execution trace in host memory (what a host that keeps the AET in RAM pays)
            t = timed_steps(lambda: prove_from(e["aet"]), 3, 1, ctx.sync)
            out["pcie_inclusive"] = {"ms_per_step": round(1e3 * t / 3, 3), "value": round(cells_per_step * 3 / t, 1), "unit": "trace-cells/s",
                                     "note": "execution trace handed over as host arrays: its upload (processor trace 327 MB at 2^20 cycles) is inside the step"}
            # (2) the reference-default low-degree test beside the headline's (Stark::default() picks STIR from 2^16 rows on)
            other = "stir" if effective_ldt == "fri" else "fri"
            if args.log2_rows >= 16:
                okw = dict(log2_expansion=args.log2_expansion, ldt=other)
                ostep = (lambda: native_host.prove_execution(ctx, host_lib, resident, padded_height, claim, PROVER_SEED, **okw)) if host_lib is not None else \
                    (lambda: Prover.from_execution(ctx, resident, padded_height, claim, PROVER_SEED, **okw).prove())
                t = timed_steps(ostep, 2, 1, ctx.sync)
                out["reference_default_ldt" if other == "stir" else "with_ldt_choice_fri"] = {
                    "ldt": other, "ms_per_step": round(1e3 * t / 2, 3), "value": round(cells_per_step * 2 / t, 1), "unit": "trace-cells/s", "host": host}
            # (2b) the reference's own formulation of the AIR on the REAL trace: every constraint on every point of the quotient domain
            if host_lib is not None:
                host_lib.tvmh_set_option(native_host.OPTION_EXACT_AIR, 1)
                try:
                    t = timed_steps(lambda: prove_from(resident), 3, 1, ctx.sync)
                    exact_proof = last["proof"]
                finally:
                    host_lib.tvmh_set_option(native_host.OPTION_EXACT_AIR, 0)
                prove_from(resident)
                out["exact_air_real"] = {"ms_per_step": round(1e3 * t / 3, 3), "value": round(cells_per_step * 3 / t, 1), "unit": "trace-cells/s",
                                         "same_proof_as_valid_trace_mode": bool(exact_proof.size == last["proof"].size and (exact_proof == last["proof"]).all()),
                                         "note": "the same step with the AIR evaluated row by row on every point of the quotient domain "
                                                 "(master_table.rs:1264-1363), not in valid-trace mode"}
            # (3) rounds 1-2's headline: the hot path alone on synthetic tables resident in HBM, exact AIR
            sp = stark_parameters(args.log2_rows, 160, args.log2_expansion, "fri")
            syn = Prover(ctx, sp, seed=1000)
            sstep = syn.prove
            if host_lib is not None:
                snat = native_host.NativeProver(ctx, host_lib, sp, syn.main.d_trace, syn.main.d_randomizers, syn.aux.d_trace,
                                                syn.aux.d_randomizers, syn.quotient_randomizer)
                sstep = lambda: snat.prove(parse=False)
            t = timed_steps(sstep, 3, 1, ctx.sync)
            out["synthetic_hot_path"] = {"ms_per_step": round(1e3 * t / 3, 3), "value": round(cells_per_step * 3 / t, 1), "unit": "trace-cells/s",
                                         "note": "random tables resident in HBM, FRI, exact row-by-row AIR, no fill/pad/extend: the timed step of rounds 1-2"}
            syn.release()
        # (2c) the multi-GPU code path, rank by rank in lockstep on this one GPU
        n_sim = args.simulate_gpus or (8 if extras and args.log2_rows <= 20 and args.log2_expansion == 2 and effective_ldt == "fri" else 0)
        if host_lib is not None and n_sim > 1 and args.data == "real" and world == 1 and not sharded:
            ctx.trim()   # (the simulated ranks need the pool's cached blocks)
            try:
                sim = simulate_ranks(ctx, host_lib, n_sim, resident, padded_height, claim, kw, args.column_split)
                sim["same_proof_as_single_gpu"] = bool(sim["proof"].size == last["proof"].size and (sim.pop("proof") == last["proof"]).all())
                sim.pop("proof", None)
                if not args.column_split:
                    sim["column_split_bracket"] = column_split_bracket(sim, stage_ms, params.trace.length)
                out["simulated_multi_gpu"] = sim
            except Exception as err:   # noqa: BLE001 (an extra: never lose the headline to it)
                out["simulated_multi_gpu"] = {"error": str(err)[:400]}
        if extras and args.data == "synthetic":
            # TVM_OPTION_AIR_VALID_TRACE on the same synthetic tables (the work does not depend on the contents)
            ctx.assume_valid_trace(True)
            t = timed_steps(step, 3, 1, ctx.sync)
            ctx.assume_valid_trace(False)
            out["valid_trace_mode"] = {"option": "TVM_OPTION_AIR_VALID_TRACE", "ms_per_step": round(1e3 * t / 3, 3),
                                       "value": round(cells_per_step * 3 / t, 1), "unit": "trace-cells/s"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_full if args.cpu_baseline == "full" else cpu_baseline)(args.log2_rows, args.log2_expansion)
            if args.cpu_baseline != "full":
                out["cpu_baseline"]["full_run_on_record"] = cpu_baseline_full_on_record(args.log2_rows)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
