#!/usr/bin/env python3
"""bench.py -- the hot path of Prover::prove on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-rows 20]

One "step" = one complete pass of the prover's hot path (the C++ host triton_vm_amd/host/triton_host.cpp, or the
Python mirror triton_vm_amd/prover.py with --host python; both sequence the same C-ABI calls: main LDE, hashing +
Merkle, aux LDE, hashing + Merkle, AIR/quotients, quotient segments, out-of-domain rows, linear
combination, DEEP, FRI, openings) over synthetic padded trace tables that are already resident in
HBM: 2^20 padded rows x (379 main + 91 aux columns = 652 base-field words), the shape of
BASELINE.json configs[1] (`prove_fib` at 2^20 rows, Stark::default() with FRI, expansion 4,
198 trace randomizers).  metric = padded_rows * 652 / seconds per step, summed over ranks.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).  Default: ONE proof split over
the N GPUs by cosets of the trace domain (triton_vm_amd/sharded.py; N must divide 8): all-gather of leaf digests
and of the quotient codeword -- total work fixed, "scaling": "strong".  --replicas: every rank proves its own
instance instead (no data-path collective, "scaling": "weak").  Barrier + max-over-ranks timing either way.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MASTER_WORDS = 652
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
LDE_ALGORITHMIC_BYTES_PER_CELL = 72   # read 8 B, write 8 * (L/N = 8) B per base-field trace cell (default expansion)


def cpu_baseline(log2_rows):
    """The oracle ("port" of the reference's algorithms, oracle/tvm_oracle.c, OpenMP over columns/rows)
    timed on a bounded sample of the same workload: an 8-column slice of the main table at the full
    height -- LDE onto the 8x domain, Tip5 row hashing, Merkle tree."""
    import numpy as np

    from oracle import oracle as orc

    cols, h = 8, 198
    n = 1 << log2_rows
    rng = np.random.default_rng(1)
    trace = orc.random_elements(rng, (cols, n))
    rnd = orc.random_elements(rng, (cols, h))
    ev = orc.domain_of_length(8 * n, offset=orc.lib().orc_bfe_generator())
    t0 = time.perf_counter()
    table = orc.lde_table(trace, rnd, ev, 1)
    digests = orc.hash_rows(table)
    orc.merkle_tree(digests)
    dt = time.perf_counter() - t0
    return {"value": round(n * cols / dt, 1), "unit": "trace-cells/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"oracle (C, OpenMP) LDE + Tip5 row hashing + Merkle tree of an {cols}-column main-table slice "
                      f"at 2^{log2_rows} rows (8x extension), {dt:.1f} s; AIR/DEEP/FRI not included in the sample; a textbook "
                      "restatement, NOT the Rust prover (no cargo in this image): do not use as a speed-up ratio"}


def valu_roofline(launch_ms, rows, n_words):
    """VALU-issue roofline of the row-hashing kernel (k_hash_rows_mfma), the kernel furthest from the HBM roofline by
    time.  profiles/valu_counts.json (tools/valu_static_count.py) holds the kernel's static wave-level VALU instruction
    count per row and permutation and the issue cycles those instructions cost on a SIMD according to the measured
    per-class issue rates (profiles/r02_valu_rates_microbench.txt; nominal: plain 32-bit VOP1/VOP2 ops 2, carry-out /
    VOP3 ops 4, v_mad_u64_u32 5.2 cycles per wave64 instruction).  peak = the SIMD cycles the chip has in the launch time
    (256 CUs x 4 SIMDs x 2.4 GHz); achieved = the modelled issue cycles of the kernel's VALU work: frac is the share of
    all SIMD cycles spent issuing this kernel's VALU instructions."""
    perms = n_words // 10 + 1            # absorb blocks of 10 words incl. the padding block (master_table.rs:667-716)
    try:
        with open(os.path.join(ROOT, "profiles", "valu_counts.json")) as f:
            k = json.load(f)["k_hash_rows_mfma"]
    except (OSError, KeyError, ValueError):
        return None
    instr = k["wave_valu_instructions_per_row_permutation"] * rows * perms
    cycles = k["modelled_valu_issue_cycles_per_row_permutation"] * rows * perms
    peak = 256 * 4 * 2.4                  # G SIMD-cycles per second
    achieved = cycles / (launch_ms * 1e-3) / 1e9
    return {"bound": "valu", "kernel": "k_hash_rows_mfma (main-table row hashing)", "achieved": round(achieved, 1),
            "peak": round(peak, 1), "unit": "G SIMD issue cycles/s", "frac": round(achieved / peak, 4),
            "launch_ms": round(launch_ms, 3), "permutations_per_row": perms,
            "wave_valu_instructions_per_row": round(k["wave_valu_instructions_per_row_permutation"] * perms, 1),
            "wave_valu_instructions_per_launch": int(instr),
            "wave_valu_instructions_per_s_G": round(instr / (launch_ms * 1e-3) / 1e9, 1),
            "lane_ops_per_clk_per_cu": round(instr * 64 / (launch_ms * 1e-3) / (256 * 2.4e9), 1),
            "hbm_frac": round(rows * n_words * 8 / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}


def spawn_ranks(n):
    """Re-launch this script as n ranks under torch.distributed.run (rendezvous on 127.0.0.1)."""
    import socket
    import subprocess

    if os.environ.get("TVM_BENCH_TEST_EMU") != "1":
        import torch

        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible", file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def timed_steps(step, steps, warmup, device_sync, dist=None, device="cuda"):
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by
    (device sync + barrier) on both sides; returns the MAX over ranks of the elapsed seconds.
    `dist` is torch.distributed (initialised) or None; ranks run independent proofs, so the barrier and the
    max-reduction are the only collectives of the N > 1 path (tests/test_bench_distributed.py, gloo)."""
    def barrier():
        device_sync()                            # the context's stream (the kernels of the step run there)
        if dist is not None:
            if device == "cuda":
                import torch

                torch.cuda.synchronize()         # ... and torch's streams (the all-gathers of the sharded proof)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-rows", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-ldt", action="store_true", help="skip the extra measurement of the reference-default LDT (STIR)")
    ap.add_argument("--trace-randomizers", type=int, default=198, help="Stark::default() with FRI: 198 (stark.rs:2083-2089)")
    ap.add_argument("--queries", type=int, default=173, help="FRI collinearity checks at 160 bits, expansion 4: 173")
    ap.add_argument("--log2-expansion", type=int, default=2, help="log2 of the LDT expansion factor: 2 (Stark::default()); 4 is "
                    "BASELINE config 5's FRI log-blowup (the quotient domain is then the short domain)")
    ap.add_argument("--ldt", choices=["fri", "stir"], default="fri",
                    help="low-degree test: fri (what BASELINE.json names) or stir (the reference's default from 2^16 rows on)")
    ap.add_argument("--jit-passes", type=int, default=0, help="single GPU: evaluate the extended tables coset-wise in this many "
                    "passes (triton_vm_amd/jit.py, the reference's JIT path) instead of caching them")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent proofs per GPU instead of one sharded proof")
    ap.add_argument("--host", choices=["cpp", "python"], default="cpp",
                    help="host side that sequences the C-ABI calls of the timed step: the C++ mirror of Prover::prove "
                         "(triton_vm_amd/host/, the default where it applies: FRI, cached tables, one proof per GPU) or the "
                         "Python mirror (always used for --ldt stir, --jit-passes and the sharded proof)")
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
    # TEST-ONLY switch (tests/test_bench_distributed.py): run this very script on CPU with gloo and the fiber
    # emulation of the kernels, to exercise the multi-process control flow without GPUs.  Never set in production.
    test_emu = os.environ.get("TVM_BENCH_TEST_EMU") == "1"
    dist, device = None, None
    if world > 1:
        import torch
        import torch.distributed as dist

        if test_emu:
            device = torch.device("cpu")
            dist.init_process_group(backend="gloo")
        else:
            device = torch.device("cuda", local_rank)
            torch.cuda.set_device(local_rank)  # torch first, then the Context (triton_vm_amd/sharded.py)
            dist.init_process_group(backend="nccl", device_id=device)

    from triton_vm_amd import Context
    from triton_vm_amd.prover import Prover, StarkParameters

    if test_emu:
        from tests.emu_fixture import emu_context

        ctx = emu_context()
    else:
        ctx = Context(device=local_rank)
    params = StarkParameters(args.log2_rows, num_trace_randomizers=args.trace_randomizers,
                             num_collinearity_checks=args.queries, ldt=args.ldt, log2_expansion=args.log2_expansion)
    sharded = world > 1 and not args.replicas
    if sharded:
        from triton_vm_amd.sharded import ShardedProver

        prover = ShardedProver(ctx, params, dist, device, seed=1000)
    elif args.jit_passes:
        from triton_vm_amd.jit import JitProver

        prover = JitProver(ctx, params, args.jit_passes, seed=1000 + rank)
    else:
        prover = Prover(ctx, params, seed=1000 + rank)
    cells_per_step = params.padded_height * MASTER_WORDS * (1 if sharded else world)

    step, host = prover.prove, "python"
    if args.host == "cpp" and not sharded and not args.jit_passes and not test_emu:
        from triton_vm_amd import native_host

        try:
            host_lib = native_host.load_host_library()
        except Exception as e:  # no g++ on this machine: the Python mirror sequences the same C-ABI calls
            print(f"bench.py: C++ host library unavailable ({e}); timing the Python host", file=sys.stderr)
            host_lib = None
        if host_lib is not None:
            native = native_host.NativeProver(ctx, host_lib, params, prover.main.d_trace, prover.main.d_randomizers,
                                              prover.aux.d_trace, prover.aux.d_randomizers, prover.quotient_randomizer)
            step, host = (lambda: native.prove(parse=False)), "cpp"
    elapsed = timed_steps(step, args.steps, args.warmup, ctx.sync, dist, device="cpu" if test_emu else "cuda")

    # live timing of the dominant HBM-bound kernel family (the main-table LDE: k_ntt2_pass1, k_lde_pass2,
    # k_lde_pass3) with HIP events on the context's stream, and a per-stage breakdown of one more pass
    lde_ms = []
    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    if rank == 0:
        for _ in range(3):
            ctx.timer_start()
            prover.main.maybe_low_degree_extend_all_columns()  # over the domain the prover extends in one go
            lde_ms.append(ctx.timer_stop())
        # the kernel furthest from the HBM roofline by time: main-table row hashing (k_hash_rows_mfma), VALU-issue bound
        hash_ms = []
        d_digests = ctx.alloc(5 * params.ldt.length)
        for _ in range(3):
            ctx.timer_start()
            ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, prover.main._need_table(), prover.main.ldt_domain.length, d_digests.ptr),
                       "tvm_hash_rows")
            hash_ms.append(ctx.timer_stop())
        del d_digests
        prover.main.clear_cache()
    barrier()
    t_prof = 0.0
    if rank == 0 or sharded:  # a sharded prove() contains collectives: every rank has to take part
        prover.timings, prover.wall = {}, {}
        t_prof = time.perf_counter()
        prover.prove(profile=True)
        t_prof = 1e3 * (time.perf_counter() - t_prof)
    barrier()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        lde_avg_ms = sum(lde_ms) / len(lde_ms)
        lde_cells = params.trace.length * 379
        # algorithmic bytes of one launch: read the trace, write this rank's share of the extended rows
        share = world if sharded else (args.jit_passes or 1)
        lde_bytes_per_cell = 8 + 8 * (params.ldt.length // params.trace.length) / share  # read once, write L/N values
        achieved = lde_cells * lde_bytes_per_cell / (lde_avg_ms * 1e-3) / 1e9
        traffic = None  # fabric-side bytes per LDE launch family, from the committed PMC run (profiles/lde_traffic.json)
        try:
            with open(os.path.join(ROOT, "profiles", "lde_traffic.json")) as f:
                # (measured for the default expansion only)
                traffic = int(json.load(f)["hbm_bytes_per_trace_cell"] * lde_cells) if share == 1 and args.log2_expansion == 2 else None
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "trace-cells/sec (padded_rows x master_cols) in prove()",
            "value": round(cells_per_step * args.steps / elapsed, 1),
            "unit": "trace-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u64 (F_p, p = 2^64 - 2^32 + 1, Montgomery) and its cubic extension",
            "data": "synthetic",
            "config": {"workload": f"prove() hot path, prove_fib-shaped tables: 2^{args.log2_rows} padded rows, 379 main + "
                                   "91 aux columns (652 words/row), Stark::default() with "
                                   + (f"FRI (expansion {1 << args.log2_expansion}, {params.h} trace randomizers, {args.queries} queries)" if args.ldt == "fri"
                                      else f"STIR (expansion {1 << args.log2_expansion}, {params.h} trace randomizers, {len(params.stir.round_queries)} full rounds)")
                                   + ", traces resident in HBM; the reference's transcript (ProofItem encoding, Fiat-Shamir) on the host; "
                                   "the host `gen` steps (VM, fill, pad, extend) are not part of the path",
                       "host": ("C++ mirror of Prover::prove over the C ABI (triton_vm_amd/host/triton_host.cpp)" if host == "cpp"
                                else "Python mirror of Prover::prove over the C ABI (triton_vm_amd/prover.py)"),
                       "padded_rows": params.padded_height, "master_words": MASTER_WORDS,
                       "ldt_domain": params.ldt.length, "parallelism": (f"one proof over {world} GPUs: coset sharding of the extended tables, all-gather of digests and "
                                       "quotient codeword" if sharded else f"{world} independent proofs, one per GPU" if world > 1
                                       else f"single GPU, tables evaluated coset-wise in {args.jit_passes} passes (nothing cached)"
                                       if args.jit_passes else "single GPU")},
            "roofline": {"bound": "hbm", "kernel": "main-table LDE (k_ntt2_pass1 + k_lde_pass2 + k_lde_pass3, 4 column chunks of 96)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "launch_ms": round(lde_avg_ms, 3),
                         "algorithmic_bytes_per_launch": int(lde_cells * lde_bytes_per_cell)},
            "roofline_valu": valu_roofline(sum(hash_ms) / len(hash_ms), prover.main.ldt_domain.length, 379),
            "stage_ms": {k: round(v, 3) for k, v in prover.timings.items()},
            "stage_wall_ms": {k: round(v, 3) for k, v in prover.wall.items()},
            "profiled_prove_wall_ms": round(t_prof, 3),
        }
        if world == 1 and not args.jit_passes:
            # TVM_OPTION_AIR_VALID_TRACE: what a host that feeds real (valid) execution traces would switch on.  The
            # quotient codeword is then bit-identical to the row-by-row evaluation only on VALID traces; this bench's
            # tables are synthetic, so `value` above is measured with the option OFF (identical to the reference on any
            # input) and the option's timing -- the work does not depend on the table contents -- is reported beside it.
            ctx.assume_valid_trace(True)
            t = timed_steps(step, 3, 1, ctx.sync)
            ctx.assume_valid_trace(False)
            out["valid_trace_mode"] = {"option": "TVM_OPTION_AIR_VALID_TRACE", "ms_per_step": round(1e3 * t / 3, 3),
                                       "value": round(cells_per_step * 3 / t, 1), "unit": "trace-cells/s",
                                       "note": "consistency/transition constraints on half of the quotient domain + interpolation; "
                                               "exact on valid traces only, hence not the headline"}
        if world == 1 and args.ldt == "fri" and not args.jit_passes and not args.no_default_ldt and args.log2_rows >= 16:
            # Stark::default() selects STIR from 2^16 padded rows on (stark.rs:1944-1951); BASELINE.json's configs name
            # FRI, which is what `value` is quoted on.  The reference-default variant is measured beside it.
            prover.release()
            sp = StarkParameters(args.log2_rows, ldt="stir", log2_expansion=args.log2_expansion)
            stir = Prover(ctx, sp, seed=1000)
            stir_step, stir_host = stir.prove, "python"
            if host == "cpp":
                native_stir = native_host.NativeProver(ctx, host_lib, sp, stir.main.d_trace, stir.main.d_randomizers, stir.aux.d_trace,
                                                       stir.aux.d_randomizers, stir.quotient_randomizer)
                stir_step, stir_host = (lambda: native_stir.prove(parse=False)), "cpp"
            t = timed_steps(stir_step, 2, 1, ctx.sync)
            out["reference_default_ldt"] = {"ldt": "stir", "trace_randomizers": sp.h, "ms_per_step": round(1e3 * t / 2, 3),
                                            "value": round(sp.padded_height * MASTER_WORDS * 2 / t, 1), "unit": "trace-cells/s",
                                            "host": stir_host}
            stir.release()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.log2_rows)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
