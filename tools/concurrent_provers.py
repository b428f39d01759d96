#!/usr/bin/env python3
"""Proofs of short traces are latency: a few hundred dependent launches and ~60 round trips that leave the chip almost empty
(DESIGN.md 4.5).  The backend's contract is one context per proving thread (triton_vm::prove may run on several threads,
/root/reference/triton-vm/src/lib.rs:522-532): this measures what K threads, each with its own context and its own device-resident
trace, prove per second on ONE GPU -- the same prove_fib instance and seed everywhere, every proof compared with the first one.
Usage: python tools/concurrent_provers.py [log2_rows=10] [proofs_per_thread=40] [threads=1,2,4,8,16] [start_at]  -> one JSON line
start_at (epoch seconds): every run's timed loop starts no earlier than that -- several PROCESSES of this script started together
(each with ONE thread count) measure what P processes x K threads prove at once (tools/r06_job21.sh adds their lines up)."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(log2_rows=10, per_thread=40, thread_counts="1,2,4,8,16", start_at="0"):
    import numpy as np
    import torch  # noqa: F401  (first: the ROCm runtime torch ships)

    from oracle.vm import workload   # the oracle-side VM stands in for the reference's VM: workload generation, outside every timed region
    from triton_vm_amd import Context, native_host
    from triton_vm_amd.master_table import aet_to_device
    from triton_vm_amd.proof_stream import Claim

    log2_rows, per_thread = int(log2_rows), int(per_thread)
    counts = [int(k) for k in str(thread_counts).split(",")]
    seed = bytes(range(32))
    e = workload.execution("fib", log2_rows)
    claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
    host_lib = native_host.load_host_library()
    n_ctx = max(counts)
    ctxs = [Context(device=0) for _ in range(n_ctx)]
    aets = [aet_to_device(c, e["aet"]) for c in ctxs]

    def prove(k):
        return native_host.prove_execution(ctxs[k], host_lib, aets[k], e["padded_height"], claim, seed, ldt="fri")

    reference = prove(0)
    for k in range(n_ctx):          # warm every context (pool, tables, fork lanes)
        for _ in range(3):
            assert np.array_equal(prove(k), reference)
    out = {"workload": f"prove_fib, padded height 2^{log2_rows}, FRI, one GPU, one context + device-resident trace per thread",
           "proofs_per_thread": per_thread, "runs": []}
    for K in counts:
        start, errors = threading.Barrier(K + 1), []

        def work(k):
            try:
                start.wait()
                for _ in range(per_thread):
                    if not np.array_equal(prove(k), reference):
                        raise RuntimeError("a concurrent proof differs from the solo proof")
            except Exception as ex:  # noqa: BLE001
                errors.append(repr(ex))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in threads:
            t.start()
        while time.time() < float(start_at):
            time.sleep(0.001)
        start.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        if errors:
            raise SystemExit(f"{K} threads: {errors[0]}")
        out["runs"].append({"threads": K, "window": [round(time.time() - dt, 3), round(time.time(), 3)], "proofs_per_s": round(K * per_thread / dt, 1), "ms_per_proof_per_thread": round(1e3 * dt / per_thread, 3),
                            "trace_cells_per_s": round(K * per_thread * e["padded_height"] * 652 / dt, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
