"""triton_vm::prove may run on several threads at once (/root/reference/triton-vm/src/lib.rs:522-532): the
backend's contract is one context per proving thread, each with its own stream, pool and tables.  Two threads prove
different instances concurrently on one GPU; each proof must equal the one the same instance produces alone."""
import threading

import numpy as np
import pytest


@pytest.mark.gpu
def test_two_threads_prove_concurrently_on_one_gpu():
    import torch  # noqa: F401  (first: tests/conftest.py)

    from triton_vm_amd import Context
    from triton_vm_amd.prover import Prover, StarkParameters

    def proof_words(ctx, seed, log2_rows):
        prover = Prover(ctx, StarkParameters(log2_rows), seed=seed)
        words = np.array(prover.prove().proof().words)
        prover.release()
        return words

    jobs = [(11, 12), (12, 13)]                      # (seed, log2 padded height): different shapes on purpose
    solo = []
    for seed, log2_rows in jobs:
        ctx = Context(device=0)
        solo.append(proof_words(ctx, seed, log2_rows))
        ctx.close()

    for attempt in range(3):
        results, errors = [None] * len(jobs), []
        start = threading.Barrier(len(jobs))

        def work(k):
            try:
                ctx = Context(device=0)              # ctypes releases the GIL inside every C-ABI call
                start.wait()
                results[k] = proof_words(ctx, *jobs[k])
                ctx.close()
            except Exception as e:                   # noqa: BLE001 -- reported below, in the main thread
                errors.append((k, repr(e)))

        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k in range(len(jobs)):
            assert results[k].size == solo[k].size and (results[k] == solo[k]).all(), (attempt, k)
