#!/usr/bin/env python3
"""Writes tests/golden/oracle_proof_digests.json: Tip5::hash(proof) -- and the proof's length -- of the ORACLE prover's proofs
(oracle/real_prover.py: the prover that reproduces both proof digests the reference holds, proof.rs:200-226 and stark.rs:2434-2460)
of the loop programs of BASELINE.json at padded heights where running it inside a test would take minutes: 2^16 ... 2^18 rows, FRI and
-- at 2^16, where the reference switches to it by default -- STIR.  CPU only; nothing of the product takes part except the STIR
parameter arithmetic (triton_vm_amd/low_degree_test.py, pinned by the reference's constants).  The GPU suite
(tests/test_wider_pins.py) requires the DEVICE proof of the same (program, input, seed) to hash to these digests: a digest over every
word of the proof, at heights whose low-degree extension runs the 256- / 512- / 1024-point row kernels.

    python tests/golden/make_oracle_proof_digests.py [kind:log2_rows:ldt ...]      (default: the four cases below; ~1 h on 8 cores)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
DEFAULT = ["fib:16:fri", "fib:16:stir", "fib:17:fri", "fib:18:fri"]


def main(cases):
    from oracle import real_prover
    from oracle.vm import workload
    from tests import test_proof_snapshot as snap
    from tests.test_wider_pins import stir_numbers

    out = os.path.join(ROOT, "tests", "golden", "oracle_proof_digests.json")
    try:
        with open(out) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        rec = {}
    for case in cases:
        kind, log2_rows, ldt = case.split(":")
        t0 = time.time()
        e = workload.execution(kind, int(log2_rows))
        stir = stir_numbers(e["padded_height"], 160) if ldt == "stir" else None
        spill = None
        if int(log2_rows) >= 19:   # the extended tables as files (oracle/real_prover.py: spill_dir)
            import tempfile

            spill = tempfile.mkdtemp(prefix="oracle_tables_", dir=os.environ.get("TVM_ORACLE_SPILL_DIR", "/tmp"))
        proof = real_prover.prove(e["program"], [e["index"]], seed_u64=snap.SEED_U64, stir=stir, spill_dir=spill)
        if spill:
            import shutil

            shutil.rmtree(spill, ignore_errors=True)
        rec[case] = {"program": kind, "log2_padded_height": int(log2_rows), "ldt": ldt, "public_input": [int(e["index"])],
                     "seed_u64": snap.SEED_U64, "security_level": 160, "digest": [int(v) for v in proof["digest"]],
                     "proof_words": len(proof["proof"]), "oracle_seconds": round(time.time() - t0, 1)}
        with open(out, "w") as f:
            json.dump(rec, f, indent=1)
        print(case, rec[case], flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or DEFAULT)
