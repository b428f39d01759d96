#!/usr/bin/env python3
"""Writes tests/golden/stir_regression_digests.json: Tip5::hash of the proofs of the reference's two snapshot programs
(proof.rs:200-226, stark.rs:2434-2460: same programs, claims and prover seeds as tests/test_proof_snapshot.py) with
LdtChoice::Stir forced, as the device path produces them.  REGRESSION values (the reference holds no STIR vector): they freeze
the STIR proofs of the commit that generated them; both restated verifiers accepted those proofs.

    python tests/golden/make_stir_regression_digests.py [emu|gpu] [out.json]     (gpu: on an MI355X; emu: the tiny program only)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: tests/conftest.py)


def main(kind="gpu", out=None):
    from oracle import oracle as orc
    from tests import test_proof_snapshot as snap
    from tests import vm_fixture as vf
    from triton_vm_amd.verifier import Verifier

    out = out or os.path.join(ROOT, "tests", "golden", "stir_regression_digests.json")
    if kind == "emu":
        from tests.emu_fixture import emu_context

        ctx = emu_context()
    else:
        from triton_vm_amd import Context

        ctx = Context(device=0)
    known = json.load(open(out)) if os.path.exists(out) else {}
    for which, seed, level in (("tiny", snap.SEED_U64, 160), ("every", snap.SEED_U64_EVERY, 32)):
        if kind == "emu" and which == "every":
            continue
        program, aet, public_input, output = vf.run(which)
        proof = snap.device_proof(ctx, orc, which, seed, level, ldt="stir")
        Verifier(ctx, ldt="stir", security_level=level).verify(snap.claim_of(orc, program, public_input, output), proof.words)
        known[which] = {"digest": [int(w) for w in proof.digest(ctx.lib)], "proof_words": int(proof.words.size), "security_level": level,
                        "generated_on": kind}
    with open(out, "w") as f:
        json.dump(known, f, indent=1)
    print(json.dumps(known))


if __name__ == "__main__":
    main(*sys.argv[1:])
