"""The reference's proof-digest snapshot (`current_proof_version_is_still_current`,
/root/reference/triton-vm/src/proof.rs:200-226): Tip5::hash of the proof of a fixed program under a fixed prover seed.
A digest over EVERY word of the proof: it pins trace generation, padding, extension, randomization, the low-degree
extension, every Merkle tree, the AIR quotients, the out-of-domain rows, the DEEP combination, FRI and the transcript.

  * the oracle (oracle/real_prover.py: oracle VM + C oracle + restated transcript) reproduces it -- so the oracle is the
    reference's prover, bit for bit, on this execution;
  * the product (Prover.from_execution over the C ABI: fill, pad, extend, LDE, hashing, AIR, FRI on the device)
    reproduces it -- the device path emits the reference's proof.
"""
import numpy as np
import pytest

from tests import vm_fixture as vf
from tests.test_fill import aet_arrays

# proof.rs:218-225
SNAPSHOT = [2390426207231576512, 11357322246033024133, 15595568858844533957, 10807389618517394866, 11786266879565336160]
SEED_U64 = 4742841043836029231   # proof.rs:212


def prover_seed(seed_u64):
    """`StdRng::seed_from_u64(..).random::<[u8; 32]>()`: one u32 draw per byte"""
    from oracle import ref_rng

    rng = ref_rng.StdRng.seed_from_u64(seed_u64)
    return bytes(rng.next_u32() & 0xFF for _ in range(32))


def test_oracle_prover_reproduces_the_reference_proof_digest():
    from oracle import real_prover

    program, _, public_input, _ = vf.run("tiny")
    proof = real_prover.prove(program, public_input, seed_u64=SEED_U64)
    assert proof["params"] == dict(n=512, h=198, ldt=4096, checks=173, fri_rounds=2)
    assert proof["digest"] == SNAPSHOT


def test_device_prover_reproduces_the_reference_proof_digest(ctx, orc):
    from triton_vm_amd.prover import Claim, Prover

    program, aet, public_input, output = vf.run("tiny")
    digest = orc.hash_varlen(orc.to_mont(np.array(program.to_bwords(), dtype=object)))
    claim = Claim(digest, orc.to_mont(np.array(public_input, dtype=object)), orc.to_mont(np.array(output, dtype=object)) if output else ())
    prover = Prover.from_execution(ctx, aet_arrays(orc, aet), aet.padded_height(), claim, prover_seed(SEED_U64))
    proof = prover.prove().proof()
    assert proof.digest(ctx.lib) == SNAPSHOT
    if ctx.kind == "emu":
        from oracle import real_prover

        want = real_prover.prove(program, public_input, seed_u64=SEED_U64)["proof"]
        assert [int(v) for v in orc.from_mont(proof.words)] == want
