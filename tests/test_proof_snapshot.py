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

# proof.rs:212, 218-225: Stark::default() (security level 160), the tiny program
SNAPSHOT = [2390426207231576512, 11357322246033024133, 15595568858844533957, 10807389618517394866, 11786266879565336160]
SEED_U64 = 4742841043836029231
# `supplying_prover_randomness_seed_fully_derandomizes_produced_proof` (stark.rs:2434-2460): Stark::low_security() (security
# level 32, stark.rs:2294-2299) on `program_executing_every_instruction` (stark.rs:4639-4803) -- every instruction, every table
SNAPSHOT_EVERY = [8369583593597337114, 14430538234814724839, 9910198730687648118, 13547514320109628452, 7746148481830452917]
SEED_U64_EVERY = 3351975627407608972


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


def test_oracle_prover_reproduces_the_second_reference_proof_digest():
    from oracle import real_prover

    program, _, public_input, _ = vf.run("every")
    secret_input, secret_digests, ram = vf.non_determinism("every")
    proof = real_prover.prove(program, public_input, secret_input, secret_digests, ram, seed_u64=SEED_U64_EVERY, security_level=32)
    assert proof["params"] == dict(n=2048, h=60, ldt=16384, checks=35, fri_rounds=6)
    assert proof["digest"] == SNAPSHOT_EVERY


def claim_of(orc, program, public_input, output):
    from triton_vm_amd.prover import Claim

    mont = lambda values: orc.to_mont(np.array(values, dtype=object)) if len(values) else ()
    return Claim(orc.hash_varlen(mont(program.to_bwords())), mont(public_input), mont(output))


def device_proof(ctx, orc, which, seed_u64, security_level, ldt="fri"):
    from triton_vm_amd.prover import Prover

    program, aet, public_input, output = vf.run(which)
    prover = Prover.from_execution(ctx, aet_arrays(orc, aet), aet.padded_height(), claim_of(orc, program, public_input, output),
                                   prover_seed(seed_u64), security_level=security_level, ldt=ldt)
    return prover.prove().proof()


@pytest.mark.gpu
def test_device_prover_reproduces_the_second_reference_proof_digest(orc):
    """every instruction and every table through fill / pad / extend and the hot path on the device (GPU only: the
    emulation would take ten minutes)"""
    from triton_vm_amd import Context

    ctx = Context(device=0)
    try:
        assert device_proof(ctx, orc, "every", SEED_U64_EVERY, 32).digest(ctx.lib) == SNAPSHOT_EVERY
    finally:
        ctx.close()


def test_device_prover_reproduces_the_reference_proof_digest(ctx, orc):
    """Prover.from_execution over the C ABI -- on the emulation of the kernel sources and on the MI355X"""
    proof = device_proof(ctx, orc, "tiny", SEED_U64, 160)
    assert proof.digest(ctx.lib) == SNAPSHOT
    if ctx.kind == "emu":   # and word for word the oracle prover's proof
        from oracle import real_prover

        program, _, public_input, _ = vf.run("tiny")
        assert [int(v) for v in orc.from_mont(proof.words)] == real_prover.prove(program, public_input, seed_u64=SEED_U64)["proof"]
