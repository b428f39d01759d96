"""Pins beyond the reference's two FRI-sized proof snapshots (tests/test_proof_snapshot.py), at no cargo cost:

  * whole-proof equality, word for word, between the device path (`Prover.from_execution` over the C ABI, and the C++ host) and
    the oracle prover (oracle/real_prover.py -- which reproduces BOTH of the reference's proof digests, so on these executions
    it IS the reference's prover as far as the repository can know) at padded heights 2^12 and 2^14: larger domains, more FRI
    rounds, 1024- and 4096-point transform axes, multi-chunk tables;
  * STIR regression digests: `Tip5::hash(proof)` of the two snapshot programs with `LdtChoice::Stir` forced.  The reference holds
    no STIR vector (its snapshots are FRI-sized), so these are NOT reference pins -- but since round 6 they are not the product's own
    word either: the oracle prover's coefficient-form STIR (oracle/real_prover.py `stir=`, oracle/stir_oracle.py) arrives at the same
    digests, and at 2^12 / 2^14 rows the device's STIR proofs equal the oracle prover's word for word.
"""
import numpy as np
import pytest

from tests import test_proof_snapshot as snap
from tests import vm_fixture as vf


def _golden():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "stir_regression_digests.json")) as f:
        return json.load(f)


def stir_numbers(padded_height, security_level):
    """the STIR instance the reference's Stark::stir picks (host arithmetic, pinned by tests/test_stir_parameters.py), as plain numbers
    for the oracle prover"""
    from triton_vm_amd.low_degree_test import stark_stir

    st = stark_stir(padded_height, security_level=security_level)
    return dict(initial_domain_length=st.initial_domain.length, num_trace_randomizers=st.num_trace_randomizers(), folding_factor=st.folding_factor,
                round_queries=list(st.round_queries), final_num_in_domain_queries=st.final_num_in_domain_queries)


@pytest.mark.parametrize("which,seed,security_level", [("tiny", snap.SEED_U64, 160), ("every", snap.SEED_U64_EVERY, 32)])
def test_oracle_stir_prover_reproduces_the_frozen_stir_digests(which, seed, security_level):
    """Round 6: the frozen STIR digests are no longer the product's word alone.  oracle/real_prover.py with `stir=` runs Stir::prove
    (stir.rs:885-993) statement by statement on polynomials in COEFFICIENT form -- Lagrange interpolation, an explicit zerofier, long
    division, schoolbook multiplication (oracle/stir_oracle.py): nothing of csrc/stir.hip's evaluation-form round -- and arrives at
    the very digests the device proofs were frozen with ("every": two full rounds + the final one; "tiny": the final round only).
    Still no reference-held STIR vector (that needs cargo): what the digests freeze is now an ORACLE-EQUAL proof."""
    from oracle import real_prover

    program, aet, public_input, _ = vf.run(which)
    extra = vf.non_determinism(which) if which == "every" else ()
    proof = real_prover.prove(program, public_input, *extra, seed_u64=seed, security_level=security_level,
                              stir=stir_numbers(aet.padded_height(), security_level))
    assert proof["digest"] == _golden()[which]["digest"]
    assert len(proof["proof"]) == _golden()[which]["proof_words"]


def test_stir_proof_of_the_first_snapshot_program_has_not_drifted(ctx, orc):
    from triton_vm_amd.verifier import Verifier

    program, aet, public_input, output = vf.run("tiny")
    claim = snap.claim_of(orc, program, public_input, output)
    proof = snap.device_proof(ctx, orc, "tiny", snap.SEED_U64, 160, ldt="stir")
    assert [int(w) for w in proof.digest(ctx.lib)] == _golden()["tiny"]["digest"]
    Verifier(ctx, security_level=160, ldt="stir").verify(claim, proof.words)   # raises on rejection


@pytest.mark.gpu
def test_stir_proof_of_the_second_snapshot_program_has_not_drifted(orc):
    """every instruction, every table, security level 32, STIR forced (GPU only: ten minutes on the emulation)"""
    from triton_vm_amd import Context

    ctx = Context(device=0)
    try:
        proof = snap.device_proof(ctx, orc, "every", snap.SEED_U64_EVERY, 32, ldt="stir")
        assert [int(w) for w in proof.digest(ctx.lib)] == _golden()["every"]["digest"]
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log2_rows", [12, 14])
def test_device_stir_proof_equals_the_oracle_provers_at_larger_heights(orc, log2_rows):
    """LdtChoice::Stir on prove_fib at 2^12 rows (one full STIR round of 246 + 1 queries) and 2^14 rows (three): every word of the
    device proof (Python host and C++ host) equals the oracle prover's coefficient-form STIR (see the test above)."""
    from oracle import real_prover
    from oracle.vm import workload
    from triton_vm_amd import Context, native_host
    from triton_vm_amd.prover import Claim, Prover

    e = workload.execution("fib", log2_rows)
    want = real_prover.prove(e["program"], [e["index"]], seed_u64=snap.SEED_U64, stir=stir_numbers(e["padded_height"], 160))
    seed = snap.prover_seed(snap.SEED_U64)
    claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
    ctx = Context(device=0)
    try:
        words = Prover.from_execution(ctx, e["aet"], e["padded_height"], claim, seed, ldt="stir").prove().proof().words
        assert [int(v) for v in orc.from_mont(words)] == want["proof"]
        native = native_host.prove_execution(ctx, native_host.load_host_library(), e["aet"], e["padded_height"], claim, seed, ldt="stir")
        assert native.size == words.size and (native == words).all()
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log2_rows", [12, 14])
@pytest.mark.parametrize("kind", ["fib", "u32", "ram"])   # (the sponge loop fills the cascade table: its padded height is 2^16 at least -- minutes of oracle prover)
def test_device_proof_equals_the_oracle_provers_at_larger_heights(orc, kind, log2_rows):
    """prove_fib -- and the u32 / ram loops of BASELINE.json's configs[3], whose U32 and RAM tables (Bezout coefficients) are NOT small
    here (round 6) -- padded to 2^12 and 2^14 rows: every word of the device proof (Python host and C++ host) equals the oracle
    prover's -- the prover that reproduces the reference's two proof digests.  (The hash-heavy sponge loop of configs[4] cannot be
    this small: it fills the cascade table, 2^16 rows at least; tests/test_gpu_baseline_configs.py proves and verifies it at 2^20.)"""
    from oracle import real_prover
    from oracle.vm import workload
    from triton_vm_amd import Context, native_host
    from triton_vm_amd.prover import Claim, Prover

    e = workload.execution(kind, log2_rows)
    want = real_prover.prove(e["program"], [e["index"]], seed_u64=snap.SEED_U64)
    seed = snap.prover_seed(snap.SEED_U64)
    claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
    ctx = Context(device=0)
    try:
        prover = Prover.from_execution(ctx, e["aet"], e["padded_height"], claim, seed, ldt="fri")
        words = prover.prove().proof().words
        assert [int(v) for v in orc.from_mont(words)] == want["proof"]
        native = native_host.prove_execution(ctx, native_host.load_host_library(), e["aet"], e["padded_height"], claim, seed, ldt="fri")
        assert native.size == words.size and (native == words).all()
    finally:
        ctx.close()


def _oracle_digests():
    import json
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", "oracle_proof_digests.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(_oracle_digests()) or ["none"])
def test_device_proof_hashes_to_the_oracle_provers_digest_at_2p16_to_2p18(orc, case):
    """Whole-proof parity where the oracle prover takes minutes to hours (tests/golden/make_oracle_proof_digests.py ran it on the CPU and
    committed `Tip5::hash(proof)` + the proof's length): prove_fib at 2^16 ... 2^20 rows with FRI -- 2^20 is BASELINE configs[1], the
    headline workload at full size: 1 h 53 min of oracle prover -- and at 2^16 rows with STIR (the reference's default from there on);
    the u32, ram and sponge loops at 2^16 rows.  The device proof of the same (program, input, prover seed) -- C++ host -- must hash to the
    same digest: every word of the proof, at the heights whose extension runs the 256-, 512- and 1024-point row kernels of round 6."""
    if case == "none":
        pytest.skip("tests/golden/oracle_proof_digests.json has not been generated")
    from oracle.vm import workload
    from triton_vm_amd import Context, native_host
    from triton_vm_amd.proof_stream import Proof
    from triton_vm_amd.prover import Claim

    want = _oracle_digests()[case]
    e = workload.execution(want["program"], want["log2_padded_height"])
    assert [int(v) for v in orc.from_mont(e["public_input"])] == want["public_input"]
    claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
    ctx = Context(device=0)
    try:
        words = native_host.prove_execution(ctx, native_host.load_host_library(), e["aet"], e["padded_height"], claim,
                                            snap.prover_seed(want["seed_u64"]), ldt=want["ldt"])
        assert words.size == want["proof_words"]
        assert [int(w) for w in Proof(words).digest(ctx.lib)] == want["digest"]
    finally:
        ctx.close()
