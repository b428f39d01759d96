"""Proofs go through the reference's acceptance procedure: `Verifier::verify` (/root/reference/triton-vm/src/stark.rs:1388-1763)
restated in oracle/real_verifier.py over the decoded proof (`ProofStream::try_from(&Proof)`, triton_vm_amd/proof_stream.py).

  * anchor: the oracle prover's proof of the snapshot program -- whose digest equals the reference's (tests/test_proof_snapshot.py),
    i.e. a proof the reference's own verifier accepts -- is accepted; single-word corruptions of every kind of item, a wrong
    claim, a truncated and an extended proof are rejected;
  * device proofs (GPU): the reference's headline program prove_fib (index 100: 2^10 padded rows) with FRI and with STIR,
    are accepted; the FRI one is also the oracle prover's proof, word for word.  (The same at 2^20 padded rows, FRI and the
    STIR the reference picks there, stark.rs:1944-1951: tests/perf/prove_fib.py, results under profiles/r02_l_*.)
"""
import functools

import numpy as np
import pytest

from tests import test_proof_snapshot as snap
from tests import vm_fixture as vf


@functools.lru_cache(maxsize=None)
def oracle_proof(which, seed_u64, security_level):
    from oracle import oracle as orc, real_prover

    program, _, public_input, output = vf.run(which)
    proof = real_prover.prove(program, public_input, *vf.non_determinism(which), seed_u64=seed_u64, security_level=security_level)
    return orc.to_mont(np.array(proof["proof"], dtype=object)), snap.claim_of(orc, program, public_input, output), proof["indices"]


@pytest.fixture(scope="module")
def host_lib():
    """the host-side helpers of the C ABI (Tip5 on the host): the emulation build exports the same ones"""
    from tests.emu.build_emu import build
    from triton_vm_amd.capi import load_library

    return load_library(build())


def verify(host_lib, words, claim, **kw):
    """the restated Verifier::verify over the oracle's OWN decoding of the proof and its own sponge (oracle/proof_decode.py:
    nothing of the product's proof_stream.py or host-side Tip5 takes part); the product's decoder must read the same items"""
    from oracle import proof_decode, real_verifier
    from triton_vm_amd.proof_stream import ProofDecodingError, ProofStream

    try:
        view = proof_decode.VerifierView(words)
    except proof_decode.DecodingError as e:
        with pytest.raises(ProofDecodingError):       # both decoders refuse the same proofs
            ProofStream.from_proof(host_lib, words)
        raise ProofDecodingError(str(e))
    theirs = ProofStream.from_proof(host_lib, words).log
    assert len(theirs) == len(view.pending)
    for (name, _enc, _k, payload), (label, other, _fs) in zip(view.pending, theirs):
        assert np.array_equal(np.asarray(payload).reshape(-1), np.asarray(other).reshape(-1)), (name, label)
    return real_verifier.verify(view, claim, **kw)


def test_verifier_accepts_the_reference_pinned_proof(host_lib):
    from triton_vm_amd.proof_stream import ProofStream

    words, claim, indices = oracle_proof("tiny", snap.SEED_U64, 160)
    assert verify(host_lib, words, claim) == indices
    # decoding and re-encoding is the identity (BFieldCodec round trip, proof_stream.rs decode tests)
    assert (ProofStream.from_proof(host_lib, words).proof().words == words).all()
    # Proof::padded_height (proof.rs:45-59, tests proof.rs:172-189)
    from triton_vm_amd.proof_stream import Proof, ProofDecodingError

    assert Proof(words).padded_height(host_lib) == 256
    stream = ProofStream.from_proof(host_lib, words)
    stream.log = [entry for entry in stream.log if entry[0] != "Log2PaddedHeight"]
    with pytest.raises(ProofDecodingError, match="NoLog2PaddedHeight"):
        stream.proof().padded_height(host_lib)
    stream = ProofStream.from_proof(host_lib, words)
    stream.log.append(stream.log[0])
    with pytest.raises(ProofDecodingError, match="TooManyLog2PaddedHeights"):
        stream.proof().padded_height(host_lib)


def item_offsets(host_lib, words):
    """word offset of the first payload word of every item of the proof, by variant name"""
    from triton_vm_amd import field
    from triton_vm_amd.proof_stream import PROOF_ITEMS

    w = [field.from_mont(int(x)) for x in words]
    out, pos = {}, 2
    for _ in range(w[1]):
        size, name = w[pos], PROOF_ITEMS[w[pos + 1]][0]
        out.setdefault(name, []).append((pos + 2, size - 1))
        pos += 1 + size
    return out


def test_verifier_rejects_corrupted_proofs_and_wrong_claims(host_lib):
    from oracle.real_verifier import VerificationError
    from triton_vm_amd.proof_stream import Claim, ProofDecodingError

    words, claim, _ = oracle_proof("tiny", snap.SEED_U64, 160)
    offsets = item_offsets(host_lib, words)
    assert set(offsets) == {"Log2PaddedHeight", "MerkleRoot", "OutOfDomainMainRow", "OutOfDomainAuxRow", "OutOfDomainQuotientSegments",
                            "FriCodeword", "Polynomial", "FriResponse", "MasterMainTableRows", "MasterAuxTableRows",
                            "QuotientSegmentsElements", "AuthenticationStructure"}
    rng = np.random.default_rng(5)
    for name, places in offsets.items():
        if name == "Log2PaddedHeight":
            continue   # changes the parameters: covered by the claim / structure checks below
        start, size = places[int(rng.integers(len(places)))]
        # a payload word well inside the item (behind any length prefixes), flipped in its lowest bit
        k = start + (size // 2 if size > 8 else size - 1)
        bad = words.copy()
        bad[k] ^= np.uint64(1)
        with pytest.raises((VerificationError, ProofDecodingError, ValueError)):
            verify(host_lib, bad, claim)
    for other in (Claim(claim.program_digest, claim.input[::-1].copy(), claim.output), Claim(claim.program_digest, claim.input, claim.output, version=5),
                  Claim(claim.program_digest[::-1].copy(), claim.input, claim.output)):
        with pytest.raises(VerificationError):
            verify(host_lib, words, other)
    with pytest.raises((ProofDecodingError, ValueError)):
        verify(host_lib, words[:-7], claim)
    with pytest.raises((ProofDecodingError, ValueError)):
        verify(host_lib, np.concatenate([words, words[-3:]]), claim)
    with pytest.raises(VerificationError):   # the right items under another security level: other parameters, other indices
        verify(host_lib, words, claim, security_level=128)


@pytest.fixture(scope="module")
def gpu_ctx():
    from triton_vm_amd import Context

    ctx = Context(device=0)
    yield ctx
    ctx.close()


@pytest.mark.gpu
def test_prove_fib_100_on_the_device_is_the_oracle_provers_proof_and_is_accepted(gpu_ctx, orc):
    seed = 20260926
    proof = snap.device_proof(gpu_ctx, orc, ("fib", 100), seed, 160)
    words, claim, indices = oracle_proof(("fib", 100), seed, 160)
    assert proof.words.size == words.size and (proof.words == words).all()
    assert verify(gpu_ctx.lib, proof.words, claim) == indices


@pytest.mark.gpu
@pytest.mark.parametrize("index,log2_padded_height", [(100, 10)])   # 2^16 and 2^20 rows: tests/perf/prove_fib.py (profiles/r02_l_*)
def test_prove_fib_with_stir_on_the_device_is_accepted(gpu_ctx, orc, index, log2_padded_height):
    """STIR is what Stark::ldt picks from 2^16 padded rows on (stark.rs:1944-1951)"""
    program, aet, public_input, output = vf.run(("fib", index))
    assert aet.padded_height() == 1 << log2_padded_height
    proof = snap.device_proof(gpu_ctx, orc, ("fib", index), 7, 160, ldt="stir")
    claim = snap.claim_of(orc, program, public_input, output)
    assert len(verify(gpu_ctx.lib, proof.words, claim, ldt_choice="stir")) > 0
    from oracle.real_verifier import VerificationError

    with pytest.raises(VerificationError):
        verify(gpu_ctx.lib, proof.words, snap.claim_of(orc, program, [index + 1], output), ldt_choice="stir")
    # the product's Verifier::verify with STIR: the same verdicts
    from triton_vm_amd import verifier as product

    assert product.Verifier(gpu_ctx, ldt="stir").verify(claim, proof.words) == verify(gpu_ctx.lib, proof.words, claim, ldt_choice="stir")
    with pytest.raises(product.VerificationError):
        product.Verifier(gpu_ctx, ldt="stir").verify(snap.claim_of(orc, program, [index + 1], output), proof.words)


def test_arbitrary_corruptions_never_escape_as_other_errors(host_lib):
    """`decoding_arbitrary_proof_data_does_not_panic` / `verifying_arbitrary_proof_does_not_panic` (proof.rs:191-196,
    stark.rs:4319-4326) in spirit: random overwrites, truncations and splices of a valid proof are either rejected with a
    decoding / verification error or (never observed) accepted -- nothing else escapes"""
    from oracle.real_verifier import VerificationError

    words, claim, _ = oracle_proof("tiny", snap.SEED_U64, 160)
    rng = np.random.default_rng(99)
    outcomes = {"rejected": 0, "accepted": 0}
    for trial in range(60):
        bad = words.copy()
        kind = trial % 4
        if kind == 0:      # a few random words anywhere (length prefixes included)
            for k in rng.integers(0, bad.size, int(rng.integers(1, 4))):
                bad[k] = np.uint64(rng.integers(0, 2**63))
        elif kind == 1:    # truncation
            bad = bad[:int(rng.integers(0, bad.size))]
        elif kind == 2:    # small canonical values where length prefixes tend to live: the head of the proof
            bad[int(rng.integers(0, 64))] = np.uint64(int(orc_mont(int(rng.integers(0, 40)))))
        else:              # a block moved elsewhere
            a, b = sorted(int(v) for v in rng.integers(0, bad.size, 2))
            bad = np.concatenate([bad[:a], bad[b:], bad[a:b]])
        try:
            verify(host_lib, bad, claim)
            outcomes["accepted"] += 1
        except (VerificationError, ValueError):   # ProofDecodingError is a ValueError
            outcomes["rejected"] += 1
    assert outcomes == {"rejected": 60, "accepted": 0}


def orc_mont(v):
    from triton_vm_amd import field

    return field.to_mont(v)
