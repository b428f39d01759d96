"""triton_vm_amd.verifier.Verifier -- the product's Verifier::verify (/root/reference/triton-vm/src/stark.rs:1388-1763, FRI):
row hashing and the per-row combination values on the device, the AIR at the out-of-domain rows through
tvm_host_air_constraints, Fiat-Shamir and the decisions on the host -- against the reference-pinned proof and against the
oracle's independent restatement of the same procedure (oracle/real_verifier.py): same verdict on every input tried."""
import numpy as np
import pytest

from tests import test_proof_snapshot as snap
from tests.test_verify_proof import item_offsets, oracle_proof


def test_air_constraints_at_one_row_pair_equal_the_oracle_circuit(ctx, orc):
    rng = np.random.default_rng(8)
    rows = [orc.random_elements(rng, (n, 3)) for n in (379, 91, 379, 91, 63)]
    got = np.zeros((604, 3), np.uint64)
    ctx._check(ctx.lib.tvm_host_air_constraints(*[r.ctypes.data for r in rows], got.ctypes.data), "tvm_host_air_constraints")
    main_cur, aux_cur, main_next, aux_next, challenges = rows
    assert (got == orc.air_constraint_values(main_cur, main_next, aux_cur, aux_next, challenges)).all()


def test_product_verifier_accepts_the_reference_pinned_proof_and_agrees_with_the_oracle_on_rejections(ctx):
    from oracle import proof_decode, real_verifier
    from triton_vm_amd.proof_stream import Claim, ProofDecodingError
    from triton_vm_amd.verifier import VerificationError, Verifier

    words, claim, indices = oracle_proof("tiny", snap.SEED_U64, 160)
    verifier = Verifier(ctx)
    assert verifier.verify(claim, words) == indices

    def verdicts(bad_words, bad_claim, **kw):
        out = []
        for run in (lambda: (Verifier(ctx, **kw) if kw else verifier).verify(bad_claim, bad_words),
                    lambda: real_verifier.verify(proof_decode.VerifierView(bad_words), bad_claim, **kw)):
            try:
                run()
                out.append("accepted")
            except (VerificationError, real_verifier.VerificationError, ProofDecodingError, proof_decode.DecodingError, ValueError):
                out.append("rejected")
        return out

    rng = np.random.default_rng(6)
    for name, places in item_offsets(ctx.lib, words).items():
        if name == "Log2PaddedHeight":
            continue
        start, size = places[int(rng.integers(len(places)))]
        bad = words.copy()
        bad[start + (size // 2 if size > 8 else size - 1)] ^= np.uint64(1)
        assert verdicts(bad, claim) == ["rejected", "rejected"], name
    assert verdicts(words, Claim(claim.program_digest, claim.input, claim.output, version=5)) == ["rejected", "rejected"]
    assert verdicts(words[:-7], claim) == ["rejected", "rejected"]
    assert verdicts(words, claim, security_level=128) == ["rejected", "rejected"]


def test_product_stir_verifier_agrees_with_the_oracle_restatement(ctx, orc):
    """Verifier._stir_verify (stir.rs:995-1340) on the transcripts of tests/test_ldt_verifiers.py: accepted with the prover's
    first-round indices and the values of the codeword there; high degree and tampering rejected"""
    from tests.test_ldt_verifiers import odom, small_stir
    from triton_vm_amd.prover import ProofStream
    from triton_vm_amd.verifier import VerificationError, Verifier, _xfe

    verifier = Verifier(ctx, ldt="stir")
    X = _xfe(ctx.lib)

    def run(stream, stir):
        view = stream.verifier_view()
        return verifier._stir_verify(view, view.dequeue, stir, X)

    for log2_bound, queries in [(6, [(3, 1), (2, 0)]), (8, [(5, 2), (3, 1), (4, 0)]), (4, [(3, 0)])]:
        rng = np.random.default_rng(log2_bound)
        stir = small_stir(log2_bound, queries)
        poly = orc.random_elements(rng, (1 << log2_bound, 3))
        codeword = orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3)
        ps = ProofStream(ctx.lib)
        first = stir.prove(ctx, ctx.to_device(codeword), ps)
        indices, values = run(ps, stir)
        assert indices == first and (values == codeword[first]).all()
    # too high a degree
    poly = orc.random_elements(rng, (stir.initial_domain.length, 3))
    codeword = orc.coset_evaluate(poly, odom(orc, stir.initial_domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    stir.prove(ctx, ctx.to_device(codeword), ps)
    with pytest.raises(VerificationError):
        run(ps, stir)
    # tampering
    stir = small_stir(6, [(3, 1), (2, 0)])
    codeword = orc.coset_evaluate(orc.random_elements(rng, (1 << 6, 3)), odom(orc, stir.initial_domain), 3).reshape(-1, 3)
    ps = ProofStream(ctx.lib)
    stir.prove(ctx, ctx.to_device(codeword), ps)
    run(ps, stir)
    for victim in ("stir response leafs", "stir response auth", "stir ood values", "stir final polynomial", "stir root"):
        bad = ProofStream(ctx.lib)
        bad.log = [(n, pl.copy(), fs) for n, pl, fs in ps.log]
        k = next(i for i, (n, pl, _) in enumerate(bad.log) if n == victim and pl.size)
        bad.log[k][1].reshape(-1)[0] ^= np.uint64(1)
        with pytest.raises(VerificationError):
            run(bad, stir)


def test_statically_sized_items_of_the_wrong_length_do_not_decode(ctx):
    """BFieldCodec rejects a MerkleRoot, Log2PaddedHeight or out-of-domain row whose payload is not exactly the type's static
    length; so must ProofStream::try_from(&Proof) here -- as ProofDecodingError, never a numpy error, and never an
    acceptance (a decoder that takes any length makes proofs malleable).  Items are re-encoded with one word appended /
    removed and the enclosing length prefixes fixed up."""
    from triton_vm_amd import field
    from triton_vm_amd.proof_stream import PROOF_ITEMS, STATIC_WORDS, Proof, ProofDecodingError, ProofStream
    from triton_vm_amd.verifier import Verifier

    words, claim, indices = oracle_proof("tiny", snap.SEED_U64, 160)
    assert Verifier(ctx).verify(claim, words) == indices
    w = [field.from_mont(int(x)) for x in words]
    items, pos = [], 2
    for _ in range(w[1]):
        size = w[pos]
        items.append(np.array(words[pos + 1:pos + 1 + size]))
        pos += 1 + size

    def proof_of(item_list):
        parts = [np.array([field.to_mont(len(item_list))], np.uint64)]
        for it in item_list:
            parts += [np.array([field.to_mont(it.size)], np.uint64), it]
        body = np.concatenate(parts)
        return np.concatenate([[np.uint64(field.to_mont(body.size))], body])

    assert (proof_of(items) == words).all()
    seen = set()
    for k, it in enumerate(items):
        name, kind, _ = PROOF_ITEMS[field.from_mont(int(it[0]))]
        if kind != "static" or name in seen:
            continue
        seen.add(name)
        assert it.size - 1 == STATIC_WORDS[name]
        for mutated in (np.concatenate([it, it[-1:]]), it[:-1]):
            bad = proof_of(items[:k] + [mutated] + items[k + 1:])
            with pytest.raises(ProofDecodingError):
                ProofStream.from_proof(ctx.lib, bad)
            with pytest.raises(ProofDecodingError):
                Verifier(ctx).verify(claim, bad)
    assert seen == set(STATIC_WORDS)
    # a u32 that is not one
    k = next(i for i, it in enumerate(items) if PROOF_ITEMS[field.from_mont(int(it[0]))][0] == "Log2PaddedHeight")
    big = items[k].copy()
    big[1] = np.uint64(field.to_mont(1 << 32))
    with pytest.raises(ProofDecodingError):
        ProofStream.from_proof(ctx.lib, proof_of(items[:k] + [big] + items[k + 1:]))
    del Proof
