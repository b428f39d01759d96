"""The reference's prove-and-verify tests (`prove_and_verify_*`, /root/reference/triton-vm/src/stark.rs:4257-4317) over the
device path: Prover.from_execution (fill, pad, extend and the hot path through the C ABI) proves, the restated
Verifier::verify (oracle/real_verifier.py, anchored to the reference-pinned proofs in tests/test_verify_proof.py) accepts
-- and rejects the same proof under another claim; so does the product's own verifier (triton_vm_amd/verifier.py).  Stark::low_security() (security level 32) like the reference's
TestableProgram, plus other parameter sets for `halt`."""
import pytest

from tests import test_proof_snapshot as snap
from tests import vm_fixture as vf
from tests.test_verify_proof import verify

CASES = [
    # program, security level, log2 expansion, low-degree test, runs on the emulation too
    ("halt", 32, 2, "fri", True), ("halt", 64, 3, "fri", False), ("halt", 160, 2, "fri", False), ("halt", 48, 4, "fri", False),
    ("many_u32", 32, 2, "fri", False), ("pick_and_place", 32, 2, "fri", False), (("fib", 100), 32, 2, "fri", False),
    ("every", 32, 2, "fri", False), ("every", 64, 3, "fri", False), (("u32", 100), 32, 2, "fri", False),
    ("halt", 32, 2, "stir", False), ("every", 48, 2, "stir", False), (("fib", 100), 160, 2, "stir", False),
    (("ram", 3000), 32, 2, "fri", False),   # 3000 distinct RAM pointers: the Bezout coefficient polynomials come from the device
    (("sponge", 60), 32, 4, "fri", False),  # sponge loop (hash and cascade tables), the recursive-verifier config's FRI log-blowup
]


@pytest.mark.parametrize("which,security_level,log2_expansion,ldt,on_emulation", CASES)
def test_prove_and_verify(ctx, orc, which, security_level, log2_expansion, ldt, on_emulation):
    from oracle.real_verifier import VerificationError
    from tests.test_fill import aet_arrays
    from triton_vm_amd.prover import Prover

    if ctx.kind == "emu" and not on_emulation:
        pytest.skip("CPU suite time: this case runs on the GPU")
    program, aet, public_input, output = vf.run(which)
    if which == "pick_and_place":
        assert output == [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    claim = snap.claim_of(orc, program, public_input, output)
    host_bezout = not (isinstance(which, tuple) and which[0] == "ram")
    prover = Prover.from_execution(ctx, aet_arrays(orc, aet, host_bezout), aet.padded_height(), claim, snap.prover_seed(len(str(which))),
                                   security_level=security_level, log2_expansion=log2_expansion, ldt=ldt)
    proof = prover.prove().proof()
    kw = dict(security_level=security_level, log2_expansion=log2_expansion, ldt_choice=ldt)
    accepted_at = verify(ctx.lib, proof.words, claim, **kw)
    assert len(accepted_at) > 0
    wrong = snap.claim_of(orc, program, public_input, list(output) + [1])
    with pytest.raises(VerificationError):
        verify(ctx.lib, proof.words, wrong, **kw)
    # ... and the product's own Verifier::verify (device batch work) gives the same verdicts
    from triton_vm_amd import verifier as product

    kw = dict(security_level=security_level, log2_expansion=log2_expansion, ldt=ldt)
    assert product.Verifier(ctx, **kw).verify(claim, proof.words) == accepted_at
    with pytest.raises(product.VerificationError):
        product.Verifier(ctx, **kw).verify(wrong, proof.words)
