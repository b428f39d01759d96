"""prove_fib for real (the reference's headline benchmark, benches/prove_fib.rs): run the Fibonacci program in the
oracle-side VM up to 2^k cycles, hand the algebraic execution trace to Prover.from_execution -- fill, pad, extend and the
hot path on the device, the reference's transcript on the host -- and put the proof through the restated Verifier::verify.
The VM run stands in for the reference's Rust VM (host work there too); everything after it is the product.
With `u32` instead: a loop of u32 operations whose U32 table fills the padded height (BASELINE.json's many-u32-ops shape);
with `ram`: a loop that writes a fresh RAM address per iteration (the RAM table's Bezout coefficient polynomials, one
coefficient per distinct pointer, are then computed on the device: tvm_bezout_coefficients inside tvm_fill_main_table);
with `sponge`: a loop of sponge_squeeze / sponge_absorb -- half a million rows of the hash table at 2^20, the full cascade
table: the hash-heavy shape standing in for BASELINE.json's recursive verifier.  `--log2-expansion 4`: Stark::new(160, 4),
that config's FRI log-blowup.
usage: python tests/perf/prove_fib.py [log2_padded_height=20] [fri|stir] [u32|ram|sponge] [--log2-expansion K] [--no-verify]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402  (first: see tests/conftest.py)

from oracle import oracle as orc  # noqa: E402
from tests import test_proof_snapshot as snap, vm_fixture as vf  # noqa: E402
from tests.test_fill import aet_arrays  # noqa: E402
from triton_vm_amd import Context  # noqa: E402
from triton_vm_amd.prover import Prover  # noqa: E402

log2 = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
ldt = "stir" if "stir" in sys.argv else "fri"
u32, ram, sponge = "u32" in sys.argv, "ram" in sys.argv, "sponge" in sys.argv
log2_expansion = int(sys.argv[sys.argv.index("--log2-expansion") + 1]) if "--log2-expansion" in sys.argv else 2
# fib: ten instructions per iteration, a dozen around the loop; u32: 33 rows of the U32 table per iteration; ram: 14 cycles
index = (1 << log2) // 33 if u32 else ((1 << log2) - 20) // 14 if ram else ((1 << log2) - 40) // 24 if sponge else ((1 << log2) - 20) // 10
t = {}
t0 = time.perf_counter()
program, aet, public_input, output = vf.run(("u32" if u32 else "ram" if ram else "sponge" if sponge else "fib", index))
t["vm_s"] = time.perf_counter() - t0
padded_height = aet.padded_height()      # (the oracle-side AET recomputes its table heights on every call: not in the timed regions)
assert padded_height == 1 << log2, padded_height
t0 = time.perf_counter()
arrays = aet_arrays(orc, aet, host_bezout=not ram)
t["aet_arrays_s"] = time.perf_counter() - t0
claim = snap.claim_of(orc, program, public_input, output)
ctx = Context(device=0)
seed = snap.prover_seed(1)
result = {}
for attempt in range(2):                    # the second pass is the warm one
    ctx.sync()
    t0 = time.perf_counter()
    prover = Prover.from_execution(ctx, arrays, padded_height, claim, seed, ldt=ldt, log2_expansion=log2_expansion)
    ctx.sync()
    t1 = time.perf_counter()
    stream = prover.prove(profile="--profile" in sys.argv)
    ctx.sync()
    t2 = time.perf_counter()
    proof = stream.proof()
    if "--profile" in sys.argv:
        print({k: round(v, 1) for k, v in prover.timings.items()}, file=sys.stderr)
    result = {"fill_pad_randomizers_ms": 1e3 * (t1 - t0), "extend_and_hot_path_ms": 1e3 * (t2 - t1), "proof_words": int(proof.words.size)}
    prover.release()
    del prover
out = {"program": f"u32 loop, {index} iterations" if u32 else f"RAM loop, {index} distinct pointers (Bezout coefficients on the device)" if ram
       else f"sponge loop, {index} x (squeeze, absorb)" if sponge else f"fibonacci_sequence, index {index}",
       "log2_expansion": log2_expansion,
       "table_heights": {name: aet.height_of_table(name) for name in ("Processor", "OpStack", "Ram", "U32", "Hash", "Cascade")},
       "cycles": aet.height_of_table("Processor"), "padded_height": padded_height,
       "ldt": ldt, **{k: round(v, 2) for k, v in t.items()}, **{k: round(v, 1) if isinstance(v, float) else v for k, v in result.items()},
       "proof_digest": proof.digest(ctx.lib)}
if ldt == "fri":   # the same through the C++ host (triton_vm::prove_execution): one call = Prover::prove(claim, aet)
    from triton_vm_amd import native_host
    from triton_vm_amd.proof_stream import Proof

    host_lib = native_host.load_host_library()
    for attempt in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        words = native_host.prove_execution(ctx, host_lib, arrays, padded_height, claim, seed, log2_expansion=log2_expansion, ldt="fri")
        out["cpp_host_whole_prove_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
    out["cpp_host_proof_equals_python_host_proof"] = bool(words.size == proof.words.size and (words == proof.words).all())
if "--no-verify" not in sys.argv:
    from oracle import proof_decode, real_verifier

    t0 = time.perf_counter()
    indices = real_verifier.verify(proof_decode.VerifierView(proof.words), claim, log2_expansion=log2_expansion, ldt_choice=ldt)
    out["verified"] = True
    out["verifier_s"] = round(time.perf_counter() - t0, 1)
    out["revealed_rows"] = len(indices)
    if True:   # the product's own Verifier::verify (device batch work)
        from triton_vm_amd.verifier import Verifier

        t0 = time.perf_counter()
        out["product_verifier_agrees"] = Verifier(ctx, log2_expansion=log2_expansion, ldt=ldt).verify(claim, proof.words) == indices
        out["product_verifier_s"] = round(time.perf_counter() - t0, 2)
print(json.dumps(out))
