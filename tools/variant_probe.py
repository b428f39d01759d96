"""EXPERIMENT: time main-table LDE + row hashing + AIR with a variant build of the library (libtriton_hip_<variant>.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from triton_vm_amd.capi import Context, load_library
from triton_vm_amd.prover import Prover, StarkParameters
variant = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else None
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "triton_vm_amd")
lib = load_library(os.path.join(here, f"libtriton_hip_{variant}.so")) if variant else load_library()
ctx = Context(device=0, lib=lib)
p = Prover(ctx, StarkParameters(int(sys.argv[2]) if len(sys.argv) > 2 else 20), seed=1)
for _ in range(2):
    p.prove(profile=False)
p.timings = {}
for _ in range(3):
    p.prove(profile=True)
print(variant, {k: round(v / 3, 2) for k, v in p.timings.items() if k in ("main LDE", "main Merkle", "aux Merkle", "AIR quotients", "FRI")})
