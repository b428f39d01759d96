import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from oracle import oracle as orc
from oracle import ldt_verifier as lv
from triton_vm_amd import Context, low_degree_test as ldt
from triton_vm_amd.prover import ProofStream
log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = Context(0)
stir = ldt.stark_stir(1 << log2)
print("initial domain", stir.initial_domain.length, "rounds", stir.round_queries, flush=True)
rng = np.random.default_rng(1)
deg_bound = stir.initial_domain.length >> 2
poly = orc.random_elements(rng, (deg_bound, 3))
d_poly = ctx.to_device(poly)
d_cw = stir.initial_domain.evaluate(ctx, d_poly, deg_bound, 3)
ps = ProofStream(ctx.lib)
t0 = time.time(); first = stir.prove(ctx, d_cw, ps); ctx.sync(); print("prove s", time.time() - t0, flush=True)
t0 = time.time(); got = lv.stir_verify(ps.verifier_view(), stir); print("verify s", time.time() - t0, got == first, flush=True)
