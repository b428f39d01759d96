"""Host-side cost of the sharded proof's plumbing on one GPU: a ONE-rank RCCL group runs ShardedProver (torch tensors,
all-gathers of one share, the Python host) next to the plain Python-host Prover on the same synthetic 2^k-row tables.
usage: python tests/perf/sharded_probe.py [log2_rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from triton_vm_amd import Context  # noqa: E402
from triton_vm_amd.prover import Prover, StarkParameters  # noqa: E402
from triton_vm_amd.sharded import ShardedProver  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
ctx = Context(device=0)
params = StarkParameters(log_n)
for name, make in (("Prover (Python host)", lambda: Prover(ctx, params, seed=1000)),
                   ("ShardedProver, one rank", lambda: ShardedProver(ctx, params, dist, device, seed=1000)),
                   ("ShardedProver, one rank, split-tree exchanges", lambda: ShardedProver(ctx, params, dist, device, seed=1000))):
    prover = make()
    if "split" in name:
        prover.split_tree_min_leaves = 1 << 21   # the production threshold (one rank: whole trees, but the exchanges run)
    prover.prove()
    ctx.sync()
    ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        prover.prove()
        ctx.sync()
        ms.append(1e3 * (time.perf_counter() - t0))
    prover.timings, prover.wall = {}, {}
    prover.prove(profile=True)
    print(f"{name}: {min(ms):.1f} ms per proof (best of 3); stages: " +
          ", ".join(f"{k} {v:.1f}" for k, v in prover.wall.items()), flush=True)
    prover.release()
    del prover
dist.destroy_process_group()
