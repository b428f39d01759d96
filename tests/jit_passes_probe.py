"""TEST-SIDE MEASUREMENT (it runs the oracle-side VM for its workload, like bench.py does): one execution trace, the C++ host's sharded entry with no communicator at several pass counts (0 = the memory policy).
usage: python tests/jit_passes_probe.py <log2 rows> <passes,passes,...> [repeats] [profile]      e.g.  23 0,4,8,2   or   23 0 2 profile
(profile: the last proof of every pass count with the host's stage clocks, its stream drained at the stage boundaries)
Prints one JSON object: per pass count the milliseconds of each proof, the pass count the host used, and whether every proof is
word for word the first one."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from oracle.vm import workload  # noqa: E402  (workload generation only: the VM stands in for the reference's Rust VM)
from triton_vm_amd import Context, native_host  # noqa: E402
from triton_vm_amd.master_table import aet_to_device  # noqa: E402
from triton_vm_amd.proof_stream import Claim  # noqa: E402

log2_rows = int(sys.argv[1])
pass_counts = [int(x) for x in sys.argv[2].split(",")]
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
profile = len(sys.argv) > 4 and sys.argv[4] == "profile"
ctx = Context(device=0)
host = native_host.load_host_library()
t0 = time.perf_counter()
e = workload.execution("fib", log2_rows)
vm_s = time.perf_counter() - t0
claim = Claim(e["program_digest"], e["public_input"], e["public_output"])
resident = aet_to_device(ctx, e["aet"])
seed = bytes(range(32))
out, first = {"log2_rows": log2_rows, "vm_seconds": round(vm_s, 1), "runs": []}, None
for k in pass_counts:
    ms, used, same, stages = [], None, True, None
    for it in range(repeats):
        ctx.sync()
        t0 = time.perf_counter()
        words, stats = native_host.prove_execution_sharded(ctx, host, None, resident, e["padded_height"], claim, seed, jit_passes=k, ldt="fri",
                                                             profile=profile and it == repeats - 1)
        ctx.sync()
        ms.append(round(1e3 * (time.perf_counter() - t0), 1))
        used = stats.get("passes")
        stages = dict(stats["stage_ms"]) if stats.get("stage_ms") else stages
        if first is None:
            first = words
        same = same and words.size == first.size and bool((words == first).all())
    out["runs"].append({"jit_passes": k, "passes_used": used, "ms": ms, "same_proof": same,
                        **({"stage_ms_of_the_last_proof": {a: round(b, 1) for a, b in stages.items()}} if stages else {})})
    print(json.dumps(out["runs"][-1]), file=sys.stderr, flush=True)
    ctx.trim()
print(json.dumps(out))
