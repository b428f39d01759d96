"""DIAGNOSTIC: some GPU boxes run the AIR stage three times slower (183 ms instead of 60 ms at 2^20 rows) while every other
stage is normal.  Print the stage times next to what could tell such a box apart (clocks, power, partition modes)."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from triton_vm_amd import Context  # noqa: E402
from triton_vm_amd.prover import Prover, StarkParameters  # noqa: E402


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return repr(e)


ctx = Context(device=0)
p = Prover(ctx, StarkParameters(20), seed=1)
p.prove()
p.timings = {}
p.prove(profile=True)
print({k: round(v, 1) for k, v in p.timings.items() if k in ("main LDE", "main Merkle", "AIR quotients", "FRI")}, flush=True)
print("during-idle:", sh("rocm-smi --showclocks --showpower --showcomputepartition --showmemorypartition --showperflevel 2>&1 | grep -v '^=\\|^$' | head -30"))
for f in ("current_compute_partition", "current_memory_partition", "pp_power_profile_mode", "power_dpm_force_performance_level"):
    print(f, sh(f"cat /sys/class/drm/card*/device/{f} 2>/dev/null | head -12"))
print("env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "ROC", "GPU_", "AMD_"))})
print("driver:", sh("cat /sys/module/amdgpu/version 2>/dev/null; uname -r"))
print("rocminfo:", sh("rocminfo | grep -i 'Compute Unit\\|Max Waves\\|Cacheline\\|Max Clock\\|Name:.*gfx\\|Marketing' | head -12"))
# the AIR stage alone, back to back, with the clocks sampled while it runs
import threading
samples = []
stop = False
def sampler():
    while not stop:
        samples.append(sh("rocm-smi --showclocks 2>&1 | grep -i 'sclk\\|mclk' | head -2 | tr '\\n' ' '"))
        time.sleep(0.2)
t = threading.Thread(target=sampler); t.start()
t0 = time.perf_counter()
for _ in range(10):
    p.timings = {}
    p.prove(profile=True)
stop = True; t.join()
print("10 proofs:", round(1e3 * (time.perf_counter() - t0) / 10, 1), "ms each; AIR of the last:", round(p.timings["AIR quotients"], 1))
print("clock samples under load:", samples[:6])
