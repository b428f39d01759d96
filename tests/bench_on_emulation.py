"""TEST-ONLY launcher: bench.py's control flow (argument handling, rank spawning, timing contract, collectives of the sharded
proof, the JSON line) on CPU -- gloo instead of RCCL and the fiber emulation of the kernel sources instead of the GPU.  bench.py
itself has no such switch: this script replaces its three process set-up hooks and calls its main()."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: tests/conftest.py)

import bench  # noqa: E402


def _init_distributed(local_rank):
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    return dist, torch.device("cpu")


def _make_context(local_rank):
    from tests.emu_fixture import emu_context

    return emu_context()


bench.make_context = _make_context
bench.init_distributed = _init_distributed
bench.visible_devices = lambda: 64
bench.USE_CPP_HOST = False      # the C++ host library is linked against the product library

if __name__ == "__main__":
    bench.main()
