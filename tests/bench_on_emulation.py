"""TEST-ONLY launcher: bench.py's control flow (argument handling, rank spawning, timing contract, collectives of the sharded
proof, the JSON line) on CPU -- gloo instead of RCCL and the fiber emulation of the kernel sources instead of the GPU.  bench.py
itself has no such switch: this script replaces its process set-up hooks (context, process group, host library, communicator) and
calls its main()."""
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


def _load_host_library():
    """the C++ host linked against the emulation instead of the product library"""
    from tests.emu.build_emu import build as build_emu
    from triton_vm_amd import native_host

    backend = build_emu()
    return native_host.load_host_library(backend, os.path.join(os.path.dirname(backend), "libtriton_host_emu.so"))


def _make_comm(dist, device, rank, world, local_rank):
    """gloo behind the sharded C++ host's communicator table (the emulation's "device" buffers are host memory)"""
    from triton_vm_amd import native_host

    return native_host.gloo_comm(dist)


bench.make_context = _make_context
bench.init_distributed = _init_distributed
bench.visible_devices = lambda: 64
bench.load_host_library = _load_host_library
bench.make_comm = _make_comm

if __name__ == "__main__":
    bench.main()
