"""The drop-in boundary without a GPU: the hipcc-built libtriton_hip.so (gfx950 code objects inside, cross-compiled in the
GPU-less container) and the C++ host library load, and they export every function include/triton_hip.h and
triton_vm_amd/host/triton_host.hpp declare.  No compute call is made here -- that is what the `-m gpu` tests do."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_product_library_exports_every_declared_entry_point():
    from tools import gen_rust_ffi as gen
    from triton_vm_amd import build

    lib = ctypes.CDLL(build.build())                         # builds only when the sources are newer than the library
    names = [name for _, name, _ in gen.declarations(open(gen.HEADER).read())]
    assert len(names) >= 60 and len(set(names)) == len(names)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.tvm_abi_version.restype = ctypes.c_uint32
    header = open(gen.HEADER).read()
    declared = re.search(r"#define\s+TVM_ABI_VERSION\s+(\d+)", header)
    if declared:                                             # a host call, no device needed
        assert lib.tvm_abi_version() == int(declared.group(1))
    lib.tvm_status_string.restype = ctypes.c_char_p
    assert lib.tvm_status_string(0) and lib.tvm_status_string(1)


def test_host_library_exports_its_entry_points():
    from triton_vm_amd import build

    host = ctypes.CDLL(build.build_host())
    declared = re.findall(r"\b(tvmh_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "triton_vm_amd", "host", "triton_host.hpp")).read())
    assert declared, "the host header declares its C entry points"
    for name in set(declared):
        assert hasattr(host, name), name


def test_there_is_no_cpu_fallback(tmp_path):
    """a missing library and a missing GPU are errors, never a silent CPU path"""
    import pytest
    import torch

    from triton_vm_amd import capi

    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "libtriton_hip.so"))
    if not torch.cuda.is_available():   # the GPU-less container: the product library loads, a context cannot be created
        with pytest.raises(capi.TritonHipError):
            capi.Context(device=0)
