"""triton_vm_amd -- MI355X (gfx950) backend for the hot path of triton_vm::stark::Prover::prove.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels + the C ABI declared in
``include/triton_hip.h``) and a thin host-side mirror of the reference interface used by tests and
bench.py.  There is no CPU fallback anywhere in this package.
"""
from .capi import Context, DeviceBuffer, Domain, TritonHipError, load_library  # noqa: F401
from .arithmetic_domain import ArithmeticDomain  # noqa: F401
from .master_table import MasterTable  # noqa: F401

__all__ = ["Context", "DeviceBuffer", "Domain", "TritonHipError", "load_library", "ArithmeticDomain", "MasterTable"]
