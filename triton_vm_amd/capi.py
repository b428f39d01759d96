"""ctypes binding of the C ABI in include/triton_hip.h.

This is plumbing: it adds nothing to the library and holds no algorithm.  The product library is
``libtriton_hip.so`` next to this file, built by hipcc for gfx950 (``triton_vm_amd.build``).  There
is no CPU fallback: loading fails loudly when the library is missing, and creating a context fails
loudly when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libtriton_hip.so")

OK, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_DEVICE, ERR_UNSUPPORTED = range(5)

u64p = C.POINTER(C.c_uint64)


class Domain(C.Structure):
    """(offset, generator, length) exactly as ArithmeticDomain holds them (arithmetic_domain.rs:34-47)."""
    _fields_ = [("offset", C.c_uint64), ("generator", C.c_uint64), ("length", C.c_uint64)]

    def __repr__(self):
        return f"Domain(offset={self.offset}, generator={self.generator}, length={self.length})"


class Aet(C.Structure):
    """tvm_aet (include/triton_hip.h): the algebraic execution trace as plain host arrays"""
    _fields_ = [("program_words", C.c_void_p), ("instruction_multiplicities", C.c_void_p), ("program_len", C.c_uint64),
                ("processor_trace", C.c_void_p), ("processor_len", C.c_uint64),
                ("op_stack_trace", C.c_void_p), ("op_stack_len", C.c_uint64),
                ("ram_trace", C.c_void_p), ("ram_len", C.c_uint64),
                ("bezout_coefficients_0", C.c_void_p), ("bezout_coefficients_1", C.c_void_p), ("num_ram_pointers", C.c_uint64),
                ("program_hash_trace", C.c_void_p), ("program_hash_len", C.c_uint64),
                ("sponge_trace", C.c_void_p), ("sponge_len", C.c_uint64),
                ("hash_trace", C.c_void_p), ("hash_len", C.c_uint64),
                ("u32_entries", C.c_void_p), ("u32_len", C.c_uint64),
                ("cascade_entries", C.c_void_p), ("cascade_len", C.c_uint64),
                ("lookup_multiplicities", C.c_void_p)]


class TritonHipError(RuntimeError):
    def __init__(self, status, what):
        super().__init__(f"libtriton_hip status {status}: {what}")
        self.status = status


_SIGNATURES = {
    "tvm_abi_version": (C.c_int32, []),
    "tvm_ctx_create": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "tvm_ctx_destroy": (None, [C.c_void_p]),
    "tvm_last_error": (C.c_char_p, [C.c_void_p]),
    "tvm_status_string": (C.c_char_p, [C.c_int32]),
    "tvm_sync": (C.c_int32, [C.c_void_p]),
    "tvm_malloc": (C.c_int32, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tvm_free": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "tvm_ctx_trim": (C.c_int32, [C.c_void_p]),
    "tvm_ctx_set_option": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64]),
    "tvm_ctx_set_memory_limit": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "tvm_ctx_memory_held": (C.c_int32, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "tvm_ctx_memory_info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "tvm_memcpy_h2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tvm_memcpy_d2h": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tvm_memcpy_d2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tvm_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "tvm_ctx_side_stream": (C.c_void_p, [C.c_void_p]),
    "tvm_side_begin": (C.c_int32, [C.c_void_p]),
    "tvm_side_memcpy_d2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tvm_side_mark": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "tvm_side_wait": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "tvm_side_sync": (C.c_int32, [C.c_void_p]),
    "tvm_timer_start": (C.c_int32, [C.c_void_p]),
    "tvm_timer_stop": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float)]),
    "tvm_synthetic_fill": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "tvm_field_op": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "tvm_evaluate": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, Domain, C.c_void_p]),
    "tvm_interpolate": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, Domain, C.c_void_p]),
    "tvm_ntt": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64]),
    "tvm_intt": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64]),
    "tvm_lde_table": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                  Domain, Domain, C.POINTER(C.c_void_p)]),
    "tvm_lde_column_coefficients": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, Domain, C.c_uint64, C.c_uint64,
                                                C.c_void_p]),
    "tvm_lde_table_begin": (C.c_int32, [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, Domain, Domain, C.POINTER(C.c_void_p)]),
    "tvm_lde_table_add_columns": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, Domain,
                                              Domain]),
    "tvm_lde_table_end": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "tvm_table_free": (None, [C.c_void_p, C.c_void_p]),
    "tvm_table_num_rows": (C.c_uint64, [C.c_void_p]),
    "tvm_table_num_columns": (C.c_uint64, [C.c_void_p]),
    "tvm_table_field_kind": (C.c_int32, [C.c_void_p]),
    "tvm_table_export_row_major": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_table_reveal_rows": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_hash_rows": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_merkle_tree": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_table_merkle_tree": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_codeword_merkle_tree": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_out_of_domain_rows": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                           C.c_uint64, Domain, C.c_void_p, C.c_uint32, C.c_void_p]),
    "tvm_weighted_sum_of_columns": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                                C.c_uint64, Domain, C.c_void_p, C.c_void_p]),
    "tvm_xfe_add_assign": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "tvm_xfe_linear_combination": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tvm_evaluate_at_points": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "tvm_evaluate_polys_at_points": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]),
    "tvm_fill_derived_main_columns": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "tvm_fill_derived_aux_columns": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_fill_main_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_pad_main_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_extend_aux_table": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_all_quotients_combined": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, Domain, Domain, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    "tvm_air_class_cosets": (C.c_int32, [C.c_void_p, C.c_void_p, Domain, C.c_void_p]),
    "tvm_air_class_values": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, Domain, Domain, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_coset_values_to_coefficients": (C.c_int32, [C.c_void_p, Domain, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_quotient_segments": (C.c_int32, [C.c_void_p, C.c_void_p, Domain, Domain, C.c_void_p, C.c_uint64, C.c_uint64,
                                          C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64]),
    "tvm_table_linear_combination": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tvm_deep_codeword": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), Domain, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "tvm_fri_split_and_fold": (C.c_int32, [C.c_void_p, C.c_void_p, Domain, C.c_void_p, C.c_void_p]),
    "tvm_stir_merkle_tree": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "tvm_fold_polynomial": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "tvm_stir_next_polynomial": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                             Domain, C.c_void_p]),
    "tvm_host_xfe_interpolate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "tvm_xfe_interpolate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "tvm_scatter_strided": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tvm_gather_elements": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "tvm_fri_commit_phase": (C.c_int32, [C.c_void_p, C.c_void_p, Domain, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "tvm_gather_elements_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_verifier_row_digests": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tvm_verifier_deep_values": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, Domain,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_host_tip5_permutation": (None, [C.c_void_p]),
    "tvm_host_sponge_pad_and_absorb": (None, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "tvm_host_xfe_mul": (None, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tvm_host_xfe_inv": (None, [C.c_void_p, C.c_void_p]),
    "tvm_host_xfe_powers": (None, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tvm_host_xfe_poly_eval": (None, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]),
    "tvm_host_stdrng_elements": (None, [C.c_char_p, C.c_uint64, C.c_void_p]),
    "tvm_stdrng_elements": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p]),
    "tvm_stdrng_streams": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "tvm_bezout_coefficients": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "tvm_host_air_constraints": (C.c_int32, [C.c_void_p] * 6),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library(path=None):
    """dlopen the C-ABI library and attach signatures.  Raises if it is missing: there is no fallback."""
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python -m triton_vm_amd.build` (needs hipcc, gfx950); "
            "triton_vm_amd has no CPU fallback")
    lib = C.CDLL(path)
    missing = [name for name in _SIGNATURES if not hasattr(lib, name)]
    if missing and path == DEFAULT_LIB:
        # a library from before the header grew: rebuild the PRODUCT library with hipcc (still no fallback: without
        # the toolchain this raises) and load the fresh file under a new handle
        from .build import build

        rebuilt = build(force=True)
        fresh = rebuilt + f".{os.getpid()}.so"
        import shutil

        shutil.copyfile(rebuilt, fresh)  # dlopen caches by path: the stale image is still mapped under the old name
        try:
            lib = C.CDLL(fresh)
        finally:
            os.unlink(fresh)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    return lib


class DeviceBuffer:
    """A device allocation of uint64 words owned by a Context."""

    def __init__(self, ctx, n_words):
        self.ctx = ctx
        self.n_words = int(n_words)
        p = C.c_void_p()
        ctx._check(ctx.lib.tvm_malloc(ctx.handle, self.n_words * 8, C.byref(p)), "tvm_malloc")
        self.ptr = p.value

    def upload(self, array):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        assert a.size <= self.n_words
        self.ctx._check(self.ctx.lib.tvm_memcpy_h2d(self.ctx.handle, self.ptr, a.ctypes.data, a.size * 8), "h2d")
        return self

    def download(self, shape=None):
        out = np.empty(self.n_words, np.uint64)
        self.ctx._check(self.ctx.lib.tvm_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.n_words * 8), "d2h")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            self.ctx.lib.tvm_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device context (one HIP stream); create one per proving thread."""

    def __init__(self, device=0, stream=None, lib=None):
        self.lib = lib if lib is not None else load_library()
        h = C.c_void_p()
        st = self.lib.tvm_ctx_create(device, stream, C.byref(h))
        if st != OK:
            raise TritonHipError(st, "tvm_ctx_create failed -- is an MI355X visible? "
                                 + self.lib.tvm_status_string(st).decode())
        self.handle = h.value

    def _check(self, status, what):
        if status != OK:
            msg = self.lib.tvm_last_error(self.handle).decode()
            raise TritonHipError(status, f"{what}: {self.lib.tvm_status_string(status).decode()} ({msg})")

    def close(self):
        if self.handle:
            self.lib.tvm_ctx_destroy(self.handle)
            self.handle = None

    def sync(self):
        self._check(self.lib.tvm_sync(self.handle), "tvm_sync")

    def set_memory_limit(self, n_bytes):
        """cap the device bytes this context may hold (0 = no cap); beyond it allocations raise status 2"""
        self._check(self.lib.tvm_ctx_set_memory_limit(self.handle, n_bytes), "tvm_ctx_set_memory_limit")

    def memory_held(self):
        n = C.c_size_t()
        self._check(self.lib.tvm_ctx_memory_held(self.handle, C.byref(n)), "tvm_ctx_memory_held")
        return n.value

    def memory_info(self):
        """-> (bytes this context could still obtain, the device's total bytes)"""
        a, t = C.c_size_t(), C.c_size_t()
        self._check(self.lib.tvm_ctx_memory_info(self.handle, C.byref(a), C.byref(t)), "tvm_ctx_memory_info")
        return a.value, t.value

    def assume_valid_trace(self, on=True):
        """TVM_OPTION_AIR_VALID_TRACE: the tables come from a valid execution -- the quotient evaluation may use the
        degree bounds of the constraint quotients (half the rows + interpolation; identical on valid traces)"""
        self._check(self.lib.tvm_ctx_set_option(self.handle, 1, 1 if on else 0), "tvm_ctx_set_option")

    def air_fork_max_workgroups(self, n=256):
        """TVM_OPTION_AIR_FORK_MAX_WORKGROUPS: quotient domains of at most n workgroups of 256 rows run the parts of the AIR on
        four streams side by side, and valid-trace mode evaluates them row by row (0: never; the library's default is 256)"""
        self._check(self.lib.tvm_ctx_set_option(self.handle, 5, n), "tvm_ctx_set_option")

    def trim(self):
        """give the cached device blocks back to the driver"""
        self._check(self.lib.tvm_ctx_trim(self.handle), "tvm_ctx_trim")

    def timer_start(self):
        self._check(self.lib.tvm_timer_start(self.handle), "tvm_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._check(self.lib.tvm_timer_stop(self.handle, C.byref(ms)), "tvm_timer_stop")
        return ms.value

    def synthetic(self, n_words, seed):
        buf = DeviceBuffer(self, n_words)
        self._check(self.lib.tvm_synthetic_fill(self.handle, buf.ptr, n_words, seed), "tvm_synthetic_fill")
        return buf

    def alloc(self, n_words):
        return DeviceBuffer(self, n_words)

    def to_device(self, array):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        return DeviceBuffer(self, max(a.size, 1)).upload(a)
