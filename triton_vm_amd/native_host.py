"""Python binding of the C++ host side (triton_vm_amd/host/triton_host.cpp): `NativeProver.prove` runs the C++
`Prover::prove` mirror on device-resident traces and returns its transcript.  Used by the tests (the C++ and the
Python mirrors must produce the same transcript) and by bench.py, whose timed step is the C++ host by default."""
import ctypes as C

import numpy as np

from .build import build_host


def load_host_library(backend_path=None, out=None):
    lib = C.CDLL(build_host(backend_path, out))
    lib.tvmh_prove.restype = C.c_int32
    lib.tvmh_prove.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64,
                               C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    lib.tvmh_prove_execution.restype = C.c_int32
    lib.tvmh_prove_execution.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p,
                                         C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p,
                                         C.c_uint64]
    lib.tvmh_stir_prove.restype = C.c_int32
    from .capi import Domain

    lib.tvmh_stir_prove.argtypes = [C.c_void_p, Domain, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    lib.tvmh_stir_parameters.restype = C.c_uint64
    lib.tvmh_stir_parameters.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.tvmh_prove_execution_sharded.restype = C.c_int32
    lib.tvmh_prove_execution_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                                 C.c_uint64, C.POINTER(C.c_uint64), C.c_uint32, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    lib.tvmh_prove_sharded.restype = C.c_int32
    lib.tvmh_prove_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    lib.tvmh_set_option.restype = None
    lib.tvmh_set_option.argtypes = [C.c_uint32, C.c_uint64]
    lib.tvmh_get_option.restype = C.c_uint64
    lib.tvmh_get_option.argtypes = [C.c_uint32]
    lib.tvmh_local_comms_create.restype = C.c_int32
    lib.tvmh_local_comms_create.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.tvmh_local_comms_destroy.restype = None
    lib.tvmh_local_comms_destroy.argtypes = [C.c_void_p]
    lib.tvmh_local_comms_abort.restype = None
    lib.tvmh_local_comms_abort.argtypes = [C.c_void_p]
    lib.tvmh_local_comms_report.restype = C.c_uint64
    lib.tvmh_local_comms_report.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    return lib


OPTION_EXACT_AIR = 1   # tvmh_set_option: prove_execution evaluates the AIR row by row instead of in valid-trace mode
OPTION_SHARE_REPLICATED_TABLES = 2   # in-process ranks use ONE copy of the replicated trace-side tables (triton_host.hpp)
OPTION_TRACE = 3   # host wall time of the steps of prove_execution on stderr
OPTION_COLUMN_SPLIT = 4   # k > 0: the sharded prover splits the inverse transforms by columns, coefficients exchanged in k chunks


# ---- communicators for the sharded C++ host (triton_host.hpp: tvmh_comm) -----------------------------------------------
_COLLECTIVE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_MARK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_char_p)
_ABORT = C.CFUNCTYPE(None, C.c_void_p)
_SHARE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p)
_ASYNC = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32)
_WAIT = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32)


class CommStruct(C.Structure):
    """struct tvmh_comm"""
    _fields_ = [("self", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32), ("all_gather", _COLLECTIVE), ("all_to_all", _COLLECTIVE),
                ("begin", _HOOK), ("mark", _MARK), ("end", _HOOK), ("abort", _ABORT), ("share", _SHARE),
                ("all_gather_async", _ASYNC), ("wait", _WAIT)]


class LocalComms:
    """`world` communicators between the contexts of THIS process, one proving thread per rank (tvmh_local_comms_create):
    collectives are a rendezvous plus device-to-device copies.  lockstep: one rank computes at a time and the per-rank,
    per-stage compute time is recorded (`report()`), which on one GPU measures what the ranks of a real run do concurrently."""

    def __init__(self, host_lib, world, lockstep=False):
        self.lib, self.world = host_lib, world
        arr = (C.c_void_p * world)()
        if host_lib.tvmh_local_comms_create(world, 1 if lockstep else 0, arr) != 0:
            raise RuntimeError("tvmh_local_comms_create failed")
        self.ptrs = [arr[r] for r in range(world)]

    def report(self):
        import json

        buf = C.create_string_buffer(1 << 16)
        self.lib.tvmh_local_comms_report(self.ptrs[0], buf, len(buf))
        return json.loads(buf.value.decode() or "{}")

    def abort(self):
        """a rank failed: release the ranks that wait for it in a collective (they report a device error)"""
        if self.ptrs:
            self.lib.tvmh_local_comms_abort(self.ptrs[0])

    def close(self):
        if self.ptrs:
            self.lib.tvmh_local_comms_destroy(self.ptrs[0])
            self.ptrs = []


class RcclComm:
    """The production communicator: RCCL collectives on the context's stream (triton_vm_amd/host/rccl_comm.cpp).  One per
    rank; `unique_id` is drawn on rank 0 (RcclComm.unique_id) and handed to the other ranks by the launcher's channel."""

    _lib = None

    @classmethod
    def library(cls):
        if cls._lib is None:
            from .build import build_rccl

            lib = C.CDLL(build_rccl())
            lib.tvmh_rccl_unique_id.restype = C.c_int32
            lib.tvmh_rccl_unique_id.argtypes = [C.c_void_p]
            lib.tvmh_rccl_comm_create.restype = C.c_int32
            lib.tvmh_rccl_comm_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]
            lib.tvmh_rccl_comm_destroy.restype = None
            lib.tvmh_rccl_comm_destroy.argtypes = [C.c_void_p]
            lib.tvmh_rccl_last_error.restype = C.c_char_p
            cls._lib = lib
        return cls._lib

    @classmethod
    def unique_id(cls):
        out = np.zeros(128, np.uint8)
        if cls.library().tvmh_rccl_unique_id(out.ctypes.data) != 0:
            raise RuntimeError("ncclGetUniqueId: " + cls.library().tvmh_rccl_last_error().decode())
        return out

    def __init__(self, unique_id, rank, world, device):
        lib = self.library()
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        ptr = C.c_void_p()
        if lib.tvmh_rccl_comm_create(uid.ctypes.data, rank, world, device, C.byref(ptr)) != 0:
            raise RuntimeError("tvmh_rccl_comm_create: " + lib.tvmh_rccl_last_error().decode())
        self.ptr, self.rank, self.world = ptr.value, rank, world

    def close(self):
        if self.ptr:
            self.library().tvmh_rccl_comm_destroy(self.ptr)
            self.ptr = None


class CallbackComm:
    """A tvmh_comm whose collectives are Python callables over (send pointer, receive pointer, words) -- the CPU tests put
    torch.distributed's gloo backend behind it (the "device" buffers of the emulation are host memory)."""

    def __init__(self, rank, world, all_gather, all_to_all):
        def wrap(fn):
            def call(_self, _ctx, send, recv, words):
                try:
                    fn(send, recv, int(words))
                    return 0
                except Exception:   # noqa: BLE001 -- reported to the C++ host as a device error, with the traceback on stderr
                    import traceback

                    traceback.print_exc()
                    return 3
            return _COLLECTIVE(call)

        self._keep = (wrap(all_gather), wrap(all_to_all))
        self.struct = CommStruct(None, rank, world, self._keep[0], self._keep[1], _HOOK(), _MARK(), _HOOK(), _ABORT(), _SHARE(), _ASYNC(), _WAIT())
        self.ptr = C.addressof(self.struct)


def gloo_comm(dist):
    """CallbackComm over an initialised torch.distributed gloo group, for buffers in host memory (the CPU emulation)"""
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()

    def view(ptr, words):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(words,)))

    def all_gather(send, recv, words):
        dist.all_gather_into_tensor(view(recv, words * world), view(send, words).clone())

    def all_to_all(send, recv, words):   # gloo has no all-to-all on CPU tensors: one all-gather of everything, then this rank's blocks
        everything = torch.empty(world * world * words, dtype=torch.int64)
        dist.all_gather_into_tensor(everything, view(send, words * world).clone())
        out = view(recv, words * world)
        for r in range(world):
            out[r * words:(r + 1) * words] = everything[(r * world + rank) * words:(r * world + rank + 1) * words]

    return CallbackComm(rank, world, all_gather, all_to_all)


def prove_execution_sharded(ctx, host_lib, comm_ptr, aet, padded_height, claim, randomness_seed, security_level=160, log2_expansion=2,
                            ldt=None, jit_passes=0, split_tree_min_leaves=1 << 21, profile=False):
    """The C++ host's Prover::prove(claim, aet) over the ranks of a communicator (comm_ptr: a tvmh_comm*, or None for this
    process alone) and / or coset by coset (jit_passes; 0 = the reference's memory policy: cached, else as few passes as fit).
    -> (proof words, stats dict)"""
    import json

    from .master_table import aet_struct

    s, keep = aet_struct(aet)
    log2 = padded_height.bit_length() - 1
    err, stats, n = C.create_string_buffer(512), C.create_string_buffer(1 << 14), C.c_uint64(0)
    out = getattr(_PROOF_BUFFERS, "buffer", None)   # one buffer per proving thread, kept between calls (its pages are mapped)
    if out is None:
        out = np.empty(1 << 20, np.uint64)
    while True:
        rc = host_lib.tvmh_prove_execution_sharded(ctx.handle, comm_ptr, jit_passes, split_tree_min_leaves, C.addressof(s), log2, security_level,
                                                   log2_expansion, {"fri": 0, "stir": 1, None: 2}[ldt], bytes(randomness_seed),
                                                   claim.program_digest.ctypes.data, claim.input.ctypes.data, claim.input.size,
                                                   claim.output.ctypes.data, claim.output.size, out.ctypes.data, out.size, C.byref(n),
                                                   1 if profile else 0, stats, len(stats), err, len(err))
        if rc != 0:
            raise NativeHostError(rc, f"tvmh_prove_execution_sharded failed ({rc}): {err.value.decode()}")
        if n.value <= out.size:
            _PROOF_BUFFERS.buffer = out
            return out[:n.value].copy(), json.loads(stats.value.decode() or "{}")
        out = np.empty(int(n.value), np.uint64)


def prove_sharded(ctx, host_lib, comm_ptr, params, d_main_trace, d_main_randomizers, d_aux_trace, d_aux_randomizers, quotient_randomizer,
                  jit_passes=1, split_tree_min_leaves=1 << 21, stir_security_level=160):
    """The hot path alone (as NativeProver.prove) through the sharded / coset-wise C++ prover -> proof words"""
    qr = np.ascontiguousarray(quotient_randomizer, dtype=np.uint64).reshape(-1, 3)
    err, n = C.create_string_buffer(512), C.c_uint64(0)
    out = np.empty(1 << 20, np.uint64)
    log2 = params.padded_height.bit_length() - 1
    while True:
        rc = host_lib.tvmh_prove_sharded(ctx.handle, comm_ptr, jit_passes, split_tree_min_leaves, log2, params.h, params.num_collinearity_checks,
                                         params.log2_expansion, d_main_trace.ptr, d_main_randomizers.ptr, d_aux_trace.ptr, d_aux_randomizers.ptr,
                                         qr.ctypes.data, 1 if params.stir is not None else 0, stir_security_level, out.ctypes.data, out.size,
                                         C.byref(n), err, len(err))
        if rc != 0:
            raise NativeHostError(rc, f"tvmh_prove_sharded failed ({rc}): {err.value.decode()}")
        if n.value <= out.size:
            return out[:n.value].copy()
        out = np.empty(int(n.value), np.uint64)


class NativeHostError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def stir_parameters(host_lib, padded_height, security_level=160, log2_expansion=2):
    """Stark::stir's instance as the C++ host derives it -> dict"""
    out = np.zeros(128, np.uint64)
    n = host_lib.tvmh_stir_parameters(padded_height, security_level, log2_expansion, out.ctypes.data, out.size)
    if not n:
        raise RuntimeError("tvmh_stir_parameters failed")
    rounds = int(out[4])
    return dict(domain_length=int(out[0]), folding_factor=int(out[1]), final_num_in_domain_queries=int(out[2]), final_degree=int(out[3]),
                round_queries=[(int(out[5 + 2 * r]), int(out[6 + 2 * r])) for r in range(rounds)])


def stir_prove(ctx, host_lib, stir, d_codeword):
    """the C++ host's Stir::prove for the instance `stir` (low_degree_test.Stir) -> (first-round indices, proof words of a
    stream that holds only the STIR items)"""
    rounds = np.array(stir.round_queries, np.uint64).reshape(-1, 2)
    n_first = stir.num_first_round_queries()
    first, out = np.zeros(n_first, np.uint64), np.empty(1 << 20, np.uint64)
    err, n = C.create_string_buffer(512), C.c_uint64(0)
    rc = host_lib.tvmh_stir_prove(ctx.handle, stir.initial_domain.c(), stir.folding_factor, rounds.ctypes.data, len(rounds),
                                  stir.final_num_in_domain_queries, stir.final_degree, d_codeword.ptr, first.ctypes.data, out.ctypes.data,
                                  out.size, C.byref(n), err, len(err))
    if rc != 0 or n.value > out.size:
        raise RuntimeError(f"tvmh_stir_prove failed ({rc}): {err.value.decode()}")
    return [int(i) for i in first], out[:n.value].copy()


_PROOF_BUFFERS = __import__("threading").local()   # one buffer per proving thread: ctypes releases the GIL inside the C++ host,
#                                                     and the backend's contract is one context per proving thread


def prove_execution(ctx, host_lib, aet, padded_height, claim, randomness_seed, security_level=160, log2_expansion=2, ldt=None):
    """The C++ host's Prover::prove(claim, aet) (triton_vm::prove_execution): fill, pad, extend and the hot path on the
    device, the seeded randomness and the transcript in C++.  aet: the arrays master_table.fill takes (host arrays, or
    master_table.aet_to_device's device-resident copy).  ldt: "fri", "stir" or None = Stark::ldt's rule (STIR from 2^16 padded
    rows on, stark.rs:1944-1951).  -> the proof words"""
    from .master_table import aet_struct

    s, keep = aet_struct(aet)
    log2 = padded_height.bit_length() - 1
    err, n = C.create_string_buffer(512), C.c_uint64(0)
    out = getattr(_PROOF_BUFFERS, "buffer", None)
    if out is None:
        out = np.empty(1 << 20, np.uint64)   # a 2^20-row proof is ~0.3 M words
    while True:
        rc = host_lib.tvmh_prove_execution(ctx.handle, C.addressof(s), log2, security_level, log2_expansion, {"fri": 0, "stir": 1, None: 2}[ldt],
                                           bytes(randomness_seed),
                                           claim.program_digest.ctypes.data, claim.input.ctypes.data, claim.input.size,
                                           claim.output.ctypes.data, claim.output.size, out.ctypes.data, out.size, C.byref(n), err, len(err))
        if rc != 0:
            raise RuntimeError(f"tvmh_prove_execution failed ({rc}): {err.value.decode()}")
        if n.value <= out.size:
            _PROOF_BUFFERS.buffer = out               # (kept: its pages are mapped by now)
            return out[:n.value].copy()
        out = np.empty(int(n.value), np.uint64)   # the proof did not fit: grow and run again


class NativeProver:
    """Same constructor data as triton_vm_amd.prover.Prover (device traces, device randomizers, the quotient randomizer,
    the claim); `prove()` returns the proof: the words of the reference's `Proof`."""

    def __init__(self, ctx, host_lib, params, d_main_trace, d_main_randomizers, d_aux_trace, d_aux_randomizers,
                 quotient_randomizer, claim=None):
        from .proof_stream import Claim

        self.ctx, self.lib, self.p = ctx, host_lib, params
        self.bufs = (d_main_trace, d_main_randomizers, d_aux_trace, d_aux_randomizers)  # keep them alive
        self.qr = np.ascontiguousarray(quotient_randomizer, dtype=np.uint64).reshape(-1, 3)
        assert self.qr.shape[0] == params.num_quotient_randomizers
        self.claim = claim or Claim()
        self.capacity = 1 << 21   # words; a 2^20-row proof (173 opened rows of 3 tables + authentication nodes) is ~0.4 M words
        self.out = np.empty(self.capacity, np.uint64)

    def prove(self, parse=True):
        p, claim = self.p, self.claim
        log2 = p.padded_height.bit_length() - 1
        err = C.create_string_buffer(512)
        n = C.c_uint64(0)
        while True:
            rc = self.lib.tvmh_prove(self.ctx.handle, log2, p.h, p.num_collinearity_checks, p.log2_expansion, self.bufs[0].ptr, self.bufs[1].ptr,
                                     self.bufs[2].ptr, self.bufs[3].ptr, self.qr.ctypes.data, claim.program_digest.ctypes.data,
                                     claim.input.ctypes.data, claim.input.size, claim.output.ctypes.data, claim.output.size,
                                     1 if p.stir is not None else 0, self.out.ctypes.data, self.capacity, C.byref(n), err, len(err))
            if rc != 0:
                raise RuntimeError(f"tvmh_prove failed ({rc}): {err.value.decode()}")
            if n.value <= self.capacity:
                break
            self.capacity = int(n.value)      # the proof did not fit: grow and run again
            self.out = np.empty(self.capacity, np.uint64)
        return self.out[:n.value].copy() if parse else None
