"""Host mirror of the reference's MasterTable trait (/root/reference/triton-vm/src/table/master_table.rs:190-610)
for the methods on the hot path.  The trace, the randomizers and the extended table live in HBM."""
import ctypes as C

import numpy as np


class MasterTable:
    """A padded master main (field_kind 1) or auxiliary (field_kind 3) table on the device.

    trace:        numpy [n_cols, n_rows(, 3)] -- column-major like the reference (master_table.rs:888,1013)
    randomizers:  numpy [n_cols, h(, 3)]      -- coefficients of trace_randomizer_for_column (:423-434),
                  produced by the host RNG (they never come from this library)
    """

    def __init__(self, ctx, trace, randomizers, trace_domain, quotient_domain, ldt_domain, field_kind=1):
        self.ctx, self.fk = ctx, field_kind
        trace = np.ascontiguousarray(trace, dtype=np.uint64)
        randomizers = np.ascontiguousarray(randomizers, dtype=np.uint64)
        self.n_cols, self.n_rows = trace.shape[0], trace.shape[1]
        self.num_trace_randomizers = randomizers.shape[1]
        self.trace_domain, self.quotient_domain, self.ldt_domain = trace_domain, quotient_domain, ldt_domain
        self.d_trace = ctx.to_device(trace)
        self.d_randomizers = ctx.to_device(randomizers)
        self._table = None

    @classmethod
    def from_device(cls, ctx, d_trace, d_randomizers, n_cols, n_rows, num_trace_randomizers, trace_domain, quotient_domain,
                    ldt_domain, field_kind=1, into=None):
        """a table whose trace [n_cols][n_rows](, [3]) and randomizers [n_cols][h](, [3]) are already in HBM"""
        mt = into if into is not None else cls.__new__(cls)
        mt.ctx, mt.fk, mt.n_cols, mt.n_rows, mt.num_trace_randomizers = ctx, field_kind, n_cols, n_rows, num_trace_randomizers
        mt.trace_domain, mt.quotient_domain, mt.ldt_domain, mt._table = trace_domain, quotient_domain, ldt_domain, None
        mt.d_trace, mt.d_randomizers = d_trace, d_randomizers
        return mt

    # master_table.rs:215-222
    def evaluation_domain(self):
        return self.quotient_domain if self.quotient_domain.length > self.ldt_domain.length else self.ldt_domain

    # master_table.rs:258-322
    def maybe_low_degree_extend_all_columns(self):
        h = C.c_void_p()
        ev = self.evaluation_domain()
        self.clear_cache()  # the old table's block goes back to the context's pool and is reused right away
        self.ctx._check(self.ctx.lib.tvm_lde_table(self.ctx.handle, self.fk, self.d_trace.ptr, self.n_rows, self.n_cols,
                                                   self.d_randomizers.ptr, self.num_trace_randomizers,
                                                   self.trace_domain.c(), ev.c(), C.byref(h)), "tvm_lde_table")
        self._table = h.value

    def low_degree_extend_by_column_blocks(self, n_blocks, n_chunks=1, exchange=None):
        """The same table as maybe_low_degree_extend_all_columns, built the way the COLUMN sharding builds it (SURVEY 8(e);
        include/triton_hip.h: tvm_lde_column_coefficients ...): the virtual columns are dealt to `n_blocks` owners in equal blocks
        (the last one short), every owner's block is interpolated on its own and cut into `n_chunks` chunks, `exchange(chunk
        buffers)` stands for the all-gather, and the table is assembled chunk by chunk from the gathered coefficients.  One
        process plays every owner here (tests; the sharded C++ host does it over a communicator)."""
        lib, ctx, n = self.ctx.lib, self.ctx, self.n_rows
        W = self.n_cols * self.fk
        cpc = -(-W // (n_blocks * n_chunks))          # virtual columns per (owner, chunk)
        per = cpc * n_chunks
        ev = self.evaluation_domain()
        self.clear_cache()
        blobs = []
        for b in range(n_blocks):
            first, count = min(b * per, W), max(0, min(W - b * per, per))
            blob = ctx.alloc(per * n)
            ctx._check(lib.tvm_lde_column_coefficients(ctx.handle, self.fk, self.d_trace.ptr, n, self.n_cols, self.trace_domain.c(), first, count,
                                                       blob.ptr), "tvm_lde_column_coefficients")
            blobs.append(blob)
        if exchange is not None:
            blobs = exchange(blobs)
        h = C.c_void_p()
        ctx._check(lib.tvm_lde_table_begin(ctx.handle, self.fk, n, self.n_cols, self.num_trace_randomizers, self.trace_domain.c(), ev.c(),
                                           C.byref(h)), "tvm_lde_table_begin")
        self._table = h.value
        for c in range(n_chunks):
            for b in range(n_blocks):
                first = b * per + c * cpc
                count = max(0, min(W - first, cpc))
                if not count:
                    continue
                ctx._check(lib.tvm_lde_table_add_columns(ctx.handle, self._table, blobs[b].ptr + 8 * c * cpc * n, first, count,
                                                         self.d_randomizers.ptr, self.num_trace_randomizers, self.trace_domain.c(), ev.c()),
                           "tvm_lde_table_add_columns")
        ctx._check(lib.tvm_lde_table_end(ctx.handle, self._table), "tvm_lde_table_end")
        for blob in blobs:
            blob.free()

    def clear_cache(self):
        if self._table:
            self.ctx.lib.tvm_table_free(self.ctx.handle, self._table)
            self._table = None

    def _need_table(self):
        if not self._table:
            raise RuntimeError("low-degree extend first (maybe_low_degree_extend_all_columns)")
        return self._table

    def low_degree_extended_table(self):
        """The reference's row-major Array2 [rows, n_cols(, 3)] (master_table.rs:304-305), on the host."""
        t = self._need_table()
        rows = self.ctx.lib.tvm_table_num_rows(t)
        buf = self.ctx.alloc(rows * self.n_cols * self.fk)
        self.ctx._check(self.ctx.lib.tvm_table_export_row_major(self.ctx.handle, t, buf.ptr), "export")
        shape = (rows, self.n_cols) + ((3,) if self.fk == 3 else ())
        return buf.download(shape)

    # master_table.rs:455-468
    def hash_all_ldt_domain_rows(self):
        n = self.ldt_domain.length
        d = self.ctx.alloc(5 * n)
        self.ctx._check(self.ctx.lib.tvm_hash_rows(self.ctx.handle, self._need_table(), n, d.ptr), "tvm_hash_rows")
        return d.download((n, 5))

    # master_table.rs:443-453; returns the node array [2n, 5] (node 1 = root)
    def merkle_tree(self):
        n = self.ldt_domain.length
        d = self.ctx.alloc(10 * n)
        self.ctx._check(self.ctx.lib.tvm_table_merkle_tree(self.ctx.handle, self._need_table(), n, d.ptr), "merkle")
        return d.download((2 * n, 5))

    # master_table.rs:548-555 (cached branch)
    def reveal_rows(self, row_indices):
        idx = np.ascontiguousarray(row_indices, dtype=np.uint64)
        out = np.empty((idx.size, self.n_cols * self.fk), np.uint64)
        self.ctx._check(self.ctx.lib.tvm_table_reveal_rows(self.ctx.handle, self._need_table(), self.ldt_domain.length,
                                                           idx.ctypes.data, idx.size, out.ctypes.data), "reveal_rows")
        return out.reshape((idx.size, self.n_cols) + ((3,) if self.fk == 3 else ()))

    # master_table.rs:348-390, for several indeterminates at once -> [n_points, n_cols, 3]
    def out_of_domain_rows(self, points, first_col=0, n_cols=None):
        """all columns, or the n_cols columns from first_col on (a rank's share when the columns are split)"""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 3)
        n_cols = self.n_cols - first_col if n_cols is None else n_cols
        out = np.empty((pts.shape[0], n_cols, 3), np.uint64)
        if n_cols == 0:
            return out
        word = 8 * self.fk
        self.ctx._check(self.ctx.lib.tvm_out_of_domain_rows(
            self.ctx.handle, self.fk, self.d_trace.ptr + first_col * self.n_rows * word, self.n_rows, n_cols,
            self.d_randomizers.ptr + first_col * self.num_trace_randomizers * word, self.num_trace_randomizers,
            self.trace_domain.c(), pts.ctypes.data, pts.shape[0], out.ctypes.data), "ood rows")
        return out

    def out_of_domain_row(self, indeterminate):
        return self.out_of_domain_rows([indeterminate])[0]

    # master_table.rs:512-542 -> DeviceBuffer of 2*n_rows XFE coefficients
    def weighted_sum_of_columns(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.uint64).reshape(self.n_cols, 3)
        poly = self.ctx.alloc(2 * self.n_rows * 3)
        self.ctx._check(self.ctx.lib.tvm_weighted_sum_of_columns(
            self.ctx.handle, self.fk, self.d_trace.ptr, self.n_rows, self.n_cols, self.d_randomizers.ptr,
            self.num_trace_randomizers, self.trace_domain.c(), w.ctypes.data, poly.ptr), "weighted sum")
        return poly

    def __del__(self):
        try:
            self.clear_cache()
        except Exception:
            pass


def extend(ctx, d_main_trace, d_aux_trace, n_rows, challenges):
    """MasterMainTable::extend (/root/reference/triton-vm/src/table/master_table.rs:1006-1075) on the device: the 49
    cross-table-argument columns of the nine tables (tvm_extend_aux_table), then the 41 degree-lowering columns
    (tvm_fill_derived_aux_columns).  d_main_trace: DeviceBuffer [379][n_rows] (padded, derived columns filled);
    d_aux_trace: DeviceBuffer [91][n_rows][3], column 90 (the batch randomizer, host RNG) is left untouched;
    challenges: 63 XFE (Montgomery words)."""
    import ctypes as C

    from . import degree_lowering

    ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64).reshape(63, 3))
    if d_main_trace.n_words < 379 * n_rows or d_aux_trace.n_words < 91 * n_rows * 3:
        raise ValueError("the traces need 379 main and 91 auxiliary columns")
    ctx._check(ctx.lib.tvm_extend_aux_table(ctx.handle, d_main_trace.ptr, d_aux_trace.ptr, n_rows,
                                            ch.ctypes.data_as(C.c_void_p)), "tvm_extend_aux_table")
    degree_lowering.fill_derived_aux_columns(ctx, d_main_trace, d_aux_trace, n_rows, ch)


TABLE_ORDER = ("Program", "Processor", "OpStack", "Ram", "JumpStack", "Hash", "Cascade", "Lookup", "U32")


def pad(ctx, d_main_trace, n_rows, table_lengths):
    """MasterMainTable::pad (/root/reference/triton-vm/src/table/master_table.rs:932-983) on the device: the nine
    table-specific padding rules (tvm_pad_main_table), then the degree-lowering fill of the main table
    (tvm_fill_derived_main_columns).  d_main_trace: DeviceBuffer [379][n_rows] with the filled, unpadded tables in
    columns 0..148; table_lengths: the nine lengths in TABLE_ORDER (all_table_lengths, master_table.rs:985-1001)."""
    from . import degree_lowering

    lengths = np.ascontiguousarray(table_lengths, dtype=np.uint64)
    if lengths.size != 9 or d_main_trace.n_words < 379 * n_rows:
        raise ValueError("pad needs nine table lengths and a 379-column main trace")
    ctx._check(ctx.lib.tvm_pad_main_table(ctx.handle, d_main_trace.ptr, n_rows, lengths.ctypes.data), "tvm_pad_main_table")
    degree_lowering.fill_derived_main_columns(ctx, d_main_trace, n_rows)


class DeviceAet:
    """An algebraic execution trace whose arrays are resident on the device (aet_to_device): tvm_fill_main_table reads
    them where they lie, so nothing crosses PCIe inside Prover::prove.  Accepted wherever the dict of numpy arrays is."""

    def __init__(self, struct, buffers):
        self.struct, self.buffers = struct, buffers


def aet_to_device(ctx, aet):
    """upload the arrays of `aet` (see aet_struct) once -> DeviceAet.  instruction_multiplicities (uint32) travel padded to
    whole words."""
    s, keep = aet_struct(aet)
    buffers = {}
    for name, a in keep.items():
        if a.size == 0:
            continue
        raw = np.frombuffer(a.tobytes() + b"\0" * (-a.nbytes % 8), np.uint64)
        buffers[name] = ctx.to_device(raw)
        setattr(s, name, buffers[name].ptr)
    ctx.sync()
    return DeviceAet(s, buffers)


def aet_struct(aet):
    """the C ABI's `tvm_aet` over a dict of numpy arrays shaped like AlgebraicExecutionTrace's fields (aet.rs:41-96):
    program_words [p], instruction_multiplicities [p] (uint32), processor_trace [c][39], op_stack_trace [k][4],
    ram_trace [k][7], bezout_coefficients_0/1 [u] (optional: computed on the device when absent), program_hash_trace / sponge_trace / hash_trace [k][67],
    u32_entries [k][4], cascade_entries [k][2], lookup_multiplicities [256].  -> (struct, the arrays it points into)"""
    from .capi import Aet

    if isinstance(aet, DeviceAet):
        return aet.struct, aet.buffers
    keep = {}

    def arr(name, dtype=np.uint64, width=None):
        a = np.ascontiguousarray(aet.get(name, np.zeros(0, dtype)), dtype=dtype)
        if width is not None:
            a = a.reshape(-1, width)
        keep[name] = a
        return a

    s = Aet()
    words, mult = arr("program_words"), arr("instruction_multiplicities", np.uint32)
    if words.size != mult.size:
        raise ValueError("one multiplicity per program word")
    s.program_words, s.instruction_multiplicities, s.program_len = words.ctypes.data, mult.ctypes.data, words.size
    for name, len_, width in (("processor_trace", "processor_len", 39), ("op_stack_trace", "op_stack_len", 4), ("ram_trace", "ram_len", 7),
                              ("program_hash_trace", "program_hash_len", 67), ("sponge_trace", "sponge_len", 67), ("hash_trace", "hash_len", 67),
                              ("u32_entries", "u32_len", 4), ("cascade_entries", "cascade_len", 2)):
        a = arr(name, width=width)
        setattr(s, name, a.ctypes.data)
        setattr(s, len_, a.shape[0])
    if "bezout_coefficients_0" in aet or "bezout_coefficients_1" in aet:
        b0, b1 = arr("bezout_coefficients_0"), arr("bezout_coefficients_1")
        if b0.size != b1.size:
            raise ValueError("the two Bezout coefficient vectors have the same length")
        s.bezout_coefficients_0, s.bezout_coefficients_1, s.num_ram_pointers = b0.ctypes.data, b1.ctypes.data, b0.size
    else:   # left to the device (csrc/bezout.hip)
        s.bezout_coefficients_0 = s.bezout_coefficients_1 = None
        s.num_ram_pointers = 0
    lk = arr("lookup_multiplicities")
    if lk.size != 256:
        raise ValueError("256 lookup multiplicities")
    s.lookup_multiplicities = lk.ctypes.data
    return s, keep


def fill(ctx, d_main_trace, n_rows, aet):
    """MasterMainTable::new's table fills (/root/reference/triton-vm/src/table/master_table.rs:881-931) on the device
    (tvm_fill_main_table).  `aet`: dict of numpy arrays (see aet_struct).  Returns the nine table lengths (TABLE_ORDER),
    the argument of `pad`."""
    import ctypes as C

    s, keep = aet_struct(aet)
    lengths = np.zeros(9, np.uint64)
    ctx._check(ctx.lib.tvm_fill_main_table(ctx.handle, C.byref(s), d_main_trace.ptr, n_rows, lengths.ctypes.data), "tvm_fill_main_table")
    del keep
    return [int(v) for v in lengths]
