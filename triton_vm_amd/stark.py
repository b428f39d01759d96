"""Host mirror of the hot private helpers of the reference's Prover (/root/reference/triton-vm/src/stark.rs)
and of the FRI prover round (/root/reference/triton-vm/src/low_degree_test/fri.rs), over the C ABI.
XFE vectors are DeviceBuffers of 3-word elements."""
import ctypes as C

import numpy as np

from . import field

ZETA = field.to_mont(3)  # Stark::ZETA, stark.rs:1801
NUM_QUOTIENT_SEGMENTS = 4
NUM_RANDOMIZED_QUOTIENT_SEGMENTS = 5


def _h(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def all_quotients_combined(ctx, main_table, aux_table, trace_domain, quotient_domain, challenges, weights):
    """all_quotients_combined (master_table.rs:1264-1363) on the two extended MasterTables -> DeviceBuffer of
    quotient_domain.length XFE."""
    ch, w = _h(challenges).reshape(63, 3), _h(weights).reshape(604, 3)
    out = ctx.alloc(quotient_domain.length * 3)
    ctx._check(ctx.lib.tvm_all_quotients_combined(ctx.handle, main_table._need_table(), aux_table._need_table(),
                                                  trace_domain.c(), quotient_domain.c(), ch.ctypes.data, w.ctypes.data,
                                                  out.ptr), "all_quotients_combined")
    return out


class QuotientSegments:
    """Result of compute_quotient_segments' tail + randomize_quotient_segments (stark.rs:784-792,1302-1356)."""

    def __init__(self, ctx, table, polys, poly_len, ldt_length):
        self.ctx, self.table, self.polys, self.poly_len, self.ldt_length = ctx, table, polys, poly_len, ldt_length

    def codewords(self):
        """[ldt.length, 5, 3] on the host (reference layout of the randomized segment table)."""
        buf = self.ctx.alloc(self.ldt_length * 15)
        self.ctx._check(self.ctx.lib.tvm_table_export_row_major(self.ctx.handle, self.table, buf.ptr), "export")
        return buf.download((self.ldt_length, 5, 3))

    def merkle_tree(self):
        d = self.ctx.alloc(10 * self.ldt_length)
        self.ctx._check(self.ctx.lib.tvm_table_merkle_tree(self.ctx.handle, self.table, self.ldt_length, d.ptr), "merkle")
        return d.download((2 * self.ldt_length, 5))

    def linear_combination(self, weights, view_length=None):
        """over the LDT-domain rows, or over the stride view of `view_length` rows (the quotient domain when it is the
        shorter one: the same points, stark.rs:501-506)"""
        w = _h(weights).reshape(5, 3)
        n = view_length or self.ldt_length
        out = self.ctx.alloc(n * 3)
        self.ctx._check(self.ctx.lib.tvm_table_linear_combination(self.ctx.handle, self.table, n, w.ctypes.data, out.ptr),
                        "lincomb")
        return out

    def free(self):
        if self.table:
            self.ctx.lib.tvm_table_free(self.ctx.handle, self.table)
            self.table = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def quotient_segments(ctx, d_quotient_codeword, quotient_domain, ldt_domain, randomizer, poly_len=None):
    rnd = _h(randomizer).reshape(-1, 3)
    poly_len = poly_len or max(quotient_domain.length // 4, rnd.shape[0])
    polys = ctx.alloc(5 * poly_len * 3)
    t = C.c_void_p()
    ctx._check(ctx.lib.tvm_quotient_segments(ctx.handle, d_quotient_codeword.ptr, quotient_domain.c(), ldt_domain.c(),
                                             rnd.ctypes.data, rnd.shape[0], ZETA, C.byref(t), polys.ptr, poly_len),
               "quotient_segments")
    return QuotientSegments(ctx, t.value, polys, poly_len, ldt_domain.length)


def evaluate_at_points(ctx, d_coeffs, n, points):
    pts = _h(points).reshape(-1, 3)
    out = np.empty((pts.shape[0], 3), np.uint64)
    ctx._check(ctx.lib.tvm_evaluate_at_points(ctx.handle, d_coeffs.ptr, n, pts.ctypes.data, pts.shape[0], out.ctypes.data),
               "evaluate_at_points")
    return out


def deep_codeword(ctx, d_codewords, domain, points, values, weights):
    """Weighted sum of the DEEP components (stark.rs:566-625); a single component with weight one is
    Prover::deep_codeword (stark.rs:1360-1379)."""
    k = len(d_codewords)
    ptrs = (C.c_void_p * k)(*[b.ptr for b in d_codewords])
    pts, vals, ws = _h(points).reshape(k, 3), _h(values).reshape(k, 3), _h(weights).reshape(k, 3)
    out = ctx.alloc(domain.length * 3)
    ctx._check(ctx.lib.tvm_deep_codeword(ctx.handle, k, ptrs, domain.c(), pts.ctypes.data, vals.ctypes.data,
                                         ws.ctypes.data, out.ptr), "deep")
    return out


def split_and_fold(ctx, d_codeword, domain, folding_challenge):
    """ProverRound::split_and_fold (fri.rs:349-366)."""
    ch = _h(folding_challenge).reshape(3)
    out = ctx.alloc(domain.length // 2 * 3)
    ctx._check(ctx.lib.tvm_fri_split_and_fold(ctx.handle, d_codeword.ptr, domain.c(), ch.ctypes.data, out.ptr), "fold")
    return out


def merkle_tree_from_codeword(ctx, d_codeword, length):
    """ProverRound::merkle_tree_from_codeword (fri.rs:343-347) -> node array [2n, 5] on the device."""
    nodes = ctx.alloc(10 * length)
    ctx._check(ctx.lib.tvm_codeword_merkle_tree(ctx.handle, d_codeword.ptr, length, nodes.ptr), "codeword tree")
    return nodes


def auth_node_indices(n_leaves, indices):
    """[twenty-first MerkleTree::authentication_structure, restated] the nodes a verifier cannot compute from the
    revealed leaves -- the siblings along the paths that are not themselves on a path -- in descending heap order"""
    k = np.unique(np.asarray(indices, dtype=np.uint64) + np.uint64(n_leaves))
    needed, computable = [], []
    while k.size and k[0] > 1:
        computable.append(k)
        needed.append(k ^ np.uint64(1))
        k = np.unique(k >> np.uint64(1))
    if not needed:
        return np.zeros(0, np.uint64)
    idx = np.setdiff1d(np.concatenate(needed), np.concatenate(computable))
    return np.ascontiguousarray(idx[::-1])


def auth_nodes(ctx, d_nodes, n_leaves, indices):
    """MerkleTree::authentication_structure for the opened leaves of a device node array, gathered to the host"""
    idx = auth_node_indices(n_leaves, indices)
    out = np.empty((idx.size, 5), np.uint64)
    if idx.size:
        ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, d_nodes.ptr, 5, idx.ctypes.data, idx.size, out.ctypes.data),
                   "auth nodes")
    return out


def merkle_root(ctx, d_nodes):
    """node 1 of a device node array (MerkleTree::root): one 40-byte copy, which also drains the stream"""
    out = np.empty(5, np.uint64)
    ctx._check(ctx.lib.tvm_memcpy_d2h(ctx.handle, out.ctypes.data, d_nodes.ptr + 40, 40), "root")
    return out


def fri_prove(ctx, ldt_domain, num_rounds, num_collinearity_checks, d_codeword, ps, trees=None):
    """Fri::prove (fri.rs:212-319, 754-772): commit and fold round by round, send the last codeword and polynomial,
    answer the queries.  ps: the transcript (enqueue / sample_scalars / sample_indices).  trees: optional provider
    of `_codeword_tree(ctx, codeword, length)`, `_root(nodes)`, `_auth_nodes(nodes, n_leaves, indices)`.
    -> (first-round indices, last codeword, last polynomial, last domain)"""
    from .arithmetic_domain import ArithmeticDomain

    lib = ctx.lib
    tree_of = trees._codeword_tree if trees is not None else merkle_tree_from_codeword
    root_of = trees._root if trees is not None else (lambda nodes: merkle_root(ctx, nodes))
    auth_of = trees._auth_nodes if trees is not None else (lambda nodes, n, ix: auth_nodes(ctx, nodes, n, ix))
    dom, cw, rounds = ldt_domain, d_codeword, []
    for r in range(num_rounds + 1):
        nodes = tree_of(ctx, cw, dom.length)
        ps.enqueue(f"fri root {r}", root_of(nodes))
        rounds.append((dom, cw, nodes))
        if r == num_rounds:
            break
        challenge = ps.sample_scalars(1)[0]
        cw = split_and_fold(ctx, cw, dom, challenge)
        dom = dom.pow(2)
    last = cw.download((dom.length, 3))
    ps.enqueue("fri last codeword", last, fiat_shamir=False)
    # the interpolant over the last domain with its offset set to one (fri.rs:292-300)
    last_poly = ArithmeticDomain.of_length(dom.length).interpolate(ctx, cw, 3).download((dom.length, 3))
    ps.enqueue("fri last polynomial", last_poly)
    a_indices = ps.sample_indices(ldt_domain.length, num_collinearity_checks)
    for r, (rdom, rcw, rnodes) in enumerate(rounds):
        idxs = a_indices if r == 0 else []
        b_idx = [(a + rdom.length // 2) % rdom.length for a in (i % rdom.length for i in a_indices)]
        for which in ((idxs, b_idx) if r == 0 else (b_idx,)):
            if r == len(rounds) - 1 and which is b_idx:
                continue
            ix = np.array(which, np.uint64)
            leaves = np.empty((ix.size, 3), np.uint64)
            ctx._check(lib.tvm_gather_elements(ctx.handle, rcw.ptr, 3, ix.ctypes.data, ix.size, leaves.ctypes.data), "leaves")
            ps.enqueue(f"fri response {r}", leaves, fiat_shamir=False)
            ps.enqueue(f"fri auth {r}", auth_of(rnodes, rdom.length, which), fiat_shamir=False)
    ps.sample_scalars(1)
    return a_indices, last, last_poly, dom
