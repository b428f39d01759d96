"""One proof across the GPUs of a node: coset (row) sharding of the extended master tables.

ROLE (round 6): the PRODUCTION host is the C++ one (triton_vm_amd/host/: what bench.py times and what a Rust binding would call);
this Python module is its mirror for the parity tests -- test scaffolding above the C ABI, kept word-for-word equal to the C++ host by
tests/test_native_host.py and tests/test_sharded_host.py.  No algorithm lives here.

SURVEY.md section 8(e): the LDT / quotient domain g*<w_L> is X = L/N cosets of the trace domain; rank r of R
owns the cosets k = r (mod R), i.e. the extended rows i = r (mod R).  Those rows are themselves an arithmetic
domain -- offset g*w_L^r, generator w_L^R, length L/R -- so a rank's share of
`maybe_low_degree_extend_all_columns`, `hash_all_ldt_domain_rows` and `all_quotients_combined` (the next row
i + L/N stays on the same rank) are the *same* C-ABI calls on that local domain: no kernel knows about ranks.
The trace (5 GiB at 2^20 rows) is replicated.  Exchanges (RCCL all-gather over xGMI; gloo in the CPU tests):

    leaf digests of each master table   all-to-all: rank t receives the digests of ITS contiguous leaf range (L/R x 40 B:
                                        42 MB per rank at 2^20 rows and 8 ranks), interleaved into row order
    the quotient codeword               L x 24 B  (201 MB), likewise
    the opened rows (173 x 652 words)   all-gather of each owner's rows, put back into query order
    the out-of-domain rows              columns split over the ranks, all-gather of the shares (2 x 470 XFE)

    Merkle subtree roots and authentication nodes      a few KB (the trees are built split, see _SplitTree)

Everything after the quotient codeword (segments, combination, DEEP, the FRI folds: ~25 ms of a 263 ms proof at
2^20 rows) is computed redundantly on every rank from identical inputs, so every rank derives the same transcript
and no broadcast is needed.  The large Merkle trees (the three table trees, the first FRI rounds) are not built
redundantly: every rank has all the leaves, builds the subtree over its contiguous 1/R of them, and the R subtree
roots are exchanged; authentication nodes are fetched from the rank that holds them.

Process set-up order matters: import torch (and select the device) BEFORE creating the Context -- torch brings
its own HIP runtime, and a process that has already initialised the system runtime through libtriton_hip.so
makes torch.cuda report "No HIP GPUs are available".  bench.py and the tests do it in that order.
"""
import numpy as np

from . import field, stark
from .arithmetic_domain import ArithmeticDomain
from .master_table import MasterTable
from .prover import NUM_AUX, NUM_MAIN, Prover


class _TensorBuffer:
    """A torch tensor of int64 words used where the C ABI wants a device pointer (`.ptr`)."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()
        self.n_words = tensor.numel()

    def download(self, shape=None):
        a = self.tensor.cpu().numpy().view(np.uint64)
        return a.reshape(shape) if shape is not None else a


def split_tree_locations(node_indices, world):
    """heap indices of a Merkle tree whose lowest levels are split into `world` subtrees by contiguous leaf ranges ->
    (in the top tree?, owning rank, heap index within that rank's subtree).  Node k sits at depth d = floor(log2 k); at
    depths >= log2(world) its position p = k - 2^d within the level selects subtree p >> (d - log2 world), where it is node
    2^(d - log2 world) + (p mod 2^(d - log2 world)).  The subtree roots themselves (depth log2 world) count as top-tree nodes."""
    idx = np.asarray(node_indices, dtype=np.int64)
    log_r = int(world).bit_length() - 1
    depth = np.floor(np.log2(np.maximum(idx, 1).astype(np.float64))).astype(np.int64)
    depth -= (np.int64(1) << depth) > idx                                         # (guards the float rounding)
    depth += (np.int64(2) << depth) <= idx
    in_top = idx < 2 * world
    dd = np.maximum(depth - log_r, 0)
    pos = idx - (np.int64(1) << depth)
    owner = np.where(in_top, -1, pos >> dd)
    local = np.where(in_top, idx, (np.int64(1) << dd) + (pos & ((np.int64(1) << dd) - 1)))
    return in_top, owner, local


class _SplitTree:
    """A Merkle tree over n leaves whose lower levels are split over the R ranks by contiguous leaf ranges: this rank
    holds the node array of subtree `rank` (device, heap order, [2 n/R][5]), every rank the top tree over the R subtree
    roots (host, heap order, [2R][5]).  Same nodes as MerkleTree::par_new over all the leaves."""

    def __init__(self, prover, sub_nodes, n_leaves):
        from .verifier import hash_pair

        self.prover, self.sub, self.n = prover, sub_nodes, n_leaves
        prover.split_trees_built = getattr(prover, "split_trees_built", 0) + 1
        ctx, R = prover.ctx, prover.world
        mine = np.empty((1, 5), np.uint64)
        ctx._check(ctx.lib.tvm_memcpy_d2h(ctx.handle, mine.ctypes.data, sub_nodes.ptr + 40, 40), "subtree root")
        self.top = np.zeros((2 * R, 5), np.uint64)
        self.top[R:] = prover._all_gather_host(mine, 1).reshape(R, 5)
        for k in range(R - 1, 0, -1):
            self.top[k] = hash_pair(ctx.lib, self.top[2 * k], self.top[2 * k + 1])

    def root(self):
        return self.top[1].copy()

    def authentication_nodes(self, indices):
        """stark.auth_nodes for the split tree: node k of the whole tree at depth d >= log2 R is node
        2^(d - log2 R) + (its position within the subtree's level) of subtree (position >> (d - log2 R))"""
        prover, R = self.prover, self.prover.world
        ctx, log_r = prover.ctx, self.prover.world.bit_length() - 1
        idx = stark.auth_node_indices(self.n, indices).astype(np.int64)
        out = np.empty((idx.size, 5), np.uint64)
        if not idx.size:
            return out
        in_top, owner_all, local_all = split_tree_locations(idx, R)
        out[in_top] = self.top[idx[in_top]]
        low = np.nonzero(~in_top)[0]
        owner, local = owner_all[low], local_all[low]
        mine = low[owner == prover.rank]
        got = np.zeros((mine.size, 5), np.uint64)
        if mine.size:
            lix = np.ascontiguousarray(local[owner == prover.rank].astype(np.uint64))
            ctx._check(ctx.lib.tvm_gather_elements(ctx.handle, self.sub.ptr, 5, lix.ctypes.data, lix.size, got.ctypes.data),
                       "auth nodes")
        cap = max(int(np.bincount(owner, minlength=R).max()) if low.size else 0, 1)
        gathered = prover._all_gather_host(got, cap)                              # [R, cap, 5]
        for r in range(R):
            sel = low[owner == r]
            out[sel] = gathered[r, :sel.size]
        return out


def local_domain(domain, rank, world):
    """rows i = rank (mod world) of `domain`, as a domain"""
    return ArithmeticDomain(field.mont_mul(domain.offset, field.mont_pow(domain.generator, rank)),
                            field.mont_pow(domain.generator, world), domain.length // world)


class ShardedProver(Prover):
    """Prover whose extended master tables are split by cosets over the ranks of a torch.distributed group."""

    def __init__(self, ctx, params, dist, device, main_trace=None, aux_trace=None, seed=1):
        self._check_sharding(params, dist)
        super().__init__(ctx, params, main_trace, aux_trace, seed)  # every rank holds the same traces (same seed)
        self._init_sharding(dist, device)

    @classmethod
    def from_execution(cls, ctx, dist, device, aet, padded_height, claim, randomness_seed, **kw):
        """Prover::prove(claim, aet) over the ranks: every rank fills, pads and (after the challenges) extends the SAME trace
        tables from the same execution trace and the same seed (replicated: 16 ms at 2^20 rows), then owns its cosets of
        the extended tables.  Arguments as Prover.from_execution."""
        self = super().from_execution(ctx, aet, padded_height, claim, randomness_seed, **kw)
        self._check_sharding(self.p, dist)
        self._init_sharding(dist, device)
        return self

    @staticmethod
    def _check_sharding(params, dist):
        world = dist.get_world_size()
        if (params.ldt.length // params.trace.length) % world or (params.quotient.length // params.trace.length) % world \
                or params.quotient.length > params.ldt.length:
            raise ValueError("coset sharding needs a world size that divides |LDT| / |trace| and |quotient| / |trace|")

    def _init_sharding(self, dist, device):
        import torch

        self.torch, self.dist, self.device = torch, dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ldt_local = local_domain(self.p.ldt, self.rank, self.world)
        # trees of at least this many leaves are built split (_SplitTree); below it the two small exchanges cost
        # more than the redundant hashing saves
        self.split_tree_min_leaves = (1 << 21) if self.world > 1 else (1 << 62)
        for mt in (self.main, self.aux):
            if mt is not None:
                mt.quotient_domain = mt.ldt_domain = self.ldt_local

    def _extend(self, challenges):
        aux = super()._extend(challenges)       # MasterMainTable::extend, replicated; this rank extends it onto its cosets
        aux.quotient_domain = aux.ldt_domain = self.ldt_local
        return aux

    # -- collectives ---------------------------------------------------------------------------------
    def _empty(self, n_words):
        return self.torch.empty(n_words, dtype=self.torch.int64, device=self.device)

    def _sync_device(self):
        self.ctx.sync()
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def _all_gather_rows(self, local, elem_words):
        """local: tensor [L/R * elem_words] holding this rank's rows -> tensor [L * elem_words] in row order
        (global row i = local row i // R of rank i % R)"""
        self._sync_device()  # the producing kernels ran on the context's stream
        gathered = self._empty(local.numel() * self.world)
        self.dist.all_gather_into_tensor(gathered, local)
        out = gathered.view(self.world, -1, elem_words).permute(1, 0, 2).contiguous().view(-1)
        self._sync_device()
        return out

    # -- the sharded steps -----------------------------------------------------------------------------
    def _commit_master_table(self, mt):
        ctx, L, R = self.ctx, self.p.ldt.length, self.world
        local_rows = self.ldt_local.length
        digests = self._empty(5 * local_rows)
        ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, mt._need_table(), local_rows, digests.data_ptr()), "hash_rows")
        if self._splits(L) and local_rows % R == 0:
            # The tree is built split: rank t needs the leaves of ITS contiguous range [t L/R, (t+1) L/R) only.  Of this
            # rank's rows (global row rank + R a) those are the local rows a in [t L/R^2, (t+1) L/R^2): one all-to-all
            # moves every digest once (1/R of what an all-gather of all leaves moves), then the R received blocks are
            # interleaved into row order.
            self._sync_device()
            received = self._empty(5 * local_rows)
            self.dist.all_to_all_single(received, digests)
            mine = received.view(R, local_rows // R, 5).permute(1, 0, 2).contiguous().view(-1)
            self._sync_device()
            self._keep = mine  # until the stream has consumed it
            self.leaf_exchange = "all_to_all"
            per = L // R
            sub = ctx.alloc(10 * per)
            ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, mine.data_ptr(), per, sub.ptr), "merkle_tree")
            return _SplitTree(self, sub, L)
        leaves = self._all_gather_rows(digests, 5)
        self._keep = leaves  # until the stream has consumed it
        self.leaf_exchange = "all_gather"
        return self._tree_of_leaf_digests(leaves.data_ptr(), L)

    def _splits(self, n_leaves):
        return n_leaves >= max(self.split_tree_min_leaves, 2 * self.world)

    def _tree_of_leaf_digests(self, leaves_ptr, n_leaves):
        """leaf digests on the device (all of them, on every rank) -> node array, or a _SplitTree"""
        ctx = self.ctx
        if not self._splits(n_leaves):
            nodes = ctx.alloc(10 * n_leaves)
            ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, leaves_ptr, n_leaves, nodes.ptr), "merkle_tree")
            return nodes
        per = n_leaves // self.world
        sub = ctx.alloc(10 * per)
        ctx._check(ctx.lib.tvm_merkle_tree(ctx.handle, leaves_ptr + self.rank * per * 40, per, sub.ptr), "merkle_tree")
        return _SplitTree(self, sub, n_leaves)

    def _table_tree(self, table_handle, n):
        """the quotient-segment table's tree: rows hashed on every rank (the table is replicated), the tree split"""
        if not self._splits(n):
            return super()._table_tree(table_handle, n)
        ctx = self.ctx
        digests = ctx.alloc(5 * n)
        ctx._check(ctx.lib.tvm_hash_rows(ctx.handle, table_handle, n, digests.ptr), "hash_rows")
        tree = self._tree_of_leaf_digests(digests.ptr, n)
        ctx.sync()  # the digests are read by the stream until here
        return tree

    def _codeword_tree(self, ctx, d_codeword, length):
        """merkle_tree_from_codeword (fri.rs:343-347) for the FRI rounds"""
        if not self._splits(length):
            return stark.merkle_tree_from_codeword(ctx, d_codeword, length)
        per = length // self.world
        sub = ctx.alloc(10 * per)
        ctx._check(ctx.lib.tvm_codeword_merkle_tree(ctx.handle, d_codeword.ptr + self.rank * per * 24, per, sub.ptr), "codeword tree")
        return _SplitTree(self, sub, length)

    def _root(self, nodes):
        return nodes.root() if isinstance(nodes, _SplitTree) else super()._root(nodes)

    def _auth_nodes(self, nodes, n_leaves, indices):
        if isinstance(nodes, _SplitTree):
            return nodes.authentication_nodes(indices)
        return super()._auth_nodes(nodes, n_leaves, indices)

    def _fri(self, combination, ps):
        p = self.p
        a_indices, self.last_codeword, self.last_polynomial, self.last_domain = stark.fri_prove(
            self.ctx, p.ldt, p.fri_rounds, p.num_collinearity_checks, combination, ps, trees=self)
        return a_indices

    def _quotient_codeword(self, challenges, quotient_weights):
        """this rank's rows q = rank (mod world) of the quotient domain, all-gathered into row order.  With |quotient| ==
        |LDT| (Stark::default()) those are rows of the cached tables; with a shorter quotient domain (LDT expansion 16:
        the quotient domain is the stride-4 view of the LDT domain, whose cosets k = 0 (mod 4) sit on a quarter of the
        ranks) every rank extends the traces once more onto ITS share of the quotient domain -- 1/4 of its LDT share --
        so that the AIR, the one stage that reads the quotient rows, stays spread evenly over the ranks."""
        ctx, p = self.ctx, self.p
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(63, 3)
        w = np.ascontiguousarray(quotient_weights, dtype=np.uint64).reshape(604, 3)
        valid = getattr(self, "assume_valid_trace", False)
        if valid:
            ctx.assume_valid_trace(True)
        try:
            if p.quotient.length == p.ldt.length:
                q_local, tables = self.ldt_local, (self.main, self.aux)
            else:
                q_local = local_domain(p.quotient, self.rank, self.world)
                tables = tuple(MasterTable.from_device(ctx, mt.d_trace, mt.d_randomizers, mt.n_cols, mt.n_rows, mt.num_trace_randomizers,
                                                       mt.trace_domain, q_local, q_local, mt.fk) for mt in (self.main, self.aux))
                for t in tables:
                    t.maybe_low_degree_extend_all_columns()
            local = self._empty(3 * q_local.length)
            ctx._check(ctx.lib.tvm_all_quotients_combined(ctx.handle, tables[0]._need_table(), tables[1]._need_table(),
                                                          p.trace.c(), q_local.c(), ch.ctypes.data, w.ctypes.data,
                                                          local.data_ptr()), "all_quotients_combined")
            if tables[0] is not self.main:
                for t in tables:
                    t.clear_cache()
        finally:
            if valid:
                ctx.assume_valid_trace(False)
        return _TensorBuffer(self._all_gather_rows(local, 3))

    def _all_gather_host(self, mine, cap):
        """all-gather of one equally sized (zero-padded to `cap` leading entries) block per rank: numpy [k, ...] ->
        numpy [world, cap, ...].  The payloads (out-of-domain rows, opened rows) are produced on the host by the C ABI
        and are a few hundred KB; the collective itself runs on device tensors (RCCL) / CPU tensors (gloo)."""
        block = np.zeros((cap,) + mine.shape[1:], np.uint64)
        block[:mine.shape[0]] = mine
        t = self.torch.from_numpy(block.view(np.int64).reshape(-1)).to(self.device)
        gathered = self._empty(t.numel() * self.world)
        self.dist.all_gather_into_tensor(gathered, t)
        return gathered.cpu().numpy().view(np.uint64).reshape((self.world,) + block.shape)

    def _out_of_domain_rows(self, mt, points):
        """columns split evenly over the ranks (the trace is replicated): rank r evaluates columns [r * per, (r+1) * per)
        at both points, the shares are all-gathered"""
        per = -(-mt.n_cols // self.world)
        c0 = min(self.rank * per, mt.n_cols)
        c1 = min(c0 + per, mt.n_cols)
        mine = mt.out_of_domain_rows(points, c0, c1 - c0)                      # [n_points, c1 - c0, 3]
        gathered = self._all_gather_host(np.ascontiguousarray(mine.transpose(1, 0, 2)), per)   # [world, per, n_points, 3]
        cols = gathered.reshape(self.world * per, len(points), 3)[:mt.n_cols]
        return np.ascontiguousarray(cols.transpose(1, 0, 2))

    def _reveal_master_rows(self, mt, row_indices):
        """row i of the extended table lives on rank i mod world (as its local row i // world); every rank opens the
        rows it owns, the shares are all-gathered and put back into query order"""
        idx = np.asarray(row_indices, dtype=np.uint64)
        owner = (idx % np.uint64(self.world)).astype(np.int64)
        width = mt.n_cols * mt.fk
        mine = np.nonzero(owner == self.rank)[0]
        rows = np.zeros((0, width), np.uint64)
        if mine.size:
            rows = mt.reveal_rows(idx[mine] // np.uint64(self.world)).reshape(mine.size, width)
        cap = max(int(np.bincount(owner, minlength=self.world).max()), 1)
        gathered = self._all_gather_host(rows, cap)                            # [world, cap, width]
        out = np.empty((idx.size, width), np.uint64)
        for r in range(self.world):
            pos = np.nonzero(owner == r)[0]
            out[pos] = gathered[r, :pos.size]
        return out.reshape((idx.size, mt.n_cols) + ((3,) if mt.fk == 3 else ()))
