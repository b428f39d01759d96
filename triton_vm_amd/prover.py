"""Host mirror of the hot path of ``Prover::prove`` (/root/reference/triton-vm/src/stark.rs:331-719),
step for step, over the C ABI.  Everything bulky stays in HBM; the host only sees Merkle roots,
out-of-domain rows, the last FRI codeword and the opened rows.

ROLE (round 6): the PRODUCTION host is the C++ one (triton_vm_amd/host/: what bench.py times and what a Rust binding would call);
this Python module is its mirror for the parity tests -- test scaffolding above the C ABI, kept word-for-word equal to the C++ host by
tests/test_native_host.py and tests/test_sharded_host.py.  No algorithm lives here.

The Fiat-Shamir transcript is the reference's (triton_vm_amd/proof_stream.py: ProofItem encoding, Claim, sampling), the
prover's randomness is the reference's when a 32-byte seed is given (triton_vm_amd/randomness.py): `Prover.from_execution`
on the reference's own snapshot program yields the reference's proof, word for word (tests/test_proof_snapshot.py).
"""
import ctypes as C
import time

import numpy as np

from . import field, master_table, randomness, stark
from .arithmetic_domain import ArithmeticDomain
from .master_table import MasterTable
from .proof_stream import Claim, ProofStream  # noqa: F401  (re-exported: the tests and the other provers import it from here)

NUM_MAIN, NUM_AUX = 379, 91
NUM_CHALLENGES, NUM_SAMPLED_CHALLENGES, NUM_CONSTRAINTS = 63, 59, 604
NUM_DEEP = 4
# challenges.rs:31-83: the indeterminates the four derived challenges are evaluated at
CH_COMPRESS_PROGRAM_DIGEST, CH_STANDARD_INPUT, CH_STANDARD_OUTPUT, CH_LOOKUP_TABLE_PUBLIC = 0, 1, 2, 54
TIP5_LOOKUP_TABLE = [(pow(x + 1, 3, 257) - 1) % 257 for x in range(256)]   # [Tip5: L(x) = (x + 1)^3 - 1 mod 257]


def derive_challenges(lib, sampled, claim):
    """Challenges::new (challenges.rs:85-121): the 59 sampled challenges, then the terminals of the public input, the
    public output, the lookup table and the program digest -- EvalArg::compute_terminal(symbols, 1, indeterminate)."""
    sampled = np.ascontiguousarray(sampled, dtype=np.uint64).reshape(NUM_SAMPLED_CHALLENGES, 3)

    def terminal(symbols, x):
        acc = np.array([field.ONE, 0, 0], np.uint64)
        for s in symbols:
            acc = xfe_mul(lib, acc, x)
            acc[0] = (int(acc[0]) + int(s)) % field.P
        return acc

    lut = [field.to_mont(v) for v in TIP5_LOOKUP_TABLE]
    derived = [terminal(claim.input, sampled[CH_STANDARD_INPUT]), terminal(claim.output, sampled[CH_STANDARD_OUTPUT]),
               terminal(lut, sampled[CH_LOOKUP_TABLE_PUBLIC]), terminal(claim.program_digest, sampled[CH_COMPRESS_PROGRAM_DIGEST])]
    return np.concatenate([sampled, np.array(derived, np.uint64)])


def xfe_powers(lib, x, first, n):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.empty((n, 3), np.uint64)
    lib.tvm_host_xfe_powers(x.ctypes.data, first, n, out.ctypes.data)
    return out


def xfe_mul(lib, a, b):
    a, b, o = (np.ascontiguousarray(v, dtype=np.uint64) for v in (a, b, np.zeros(3)))
    lib.tvm_host_xfe_mul(a.ctypes.data, b.ctypes.data, o.ctypes.data)
    return o


def xfe_add(a, b):
    return np.array([(int(x) + int(y)) % field.P for x, y in zip(a, b)], np.uint64)


class ZeroKnowledgeViolation(RuntimeError):
    """ProvingError::ZeroKnowledgeViolation (error.rs:150-186, stark.rs:645-663)"""


class StarkParameters:
    """Domains for a padded height, as Stark::default() with LdtChoice::Fri derives them
    (stark.rs:263-286, 1885-1916, 2083-2089; fri.rs:832-836, 907-920).  expansion = 4."""

    def __init__(self, log2_padded_height, num_trace_randomizers=198, num_collinearity_checks=173, log2_expansion=2,
                 ldt="fri", security_level=160):
        """ldt = "fri" (LdtChoice::Fri, what BASELINE.json's configs name) or "stir" (the reference's automatic choice
        from 2^16 rows on, stark.rs:1944-1951): then the STIR instance fixes the number of trace randomizers and the
        LDT domain (Stark::stir, stark.rs:1972-2032) and the two query-count arguments are ignored."""
        self.padded_height = 1 << log2_padded_height
        self.log2_expansion = log2_expansion
        self.stir = None
        if ldt == "stir":
            from .low_degree_test import stark_stir

            self.stir = stark_stir(self.padded_height, security_level=security_level, log2_ldt_expansion_factor=log2_expansion)
            num_trace_randomizers = self.stir.num_trace_randomizers()
        elif ldt != "fri":
            raise ValueError("ldt must be 'fri' or 'stir'")
        self.h = num_trace_randomizers
        self.num_collinearity_checks = num_collinearity_checks
        h = self.h
        rtl = max(self.padded_height + h, 2 * h + 1, (h + 1) * 5)
        self.randomized_trace_len = 1 << (rtl - 1).bit_length()
        self.trace = ArithmeticDomain.of_length(self.randomized_trace_len // 2)
        max_degree = 4 * (self.randomized_trace_len - 1) - 1
        quotient_len = 1 << (max_degree - 1).bit_length()
        g = field.generator()
        self.ldt = ArithmeticDomain.of_length(self.randomized_trace_len << log2_expansion).with_offset(g)
        self.quotient = ArithmeticDomain.of_length(quotient_len).with_offset(g)
        first_round_dim = self.randomized_trace_len
        max_rounds = (first_round_dim - 1).bit_length()
        self.fri_rounds = max(0, max_rounds - (num_collinearity_checks.bit_length() - 1) - 1)
        self.num_quotient_randomizers = (h + 1) * 5
        if self.stir is not None and self.stir.initial_domain.length != self.ldt.length:
            self.ldt = self.stir.initial_domain  # Stark::stir may have grown the domain (tiny padded heights)
            self.quotient = ArithmeticDomain.of_length(max(quotient_len, 1)).with_offset(g)


def stark_parameters(log2_padded_height, security_level=160, log2_expansion=2, ldt=None):
    """Stark::new(security_level, log2_expansion) for a padded height (stark.rs:1815-1830, 1944-2089): ldt "fri", "stir" or None
    = Stark::ldt's rule (STIR from 2^16 padded rows on).  fri.rs:832-836: the number of collinearity checks;
    stark.rs:2083-2089: the number of trace randomizers."""
    import math

    from .low_degree_test import ReedSolomonCode

    if ldt is None:
        ldt = "fri" if log2_padded_height < 16 else "stir"
    checks = math.ceil(-security_level / math.log2(1.0 - ReedSolomonCode(log2_expansion).proximity_parameter()))
    return StarkParameters(log2_padded_height, num_trace_randomizers=checks + 4 * 3 * 2 + 1, num_collinearity_checks=checks,
                           log2_expansion=log2_expansion, ldt=ldt, security_level=security_level)


class Prover:
    """Runs the prover's hot path on synthetic or caller-provided padded trace tables, or -- `from_execution` -- the
    whole of Prover::prove from an algebraic execution trace."""

    def __init__(self, ctx, params, main_trace=None, aux_trace=None, seed=1, claim=None):
        self.ctx, self.p = ctx, params
        n, h = params.trace.length, params.h
        rng = np.random.default_rng(seed)
        dom = (params.trace, params.quotient, params.ldt)
        if main_trace is None:
            self.main = MasterTable.__new__(MasterTable)
            self._init_synthetic(self.main, 1, NUM_MAIN, n, h, dom, seed)
            self.aux = MasterTable.__new__(MasterTable)
            self._init_synthetic(self.aux, 3, NUM_AUX, n, h, dom, seed + 100)
        else:
            rnd = lambda *s: rng.integers(0, field.P, size=s, dtype=np.uint64)
            self.main = MasterTable(ctx, main_trace, rnd(NUM_MAIN, h), *dom, 1)
            self.aux = MasterTable(ctx, aux_trace, rnd(NUM_AUX, h, 3), *dom, 3)
        self.quotient_randomizer = rng.integers(0, field.P, size=(params.num_quotient_randomizers, 3), dtype=np.uint64)
        self.claim = claim or Claim()
        self.randomness_seed = None
        self.timings, self.wall, self.opened = {}, {}, {}
        self.capture = None

    @classmethod
    def from_execution(cls, ctx, aet, padded_height, claim, randomness_seed, log2_expansion=2, ldt=None, security_level=160,
                       assume_valid_trace=True):
        """Prover::prove from the start (stark.rs:331-400): the master main table is filled from the algebraic
        execution trace and padded on the device (MasterMainTable::new + pad, master_table.rs:881-983), all randomness
        comes from `randomness_seed` (32 bytes) the way the reference draws it, and the auxiliary table is extended on
        the device once the challenges are sampled (MasterMainTable::extend, master_table.rs:1006-1075).
        aet: the arrays master_table.fill takes; padded_height: AlgebraicExecutionTrace::padded_height (aet.rs:141-146);
        security_level, log2_expansion: Stark::new's (stark.rs:1815-1830; Stark::default() is 160, 2); ldt: "fri", "stir" or
        None (the default, like Stark::default()) for the reference's automatic choice (STIR from 2^16 padded rows on);
        assume_valid_trace: the tables come from an execution trace, so the quotient evaluation may use the degree
        bounds of the constraint quotients (TVM_OPTION_AIR_VALID_TRACE) -- the same proof, word for word, as
        long as the trace satisfies the AIR (both reference snapshots are reproduced this way); an invalid trace gives a
        proof that differs from the reference's equally unverifiable one."""
        self = cls.__new__(cls)
        log2 = padded_height.bit_length() - 1
        if padded_height != 1 << log2:
            raise ValueError("the padded height is a power of two")
        p = stark_parameters(log2, security_level, log2_expansion, ldt)
        self.ctx, self.p, self.claim, self.randomness_seed = ctx, p, claim, bytes(randomness_seed)
        self.assume_valid_trace = assume_valid_trace
        n, h, lib = p.trace.length, p.h, ctx.lib
        d_main = ctx.alloc(NUM_MAIN * n)
        lengths = master_table.fill(ctx, d_main, n, aet)
        if max(lengths) > padded_height:
            raise ValueError("a table is longer than the padded height")
        master_table.pad(ctx, d_main, n, lengths)
        rnd = randomness.trace_randomizers(lib, self.randomness_seed, NUM_MAIN, h, 1)
        self.main = MasterTable.from_device(ctx, d_main, ctx.to_device(rnd), NUM_MAIN, n, h, p.trace, p.quotient, p.ldt, 1)
        self.aux = None     # `extend` needs the challenges
        self.quotient_randomizer = randomness.quotient_randomizer(lib, self.randomness_seed, p.num_quotient_randomizers)
        self.timings, self.wall, self.opened = {}, {}, {}
        self.capture = None
        return self

    def _extend(self, challenges):
        """MasterMainTable::extend (master_table.rs:1006-1075) -> the auxiliary MasterTable"""
        ctx, p, lib = self.ctx, self.p, self.ctx.lib
        n, h = p.trace.length, p.h
        d_aux = ctx.alloc(NUM_AUX * n * 3)
        # the batch-randomizer column (master_table.rs:1017-1024): n XFieldElements of the seeded stream, made on the device
        ctx._check(lib.tvm_stdrng_elements(ctx.handle, randomness.batch_randomizer_seed(self.randomness_seed), 3 * n,
                                           d_aux.ptr + 8 * (NUM_AUX - 1) * n * 3), "tvm_stdrng_elements")
        master_table.extend(ctx, self.main.d_trace, d_aux, n, challenges)
        rnd = randomness.trace_randomizers(lib, randomness.aux_seed(self.randomness_seed), NUM_AUX, h, 3)
        return MasterTable.from_device(ctx, d_aux, ctx.to_device(rnd), NUM_AUX, n, h, p.trace, p.quotient, p.ldt, 3)

    def _init_synthetic(self, mt, fk, n_cols, n, h, dom, seed):
        MasterTable.from_device(self.ctx, self.ctx.synthetic(n_cols * n * fk, seed), self.ctx.synthetic(n_cols * h * fk, seed + 1),
                                n_cols, n, h, *dom, fk, into=mt)

    def _timed(self, name):
        prover = self

        class T:
            def __enter__(self):
                if prover.profile:
                    prover.ctx.sync()
                    self.t0 = time.perf_counter()
                    prover.ctx.timer_start()

            def __exit__(self, *a):
                if prover.profile:
                    prover.timings[name] = prover.timings.get(name, 0.0) + prover.ctx.timer_stop()
                    prover.wall[name] = prover.wall.get(name, 0.0) + 1e3 * (time.perf_counter() - self.t0)
        return T()

    def _root(self, d_nodes):
        """node 1 of a device node array (MerkleTree::root): one 40-byte copy, which also drains the stream"""
        out = np.empty(5, np.uint64)
        self.ctx._check(self.ctx.lib.tvm_memcpy_d2h(self.ctx.handle, out.ctypes.data, d_nodes.ptr + 40, 40), "root")
        return out

    def _table_tree(self, table_handle, n):
        d = self.ctx.alloc(10 * n)
        self.ctx._check(self.ctx.lib.tvm_table_merkle_tree(self.ctx.handle, table_handle, n, d.ptr), "merkle")
        return d

    # -- the steps that touch whole extended master tables; triton_vm_amd/sharded.py overrides them to split the
    #    extended rows across GPUs, triton_vm_amd/jit.py to evaluate them coset by coset ----------------------
    def _extend_master_table(self, mt):
        """maybe_low_degree_extend_all_columns (master_table.rs:258-322): the cached path"""
        mt.maybe_low_degree_extend_all_columns()

    def _commit_master_table(self, mt):
        """hash_all_ldt_domain_rows + merkle_tree (master_table.rs:443-468) -> device node array [2L][5]"""
        return self._table_tree(mt._need_table(), self.p.ldt.length)

    def _quotient_codeword(self, challenges, quotient_weights):
        """all_quotients_combined over the quotient domain (master_table.rs:1264-1363) -> device XFE vector"""
        valid = getattr(self, "assume_valid_trace", False)
        if valid:
            self.ctx.assume_valid_trace(True)
        try:
            return stark.all_quotients_combined(self.ctx, self.main, self.aux, self.p.trace, self.p.quotient, challenges,
                                                quotient_weights)
        finally:
            if valid:
                self.ctx.assume_valid_trace(False)

    def _out_of_domain_rows(self, mt, points):
        """out_of_domain_row at several indeterminates (master_table.rs:348-390) -> host [n_points][n_cols][3]"""
        return mt.out_of_domain_rows(points)

    def _reveal_master_rows(self, mt, row_indices):
        """reveal_rows (master_table.rs:548-555) -> host array"""
        return mt.reveal_rows(row_indices)

    def _auth_nodes(self, d_nodes, n_leaves, indices):
        return stark.auth_nodes(self.ctx, d_nodes, n_leaves, indices)

    def release(self):
        """give the device memory of both master tables (traces, randomizers, cached extensions) back to the context
        now, without waiting for the garbage collector"""
        for mt in (self.main, self.aux):
            mt.clear_cache()
            for buf in (mt.d_trace, mt.d_randomizers):
                buf.free()

    def _fri(self, combination, ps):
        p = self.p
        a_indices, self.last_codeword, self.last_polynomial, self.last_domain = stark.fri_prove(
            self.ctx, p.ldt, p.fri_rounds, p.num_collinearity_checks, combination, ps)
        return a_indices

    def prove(self, profile=False):
        """One pass of the hot path.  Returns the proof stream (`.proof()` is the reference's Proof)."""
        self.profile = profile
        ctx, lib, p = self.ctx, self.ctx.lib, self.p
        ps = self.transcript = ProofStream(lib)
        ps.alter_fiat_shamir_state_with(self.claim.encode())                                       # stark.rs:336-339
        ps.enqueue("log2 padded height", [field.to_mont(p.padded_height.bit_length() - 1)])        # stark.rs:354
        L = p.ldt.length
        short = p.ldt if p.ldt.length <= p.quotient.length else p.quotient

        # 4-6: main table LDE, Merkle tree, challenges  (stark.rs:367-377)
        with self._timed("main LDE"):
            self._extend_master_table(self.main)
        with self._timed("main Merkle"):
            main_nodes = self._commit_master_table(self.main)
        ps.enqueue("main root", self._root(main_nodes))
        challenges = derive_challenges(lib, ps.sample_scalars(NUM_SAMPLED_CHALLENGES), self.claim)

        # 8-9: aux table (`extend` runs on the device when the prover started from an execution trace; otherwise the
        # caller's auxiliary trace is already resident)
        if self.aux is None:
            with self._timed("extend"):
                self.aux = self._extend(challenges)
        with self._timed("aux LDE"):
            self._extend_master_table(self.aux)
        with self._timed("aux Merkle"):
            aux_nodes = self._commit_master_table(self.aux)
        ps.enqueue("aux root", self._root(aux_nodes))
        quotient_weights = xfe_powers(lib, ps.sample_scalars(1)[0], 0, NUM_CONSTRAINTS)

        # 10: quotient codeword, segments, randomization  (stark.rs:405-423)
        with self._timed("AIR quotients"):
            d_quot = self._quotient_codeword(challenges, quotient_weights)
        with self._timed("quotient segments LDE"):
            qs = stark.quotient_segments(ctx, d_quot, p.quotient, p.ldt, self.quotient_randomizer)
        del d_quot
        # 12: quotient Merkle tree  (stark.rs:425-446)
        with self._timed("quotient Merkle"):
            quot_nodes = self._table_tree(qs.table, L)
        ps.enqueue("quot root", self._root(quot_nodes))

        # 13: out-of-domain rows  (stark.rs:450-495)
        alpha = ps.sample_scalars(1)[0]
        alpha_next = np.array([field.mont_mul(int(c), p.trace.generator) for c in alpha], np.uint64)
        with self._timed("out-of-domain rows"):
            ood_main = self._out_of_domain_rows(self.main, [alpha, alpha_next])
            ood_aux = self._out_of_domain_rows(self.aux, [alpha, alpha_next])
            a4 = xfe_powers(lib, alpha, 4, 1)[0]
            zeta_alpha = np.array([field.mont_mul(int(c), stark.ZETA) for c in alpha], np.uint64)
            za4 = xfe_powers(lib, zeta_alpha, 4, 1)[0]
            seg_ood = np.zeros((5, 2, 3), np.uint64)
            for k in range(5):
                view = _Slice(qs.polys, k * qs.poly_len * 3)
                seg_ood[k] = stark.evaluate_at_points(ctx, view, qs.poly_len, [a4, za4])
        ps.enqueue("ood main", ood_main[0]); ps.enqueue("ood aux", ood_aux[0])
        ps.enqueue("ood main next", ood_main[1]); ps.enqueue("ood aux next", ood_aux[1])
        ps.enqueue("ood quot p", seg_ood[:4, 0]); ps.enqueue("ood quot r", seg_ood[1:, 1])

        # 14-15: combination weights, linear combinations  (stark.rs:497-543)
        w_main_aux, w_quot, w_deep = ps.sample_scalars(3)
        weights_ma = xfe_powers(lib, w_main_aux, 0, NUM_MAIN + NUM_AUX)
        weights_q = xfe_powers(lib, w_quot, 0, 5)
        weights_d = xfe_powers(lib, w_deep, 0, NUM_DEEP)
        with self._timed("linear combination"):
            comb = self.main.weighted_sum_of_columns(weights_ma[:NUM_MAIN])
            comb_aux = self.aux.weighted_sum_of_columns(weights_ma[NUM_MAIN:])
            ctx._check(lib.tvm_xfe_add_assign(ctx.handle, comb.ptr, comb_aux.ptr, 2 * p.trace.length), "add")
            n_comb = p.trace.length + p.h
            main_aux_codeword = short.evaluate(ctx, comb, n_comb, 3)
            wp, wr = weights_q.copy(), weights_q.copy()
            wp[4] = 0
            wr[0] = 0
            # values of the P and R polynomials on the short domain (stark.rs:536-539): its points are the rows
            # i * L/|short| of the segment table, which was evaluated on the LDT domain
            cw_p, cw_r = qs.linear_combination(wp, short.length), qs.linear_combination(wr, short.length)
            ma_values = stark.evaluate_at_points(ctx, comb, n_comb, [alpha, alpha_next])
        p_value, r_value = np.zeros(3, np.uint64), np.zeros(3, np.uint64)
        for k in range(4):
            p_value = xfe_add(p_value, xfe_mul(lib, weights_q[k], seg_ood[k, 0]))
        for k in range(1, 5):
            r_value = xfe_add(r_value, xfe_mul(lib, weights_q[k], seg_ood[k, 1]))

        # 16: DEEP  (stark.rs:545-639)
        with self._timed("DEEP"):
            combination = stark.deep_codeword(ctx, [main_aux_codeword, main_aux_codeword, cw_p, cw_r], short,
                                              [alpha, alpha_next, a4, za4], [ma_values[0], ma_values[1], p_value, r_value],
                                              weights_d)
        if self.capture is not None:
            self.capture.update(challenges=challenges, quotient_weights=quotient_weights, alpha=alpha,
                                weights_ma=weights_ma, weights_q=weights_q, weights_d=weights_d,
                                main_root=self._root(main_nodes), aux_root=self._root(aux_nodes),
                                quot_root=self._root(quot_nodes), ood_main=ood_main, ood_aux=ood_aux, seg_ood=seg_ood,
                                combination=combination.download((short.length, 3)))
        del main_aux_codeword, cw_p, cw_r, comb, comb_aux
        if short.length != L:  # stark.rs:629-639: the quotient domain was the short one -- extend to the LDT domain
            with self._timed("DEEP"):
                combination = p.quotient.low_degree_extension(ctx, combination, p.ldt, 3)

        # 17: the low-degree test  (stark.rs:641-663)
        if p.stir is not None:
            with self._timed("STIR"):
                a_indices = p.stir.prove(ctx, combination, ps)
                self.last_polynomial = p.stir.final_polynomial
        else:
            with self._timed("FRI"):
                a_indices = self._fri(combination, ps)

        # 18: the out-of-domain point must not collide with a revealed in-domain point  (stark.rs:645-663)
        if not a4[1] and not a4[2]:
            zeta4 = field.mont_pow(stark.ZETA, 4)
            revealed = {p.ldt.value(int(i)) for i in a_indices}
            if int(a4[0]) in revealed or field.mont_mul(int(a4[0]), zeta4) in revealed:
                raise ZeroKnowledgeViolation("the out-of-domain point conflicts with a revealed in-domain row")

        # 19: open the trace leafs  (stark.rs:665-716)
        with self._timed("open trace leafs"):
            for name, mt, nodes in (("main", self.main, main_nodes), ("aux", self.aux, aux_nodes)):
                rows = self._reveal_master_rows(mt, a_indices)
                self.opened[name] = rows
                ps.enqueue(f"{name} rows", rows, fiat_shamir=False)
                ps.enqueue(f"{name} auth", self._auth_nodes(nodes, L, a_indices), fiat_shamir=False)
            ix = np.array(a_indices, np.uint64)
            qrows = np.empty((ix.size, 15), np.uint64)
            ctx._check(lib.tvm_table_reveal_rows(ctx.handle, qs.table, L, ix.ctypes.data, ix.size, qrows.ctypes.data), "q rows")
            ps.enqueue("quot rows", qrows, fiat_shamir=False)
            self.opened_at, self.opened_quotient_rows = list(a_indices), qrows
            ps.enqueue("quot auth", self._auth_nodes(quot_nodes, L, a_indices), fiat_shamir=False)
        self.main.clear_cache()
        self.aux.clear_cache()
        qs.free()
        ctx.sync()
        return ps


class _Slice:
    """A view into a DeviceBuffer (word offset), enough for entry points that take `.ptr`."""

    def __init__(self, buf, word_offset):
        self.buf, self.ptr = buf, buf.ptr + 8 * word_offset
