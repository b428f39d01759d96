"""ctypes binding of the CPU ORACLE (oracle/tvm_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by the product package ``triton_vm_amd``.

All arrays are numpy ``uint64`` holding Montgomery raw words (see tvm_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtvm_oracle.so")

P = 2**64 - 2**32 + 1


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("tvm_oracle.c", "tvm_oracle_fast.c", "tvm_oracle.h", "tip5_constants.h", "air_circuit.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Domain(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("generator", C.c_uint64), ("length", C.c_uint64)]

    def __repr__(self):
        return f"Domain(offset={self.offset}, generator={self.generator}, length={self.length})"


_lib = None
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u = C.c_uint64
        for name, res, args in [
            ("orc_bfe_new", u, [u]), ("orc_bfe_value", u, [u]), ("orc_bfe_add", u, [u, u]),
            ("orc_bfe_sub", u, [u, u]), ("orc_bfe_mul", u, [u, u]), ("orc_bfe_inv", u, [u]),
            ("orc_bfe_pow", u, [u, u]), ("orc_bfe_generator", u, []), ("orc_bfe_primitive_root", u, [u]),
            ("orc_domain_of_length", Domain, [u]), ("orc_domain_pow", Domain, [Domain, u]),
            ("orc_domain_value", u, [Domain, u]),
        ]:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(u64p)


def _arr(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


# ---- scalar field helpers -------------------------------------------------------------------
def bfe(v):
    return lib().orc_bfe_new(int(v) % P)


def value(raw):
    return lib().orc_bfe_value(int(raw))


def to_mont(a):
    """canonical values (any integers; reduced mod p) -> Montgomery words"""
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint64):   # python ints: never let numpy guess a float dtype
        a = np.asarray(a, dtype=object)
        a = np.array(a % P, dtype=np.uint64).reshape(a.shape)
    out = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_bfe_new_array(_p(out.reshape(-1)), C.c_uint64(out.size))
    return out


def from_mont(a):
    out = np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).copy()
    lib().orc_bfe_value_array(_p(out.reshape(-1)), C.c_uint64(out.size))
    return out


def random_elements(rng, shape):
    """Uniform canonical Montgomery words < p."""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    return np.ascontiguousarray(a)


def xfe_mul(a, b):
    a, b, o = _arr(a), _arr(b), np.zeros(3, np.uint64)
    lib().orc_xfe_mul(_p(a), _p(b), _p(o))
    return o


def xfe_inv(a):
    a, o = _arr(a), np.zeros(3, np.uint64)
    lib().orc_xfe_inv(_p(a), _p(o))
    return o


def xfe_pow(a, e):
    a, o = _arr(a), np.zeros(3, np.uint64)
    lib().orc_xfe_pow(_p(a), C.c_uint64(e), _p(o))
    return o


def xfe_add(a, b):
    a, b, o = _arr(a), _arr(b), np.zeros(3, np.uint64)
    lib().orc_xfe_add(_p(a), _p(b), _p(o))
    return o


def xfe_sub(a, b):
    a, b, o = _arr(a), _arr(b), np.zeros(3, np.uint64)
    lib().orc_xfe_sub(_p(a), _p(b), _p(o))
    return o


# ---- domains --------------------------------------------------------------------------------
def domain_of_length(n, offset=None):
    d = lib().orc_domain_of_length(n)
    if offset is not None:
        d.offset = int(offset)
    return d


def domain_pow(d, e):
    return lib().orc_domain_pow(d, e)


def domain_values(d):
    out = np.zeros(d.length, np.uint64)
    lib().orc_domain_values(d, _p(out))
    return out


def ntt(a, fk=1):
    a = _arr(a).copy()
    lib().orc_ntt(_p(a), C.c_uint64(a.size // fk), fk)
    return a


def intt(a, fk=1):
    a = _arr(a).copy()
    lib().orc_intt(_p(a), C.c_uint64(a.size // fk), fk)
    return a


def coset_evaluate(coeffs, d, fk=1):
    coeffs = _arr(coeffs)
    out = np.zeros(d.length * fk, np.uint64)
    lib().orc_coset_evaluate(fk, _p(coeffs), C.c_uint64(coeffs.size // fk), d, _p(out))
    return out


def coset_interpolate(values, d, fk=1):
    values = _arr(values)
    out = np.zeros(d.length * fk, np.uint64)
    lib().orc_coset_interpolate(fk, _p(values), d, _p(out))
    return out


# ---- LDE ------------------------------------------------------------------------------------
def randomized_column_interpolant(column, randomizer, fk=1):
    column, randomizer = _arr(column), _arr(randomizer)
    n = column.size // fk
    out = np.zeros(2 * n * fk, np.uint64)
    lib().orc_randomized_column_interpolant(fk, _p(column), C.c_uint64(n), _p(randomizer),
                                            C.c_uint64(randomizer.size // fk), _p(out))
    return out


def lde_table(trace, randomizers, eval_domain, fk=1, out=None):
    """trace [n_cols, n_rows(, 3)] column-major; randomizers [n_cols, h(, 3)] -> [L, n_cols(, 3)].  `out`: an array of that shape to
    fill instead of a new one (a file-backed np.memmap for the heights whose tables do not fit the host's memory)."""
    trace, randomizers = _arr(trace), _arr(randomizers)
    n_cols, n_rows = trace.shape[0], trace.shape[1]
    h = randomizers.shape[1]
    shape = (eval_domain.length, n_cols) + ((3,) if fk == 3 else ())
    if out is None:
        out = np.zeros(shape, np.uint64)
    assert out.shape == shape and out.dtype == np.uint64 and out.flags.c_contiguous
    lib().orc_lde_table(fk, _p(trace), C.c_uint64(n_rows), C.c_uint64(n_cols), _p(randomizers),
                        C.c_uint64(h), eval_domain, _p(out))
    return out


# ---- hashing --------------------------------------------------------------------------------
def tip5_permutation(state):
    s = _arr(state).copy()
    lib().orc_tip5_permutation(_p(s))
    return s


def tip5_trace(state):
    """[6, 16]: the state before the permutation and after each round"""
    s, out = _arr(state), np.zeros((6, 16), np.uint64)
    lib().orc_tip5_trace(_p(s), _p(out))
    return out


def hash_varlen(words):
    w = _arr(words)
    out = np.zeros(5, np.uint64)
    lib().orc_hash_varlen(_p(w) if w.size else None, C.c_size_t(w.size), _p(out))
    return out


def hash_10(words):
    w = _arr(words)
    assert w.size == 10
    out = np.zeros(5, np.uint64)
    lib().orc_hash_10(_p(w), _p(out))
    return out


def hash_pair(left, right):
    l, r, out = _arr(left), _arr(right), np.zeros(5, np.uint64)
    lib().orc_hash_pair(_p(l), _p(r), _p(out))
    return out


def hash_rows(rows):
    rows = _arr(rows)
    n = rows.shape[0]
    w = rows.size // n
    out = np.zeros((n, 5), np.uint64)
    lib().orc_hash_rows(_p(rows), C.c_uint64(n), C.c_uint64(w), _p(out))
    return out


def merkle_tree(leaves):
    leaves = _arr(leaves)
    n = leaves.shape[0]
    nodes = np.zeros((2 * n, 5), np.uint64)
    lib().orc_merkle_tree(_p(leaves), C.c_uint64(n), _p(nodes))
    return nodes


def xfe_to_digest(x):
    x = _arr(x)
    n = x.size // 3
    out = np.zeros((n, 5), np.uint64)
    lib().orc_xfe_to_digest(_p(x), C.c_uint64(n), _p(out))
    return out


# ---- quotient plumbing ----------------------------------------------------------------------
def zerofier_inverses(trace_domain, quotient_domain):
    n = quotient_domain.length
    outs = [np.zeros(n, np.uint64) for _ in range(4)]
    lib().orc_zerofier_inverses(trace_domain, quotient_domain, *[_p(o) for o in outs])
    return outs


def interpolate_quotient_segments(codeword, quotient_domain):
    codeword = _arr(codeword)
    out = np.zeros((4, quotient_domain.length // 4, 3), np.uint64)
    lib().orc_interpolate_quotient_segments(_p(codeword), quotient_domain, _p(out))
    return out


def randomize_quotient_segments(seg_polys, randomizer, ldt_domain, poly_len=None):
    seg_polys, randomizer = _arr(seg_polys), _arr(randomizer)
    seg_len = seg_polys.shape[1]
    n_rand = randomizer.size // 3
    poly_len = poly_len or max(seg_len, n_rand)
    polys = np.zeros((5, poly_len, 3), np.uint64)
    cws = np.zeros((ldt_domain.length, 5, 3), np.uint64)
    lib().orc_randomize_quotient_segments(_p(seg_polys), C.c_uint64(seg_len), _p(randomizer),
                                          C.c_uint64(n_rand), ldt_domain, _p(polys),
                                          C.c_uint64(poly_len), _p(cws))
    return polys, cws


def quotients_combined(main_rows, aux_rows, trace_domain, quotient_domain, challenges, weights):
    """main_rows [Q, 379], aux_rows [Q, 91, 3] (row-major quotient-domain views) -> [Q, 3]"""
    main_rows, aux_rows, challenges, weights = _arr(main_rows), _arr(aux_rows), _arr(challenges), _arr(weights)
    q = quotient_domain.length
    out = np.zeros((q, 3), np.uint64)
    lib().orc_quotients_combined(_p(main_rows), C.c_uint64(main_rows.shape[1]), _p(aux_rows),
                                 C.c_uint64(aux_rows.shape[1]), trace_domain, quotient_domain, _p(challenges),
                                 _p(weights), _p(out))
    return out


def air_constraint_values(main_cur, main_next, aux_cur, aux_next, challenges):
    """All 604 constraint values on one row pair -> [604, 3]; main rows [379] (base field) or [379, 3]."""
    main_cur, main_next, aux_cur, aux_next, challenges = map(_arr, (main_cur, main_next, aux_cur, aux_next, challenges))
    words = 3 if main_cur.ndim == 2 else 1
    out = np.zeros((604, 3), np.uint64)
    lib().orc_air_constraint_values(_p(main_cur), _p(main_next), _p(aux_cur), _p(aux_next), _p(challenges),
                                    C.c_int(words), _p(out))
    return out


# ---- combination / DEEP / FRI ---------------------------------------------------------------
def weighted_sum_of_columns(trace, randomizers, weights, fk=1):
    trace, randomizers, weights = _arr(trace), _arr(randomizers), _arr(weights)
    n_cols, n_rows = trace.shape[0], trace.shape[1]
    out = np.zeros((2 * n_rows, 3), np.uint64)
    lib().orc_weighted_sum_of_columns(fk, _p(trace), C.c_uint64(n_rows), C.c_uint64(n_cols),
                                      _p(randomizers), C.c_uint64(randomizers.shape[1]), _p(weights), _p(out))
    return out


def out_of_domain_row(trace, randomizers, point, fk=1):
    trace, randomizers, point = _arr(trace), _arr(randomizers), _arr(point)
    n_cols, n_rows = trace.shape[0], trace.shape[1]
    out = np.zeros((n_cols, 3), np.uint64)
    lib().orc_out_of_domain_row(fk, _p(trace), C.c_uint64(n_rows), C.c_uint64(n_cols), _p(randomizers),
                                C.c_uint64(randomizers.shape[1]), _p(point), _p(out))
    return out


def poly_eval_xfe(coeffs, point):
    coeffs, point, out = _arr(coeffs), _arr(point), np.zeros(3, np.uint64)
    lib().orc_poly_eval_xfe(_p(coeffs), C.c_uint64(coeffs.size // 3), _p(point), _p(out))
    return out


def deep_codeword(codeword, d, point, value_):
    codeword, point, value_ = _arr(codeword), _arr(point), _arr(value_)
    out = np.zeros((d.length, 3), np.uint64)
    lib().orc_deep_codeword(_p(codeword), d, _p(point), _p(value_), _p(out))
    return out


def fri_split_and_fold(codeword, d, challenge):
    codeword, challenge = _arr(codeword), _arr(challenge)
    out = np.zeros((d.length // 2, 3), np.uint64)
    lib().orc_fri_split_and_fold(_p(codeword), d, _p(challenge), _p(out))
    return out


# ---- the optimised restatement (oracle/tvm_oracle_fast.c): what bench.py's cpu_baseline leg times ------------------------
class fast:
    """Same results as the functions above, bit for bit (tests/test_oracle_fast.py); CPU-efficient forms."""

    @staticmethod
    def tip5_permutation(state):
        s = _arr(state).copy()
        lib().orcf_tip5_permutation(_p(s))
        return s

    @staticmethod
    def hash_rows(rows):
        rows = _arr(rows)
        n = rows.shape[0]
        out = np.zeros((n, 5), np.uint64)
        lib().orcf_hash_rows(_p(rows), C.c_uint64(n), C.c_uint64(rows.size // n), _p(out))
        return out

    @staticmethod
    def merkle_tree(leaves):
        leaves = _arr(leaves)
        n = leaves.shape[0]
        nodes = np.zeros((2 * n, 5), np.uint64)
        lib().orcf_merkle_tree(_p(leaves), C.c_uint64(n), _p(nodes))
        return nodes

    @staticmethod
    def ntt(values, generator):
        a = _arr(values).copy()
        lib().orcf_ntt(_p(a), C.c_uint64(a.size), C.c_uint64(int(generator)))
        return a

    @staticmethod
    def lde_table(trace, randomizers, eval_domain):
        """base-field columns: trace [n_cols, n_rows], randomizers [n_cols, h] -> [L, n_cols]"""
        trace, randomizers = _arr(trace), _arr(randomizers)
        n_cols, n_rows = trace.shape
        out = np.zeros((eval_domain.length, n_cols), np.uint64)
        lib().orcf_lde_table(_p(trace), C.c_uint64(n_rows), C.c_uint64(n_cols), _p(randomizers), C.c_uint64(randomizers.shape[1]),
                             eval_domain, _p(out))
        return out

    @staticmethod
    def quotients_combined(main_rows, aux_rows, trace_domain, quotient_domain, challenges, weights):
        main_rows, aux_rows, challenges, weights = _arr(main_rows), _arr(aux_rows), _arr(challenges), _arr(weights)
        out = np.zeros((quotient_domain.length, 3), np.uint64)
        lib().orcf_quotients_combined(_p(main_rows), C.c_uint64(main_rows.shape[1]), _p(aux_rows), C.c_uint64(aux_rows.shape[1]),
                                      trace_domain, quotient_domain, _p(challenges), _p(weights), _p(out))
        return out

    @staticmethod
    def deep_codeword(codeword, d, point, value_):
        codeword, point, value_ = _arr(codeword), _arr(point), _arr(value_)
        out = np.zeros((d.length, 3), np.uint64)
        lib().orcf_deep_codeword(_p(codeword), d, _p(point), _p(value_), _p(out))
        return out
