"""csrc/field.h (device arithmetic: hand-scheduled carry chains on gfx950, plain C under the emulator) against
the oracle's BFieldElement arithmetic, element by element, on random words and on the edge words around
0, 2^32 and p where the conditional corrections of the modular add / sub / Montgomery reduction switch."""
import numpy as np
import pytest

from triton_vm_amd import field

P = field.P
EDGE = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, 0xFFFFFFFE, P - 1, P - 2, P - 0xFFFFFFFF, P - 0x100000000,
        0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF, 0x8000000000000000, 0x7FFFFFFFFFFFFFFF,
        0x00000001FFFFFFFF, 0xFFFFFFFF00000000 - 1, (1 << 63) + (1 << 31), 0xFFFFFFFF << 31]


@pytest.mark.parametrize("op,name", [(0, "orc_bfe_add"), (1, "orc_bfe_sub"), (2, "orc_bfe_mul")])
def test_field_ops_match_oracle(ctx, orc, op, name):
    rng = np.random.default_rng(op)
    edge = np.array([e % P for e in EDGE], np.uint64)
    a = np.concatenate([np.repeat(edge, len(edge)), orc.random_elements(rng, 4096)])
    b = np.concatenate([np.tile(edge, len(edge)), orc.random_elements(rng, 4096)])
    out, da, db = ctx.alloc(a.size), ctx.to_device(a), ctx.to_device(b)
    ctx._check(ctx.lib.tvm_field_op(ctx.handle, op, da.ptr, db.ptr, out.ptr, a.size), "field_op")
    fn = getattr(orc.lib(), name)
    want = np.array([fn(int(x), int(y)) for x, y in zip(a, b)], np.uint64)
    assert (out.download() == want).all()


def test_pow7_matches_oracle(ctx, orc):
    rng = np.random.default_rng(7)
    a = np.concatenate([np.array([e % P for e in EDGE], np.uint64), orc.random_elements(rng, 2048)])
    out, da = ctx.alloc(a.size), ctx.to_device(a)
    ctx._check(ctx.lib.tvm_field_op(ctx.handle, 3, da.ptr, da.ptr, out.ptr, a.size), "field_op")
    want = np.array([orc.lib().orc_bfe_pow(int(x), 7) for x in a], np.uint64)
    assert (out.download() == want).all()


def test_multiplication_by_powers_of_two(ctx, orc):
    """csrc/ntt_shift.h: x * 2^s as a shift and one folding step (s < 96; 2^96 = -1 carries the sign), for every s < 192, on
    the edge words and random words, against plain integer arithmetic on the Montgomery words ((aR) 2^s = (a 2^s) R)."""
    rng = np.random.default_rng(96)
    edge = np.array([e % P for e in EDGE], np.uint64)
    base = np.concatenate([edge, orc.random_elements(rng, 200)])
    a = np.tile(base, 192)
    b = np.repeat(np.arange(192, dtype=np.uint64), base.size)
    out, da, db = ctx.alloc(a.size), ctx.to_device(a), ctx.to_device(b)
    ctx._check(ctx.lib.tvm_field_op(ctx.handle, 4, da.ptr, db.ptr, out.ptr, a.size), "field_op")
    want = np.array([int(x) * pow(2, int(s), P) % P for x, s in zip(a, b)], np.uint64)
    assert (out.download() == want).all()


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("dit,inverse", [(1, 0), (0, 1), (0, 0), (1, 1)])
def test_power_of_two_twiddle_transforms(ctx, orc, K, dit, inverse):
    """the 2^K-point transforms with shift twiddles (K <= 4) against the DFT matrix of the domain's 2^K-th root of unity
    [twenty-first BFieldElement::primitive_root_of_unity = 1753635133440165772^(2^32 / 2^K)]: decimation in time takes
    bit-reversed input, decimation in frequency yields bit-reversed output (the conventions of lds_ntt_group)."""
    rng = np.random.default_rng(K * 4 + dit * 2 + inverse)
    R, groups = 1 << K, 40
    w = pow(1753635133440165772, (1 << 32) >> K, P)
    assert pow(w, R, P) == 1 and pow(w, R // 2, P) == P - 1
    if inverse:
        w = pow(w, P - 2, P)
    brev = lambda i: int(format(i, f"0{K}b")[::-1], 2)
    edge = np.array([e % P for e in EDGE], np.uint64)
    a = np.concatenate([edge[:R * (len(edge) // R)], orc.random_elements(rng, R * groups)])
    out, da = ctx.alloc(a.size), ctx.to_device(a)
    ctx._check(ctx.lib.tvm_field_op(ctx.handle, 32 + 4 * K + 2 * dit + inverse, da.ptr, da.ptr, out.ptr, a.size), "field_op")
    got = out.download().reshape(-1, R)
    for g, x in enumerate(a.reshape(-1, R)):
        x = [int(v) for v in x]
        nat_in = [x[brev(i)] for i in range(R)] if dit else x
        dft = [sum(nat_in[j] * pow(w, i * j, P) for j in range(R)) % P for i in range(R)]   # linear: Montgomery words stay Montgomery words
        want = dft if dit else [dft[brev(i)] for i in range(R)]
        assert [int(v) for v in got[g]] == want, (g, K, dit, inverse)
