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
