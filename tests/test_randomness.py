"""The prover's seeded randomness (triton_vm_amd/randomness.py, tvm_host_stdrng_elements, tvm_stdrng_elements) against the
oracle's restatement of rand's StdRng (oracle/ref_rng.py -- the one that reproduces the reference's AIR fingerprint and both
proof-digest snapshots) and of offset_rng_seed (/root/reference/triton-vm/src/table/master_table.rs:630-662)."""
import numpy as np
import pytest

from triton_vm_amd import randomness as R


@pytest.mark.parametrize("n", [1, 3, 4, 5, 64, 1000, 4099])
def test_stdrng_elements_host_and_device_equal_the_oracle_stream(ctx, orc, n):
    from oracle import ref_rng

    seed = bytes((7 * n + k) & 0xFF for k in range(32))
    rng = ref_rng.StdRng.from_seed(seed)
    want = orc.to_mont(np.array([rng.range_canon() for _ in range(n)], dtype=object))
    assert (R.random_elements(ctx.lib, seed, n) == want).all()
    d = ctx.alloc(n + 2)
    d.upload(np.full(n + 2, 77, np.uint64))
    ctx._check(ctx.lib.tvm_stdrng_elements(ctx.handle, seed, n, d.ptr + 8), "tvm_stdrng_elements")
    got = d.download()
    assert (got[1:n + 1] == want).all() and got[0] == 77 and got[n + 1] == 77     # nothing outside the n elements


@pytest.mark.parametrize("n_streams,per_stream,seed", [
    (1, 5, bytes(range(32))), (7, 198, bytes(range(32))), (3, 594, bytes(range(50, 82))),
    (5, 9, bytes([255]) * 8 + bytes(24)),      # the stream number carries out of the first 64 bits of the seed
    (4, 4, bytes([254]) + bytes([255]) * 31)])  # ... and wraps around 2^256 (offset_rng_seed: wrapping)
def test_stdrng_streams_are_the_offset_seed_streams(ctx, n_streams, per_stream, seed):
    """tvm_stdrng_streams: stream s = StdRng::from_seed(seed + s), i.e. a table's trace randomizers column by column
    (master_table.rs:423-434 with rng_from_offset_seed, master_table.rs:630-662), in one launch"""
    d = ctx.alloc(n_streams * per_stream + 2)
    d.upload(np.full(n_streams * per_stream + 2, 77, np.uint64))
    ctx._check(ctx.lib.tvm_stdrng_streams(ctx.handle, seed, n_streams, per_stream, d.ptr + 8), "tvm_stdrng_streams")
    got = d.download()
    assert got[0] == 77 and got[-1] == 77
    for s in range(n_streams):
        want = R.random_elements(ctx.lib, R.offset_rng_seed(seed, s), per_stream)
        assert (got[1 + s * per_stream:1 + (s + 1) * per_stream] == want).all()


def test_offset_rng_seed_carries_across_the_whole_seed():
    from oracle import real_prover

    ones = bytes([255]) * 32
    for seed, offset in ((bytes(32), 379), (ones, 1), (bytes([255]) * 8 + bytes(24), 471), (bytes(range(32)), 2**64 - 1)):
        assert R.offset_rng_seed(seed, offset) == real_prover.offset_rng_seed(seed, offset)
    assert R.offset_rng_seed(ones, 1) == bytes(32)
    # linear: offsetting by a then b is offsetting by a + b (master_table.rs:614-616)
    s = bytes(range(100, 132))
    assert R.offset_rng_seed(R.offset_rng_seed(s, 379), 91) == R.offset_rng_seed(s, 470)
    assert R.batch_randomizer_seed(s) == R.offset_rng_seed(s, 379 + 91)


def test_randomizer_shapes(ctx):
    seed = bytes(range(32))
    main = R.trace_randomizers(ctx.lib, seed, 5, 7, 1)
    aux = R.trace_randomizers(ctx.lib, R.aux_seed(seed), 3, 7, 3)
    assert main.shape == (5, 7) and aux.shape == (3, 7, 3)
    # column c is the stream of the seed offset by c (master_table.rs:429-433)
    assert (main[3] == R.random_elements(ctx.lib, R.offset_rng_seed(seed, 3), 7)).all()
    assert (aux[2].reshape(-1) == R.random_elements(ctx.lib, R.offset_rng_seed(seed, 379 + 2), 21)).all()
    assert (R.quotient_randomizer(ctx.lib, seed, 4).reshape(-1) == R.random_elements(ctx.lib, R.offset_rng_seed(seed, 471), 12)).all()
