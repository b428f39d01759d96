"""The prover's seeded randomness, as the reference draws it (`StdRng::from_seed` of a 32-byte seed, offset per purpose):
trace randomizers (/root/reference/triton-vm/src/table/master_table.rs:423-434), the batch-randomizer column
(master_table.rs:1006-1024), the quotient-segment randomizer (stark.rs:1315-1322).  The generator itself is the C ABI's
host helper tvm_host_stdrng_elements (rand's StdRng is not in the reference tree; a Rust host uses rand)."""
import numpy as np

NUM_MAIN, NUM_AUX = 379, 91


def offset_rng_seed(seed, offset):
    """master_table.rs:630-662: wrapping addition of the little-endian offset into the seed bytes"""
    seed = bytes(seed)
    assert len(seed) == 32
    return ((int.from_bytes(seed, "little") + int(offset)) % (1 << 256)).to_bytes(32, "little")


def random_elements(lib, seed, n):
    """n draws of rng.random::<BFieldElement>() (an XFieldElement is three of them), Montgomery words"""
    out = np.empty(n, np.uint64)
    lib.tvm_host_stdrng_elements(bytes(seed), n, out.ctypes.data)
    return out


def aux_seed(seed):
    return offset_rng_seed(seed, NUM_MAIN)


def trace_randomizers(lib, table_seed, n_cols, h, field_kind):
    """trace_randomizer_for_column for every column -> [n_cols][h](, [3])"""
    out = np.stack([random_elements(lib, offset_rng_seed(table_seed, c), h * field_kind) for c in range(n_cols)])
    return out.reshape((n_cols, h) + ((3,) if field_kind == 3 else ()))


def batch_randomizer_seed(seed):
    return offset_rng_seed(aux_seed(seed), NUM_AUX)


def batch_randomizer_column(lib, seed, n_rows):
    return random_elements(lib, batch_randomizer_seed(seed), 3 * n_rows).reshape(n_rows, 3)


def quotient_randomizer(lib, seed, n_coefficients):
    return random_elements(lib, offset_rng_seed(seed, NUM_MAIN + NUM_AUX + 1), 3 * n_coefficients).reshape(-1, 3)
