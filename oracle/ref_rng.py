"""TEST INFRASTRUCTURE (oracle): restatement of the random number generator the reference's tests and
prover use, so that reference-held golden values that depend on `StdRng` can be reproduced here.

Third-party algorithms restated (the crates are not vendored under /root/reference; the workspace pins
`rand = "0.10.1"` and `twenty-first = "2.0.0"`, /root/reference/Cargo.toml:96,104):

* `StdRng` = ChaCha with 12 rounds (rand's documented choice), 64-bit block counter in words 12-13, stream id 0 in
  words 14-15, output = the 16 state words of consecutive blocks, little-endian u32 stream; `next_u64` = two
  consecutive u32 words, low word first (rand_core `BlockRng`).
* `SeedableRng::seed_from_u64` (rand_core): the 32 seed bytes are eight outputs of PCG32 (multiplier
  6364136223846793005, increment 11634580027462260723, state advanced before each output, XSH-RR output).
* `Rng::random_range(0..=MAX)` for u64 and the `Distribution<BFieldElement>` built on it: several variants are
  provided (rand 0.9's widening-multiply "Canon" sampler, rand 0.8's zone rejection sampler, plain rejection,
  reduction mod p).  tests/test_air_fingerprint.py records which variant reproduces the reference's golden value
  (`air_constraints_evaluators_have_not_changed`, /root/reference/triton-vm/src/table/master_table.rs:2328-2414).
"""
P = 2**64 - 2**32 + 1
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & M32


def chacha_block(key_words, counter, stream, rounds):
    """One ChaCha block: 16 output words."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, s)]


def pcg32_seed(state):
    """rand_core seed_from_u64: 32 bytes from PCG32."""
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & M32
        out += x.to_bytes(4, "little")
    return out


def splitmix64_seed(state):
    """Alternative seed expansion (SplitMix64), tried as a variant."""
    out = b""
    for _ in range(4):
        state = (state + 0x9E3779B97F4A7C15) & M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        z ^= z >> 31
        out += z.to_bytes(8, "little")
    return out


class StdRng:
    """ChaCha12-based generator with the u32 word stream semantics of rand's BlockRng."""

    def __init__(self, seed_bytes, rounds=12):
        assert len(seed_bytes) == 32
        self.key = [int.from_bytes(seed_bytes[4 * i:4 * i + 4], "little") for i in range(8)]
        self.rounds = rounds
        self.counter = 0
        self.buf = []
        self.idx = 0
        self.draws = 0

    @classmethod
    def from_seed(cls, seed_bytes, **kw):
        return cls(bytes(seed_bytes), **kw)

    @classmethod
    def seed_from_u64(cls, state, expand=pcg32_seed, **kw):
        return cls(expand(state), **kw)

    def next_u32(self):
        if self.idx >= len(self.buf):
            self.buf = chacha_block(self.key, self.counter, 0, self.rounds)
            self.counter += 1
            self.idx = 0
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self):
        self.draws += 1
        lo = self.next_u32()
        hi = self.next_u32()
        return (hi << 32) | lo

    def fill_bytes(self, n):
        """rand_core fill_bytes on a block generator: consumes whole u32 words, little-endian."""
        out = b""
        while len(out) < n:
            out += self.next_u32().to_bytes(4, "little")
        return out[:n]

    # ---- u64 range samplers for 0..=p-1 (range = p) -------------------------------------------------------
    def range_canon(self, rng_size=P):
        """rand 0.9 UniformInt::sample_single_inclusive (biased Canon's method, one extra draw)."""
        prod = self.next_u64() * rng_size
        result, lo = prod >> 64, prod & M64
        if lo > ((-rng_size) & M64):
            new_hi = (self.next_u64() * rng_size) >> 64
            if lo + new_hi > M64:
                result += 1
        return result

    def range_zone(self, rng_size=P):
        """rand 0.8 sample_single_inclusive: widening multiply with a conservative rejection zone."""
        lz = 64 - rng_size.bit_length()
        zone = ((rng_size << lz) - 1) & M64
        while True:
            prod = self.next_u64() * rng_size
            hi, lo = prod >> 64, prod & M64
            if lo <= zone:
                return hi

    def range_zone_exact(self, rng_size=P):
        """Uniform::new_inclusive(..).sample(): exact zone ints_to_reject = (2^64 - range) % range."""
        ints_to_reject = ((1 << 64) - rng_size) % rng_size
        zone = M64 - ints_to_reject
        while True:
            prod = self.next_u64() * rng_size
            hi, lo = prod >> 64, prod & M64
            if lo <= zone:
                return hi

    def range_reject(self, rng_size=P):
        while True:
            v = self.next_u64()
            if v < rng_size:
                return v

    def range_mod(self, rng_size=P):
        return self.next_u64() % rng_size


BFE_SAMPLERS = {
    "canon": StdRng.range_canon,
    "zone": StdRng.range_zone,
    "zone_exact": StdRng.range_zone_exact,
    "reject": StdRng.range_reject,
    "mod": StdRng.range_mod,
}
