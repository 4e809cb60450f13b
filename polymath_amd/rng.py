"""Python twin of polymath_amd/host/rng.hpp: the reference's random sources (rand 0.8 StdRng = ChaCha12, rand_core's
seed_from_u64, ark_std::test_rng, ark-ff's Fp::rand, ark-poly's sample_element_outside_domain), so that
Polymath.setup(circuit, rng) / Polymath.prove(pk, circuit, rng) take an rng like /root/reference/src/lib.rs:63-78."""
import struct

_M32, _M64 = 0xFFFFFFFF, (1 << 64) - 1


def _rotl(v, c):
    return ((v << c) | (v >> (32 - c))) & _M32


def chacha_block(key_words, tail_words, rounds):
    """key_words: 8 u32, tail_words: 4 u32 (state words 12..15) -> 16 output words (RFC 7539 section 2.3, `rounds` rounds)."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + list(tail_words)
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & _M32 for a, b in zip(x, s)]


class StdRng:
    """rand::rngs::StdRng (rand 0.8) == rand_chacha::ChaCha12Rng: block counter in words 12-13, stream id 0."""

    def __init__(self, seed32):
        assert len(seed32) == 32
        self.key = list(struct.unpack("<8I", bytes(seed32)))
        self.counter, self.buf, self.index = 0, [], 16

    @classmethod
    def seed_from_u64(cls, state):
        """rand_core::SeedableRng::seed_from_u64: PCG32 expands the u64 into the 32 seed bytes."""
        mul, inc = 6364136223846793005, 11634580027462260723
        seed = b""
        for _ in range(8):
            state = (state * mul + inc) & _M64
            xorshifted = (((state >> 18) ^ state) >> 27) & _M32
            rot = state >> 59
            seed += struct.pack("<I", ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32)
        return cls(seed)

    @classmethod
    def test_rng(cls):
        """ark_std::test_rng()."""
        return cls(bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16))

    def next_u32(self):
        if self.index >= 16:
            self.buf = chacha_block(self.key, [self.counter & _M32, self.counter >> 32, 0, 0], 12)
            self.counter += 1
            self.index = 0
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


def fr_rand_mont(rng, r):
    """ark-ff Fp::rand: 4 x next_u64 into the limbs (limb 0 first), masked to the modulus' bit length, rejected while >= r.
    Returns the Montgomery REPRESENTATION (the integer the limbs spell)."""
    nb = r.bit_length()
    while True:
        v = 0
        for i in range(4):
            v |= rng.next_u64() << (64 * i)
        v &= (1 << nb) - 1
        if v < r:
            return v


def fr_rand(rng, r):
    """The field ELEMENT F::rand(rng) denotes: its limbs are the Montgomery form, so value = limbs * R^-1 mod r."""
    return fr_rand_mont(rng, r) * pow(1 << 256, -1, r) % r


def sample_element_outside_domain(rng, r, n):
    """ark-poly EvaluationDomain::sample_element_outside_domain (generator.rs:72,77)."""
    while True:
        t = fr_rand(rng, r)
        if pow(t, n, r) != 1:
            return t
