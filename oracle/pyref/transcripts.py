"""ORACLE (test infrastructure only) -- Fiat-Shamir transcripts of
/root/reference/src/transcript/{mod,merlin,keccak256,blake3}.rs restated from the
public specifications of the hash functions (the crates merlin 3.0.0, sha3, blake3
are not vendored in the reference: Cargo.toml:29-31).

Pinned by published known-answer vectors (tests/test_oracle_pyref.py):
  Keccak-256("") , SHA3-256 cross-check against hashlib (same permutation),
  BLAKE3("") / BLAKE3(0..250 pattern), Merlin "test protocol" equivalence vector.
"""
import struct

# ------------------------------------------------------------ Keccak-f[1600]
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & _M64 if n else v


def keccak_f1600(state: bytearray):
    a = [[0] * 5 for _ in range(5)]
    for x in range(5):
        for y in range(5):
            a[x][y] = int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little")
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def _sponge256(data: bytes, pad: int) -> bytes:
    rate = 136
    st = bytearray(200)
    msg = bytearray(data)
    msg.append(pad)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    for off in range(0, len(msg), rate):
        for i in range(rate):
            st[i] ^= msg[off + i]
        keccak_f1600(st)
    return bytes(st[:32])


def keccak256(data: bytes) -> bytes:
    """Legacy Keccak-256 (pad 0x01), what the sha3 crate's Keccak256 computes."""
    return _sponge256(data, 0x01)


def sha3_256(data: bytes) -> bytes:
    return _sponge256(data, 0x06)


# --------------------------------------------------------------------- BLAKE3
_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
_CHUNK_START, _CHUNK_END, _PARENT, _ROOT = 1, 2, 4, 8
_M32 = 0xFFFFFFFF


def _ror32(v, n):
    return ((v >> n) | (v << (32 - n))) & _M32


def _g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & _M32
    s[d] = _ror32(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _ror32(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b] + my) & _M32
    s[d] = _ror32(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & _M32
    s[b] = _ror32(s[b] ^ s[c], 7)


def _compress(cv, block_words, counter, block_len, flags):
    s = list(cv) + _IV[:4] + [counter & _M32, (counter >> 32) & _M32, block_len, flags]
    m = list(block_words)
    for rnd in range(7):
        _g(s, 0, 4, 8, 12, m[0], m[1])
        _g(s, 1, 5, 9, 13, m[2], m[3])
        _g(s, 2, 6, 10, 14, m[4], m[5])
        _g(s, 3, 7, 11, 15, m[6], m[7])
        _g(s, 0, 5, 10, 15, m[8], m[9])
        _g(s, 1, 6, 11, 12, m[10], m[11])
        _g(s, 2, 7, 8, 13, m[12], m[13])
        _g(s, 3, 4, 9, 14, m[14], m[15])
        m = [m[_PERM[i]] for i in range(16)]
    for i in range(8):
        s[i] ^= s[i + 8]
        s[i + 8] ^= cv[i]
    return s


def _words(block: bytes):
    block = block + b"\0" * (64 - len(block))
    return list(struct.unpack("<16I", block))


def _chunk_output(chunk: bytes, counter: int):
    """Returns (cv, block_words, block_len, flags) of the chunk's LAST block, not yet compressed."""
    cv = list(_IV)
    blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
    for i, blk in enumerate(blocks):
        flags = (_CHUNK_START if i == 0 else 0) | (_CHUNK_END if i == len(blocks) - 1 else 0)
        if i == len(blocks) - 1:
            return cv, _words(blk), len(blk), flags
        cv = _compress(cv, _words(blk), counter, 64, flags)[:8]


def blake3(data: bytes) -> bytes:
    """BLAKE3 default hash mode, 32-byte output."""
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]
    if len(chunks) == 1:
        cv, bw, bl, fl = _chunk_output(chunks[0], 0)
        out = _compress(cv, bw, 0, bl, fl | _ROOT)
        return struct.pack("<8I", *out[:8])
    # chunk chaining values, then binary tree merge (left subtree = largest power of two)
    cvs = []
    for i, ch in enumerate(chunks):
        cv, bw, bl, fl = _chunk_output(ch, i)
        cvs.append(_compress(cv, bw, i, bl, fl)[:8])

    def merge(nodes, root):
        if len(nodes) == 1:
            return nodes[0]
        split = 1
        while split * 2 < len(nodes):
            split *= 2
        left = merge(nodes[:split], False)
        right = merge(nodes[split:], False)
        out = _compress(_IV, left + right, 0, 64, _PARENT | (_ROOT if root else 0))
        return out[:8]

    return struct.pack("<8I", *merge(cvs, True))


# ------------------------------------------------------- STROBE-128 / Merlin
_STROBE_R = 166
_F_I, _F_A, _F_C, _F_T, _F_M, _F_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, _STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[_STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & _F_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (_F_C | _F_K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(_F_M | _F_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(_F_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(_F_I | _F_A | _F_C, more)
        return self._squeeze(n)


class MerlinTranscriptRaw:
    """merlin::Transcript (v3.0.0): new / append_message / challenge_bytes."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(message)), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)


# ----------------------------------------- the reference's Transcript impls
def make_transcripts(curve):
    """Returns the three transcript classes bound to curve.r (Challenge = Fr)."""
    r = curve.r
    nbits = r.bit_length()

    class MerlinFieldTranscript:
        """src/transcript/merlin.rs:13-37.  F::from_random_bytes on 64 bytes
        [ark-ff, from memory]: take the first 32 bytes little-endian, mask to the
        modulus bit length, accept iff < r, else draw again."""

        def __init__(self, name):
            self.m = MerlinTranscriptRaw(name)

        def append_message(self, label, message):
            self.m.append_message(label, message)

        def challenge(self, label):
            while True:
                buf = self.m.challenge_bytes(label, 64)
                v = int.from_bytes(buf[:32], "little") & ((1 << nbits) - 1)
                if v < r:
                    return v

    class _HashTranscript:
        """src/transcript/keccak256.rs:12-43 / blake3.rs:12-43: state := bytes;
        challenge = H(state || label) big-endian mod r; state := digest.
        `name` is ignored (keccak256.rs:19-24)."""
        H = None

        def __init__(self, name):
            self.t = b""

        def append_message(self, label, message):
            self.t += label + message

        def challenge(self, label):
            d = type(self).H(self.t + label)
            self.t = d
            return int.from_bytes(d, "big") % r

    class Keccak256Transcript(_HashTranscript):
        H = staticmethod(keccak256)

    class Blake3Transcript(_HashTranscript):
        H = staticmethod(blake3)

    return {"merlin": MerlinFieldTranscript, "keccak256": Keccak256Transcript, "blake3": Blake3Transcript}
