"""Fiat-Shamir transcripts: the host-side mirror of /root/reference/src/transcript/*.rs.

trait Transcript { new(name); append_message(label, message); challenge(label) -> Fr }
(transcript/mod.rs:17-29) with the reference's three implementations: Merlin (STROBE-128 over
Keccak-f[1600], merlin 3.0.0), Keccak256 and Blake3 (hash(state || label), big-endian mod r).
Three short hashes per proof (< 300 B): host work by design.
"""
import struct

_M64 = (1 << 64) - 1
_RC = (0x1, 0x8082, 0x800000000000808A, 0x8000000080008000, 0x808B, 0x80000001, 0x8000000080008081,
       0x8000000000008009, 0x8A, 0x88, 0x80008009, 0x8000000A, 0x8000808B, 0x800000000000008B,
       0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x800A,
       0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x80000001, 0x8000000080008008)
# rho offsets indexed [x + 5*y], pi destination index for lane x + 5*y
_RHO = (0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14)
_PI = tuple((i // 5) + 5 * ((2 * (i % 5) + 3 * (i // 5)) % 5) for i in range(25))


_native = None


def _native_f1600():
    """pm_host_keccak_f1600 from the built library when it is there (40x faster than the loop below)."""
    global _native
    if _native is None:
        try:
            from . import api
            import ctypes as ct
            fn = api.load_library().pm_host_keccak_f1600
            buf = (ct.c_uint64 * 25)()
            _native = (fn, buf)
        except Exception:
            _native = False
    return _native


def keccak_f1600(lanes):
    """In-place Keccak-f[1600] on a list of 25 64-bit lanes (index x + 5*y)."""
    nat = _native_f1600()
    if nat:
        fn, buf = nat
        buf[:] = lanes
        fn(buf)
        lanes[:] = list(buf)
        return lanes
    a = lanes
    for rc in _RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _M64) for x in range(5)]
        b = [0] * 25
        for i in range(25):
            v = a[i] ^ d[i % 5]
            r = _RHO[i]
            b[_PI[i]] = ((v << r) | (v >> (64 - r))) & _M64 if r else v
        for y in range(0, 25, 5):
            row = b[y:y + 5]
            for x in range(5):
                a[y + x] = row[x] ^ ((~row[(x + 1) % 5]) & row[(x + 2) % 5] & _M64)
        a[0] ^= rc
    return a


def _sponge(data, pad, rate=136, out_len=32):
    lanes = [0] * 25
    msg = bytearray(data)
    msg.append(pad)
    msg.extend(b"\0" * (-len(msg) % rate))
    msg[-1] |= 0x80
    for off in range(0, len(msg), rate):
        for i, w in enumerate(struct.unpack("<%dQ" % (rate // 8), bytes(msg[off:off + rate]))):
            lanes[i] ^= w
        keccak_f1600(lanes)
    return struct.pack("<25Q", *lanes)[:out_len]


def keccak256(data):
    return _sponge(data, 0x01)


def sha3_256(data):
    return _sponge(data, 0x06)


# ---------------------------------------------------------------------------------- BLAKE3
_IV = (0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19)
_PERM = (2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
_M32 = 0xFFFFFFFF


def _b3_compress(cv, m, counter, blen, flags):
    s = list(cv) + list(_IV[:4]) + [counter & _M32, (counter >> 32) & _M32, blen, flags]
    m = list(m)

    def g(a, b, c, d, mx, my):
        s[a] = (s[a] + s[b] + mx) & _M32
        t = s[d] ^ s[a]
        s[d] = ((t >> 16) | (t << 16)) & _M32
        s[c] = (s[c] + s[d]) & _M32
        t = s[b] ^ s[c]
        s[b] = ((t >> 12) | (t << 20)) & _M32
        s[a] = (s[a] + s[b] + my) & _M32
        t = s[d] ^ s[a]
        s[d] = ((t >> 8) | (t << 24)) & _M32
        s[c] = (s[c] + s[d]) & _M32
        t = s[b] ^ s[c]
        s[b] = ((t >> 7) | (t << 25)) & _M32

    for _ in range(7):
        g(0, 4, 8, 12, m[0], m[1]); g(1, 5, 9, 13, m[2], m[3]); g(2, 6, 10, 14, m[4], m[5]); g(3, 7, 11, 15, m[6], m[7])
        g(0, 5, 10, 15, m[8], m[9]); g(1, 6, 11, 12, m[10], m[11]); g(2, 7, 8, 13, m[12], m[13]); g(3, 4, 9, 14, m[14], m[15])
        m = [m[p] for p in _PERM]
    return [s[i] ^ s[i + 8] for i in range(8)]


def blake3(data):
    """BLAKE3 hash mode, 32-byte digest (chunks of 1024 B, binary tree of parents)."""
    START, END, PARENT, ROOT = 1, 2, 4, 8
    words = lambda blk: struct.unpack("<16I", blk + b"\0" * (64 - len(blk)))
    chunks = [data[i:i + 1024] for i in range(0, len(data), 1024)] or [b""]

    def chunk_cv(chunk, counter, root):
        cv = _IV
        blocks = [chunk[i:i + 64] for i in range(0, len(chunk), 64)] or [b""]
        for i, blk in enumerate(blocks):
            fl = (START if i == 0 else 0) | (END if i == len(blocks) - 1 else 0)
            if root and i == len(blocks) - 1:
                fl |= ROOT
            cv = _b3_compress(cv, words(blk), counter, len(blk), fl)
        return cv

    if len(chunks) == 1:
        return struct.pack("<8I", *chunk_cv(chunks[0], 0, True))
    cvs = [chunk_cv(ch, i, False) for i, ch in enumerate(chunks)]

    def merge(nodes, root):
        if len(nodes) == 1:
            return nodes[0]
        split = 1
        while split * 2 < len(nodes):
            split *= 2
        l, r = merge(nodes[:split], False), merge(nodes[split:], False)
        return _b3_compress(_IV, l + r, 0, 64, PARENT | (ROOT if root else 0))

    return struct.pack("<8I", *merge(cvs, True))


# ------------------------------------------------------------------------------ Merlin
class _Strobe128:
    R = 166

    def __init__(self, label):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        self.st = self._f(st)
        self.pos = self.pos_begin = self.cur_flags = 0
        self.meta_ad(label, False)

    @staticmethod
    def _f(st):
        return bytearray(struct.pack("<25Q", *keccak_f1600(list(struct.unpack("<25Q", bytes(st))))))

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        self.st = self._f(self.st)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _begin(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        old = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old, flags]))
        if flags & (4 | 32) and self.pos:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin(16 | 2, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin(2, more)
        self._absorb(data)

    def prf(self, n):
        self._begin(1 | 2 | 4, False)
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)


class MerlinFieldTranscript:
    """transcript/merlin.rs:13-37 (the reference's default): 64 challenge bytes ->
    F::from_random_bytes (first 32 bytes LE masked to the modulus bit length), retry if >= r."""

    def __init__(self, name, r):
        self.r, self.s = r, _Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", name)

    def append_message(self, label, message):
        self.s.meta_ad(label, False)
        self.s.meta_ad(struct.pack("<I", len(message)), True)
        self.s.ad(message, False)

    def challenge(self, label):
        mask = (1 << self.r.bit_length()) - 1
        while True:
            self.s.meta_ad(label, False)
            self.s.meta_ad(struct.pack("<I", 64), True)
            v = int.from_bytes(self.s.prf(64)[:32], "little") & mask
            if v < self.r:
                return v


class _HashTranscript:
    H = None

    def __init__(self, name, r):   # `name` ignored, keccak256.rs:19-24 / blake3.rs:19-24
        self.r, self.t = r, b""

    def append_message(self, label, message):
        self.t += label + message

    def challenge(self, label):
        self.t = type(self).H(self.t + label)
        return int.from_bytes(self.t, "big") % self.r


class Keccak256Transcript(_HashTranscript):
    """transcript/keccak256.rs:12-43"""
    H = staticmethod(keccak256)


class Blake3Transcript(_HashTranscript):
    """transcript/blake3.rs:12-43"""
    H = staticmethod(blake3)


TRANSCRIPTS = {"merlin": MerlinFieldTranscript, "keccak256": Keccak256Transcript, "blake3": Blake3Transcript}
