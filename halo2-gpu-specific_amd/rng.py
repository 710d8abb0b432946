"""The randomness of create_proof: the `OsRng` / `thread_rng` sites of the reference (plonk/prover.rs:284 advice
blinding rows, permutation/prover.rs:141 and logup/shuffle blinding scalars, vanishing/prover.rs:51-61 the random
polynomial).

`ProverRng()` draws everything from the operating system's entropy source (os.urandom), as the reference does with
`OsRng`: blinding values are unpredictable and the proof is zero-knowledge.

`ProverRng.deterministic(seed)` (also `ProverRng(seed)`) is the TEST-ONLY mode: one xoshiro256** stream expanded from
a 64-bit seed, so that two provers fed the same seed draw the same values in the same order and their proofs can be
compared byte for byte.  A proof made in this mode hides nothing from anyone who can guess
the seed: never use it outside tests and benchmarks.

In both modes the vanishing argument's random polynomial comes from a ChaCha20 keystream under a 256-bit key
(h2_dev_random_fr on the device, `random_poly_limbs` is its host twin): the key is 32 bytes of OS entropy, or four
draws of a SECOND seeded stream.  The key does not come out of the main stream on purpose: the random polynomial and its
commitment depend on nothing the transcript hashes, so the prover computes them while the witness is still crossing
PCIe, long before the reference's draw order would reach them -- a key that is independent of the position in the
main stream keeps the two provers' draws aligned whenever it is taken.
"""
import os

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_M64 = (1 << 64) - 1
_R_INV = pow(1 << 256, -1, R_MOD)
_M253 = (1 << 253) - 1


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & _M64


def _rotl32(x, k):
    return (x << np.uint32(k)) | (x >> np.uint32(32 - k))


def chacha20_blocks(key, n, first=0, stream=0):
    """ChaCha20 blocks first .. first + n - 1 under `key` (32 bytes), 64-bit block counter in words 12-13, nonce words
    (stream, 0): (n, 16) uint32, the keystream words in order (RFC 8439 block function)."""
    k = np.frombuffer(bytes(key), dtype="<u4")
    assert k.shape == (8,)
    s = np.zeros((16, n), dtype=np.uint32)
    s[0], s[1], s[2], s[3] = 0x61707865, 0x3320646E, 0x79622D32, 0x6B206574
    for j in range(8):
        s[4 + j] = k[j]
    ctr = np.arange(n, dtype=np.uint64) + np.uint64(first)
    s[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    s[13] = (ctr >> np.uint64(32)).astype(np.uint32)
    s[14] = stream
    x = s.copy()

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] ^= x[a]; x[d] = _rotl32(x[d], 16)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = _rotl32(x[b], 12)
        x[a] += x[b]; x[d] ^= x[a]; x[d] = _rotl32(x[d], 8)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = _rotl32(x[b], 7)

    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += s
    return np.ascontiguousarray(x.T)


class ProverRng:
    """seed=None (the default): OS entropy.  An integer seed: the deterministic test stream (xoshiro256** seeded
    through splitmix64)."""

    def __init__(self, seed=None, key=None):
        """`key` (32 bytes of entropy): the same unpredictable stream on every holder of the key -- how the ranks of a
        multi-GPU proof draw identical blinding values (`shared`)."""
        self.secure = seed is None
        self.key, self._buf, self._block = (bytes(key) if key is not None else None), b"", 0
        if self.secure:
            self.s = None
            return
        self.s = self._expand(seed)
        self._poly = None                      # the second stream (random-polynomial keys), made on first use
        self._seed = seed & _M64

    @staticmethod
    def _expand(seed):
        s, st = seed & _M64, []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & _M64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            st.append(z ^ (z >> 31))
        return st

    @classmethod
    def deterministic(cls, seed):
        """TEST-ONLY: reproducible blinding from a 64-bit seed (not zero-knowledge)"""
        return cls(seed)

    @classmethod
    def from_os_entropy(cls):
        return cls(None)

    def shared(self, group=None):
        """For one proof over several ranks: every rank must blind with the same values.  The OS-entropy mode becomes a
        ChaCha20 stream under a key drawn by rank 0 and broadcast; the seeded test stream is already identical."""
        if not self.secure or self.key is not None:
            return self
        import torch
        import torch.distributed as dist

        key = torch.frombuffer(bytearray(os.urandom(32)), dtype=torch.uint8).clone()
        if dist.get_backend(group) == "nccl":
            key = key.cuda()
        dist.broadcast(key, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return ProverRng(key=bytes(key.cpu().tolist()))

    def _keyed_bytes(self, count):
        while len(self._buf) < count:
            self._buf += chacha20_blocks(self.key, 4, first=self._block, stream=1).tobytes()
            self._block += 4
        out, self._buf = self._buf[:count], self._buf[count:]
        return out

    def next_u64(self):
        if self.secure:
            return int.from_bytes(self._keyed_bytes(8) if self.key is not None else os.urandom(8), "little")
        s = self.s
        out = (_rotl((s[1] * 5) & _M64, 7) * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl(s[3], 45)
        return out

    def u16(self):
        """`u16::rand(&mut OsRng)` of the advice blinding rows (prover.rs:284)"""
        return self.next_u64() & 0xFFFF

    def fr(self):
        """`Fr::random`: 512 random bits reduced modulo r, as a canonical integer"""
        v = 0
        for i in range(8):
            v |= self.next_u64() << (64 * i)
        return v % R_MOD

    def random_poly_key(self):
        """the 256-bit ChaCha20 key of the vanishing argument's blinding polynomial"""
        if self.secure:
            return self._keyed_bytes(32) if self.key is not None else os.urandom(32)
        if self._poly is None:
            self._poly = ProverRng(self._seed ^ 0x706F6C795F6B6579)      # "poly_key": a stream of its own
        return b"".join(self._poly.next_u64().to_bytes(8, "little") for _ in range(4))

    @staticmethod
    def random_poly_values(key, n):
        """Host twin of h2_dev_random_fr (csrc/poly.hip k_random_fr): element i = lo + 2^253 hi mod r with lo / hi the
        low 253 bits of keystream words 0..7 / 8..15 of ChaCha20 block i.  Canonical integers."""
        w = chacha20_blocks(key, n)
        out = []
        for row in w:
            lo = int.from_bytes(row[:8].tobytes(), "little") & _M253
            hi = int.from_bytes(row[8:].tobytes(), "little") & _M253
            out.append((lo + (hi << 253)) % R_MOD)
        return out

    @staticmethod
    def random_poly_limbs(key, n):
        """the same values as the device stores them: (n, 4) u64 Montgomery limbs"""
        vals = ProverRng.random_poly_values(key, n)
        a = np.zeros((n, 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            m = (v << 256) % R_MOD
            for j in range(4):
                a[i, j] = (m >> (64 * j)) & _M64
        return a

    def random_poly(self, n):
        """the blinding polynomial as canonical integers (consumes one key draw)"""
        return self.random_poly_values(self.random_poly_key(), n)
