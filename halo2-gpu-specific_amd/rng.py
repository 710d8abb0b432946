"""One seeded stream standing in for the `OsRng` / `thread_rng` sites of create_proof (plonk/prover.rs:284,
permutation/prover.rs:141, vanishing/prover.rs:51-61), so that two provers fed the same seed draw the same
blinding values in the same order."""
import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
_M64 = (1 << 64) - 1
_R_INV = pow(1 << 256, -1, R_MOD)


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & _M64


class ProverRng:
    """xoshiro256** seeded through splitmix64"""

    def __init__(self, seed=0x48414C4F32):
        s, st = seed & _M64, []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & _M64
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            st.append(z ^ (z >> 31))
        self.s = st

    def next_u64(self):
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

    def random_poly_seed(self):
        """one draw keys the counter-based generator of the vanishing argument's blinding polynomial"""
        return self.next_u64()

    @staticmethod
    def random_poly_limbs(seed, n):
        """Host twin of h2_dev_random_fr (csrc/poly.hip k_random_fr): n x 4 u64 limbs, limb j of element i =
        mix64(seed + 4 i + j) with splitmix64's output function, the top limb cut to 61 bits.  The 253-bit value
        is the in-memory (Montgomery) representation."""
        with np.errstate(over="ignore"):
            z = np.uint64(seed) + np.arange(4 * n, dtype=np.uint64)
            z += np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(31)
        a = z.reshape(n, 4)
        a[:, 3] &= np.uint64((1 << 61) - 1)
        return a

    def random_poly(self, n):
        """the blinding polynomial as canonical integers (consumes one draw)"""
        a = self.random_poly_limbs(self.random_poly_seed(), n)
        return [(int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192) * _R_INV % R_MOD for r in a]
