"""Per-thread RNG of the reference, host side (scene construction only).

Mirrors /root/reference/src/init.jl:2-12 (``TRNG``, one ``Xoroshiro128Plus(i)`` per thread) and
/root/reference/src/rand.jl:2-13,24 (``reseed!``, ``trand``, ``random_between``).  The generator
is RandomNumbers.jl 1.5.3's Xoroshiro128Plus restated from its published algorithm
(constants 55/14/36, SplitMix64 seed expansion + one discarded output, low 23 / 52 bits for
Float32 / Float64) -- unverified against Julia here, see DESIGN.md section 3.

The Python host drives one "thread", so ``TRNG`` has one entry.  On the device every
(pixel, sample chunk) has its own stream; this module is NOT used there.
"""
import numpy as np

_M64 = (1 << 64) - 1


def _rotl(v, k):
    return ((v << k) | (v >> (64 - k))) & _M64


def _splitmix64(s):
    s = (s + 0x9E3779B97F4A7C15) & _M64
    z = s
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return s, z ^ (z >> 31)


class Xoroshiro128Plus:
    """``RandomNumbers.Xorshifts.Xoroshiro128Plus(seed::Integer)``."""

    def __init__(self, seed=1):
        self.seed(seed)

    def seed(self, seed):
        s = int(seed) & _M64
        s, self.x = _splitmix64(s)
        s, self.y = _splitmix64(s)
        self.next_u64()
        return self

    def next_u64(self):
        x, y = self.x, self.y
        out = (x + y) & _M64
        s1 = x ^ y
        self.x = _rotl(x, 55) ^ s1 ^ ((s1 << 14) & _M64)
        self.y = _rotl(s1, 36)
        return out

    def rand(self, T=np.float64):
        u = self.next_u64()
        if np.dtype(T) == np.float32:
            bits = np.uint32((u & 0x007FFFFF) | 0x3F800000)
            return bits.view(np.float32) - np.float32(1)
        bits = np.uint64((u & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000)
        return bits.view(np.float64) - np.float64(1)


#: src/init.jl:2 -- one generator per (host) thread
TRNG = [Xoroshiro128Plus(1)]


def reseed():
    """``reseed!()`` (src/rand.jl:2): thread i's generator is re-seeded with i."""
    for i, r in enumerate(TRNG):
        r.seed(i + 1)


def trand(T=np.float64):
    """``trand(T)`` (src/rand.jl:10-13)."""
    return TRNG[0].rand(T)


def random_between(mn, mx):
    """``random_between(min, max) = trand(T)*(max-min) + min`` (src/rand.jl:24)."""
    T = type(mn)
    return trand(T) * (mx - mn) + mn
