"""Scene-construction RNG: `rand::rngs::SmallRng::seed_from_u64` of rand 0.6.5 (src/main.rs:333).

On 64-bit targets SmallRng = rand_pcg 0.1.2 `Pcg64Mcg` (Mcg128Xsl64).  The crates are third-party
and not under /root/reference; this restates their published algorithms:

    state' = state * 0x2360ED051FC65DA44385DF649FCCF645 mod 2^128
    out    = rotr64((state' >> 64) ^ state', state' >> 122)
    seed_from_u64: PCG32 (XSH-RR) expansion into 16 little-endian seed bytes; from_seed sets state |= 1
    gen::<f32>() = (next_u32 >> 8) * 2^-24 ;  next_u32 = low 32 bits of next_u64

It is only used on the host to build scenes (never on the GPU path).
"""
import numpy as np

_M128 = (1 << 128) - 1
_M64 = (1 << 64) - 1
_MUL = 0x2360ED051FC65DA44385DF649FCCF645

f32 = np.float32


class SmallRng:
    def __init__(self, seed):
        MUL, INC = 6364136223846793005, 11634580027462260723
        s = seed & _M64
        words = []
        for _ in range(4):
            s = (s * MUL + INC) & _M64
            xorshifted = (((s >> 18) ^ s) >> 27) & 0xFFFFFFFF
            rot = s >> 59
            words.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF)
        st = 0
        for w in reversed(words):
            st = (st << 32) | w
        self.state = st | 1

    def next_u64(self):
        self.state = (self.state * _MUL) & _M128
        hi, lo = self.state >> 64, self.state & _M64
        rot = hi >> 58
        x = hi ^ lo
        return ((x >> rot) | (x << ((64 - rot) & 63))) & _M64

    def next_u32(self):
        return self.next_u64() & 0xFFFFFFFF

    def gen_f32(self):
        """rng.gen::<f32>()"""
        return f32(self.next_u32() >> 8) * f32(1.0 / 16777216.0)

    def gen_vec3(self):
        """rng.gen::<Vec3>() -- vec3.rs:209-214: x, y, z in order"""
        a = self.gen_f32()
        b = self.gen_f32()
        c = self.gen_f32()
        return np.array([a, b, c], dtype=f32)

    def gen_range_f32(self, low, high):
        """rng.gen_range(low, high) for f32 (rand 0.6.5 UniformFloat::sample_single)"""
        low, high = f32(low), f32(high)
        scale = high - low
        while True:
            bits = (self.next_u32() >> 9) | 0x3F800000
            value1_2 = np.array([bits], dtype=np.uint32).view(f32)[0]
            res = (value1_2 - f32(1.0)) * scale + low
            if res < high:
                return res

    def gen_range_usize(self, low, high):
        """rng.gen_range(low, high) for usize (rand 0.6.5 UniformInt::sample_single, 64-bit)"""
        rng_range = (high - low) & _M64
        lz = 64 - rng_range.bit_length()
        zone = (rng_range << lz) & _M64
        while True:
            v = self.next_u64()
            m = v * rng_range
            hi, lo = m >> 64, m & _M64
            if lo <= zone:
                return low + hi

    def in_unit_sphere(self):
        """vec3.rs:19-26"""
        while True:
            v = f32(2.0) * self.gen_vec3() - f32(1.0)
            d = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]
            if d < f32(1.0):
                return v


def perlin_tables(seed):
    """perlin.rs:5-29 generate_vecs / generate_perm.  The reference seeds them from thread_rng()
    (non-deterministic), so any fixed tables are admissible; these come from SmallRng(seed) in the
    declaration order VECS, PERM_X, PERM_Y, PERM_Z."""
    rng = SmallRng(seed)
    vecs = np.stack([rng.in_unit_sphere() for _ in range(256)]).astype(f32)

    def perm():
        p = list(range(256))
        for i in range(255, 0, -1):
            j = rng.gen_range_usize(0, i)
            p[i], p[j] = p[j], p[i]
        return np.array(p, dtype=np.uint8)

    return vecs, perm(), perm(), perm()
