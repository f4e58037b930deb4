"""RNG layer: third-party arithmetic restated from published algorithms (rand 0.6.5, rand_pcg 0.1.2,
Random123 Philox) -- pinned here by their published known-answer vectors."""
import ctypes as C

import numpy as np


def philox(oracle, key, ctr):
    out = (C.c_uint32 * 4)()
    oracle.lib.rto_debug_philox(C.c_uint32(key[0]), C.c_uint32(key[1]), *[C.c_uint32(c) for c in ctr], out)
    return [int(x) for x in out]


def test_philox4x32_10_random123_kat(oracle):
    # Random123 kat_vectors: philox4x32 10 rounds
    assert philox(oracle, (0, 0), (0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox(oracle, (0xffffffff,) * 2, (0xffffffff,) * 4) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox(oracle, (0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sample_stream_layout(oracle):
    """Stream (seed, pixel, sample, event): block b = philox(key=seed lo/hi, ctr=(b, sample, pixel, event))."""
    seed, pixel, sample = 0x0123456789ABCDEF, 4711, 13
    for event in (0, 1, 7):
        out = (C.c_uint32 * 10)()
        oracle.lib.rto_debug_sample_rng_u32(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(event),
                                            C.c_size_t(10), out)
        expect = []
        for blk in range(3):
            expect += philox(oracle, (seed & 0xffffffff, seed >> 32), (blk, sample, pixel, event))
        assert [int(x) for x in out] == expect[:10]


def test_mcg128xsl64_kat(oracle):
    # rand_pcg test vector: Mcg128Xsl64::new(42)
    out = (C.c_uint64 * 6)()
    oracle.lib.rto_debug_mcg128_u64(C.c_uint64(0), C.c_uint64(42), C.c_size_t(6), out)
    assert [int(x) for x in out] == [0x63b4a3a813ce700a, 0x382954200617ab24, 0xa7fd85ae3fe950ce,
                                     0xd715286aa2887737, 0x60c92fee2e59f32c, 0x84c4e96beff30017]


def test_small_rng_seed_from_u64(oracle, pkg):
    """SmallRng::seed_from_u64(0xDEADBEEF) (main.rs:333, benches/scene.rs:32).  KAT recorded in SURVEY.md 8c,
    UNVERIFIED against rustc (no Rust toolchain here): first u64 f0ba7b9d01eb7ece, first f32 0.0074995756."""
    out = (C.c_uint64 * 4)()
    oracle.lib.rto_debug_small_rng_u64(C.c_uint64(0xDEADBEEF), C.c_size_t(4), out)
    assert int(out[0]) == 0xf0ba7b9d01eb7ece
    py = pkg.small_rng.SmallRng(0xDEADBEEF)
    assert [py.next_u64() for _ in range(4)] == [int(x) for x in out]
    f = (C.c_float * 8)()
    oracle.lib.rto_debug_small_rng_f32(C.c_uint64(0xDEADBEEF), C.c_size_t(8), f)
    py = pkg.small_rng.SmallRng(0xDEADBEEF)
    assert [float(py.gen_f32()) for _ in range(8)] == [float(x) for x in f]
    assert float(f[0]) == 0.007499575614929199


def test_python_rng_float_conversions(pkg):
    r = pkg.small_rng.SmallRng(1)
    for _ in range(200):
        x = r.gen_f32()
        assert x.dtype == np.float32 and 0.0 <= x < 1.0
        y = r.gen_range_f32(0.0, 0.5)
        assert y.dtype == np.float32 and 0.0 <= y < 0.5
        k = r.gen_range_usize(0, 7)
        assert 0 <= k < 7


def test_perlin_tables_are_deterministic_permutations(pkg):
    v, px, py, pz = pkg.small_rng.perlin_tables(0xDEADBEEF)
    v2, px2, _, _ = pkg.small_rng.perlin_tables(0xDEADBEEF)
    assert np.array_equal(v, v2) and np.array_equal(px, px2)
    for p in (px, py, pz):
        assert sorted(p.tolist()) == list(range(256))
    assert ((v * v).sum(axis=1) < 1.0).all()
