"""BASELINE.json's five configs at their NAMED sizes and sample counts, HIP path (through the C ABI) against the CPU
oracle, bit for bit.

  C1 configs[0]  Cornell box + prisms 300x300x100                  whole frame vs oracle
  C2 configs[1]  book-1 1200x800x50 on one GPU                      whole frame vs oracle (48 M samples, seconds)
  C3 configs[2]  book-1 1200x800x500, pixel tiles over 8 ranks      ranks 0 and 7 of 8: the rank's WHOLE shard vs oracle at
                                                                    500 spp; all 8 shards summed == the unsharded frame
  C4 configs[3]  book-2 final 800x800x1000 on one GPU               whole frame on the GPU, a 16-row band vs oracle at 1000 spp
  C5 configs[4]  book-2 final 800x800x5000 over 8 ranks             ranks 0 and 7 of 8 at 5000 spp: two eighths of each
                                                                    shard (the rank's tiles with index % 64 == rank and
                                                                    % 64 == rank + 8) vs oracle; a rank's shard == the same
                                                                    pixels of a one-GPU frame at a prefix-exact spp

Sharding is by 16x16 pixel tiles, tile_index % nranks == rank (rtg_params; bench.py --gpus N uses exactly this), so
"rank r of 64" is a subset of "rank r of 8": the oracle checks full-spp pixels without rendering 400 M samples per case.

EVERY pixel of every config is pinned besides: tests/golden/config_hashes.json holds the SHA-256 of each 16-row band of the
oracle's full frame at the named size and sample count, of the whole frame, and of each shard of the two 8-rank tile
interleaves (tools/gen_golden.py --configs: the oracle on all cores of the build container, C5 alone 37 minutes); the tests
below hash what the GPU renders and compare (test_config_every_pixel_*).
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import assert_bit_equal
from scene_cases import build_case

pytestmark = pytest.mark.gpu


def _band(scene, cam, nx, ny, ns, band, nbands, **kw):
    """Render only the `band`-th group of 16 rows (tile_w = nx rounded up, tile_h = 16, rank = band)."""
    return scene.par_cast(cam, nx, ny, ns, tile_w=((nx + 15) // 16) * 16, tile_h=16, rank=band, nranks=nbands, **kw)


def _owned_mask(nx, ny, rank, nranks, tile=16):
    tx = (np.arange(nx) // tile)[None, :]
    ty = (np.arange(ny) // tile)[:, None]
    tiles_x = (nx + tile - 1) // tile
    return ((ty * tiles_x + tx) % nranks) == rank


def test_config_c1_cornell_300x300x100(pkg, gpu, oracle):
    """configs[0]: Cornell box + prisms 300x300x100 (list world), whole frame against the oracle."""
    nx, ny, ns = 300, 300, 100
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "cornell", nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "cornell", nx, ny)
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), so.par_cast(cam_o, nx, ny, ns), "cornell 300x300x100")


def test_config_c2_book1_1200x800x50(pkg, gpu, oracle):
    """configs[1]: book-1 1200x800x50 spp.  WHOLE frame and every counter against the oracle, run-to-run
    determinism, 8 shards summed == frame."""
    nx, ny, ns = 1200, 800, 50
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", nx, ny)
    full, st = sg.par_cast(cam_g, nx, ny, ns, stats=True)
    ref, st_o = so.par_cast(cam_o, nx, ny, ns, stats=True)
    assert_bit_equal(full, ref, "C2 whole frame")
    for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st[k] == st_o[k], (k, st[k], st_o[k])
    assert st["samples"] == nx * ny * ns and st["rays"] == st["shaded_hits"]  # sky dome: every ray hits
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns), ref, "C2 timed (non-instrumented) variant")
    acc = np.zeros_like(full)
    for r in range(8):
        acc += sg.par_cast(cam_g, nx, ny, ns, rank=r, nranks=8)
    assert_bit_equal(acc, ref, "8 shards")


@pytest.mark.parametrize("rank", [0, 7])
def test_config_c3_book1_1200x800x500_rank_of_8(pkg, gpu, oracle, rank):
    """configs[2]: the shard rank r of 8 renders of the fixed 1200x800x500 frame (what `bench.py --gpus 8` launches on
    rank r), the rank's whole shard at 500 spp against the oracle."""
    nx, ny, ns = 1200, 800, 500
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book1", nx, ny)
    part, st = sg.par_cast(cam_g, nx, ny, ns, rank=rank, nranks=8, stats=True)
    ref, st_o = so.par_cast(cam_o, nx, ny, ns, rank=rank, nranks=8, stats=True)
    assert_bit_equal(part, ref, "C3 rank %d of 8" % rank)
    for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws"):
        assert st[k] == st_o[k], (k, st[k], st_o[k])
    mask = _owned_mask(nx, ny, rank, 8)
    assert st["samples"] == int(mask.sum()) * ns
    assert (part[~mask] == 0).all(), "pixels of other ranks must stay untouched"
    assert_bit_equal(sg.par_cast(cam_g, nx, ny, ns, rank=rank, nranks=8), ref, "C3 rank %d (timed variant)" % rank)


def test_config_c3_shards_sum_to_the_one_gpu_frame(pkg, gpu):
    """configs[2]: the 8 shards at 500 spp, summed (what the RCCL reduce does: x + 0 is exact), equal the frame one GPU
    renders alone."""
    nx, ny, ns = 1200, 800, 500
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book1", nx, ny)
    full = sg.par_cast(cam_g, nx, ny, ns)
    acc = np.zeros_like(full)
    for r in range(8):
        acc += sg.par_cast(cam_g, nx, ny, ns, rank=r, nranks=8)
    assert_bit_equal(acc, full, "C3: 8 shards vs one GPU")
    assert np.isfinite(full).all() and 0.2 < full.mean() < 1.0


@pytest.mark.parametrize("name", ["book2", "book2_bvh"])
def test_config_c4_book2_800x800x1000(pkg, gpu, oracle, name):
    """configs[3]: book-2 final scene 800x800x1000 (USE_BVH false and true, main.rs:321).  The GPU renders the whole
    frame at 1000 spp; one 16-row band at 1000 spp and two more at a 16-spp prefix against the oracle."""
    nx, ny, ns = 800, 800, 1000
    sg, cam_g, _, _, _ = build_case(pkg, gpu, name, nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, name, nx, ny)
    full = sg.par_cast(cam_g, nx, ny, ns)
    band = 30 if name == "book2" else 12
    ref = _band(so, cam_o, nx, ny, ns, band, 50)
    rows = slice(band * 16, band * 16 + 16)
    assert_bit_equal(full[rows], ref[rows], "%s band %d at %d spp" % (name, band, ns))
    short = sg.par_cast(cam_g, nx, ny, 16)
    for b in (5, 44):
        ref = _band(so, cam_o, nx, ny, 16, b, 50)
        rows = slice(b * 16, b * 16 + 16)
        assert_bit_equal(short[rows], ref[rows], "%s band %d at 16 spp" % (name, b))


@pytest.mark.parametrize("rank", [0, 7])
def test_config_c5_book2_800x800x5000_rank_of_8(pkg, gpu, oracle, rank):
    """configs[4]: the shard rank r of 8 renders of the 800x800x5000 frame, bounce limit 50 (what
    `bench.py --workload book2 --gpus 8` launches on rank r).  Two eighths of the shard at the full 5000 spp against
    the oracle (tiles with index % 64 == rank and == rank + 8: subsets of the rank's tiles)."""
    nx, ny, ns = 800, 800, 5000
    sg, cam_g, _, _, _ = build_case(pkg, gpu, "book2", nx, ny)
    so, cam_o, _, _, _ = build_case(pkg, oracle, "book2", nx, ny)
    part = sg.par_cast(cam_g, nx, ny, ns, rank=rank, nranks=8, max_bounces=50)
    mask = _owned_mask(nx, ny, rank, 8)
    assert (part[~mask] == 0).all() and np.isfinite(part).all()
    for sub in (rank, rank + 8):
        ref = so.par_cast(cam_o, nx, ny, ns, rank=sub, nranks=64, max_bounces=50)
        m = _owned_mask(nx, ny, sub, 64)
        assert (m & ~mask).sum() == 0
        assert_bit_equal(part[m], ref[m], "C5 rank %d of 8, tiles %% 64 == %d" % (rank, sub))


# ---- every pixel of every config against the oracle's committed digests -------------------------------------------------------
_HASHES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_hashes.json")
BAND_ROWS = 16


def _canonical_bytes(a):
    """as tools/gen_golden.py canonical_bytes: float32 bytes, every NaN as ONE quiet NaN"""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).copy()
    u[np.isnan(u.view(np.float32))] = 0x7FC00000
    return u.tobytes()


def _sha(a):
    return hashlib.sha256(_canonical_bytes(a)).hexdigest()


def _config(key):
    with open(_HASHES) as f:
        doc = json.load(f)
    assert key in doc, "tests/golden/config_hashes.json lacks %s (tools/gen_golden.py --configs)" % key
    return doc[key]


@pytest.mark.parametrize("key", ["C1_cornell_300x300x100", "C2_book1_1200x800x50", "C3_book1_1200x800x500", "C4_book2_800x800x1000",
                                 "C4_book2_bvh_800x800x1000", "C5_book2_800x800x5000"])
def test_config_every_pixel_against_the_oracle_digests(pkg, gpu, key):
    """The WHOLE frame of each of BASELINE.json's configs at its named size: every 16-row band's SHA-256 equals the digest of the
    oracle's band (100 % of the pixels, where the direct comparisons above sample C4 / C5), and the instrumented launch's
    counters equal the oracle's."""
    ref = _config(key)
    nx, ny, ns = ref["nx"], ref["ny"], ref["ns"]
    sg, cam, _, _, _ = build_case(pkg, gpu, ref["case"], nx, ny)
    frame = sg.par_cast(cam, nx, ny, ns)
    bands = [_sha(frame[r:r + BAND_ROWS]) for r in range(0, ny, BAND_ROWS)]
    wrong = [i for i, (a, b) in enumerate(zip(bands, ref["bands"])) if a != b]
    assert len(bands) == len(ref["bands"]) and not wrong, "%s: bands %s of %d differ from the oracle's" % (key, wrong, len(bands))
    assert _sha(frame) == ref["frame"]
    if ns <= 1000:   # (the counting variant of C5 would add seconds for nothing new: C4 is the same scene)
        _, st = sg.par_cast(cam, nx, ny, ns, stats=True)
        for k, v in ref["counters"].items():
            assert st[k] == v, (key, k, st[k], v)


@pytest.mark.parametrize("key", ["C3_book1_1200x800x500", "C5_book2_800x800x5000"])
@pytest.mark.parametrize("tile", [16, 8])
def test_config_every_shard_against_the_oracle_digests(pkg, gpu, key, tile):
    """configs[2] / configs[4]: each of the 8 shards `bench.py --gpus 8` (8x8 tiles) and the 16x16 interleave render -- rank r's
    pixels in row-major order -- against the digest of the same pixels of the oracle's frame; pixels of other ranks untouched."""
    ref = _config(key)
    nx, ny, ns = ref["nx"], ref["ny"], ref["ns"]
    sg, cam, _, _, _ = build_case(pkg, gpu, ref["case"], nx, ny)
    for rank in range(8):
        part = sg.par_cast(cam, nx, ny, ns, rank=rank, nranks=8, tile_w=tile, tile_h=tile)
        tx = (np.arange(nx) // tile)[None, :]
        ty = (np.arange(ny) // tile)[:, None]
        mask = ((ty * ((nx + tile - 1) // tile) + tx) % 8) == rank
        assert _sha(part[mask]) == ref["shards_of_8_tile%d" % tile][rank], (key, tile, rank)
        assert (part[~mask] == 0).all()
