"""tests/golden/config_hashes.json (tools/gen_golden.py --configs): the digests of the oracle's frames at BASELINE.json's config
sizes that tests/test_configs_gpu.py compares the GPU's frames with.  CPU side: the fixture is complete and well-formed, and the
oracle still reproduces the one config that takes it seconds (C1) -- so generator, fixture and test hash the same bytes."""
import hashlib
import json
import os
import sys

import numpy as np

from scene_cases import build_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

EXPECT = {  # key: (case, nx, ny, ns) -- BASELINE.json configs[0..4]
    "C1_cornell_300x300x100": ("cornell", 300, 300, 100),
    "C2_book1_1200x800x50": ("book1", 1200, 800, 50),
    "C3_book1_1200x800x500": ("book1", 1200, 800, 500),
    "C4_book2_800x800x1000": ("book2", 800, 800, 1000),
    "C4_book2_bvh_800x800x1000": ("book2_bvh", 800, 800, 1000),
    "C5_book2_800x800x5000": ("book2", 800, 800, 5000),
}


def _doc():
    with open(os.path.join(ROOT, "tests", "golden", "config_hashes.json")) as f:
        return json.load(f)


def test_fixture_is_complete():
    doc = _doc()
    for key, (case, nx, ny, ns) in EXPECT.items():
        e = doc[key]
        assert (e["case"], e["nx"], e["ny"], e["ns"]) == (case, nx, ny, ns)
        assert len(e["bands"]) == (ny + 15) // 16 and all(len(h) == 64 for h in e["bands"] + [e["frame"]])
        for tile in (16, 8):
            assert len(e["shards_of_8_tile%d" % tile]) == 8
        assert e["counters"]["samples"] == nx * ny * ns and e["counters"]["rays"] >= e["counters"]["samples"]
        assert 0.0 < e["mean"] < 4.0
    # one scene, one seed: the book-1 frames at 50 and 500 spp differ, the two book-2 worlds (list / Bvh) render the SAME image
    # except where exact-t ties and medium draw order differ -- their digests must at least not be copies of each other's keys
    assert doc["C2_book1_1200x800x50"]["frame"] != doc["C3_book1_1200x800x500"]["frame"]


def test_oracle_reproduces_c1(pkg, oracle):
    import gen_golden
    e = _doc()["C1_cornell_300x300x100"]
    scene, cam, _, _, _ = build_case(pkg, oracle, e["case"], e["nx"], e["ny"])
    frame, st = scene.par_cast(cam, e["nx"], e["ny"], e["ns"], stats=True)
    h = gen_golden.frame_hashes(frame)
    assert h["bands"] == e["bands"] and h["frame"] == e["frame"]
    assert h["shards_of_8_tile16"] == e["shards_of_8_tile16"] and h["shards_of_8_tile8"] == e["shards_of_8_tile8"]
    assert {k: int(st[k]) for k in e["counters"]} == e["counters"]
    assert hashlib.sha256(gen_golden.canonical_bytes(np.float32([np.nan, -np.nan]))).hexdigest() == \
        hashlib.sha256(np.uint32([0x7FC00000, 0x7FC00000]).tobytes()).hexdigest()
