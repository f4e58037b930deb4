#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the CPU oracle (TEST INFRASTRUCTURE).

The reference holds no golden vectors for this path and cannot be compiled here, so
the fixtures pin the oracle's own outputs: they catch accidental changes of the restatement and any
host/compiler dependence, and the GPU tests compare the HIP path against the same files.
Run from the repo root:  python tools/gen_golden.py
                         python tools/gen_golden.py --configs [C1 C2 ...]   (tests/golden/config_hashes.json, below)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from scene_cases import CASES, build_case  # noqa: E402
from probe_rays import probe_rays  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def sample_keys(nx, ny, ns, n=64, seed=11):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, nx, n).astype(np.uint32), rs.randint(0, ny, n).astype(np.uint32),
            rs.randint(0, ns, n).astype(np.uint32))


def main():
    pkg = graft.load_package()
    ora = graft.load_oracle()
    os.makedirs(GOLD, exist_ok=True)
    fbs, samples, hits = {}, {}, {}
    for name in sorted(CASES):
        scene, cam, nx, ny, ns = build_case(pkg, ora, name)
        fbs[name] = scene.par_cast(cam, nx, ny, ns)
        xs, ys, ss = sample_keys(nx, ny, ns)
        rgb, info = scene.debug_samples(cam, nx, ny, ns, xs, ys, ss)
        samples[name + ".keys"] = np.stack([xs, ys, ss])
        samples[name + ".rgb"] = rgb
        samples[name + ".info"] = info
        if name in ("cornell", "book1", "book2", "volume_bvh", "checker_scale", "motion"):
            rays = probe_rays(name)
            out, mat = scene.debug_hit_top(rays, seed=5)
            hits[name + ".rays"] = rays
            hits[name + ".out"] = out
            hits[name + ".mat"] = mat
    np.savez_compressed(os.path.join(GOLD, "framebuffers.npz"), **fbs)
    np.savez_compressed(os.path.join(GOLD, "samples.npz"), **samples)
    np.savez_compressed(os.path.join(GOLD, "hit_top.npz"), **hits)
    # cast(): the deterministic sequential path of benches/scene.rs:32-36 (SmallRng 0xDEADBEEF, 10x10x4)
    scene, cam, nx, ny, ns = build_case(pkg, ora, "bench")
    np.savez_compressed(os.path.join(GOLD, "cast_bench_10x10x4.npz"), image=scene.cast(cam, 10, 10, 4, 0xDEADBEEF))
    print("wrote fixtures:", {f: os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD)})


# ---- BASELINE.json's five configs at their NAMED sizes: SHA-256 of every 16-row band of the oracle's full frame ----------------
# (C4 / C5 met the oracle on 6 % of their pixels.)  The frames themselves are 4-12 MB each and take the oracle
# minutes on all cores, so the fixture holds hashes: data, no reference text, nothing read from /root/reference at test time.
CONFIGS = {
    # key: (scene case, nx, ny, ns, BASELINE.json configs[] index)
    "C1_cornell_300x300x100": ("cornell", 300, 300, 100, 0),
    "C2_book1_1200x800x50": ("book1", 1200, 800, 50, 1),
    "C3_book1_1200x800x500": ("book1", 1200, 800, 500, 2),
    "C4_book2_800x800x1000": ("book2", 800, 800, 1000, 3),
    "C4_book2_bvh_800x800x1000": ("book2_bvh", 800, 800, 1000, 3),
    "C5_book2_800x800x5000": ("book2", 800, 800, 5000, 4),
}
BAND_ROWS = 16


def canonical_bytes(a):
    """float32 bytes with every NaN replaced by ONE quiet NaN (x86 and gfx950 differ in NaN sign / payload;
    conftest.assert_bit_equal treats NaN == NaN the same way)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).copy()
    u[np.isnan(u.view(np.float32))] = 0x7FC00000
    return u.tobytes()


def owned_mask(nx, ny, rank, nranks, tile):
    tx = (np.arange(nx) // tile)[None, :]
    ty = (np.arange(ny) // tile)[:, None]
    return ((ty * ((nx + tile - 1) // tile) + tx) % nranks) == rank


def frame_hashes(frame):
    """What tests/test_configs_gpu.py recomputes from the GPU's frame: one digest per 16-row band (row 0 = top), one per
    shard of the 8-rank interleaves bench.py / rtg_par_cast_multi use (16x16 and 8x8 tiles: the shard's pixels in row-major
    order), and one over the whole frame."""
    import hashlib
    ny, nx, _ = frame.shape
    h = {"bands": [hashlib.sha256(canonical_bytes(frame[r:r + BAND_ROWS])).hexdigest() for r in range(0, ny, BAND_ROWS)],
         "frame": hashlib.sha256(canonical_bytes(frame)).hexdigest(), "mean": float(frame.astype(np.float64).mean())}
    for tile in (16, 8):
        h["shards_of_8_tile%d" % tile] = [
            hashlib.sha256(canonical_bytes(frame[owned_mask(nx, ny, r, 8, tile)])).hexdigest() for r in range(8)]
    return h


def configs(which):
    import time
    pkg = graft.load_package()
    ora = graft.load_oracle()
    path = os.path.join(GOLD, "config_hashes.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["_about"] = ("SHA-256 of the CPU oracle's frames at BASELINE.json's config sizes (tools/gen_golden.py --configs): per "
                     "16-row band, per shard of the 8-rank tile interleaves, whole frame.  float32 little-endian bytes, row 0 = "
                     "top, NaNs canonicalised to 0x7fc00000.  seed 0xDEADBEEF, bounce cap 50.")
    for key, (case, nx, ny, ns, idx) in CONFIGS.items():
        if which and not any(key.startswith(w) for w in which):
            continue
        scene, cam, _, _, _ = build_case(pkg, ora, case, nx, ny)
        t = time.time()
        frame, st = scene.par_cast(cam, nx, ny, ns, stats=True)
        entry = frame_hashes(frame)
        entry.update(case=case, nx=nx, ny=ny, ns=ns, baseline_config=idx, oracle_seconds=round(time.time() - t, 1),
                     counters={k: int(st[k]) for k in ("samples", "aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")})
        doc[key] = entry
        with open(path, "w") as f:     # after every config: C5 alone is ~15 minutes of all cores
            json.dump(doc, f, indent=1, sort_keys=True)
        print(key, "%.0f s" % (time.time() - t), entry["frame"][:16], entry["counters"], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--configs":
        configs(sys.argv[2:])
    else:
        main()
