#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ from the CPU oracle (TEST INFRASTRUCTURE).

The reference holds no golden vectors for this path and cannot be compiled here, so
the fixtures pin the oracle's own outputs: they catch accidental changes of the restatement and any
host/compiler dependence, and the GPU tests compare the HIP path against the same files.
Run from the repo root:  python tools/gen_golden.py
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


if __name__ == "__main__":
    main()
