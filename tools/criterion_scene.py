#!/usr/bin/env python3
"""Criterion-equivalent of the reference's micro-benchmarks (benches/scene.rs:8-68): `scene/seq/10x10x4` and
`scene/par/10x10x4` -- Cornell box + prisms under bvh::from_scene, book-1 camera, 10x10 pixels, 4 spp.

This is the LATENCY path of the boundary (a 400-sample frame is nothing but launch + staging overhead on a GPU):
  scene/par/10x10x4 [gpu]     rtg_par_cast into a HOST framebuffer through the C ABI (what a Rust caller would see)
  scene/par/10x10x4 [gpu-dev] rtg_par_cast_device into a device framebuffer + stream sync (no copy back)
  scene/par/10x10x4 [oracle]  the CPU oracle's par_cast, all threads        (baseline, test infrastructure)
  scene/seq/10x10x4 [oracle]  the CPU oracle's cast() with the emulated SmallRng(0xDEADBEEF), one thread
Criterion style: warm-up, then N timed iterations; prints mean / median / min in microseconds as JSON lines.
"""
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def bench(name, fn, warmup=20, iters=300):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    print(json.dumps({"bench": name, "iters": iters, "mean_us": statistics.fmean(ts), "median_us": statistics.median(ts),
                      "min_us": min(ts), "p95_us": sorted(ts)[int(0.95 * len(ts))]}))


def main():
    import torch
    pkg = graft.load_package()
    gpu = pkg.load()
    ora = graft.load_oracle()
    NX, NY, NS = 10, 10, 4
    bg = gpu.builder()
    world, cam, _ = pkg.scenes.bench_scene(bg, NX, NY)
    sg = bg.scene(world)
    bo = ora.builder()
    world_o, cam_o, _ = pkg.scenes.bench_scene(bo, NX, NY)
    so = bo.scene(world_o)
    bench("scene/par/10x10x4 [gpu: rtg_par_cast, host framebuffer]", lambda: sg.par_cast(cam, NX, NY, NS))
    fb = torch.zeros((NY, NX, 3), dtype=torch.float32, device="cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = pkg.make_params(NX, NY, NS)

    def dev():
        sg.par_cast_device(cam, p, ctypes.c_void_p(fb.data_ptr()), stream)
        torch.cuda.synchronize()
    bench("scene/par/10x10x4 [gpu: rtg_par_cast_device + sync]", dev)
    bench("scene/par/10x10x4 [oracle par_cast, all threads]", lambda: so.par_cast(cam_o, NX, NY, NS), iters=100)
    bench("scene/seq/10x10x4 [oracle cast(), SmallRng(0xDEADBEEF), 1 thread]", lambda: so.cast(cam_o, NX, NY, NS), iters=100)
    for nx, ny, ns in ((100, 100, 4), (300, 300, 10)):   # the same scene as the frame grows: where the GPU path takes over
        cg = gpu.camera_look(pkg.scenes.v(13, 2, 3), pkg.scenes.v(0, 0, 0), pkg.scenes.v(0, 1, 0), 20.0, nx / ny, 0.1, 10.0)
        bench("scene/par/%dx%dx%d [gpu: rtg_par_cast]" % (nx, ny, ns), lambda: sg.par_cast(cg, nx, ny, ns), iters=50)
        bench("scene/par/%dx%dx%d [oracle par_cast]" % (nx, ny, ns), lambda: so.par_cast(cg, nx, ny, ns), warmup=2, iters=10)


if __name__ == "__main__":
    main()
