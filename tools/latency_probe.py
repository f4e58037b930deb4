#!/usr/bin/env python3
"""Per-call floor of the boundary on small frames (GPU box): wall time of rtg_par_cast_device + stream sync against the GPU-side
time between the two events around the launches (stats->kernel_ms), for the Criterion scene (benches/scene.rs:8-68: Cornell box +
prisms under bvh::from_scene) and, with `lean`, the book-1 scene.  usage: latency_probe.py [lean] [nx ny ns]...
Run it under `rocprofv3 --kernel-trace --stats` for the duration of every kernel of a call."""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    import torch
    pkg = graft.load_package()
    gpu = pkg.load()
    args = sys.argv[1:]
    lean = bool(args) and args[0] == "lean"
    if lean:
        args = args[1:]
    frames = [(10, 10, 4), (100, 100, 4), (300, 300, 10)]
    if args:
        frames = [(int(args[i]), int(args[i + 1]), int(args[i + 2])) for i in range(0, len(args), 3)]
    b = gpu.builder()
    world, _, _ = (pkg.scenes.random_scene if lean else pkg.scenes.bench_scene)(b, 10, 10)
    sc = b.scene(world)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for nx, ny, ns in frames:
        cam = gpu.camera_look(pkg.scenes.v(13, 2, 3), pkg.scenes.v(0, 0, 0), pkg.scenes.v(0, 1, 0), 20.0, nx / ny, 0.1, 10.0)
        fb = torch.zeros((ny, nx, 3), dtype=torch.float32, device="cuda:0")
        p = pkg.make_params(nx, ny, ns)
        ptr = ctypes.c_void_p(fb.data_ptr())
        for _ in range(30):
            sc.par_cast_device(cam, p, ptr, stream)
        torch.cuda.synchronize()
        wall, enq, kms = [], [], []
        for _ in range(200):
            t0 = time.perf_counter()
            sc.par_cast_device(cam, p, ptr, stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            wall.append((t2 - t0) * 1e6), enq.append((t1 - t0) * 1e6)
        for _ in range(50):
            kms.append(sc.par_cast_device(cam, p, ptr, stream, want_stats=True)["kernel_ms"] * 1e3)
        print("%s %dx%dx%d: call + sync median %.0f us (min %.0f), enqueue alone %.0f us, GPU-side (events) median %.0f us (min %.0f)" % (
            "book1" if lean else "criterion", nx, ny, ns, statistics.median(wall), min(wall), statistics.median(enq),
            statistics.median(kms), min(kms)))


if __name__ == "__main__":
    main()
