#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json metric).

A "step" = one full par_cast of the workload into a framebuffer resident in HBM.
  N = 1 : book-1 random-spheres, 1200x800, 50 spp (BASELINE.json configs[1]).
  N > 1 : the same frame at 50*N spp, pixel tiles sharded over the N ranks (weak scaling: per-GPU
          samples fixed), then ONE RCCL reduce(sum) of the float3 framebuffer to rank 0.
Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def usable_cores():
    """Threads this process may really run: affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 hardware threads but its container is quota-limited)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


WORKLOADS = {
    # name: (scene builder(pkg, b, nx, ny) -> (world, camera, exposure), nx, ny, spp per GPU, description)
    "book1": (lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny), 1200, 800, 50,
              "SmallRng(0xDEADBEEF) book-1 random spheres under bvh::from_scene + sky-dome emitter (SURVEY.md 8d)"),
    "book2": (lambda pkg, b, nx, ny: pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF)), 800, 800, 1000,
              "book_final_scene (src/main.rs:161-319), list world (USE_BVH = false), SmallRng(0xDEADBEEF) construction"),
    "cornell": (lambda pkg, b, nx, ny: pkg.scenes.cornell_box_scene(b, nx, ny), 300, 300, 100,
                "cornell_box_scene (src/main.rs:11-30): Cornell box + two prisms, list world"),
}


def cpu_baseline(pkg, build_scene, nx, ny, target_seconds=12.0, max_spp=1000):
    """TEST-INFRASTRUCTURE leg: time the CPU oracle (C++ restatement, row-parallel like lib.rs:326-330)
    on all host cores, on a bounded sample of the same workload (same scene/seed, reduced spp)."""
    ora = graft.load_oracle()
    b = ora.builder()
    world, cam, _ = build_scene(pkg, b, nx, ny)
    scene = b.scene(world)
    cores = usable_cores()
    scene.par_cast(cam, nx, ny, 1, threads=cores)   # warm the thread pool / page in the scene
    t0 = time.perf_counter()
    scene.par_cast(cam, nx, ny, 4, threads=cores)
    t1 = (time.perf_counter() - t0) / 4.0             # seconds per spp
    spp = int(max(1, min(max_spp, target_seconds / max(t1, 1e-4))))
    t0 = time.perf_counter()
    scene.par_cast(cam, nx, ny, spp, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": nx * ny * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d at %d spp (same scene and seed), oracle par_cast on %d threads, %.1f s"
                      % (nx, ny, spp, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="book1",
                    help="book1 = BASELINE.json's metric workload (default); book2 / cornell = configs[3] / configs[0]")
    ap.add_argument("--bvh", choices=["reference", "sah"], default="reference",
                    help="book1 only: 'reference' = Bvh::new's median split (bvh.rs:22-81, the parity mode and the "
                         "default); 'sah' = the surface-area-heuristic builder (SURVEY.md 8 f2: same image, fewer Aabb tests)")
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: the workload's spp * gpus)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xDEADBEEF)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 also renders the frame unsharded and checks the reduced frame bit-for-bit")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RTG_BENCH_BACKEND=gloo is a TEST hook: it lets the N>1 code path run with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the framebuffer is then reduced through host memory.
    backend = os.environ.get("RTG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = graft.load_package()
    gpu = pkg.load()
    build_scene, wnx, wny, wspp, wdesc = WORKLOADS[args.workload]
    if args.workload == "book1" and args.bvh == "sah":
        build_scene = lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny, use_bvh="sah")  # noqa: E731
        wdesc = wdesc.replace("bvh::from_scene", "a SAH-built Bvh (non-reference tree shape)")
    nx, ny = args.nx or wnx, args.ny or wny
    spp = args.spp or wspp * world

    b = gpu.builder()
    objs, cam, _ = build_scene(pkg, b, nx, ny)
    scene = b.scene(objs, device=dev_index)
    info = scene.info()
    info_lean = args.workload == "book1"

    fb = torch.zeros((ny, nx, 3), dtype=torch.float32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step(flags=0):
        p = pkg.make_params(nx, ny, spp, seed=args.seed, rank=rank, nranks=world, flags=flags)
        st = scene.par_cast_device(cam, p, ctypes.c_void_p(fb.data_ptr()), stream, want_stats=True)
        if world > 1:
            if backend == "nccl":
                dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)   # ONE collective: float3 framebuffer over RCCL/xGMI
            else:
                host = fb.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                fb.copy_(host)
        return st

    # counting pass (untimed): the instrumented kernel gives N/P/H for the algorithmic byte model
    fb.zero_()
    cst = step(flags=pkg.capi.FLAG_COUNTERS)
    px_rank = cst["samples"] // spp
    algo_bytes = 32 * cst["aabb_tests"] + 32 * cst["prim_tests"] + 32 * cst["shaded_hits"] + 12 * px_rank

    for _ in range(args.warmup):
        fb.zero_()
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    verified = None
    if args.verify:
        fb.zero_()
        step()
        sync()
        if rank == 0:
            whole = torch.zeros_like(fb)
            p1 = pkg.make_params(nx, ny, spp, seed=args.seed)
            scene.par_cast_device(cam, p1, ctypes.c_void_p(whole.data_ptr()), stream, want_stats=True)
            verified = bool(torch.equal(whole.view(torch.int32), fb.view(torch.int32)))

    sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        if world > 1:
            fb.zero_()
        kernel_ms.append(step()["kernel_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_samples = nx * ny * spp
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if world == 1 and os.path.exists(tpath):
            # measured separately (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
            # tools/summarize_pmc.py); counters cannot be read from inside the process
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == "%s_%dx%dx%d" % (args.workload, nx, ny, spp):
                    traffic = tj["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s (pixels*spp/s), %s %dx%d" % (
                {"book1": "book-1 random-spheres", "book2": "book-2 final scene", "cornell": "Cornell box"}[args.workload], nx, ny),
            "value": total_samples / (elapsed / args.steps) / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s_%dx%dx%dspp" % ({"book1": "book1_random_spheres", "book2": "book2_final_scene",
                                                 "cornell": "cornell_box_with_boxes"}[args.workload], nx, ny, spp),
                "scene": "%s, %d flat-program instructions, %d materials, %d B in HBM"
                         % (wdesc, info["instructions"], info["materials"], info["hbm_bytes"]),
                "max_bounces": 50, "seed": hex(args.seed),
                "sharding": "none" if world == 1 else
                            "interleaved 16x16 pixel tiles (tile %% %d == rank), spp = %d*N; RCCL reduce(sum) of the "
                            "float3 framebuffer to rank 0" % (world, wspp),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": ("rtg::render_lean_pool" if info_lean else "rtg::render_full_pool") +
                          " (+ rtg::fold_samples_kernel, <1%): HIP events around both on the launch stream",
                "kernel_ms_avg": avg_kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
                "counters_per_launch": {k: cst[k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")},
            },
        }
        if verified is not None:
            line["verified_bit_exact_vs_unsharded"] = verified
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(pkg, build_scene, nx, ny)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
