#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on MI355X (BASELINE.json metric).

A "step" = one full par_cast of the workload into a framebuffer resident in HBM.
  N = 1 : book-1 random-spheres, 1200x800, 50 spp (BASELINE.json configs[1], C2).
  N > 1 : the FIXED frame of configs[2] (C3): book-1 1200x800 at 500 spp, its 16x16 pixel tiles sharded over the N
          ranks (tile % N == rank), then ONE RCCL reduce(sum) of the float3 framebuffer to rank 0 -- strong
          scaling, as north_star states it ("1200x800x500spp reported at 1/2/4/8 MI355X").
          `--workload book2`: N = 1 renders configs[3] (C4, 800x800x1000), N > 1 the fixed frame of configs[4]
          (C5, 800x800x5000).  `--scaling weak` keeps the per-GPU samples fixed instead (spp = N x the N = 1 spp).
Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   (one rank per GPU)
              or plain `python bench.py --gpus N`, which starts those N ranks itself.  Either way WORLD_SIZE must equal
              --gpus and the host must show N GPUs, else the run exits non-zero instead of reporting another N.
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


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
    # name: (scene builder(pkg, b, nx, ny) -> (world, camera, exposure), nx, ny, spp at N = 1, spp of the fixed frame
    #        sharded at N > 1, description)
    "book1": (lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny), 1200, 800, 50, 500,
              "SmallRng(0xDEADBEEF) book-1 random spheres under bvh::from_scene + sky-dome emitter (SURVEY.md 8d)"),
    "book2": (lambda pkg, b, nx, ny: pkg.scenes.book_final_scene(b, nx, ny, pkg.small_rng.SmallRng(0xDEADBEEF)), 800, 800, 1000, 5000,
              "book_final_scene (src/main.rs:161-319), list world (USE_BVH = false), SmallRng(0xDEADBEEF) construction"),
    "cornell": (lambda pkg, b, nx, ny: pkg.scenes.cornell_box_scene(b, nx, ny), 300, 300, 100, 100,
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
            "cpu_model": cpu_model(),
            "sample": "%dx%d at %d spp (same scene and seed), oracle par_cast on %d threads, %.1f s"
                      % (nx, ny, spp, cores, dt),
            "note": "kind 'port': the Rust + Rayon reference cannot be built here or on the GPU box (no rustc / cargo, "
                    "rand / rayon not vendored, no network); this is the C++ oracle's row-parallel par_cast "
                    "(same row granularity as lib.rs:326-330) on the cores the cgroup quota allows"}


def rank_px_big(samples):
    """rtg_launch.inc: programs with a second flat program (book-2) take the pool-2 kernel from 32 M samples per launch on."""
    return samples >= (32 << 20) and os.environ.get("RTG_POOL2", "1") != "0"


def self_launch(n, backend):
    """`python bench.py --gpus N` without a launcher: run this file as N ranks under torch.distributed.run (one rank per
    GPU, rendezvous on 127.0.0.1) and pass the JSON line and the exit code through.  Refuses, non-zero, when the host has
    fewer than N GPUs (the gloo TEST hook may put several ranks on one GPU)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        print("bench.py needs a GPU: the hot path has no CPU fallback", file=sys.stderr)
        return 2
    if backend == "nccl" and have < n:
        print("bench.py: --gpus %d but this host shows %d GPU(s): one rank per GPU over RCCL" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="book1",
                    help="book1 = BASELINE.json's metric workload (default); book2 / cornell = configs[3] / configs[0]")
    ap.add_argument("--bvh4", action="store_true",
                    help="book1: traverse the 4-wide collapse of the reference Bvh (same image, other counters: not the headline)")
    ap.add_argument("--bvh", choices=["reference", "sah"], default="reference",
                    help="book1 only: 'reference' = Bvh::new's median split (bvh.rs:22-81, the parity mode and the "
                         "default); 'sah' = the surface-area-heuristic builder (SURVEY.md 8 f2: same image, fewer Aabb tests)")
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: see --scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: 'strong' (default) shards the workload's FIXED multi-GPU frame (book1: 500 spp = C3, "
                         "book2: 5000 spp = C5); 'weak' renders N x the N = 1 spp (per-GPU samples fixed)")
    ap.add_argument("--reduce-mode", choices=["reduce", "gather"], default="reduce",
                    help="N > 1: 'reduce' (default) = ONE RCCL reduce(sum) of the zero-padded float3 framebuffer to rank 0, as north_star "
                         "names it; 'gather' = ONE gather of the ranks' packed tiles (1 / N of the bytes per rank), bit-identical by construction")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xDEADBEEF)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="N = 1 default run: skip the secondary anchors (north_star's 1200x800x500 frame = C3's, and C4 "
                         "book-2 800x800x1000) that are timed in the same process after the headline loop")
    ap.add_argument("--verify", action="store_true",
                    help="rank 0 also renders the frame unsharded and checks the reduced frame bit-for-bit")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    # RTG_BENCH_BACKEND=gloo is a TEST hook: it lets the N>1 code path run with several ranks on ONE GPU
    # (RCCL refuses two ranks per device); the framebuffer is then reduced through host memory.
    backend = os.environ.get("RTG_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: launch ourselves as N ranks, one per GPU (the driver's own
        # `python -m torch.distributed.run ... bench.py --gpus N` arrives with WORLD_SIZE set and skips this).
        sys.exit(self_launch(args.gpus, backend))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # never report another N than the one asked for
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: launch with --nproc-per-node %d (or plain "
                         "`python bench.py --gpus %d`, which launches its own ranks)" % (args.gpus, world, args.gpus, args.gpus))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but this host shows %d GPU(s): one rank per GPU over RCCL"
                         % (world, torch.cuda.device_count()))
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
    build_scene, wnx, wny, wspp, wspp_multi, wdesc = WORKLOADS[args.workload]
    if args.workload == "book1" and args.bvh == "sah":
        build_scene = lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny, use_bvh="sah")  # noqa: E731
        wdesc = wdesc.replace("bvh::from_scene", "a SAH-built Bvh (non-reference tree shape)")
    nx, ny = args.nx or wnx, args.ny or wny
    if args.spp:
        spp = args.spp
    elif world == 1:
        spp = wspp
    else:
        spp = wspp_multi if args.scaling == "strong" else wspp * world

    b = gpu.builder()
    objs, cam, _ = build_scene(pkg, b, nx, ny)
    scene = b.scene(objs, device=dev_index)
    if args.bvh4:
        scene.set_option("bvh4", 1)
        wdesc = wdesc + " [traversed as 4-wide nodes: non-parity counters]"
    info = scene.info()
    info_lean = args.workload == "book1"

    from rtiow_rust_amd import parallel   # the ONE sharding implementation (also what tests/test_dist_cpu.py drives)
    frame = parallel.ShardedFrame(nx, ny, dev, via_host=(backend != "nccl"), mode=args.reduce_mode)
    fb = frame.fb
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    reduce_ms = []   # N > 1: what the ONE collective of a frame takes on this rank (events around it on the current stream)

    def step(flags=0):
        def render_shard(fb_, rank_, world_):
            p = pkg.make_params(nx, ny, spp, seed=args.seed, rank=rank_, nranks=world_, flags=flags, tile_w=frame.tile[0], tile_h=frame.tile[1])
            return scene.par_cast_device(cam, p, ctypes.c_void_p(fb_.data_ptr()), stream, want_stats=True)
        if world == 1:
            return frame.render(render_shard)
        # the same three steps as ShardedFrame.render (zero, this rank's tiles, ONE reduce(sum) to rank 0), with events round the reduce
        if frame.mode == "reduce":
            frame.fb.zero_()
        out = render_shard(frame.fb, frame.rank, frame.world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        frame.reduce(0)
        e1.record()
        reduce_ms.append((e0, e1))
        return out

    # counting pass (untimed): the instrumented kernel gives N/P/H for the algorithmic byte model
    cst = step(flags=pkg.capi.FLAG_COUNTERS)
    px_rank = cst["samples"] // spp
    algo_bytes = 32 * cst["aabb_tests"] + 32 * cst["prim_tests"] + 32 * cst["shaded_hits"] + 12 * px_rank

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    verified = None
    if args.verify:
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
        kernel_ms.append(step()["kernel_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what every rank did, so that a scaling loss can be attributed: its render kernel (HIP events inside the library), its
        # samples, and the reduce as it saw it (a rank that finishes rendering early waits for the slowest one inside the reduce)
        timed = reduce_ms[-args.steps:]
        mine = torch.tensor([sum(kernel_ms) / len(kernel_ms), float(px_rank * spp), sum(a.elapsed_time(b) for a, b in timed) / max(1, len(timed))],
                            dtype=torch.float64, device=dev)
        if backend != "nccl":
            mine = mine.cpu()
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "kernel_ms_avg": float(v[0]), "samples": int(v[1]), "reduce_ms_avg": float(v[2])} for r, v in enumerate(allr)]

    # Secondary anchors of the default N = 1 run, timed in this process AFTER the headline loop (they cannot disturb it):
    # the frame north_star's target names (C3's 1200x800x500, here on ONE GPU: what `--gpus 2/4/8` is compared against) and
    # C4.  Same step definition (one par_cast into an HBM-resident framebuffer, barrier-free at N = 1, synchronised both sides).
    also = None
    default_run = (world == 1 and args.workload == "book1" and not args.bvh4 and args.bvh == "reference" and not args.spp
                   and (nx, ny) == (wnx, wny))
    if default_run and not args.no_also:
        from rtiow_rust_amd import roofline as rl_a
        valu_costs, _ = rl_a.load_valu_costs(ROOT)

        def anchor(wl, a_spp, a_steps, profile_key=None, bvh="reference", ref_counters=None):
            a_build, anx, any_, _, _, _ = WORKLOADS[wl]
            if bvh == "sah":
                a_build = lambda pkg, b, nx, ny: pkg.scenes.random_scene(b, nx, ny, use_bvh="sah")  # noqa: E731
            ab = gpu.builder()
            a_objs, a_cam, _ = a_build(pkg, ab, anx, any_)
            a_scene = ab.scene(a_objs, device=dev_index)
            a_fb = torch.zeros((any_, anx, 3), dtype=torch.float32, device=dev)
            a_p = pkg.make_params(anx, any_, a_spp, seed=args.seed)
            one = lambda: a_scene.par_cast_device(a_cam, a_p, ctypes.c_void_p(a_fb.data_ptr()), stream, want_stats=True)  # noqa: E731
            # counting pass (untimed, instrumented variant): N / P / H of this frame for roofline.algorithmic_valu
            a_pc = pkg.make_params(anx, any_, a_spp, seed=args.seed, flags=pkg.capi.FLAG_COUNTERS)
            a_cst = a_scene.par_cast_device(a_cam, a_pc, ctypes.c_void_p(a_fb.data_ptr()), stream, want_stats=True)
            one()
            torch.cuda.synchronize()
            a_t0 = time.perf_counter()
            k_ms = [one()["kernel_ms"] for _ in range(a_steps)]
            torch.cuda.synchronize()
            a_dt = (time.perf_counter() - a_t0) / a_steps
            res = {"value": anx * any_ * a_spp / a_dt / 1e6, "unit": "Msamples/s", "ms_per_step": a_dt * 1e3,
                   "kernel_ms_avg": sum(k_ms) / len(k_ms), "steps": a_steps, "warmup": 1}
            # (a non-reference tree does less box work for the same image: its algorithmic figure keeps the REFERENCE walk's N / P / H)
            a_alg = rl_a.algorithmic_valu(valu_costs, ref_counters or a_cst, anx * any_ * a_spp, sum(k_ms) / len(k_ms) * 1e-3, root=ROOT)
            if ref_counters:
                a_alg["counters"] = "the reference tree's walk of the same frame (the headline's counters), not what this tree's walk did"
            if profile_key:
                # the same object as the headline's, from the counters collected AT this config (profiles/current.json
                # "<workload>@<spp>"), under the same staleness rule: another build -> frac null + the reason
                a_pmc, a_path = rl_a.find_profile(ROOT, profile_key, a_spp, frame=(anx, any_))
                if a_pmc is not None:
                    a_stale = rl_a.profile_staleness(a_pmc, ROOT, lib_override=os.environ.get("RTIOW_GPU_LIB"),
                                                     knobs=rl_a.knob_differences(a_pmc, os.environ))
                    r = rl_a.valu_roofline(a_pmc, sum(k_ms) / len(k_ms) * 1e-3, samples=anx * any_ * a_spp, stale=a_stale)
                    res["roofline"] = {"bound": r["bound"], "frac": r["frac"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                                       "valu_issue": {"frac": (r.get("valu_issue") or {}).get("frac")}, "lane_utilization": r.get("lane_utilization"),
                                       "traffic": r.get("traffic"), "extrapolated": r.get("extrapolated"), "pmc_source": a_path}
                    if r.get("stale_profile"):
                        res["roofline"]["stale_profile"] = r["stale_profile"]
                else:
                    res["roofline"] = {"bound": "valu", "frac": None, "traffic": None, "pmc_source": None}
            else:
                res["roofline"] = {"bound": "valu", "frac": None, "traffic": None, "pmc_source": None}
            # today's `frac` counts every issued instruction: also reported under its plain name; the fraction of USEFUL work beside it
            res["roofline"]["valu_lane_utilisation"] = res["roofline"].get("frac")
            res["roofline"]["algorithmic_valu"] = a_alg
            res["roofline"]["counters_per_launch"] = {k: a_cst[k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")}
            return res
        also = {"book1_random_spheres_1200x800x500spp": dict(anchor("book1", 500, 5, "book1"), baseline_config="configs[2] on ONE GPU (north_star target frame)"),
                "book2_final_scene_800x800x1000spp": dict(anchor("book2", 1000, 3, "book2"), baseline_config="configs[3]"),
                # NOT the reference's tree: the surface-area-heuristic builder renders the identical image
                # (tests/test_parity_gpu.py::test_sah_tree_renders_the_reference_tree_frame_at_c2) with fewer Aabb::hit calls
                "book1_random_spheres_1200x800x50spp_sah_tree": dict(anchor("book1", 50, 5, None, bvh="sah", ref_counters=cst), baseline_config=None,
                                                                     note="non-reference Bvh shape (SAH builder), identical image; not the headline")}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_samples = nx * ny * spp
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        # ---- roofline of the dominant kernel ------------------------------------------------------------------
        # Binding resource: VALU issue (the scene is LDS-resident, path state LDS / L2-resident: neither HBM nor MFMA
        # can bound this kernel).  achieved = SQ_INSTS_VALU per launch (rocprofv3 PMC pass of THIS command, committed
        # under profiles/, scaled per sample when the spp differs) / the kernel time measured LIVE here with HIP
        # events on the launch stream; peak = 256 CUs x 4 SIMDs x shader clock / issue cycles per wave64 instruction.
        # tools/roofline.py recomputes the same object from profiles/<tag>/pmc_summary.json + kernel_stats.csv.
        from rtiow_rust_amd import roofline as rl
        # (book-2: frames of >= 32 M samples run on the pool-2 kernel, smaller ones on the first full-feature kernel: rtg_launch.inc)
        kernel_name = {"book1": "rtg::render_lean_pool", "book2": "rtg::render_full_pool2" if rank_px_big(px_rank * spp) else "rtg::render_full_pool",
                       "cornell": "rtg::render_full_sync"}[args.workload]
        kernel_s = avg_kernel_ms * 1e-3
        rank_samples = px_rank * spp
        pmc, pmc_path = rl.find_profile(ROOT, (args.workload if args.bvh == "reference" else args.workload + "_" + args.bvh) + ("_bvh4" if args.bvh4 else ""), spp, frame=(nx, ny))
        if pmc is not None:
            # counters are only quoted for the build and schedule they were collected on (profiles carry the git blob hashes
            # of csrc/*): another build, or an RTG_* option set on one side only -> achieved / frac = null + the reason
            knobs = rl.knob_differences(pmc, os.environ)
            stale = rl.profile_staleness(pmc, ROOT, lib_override=os.environ.get("RTIOW_GPU_LIB"), knobs=knobs)
            pf = pmc.get("frame") or {"nx": wnx, "ny": wny}
            if stale is None and (nx, ny) != (pf["nx"], pf["ny"]):
                stale = "frame geometry %dx%d differs from the profiled %dx%d" % (nx, ny, pf["nx"], pf["ny"])
            roof = rl.valu_roofline(pmc, kernel_s, samples=rank_samples, stale=stale)
            roof["pmc_source"] = pmc_path
            roof["pmc_build"] = pmc.get("build", {}).get("digest")
        else:   # no counter profile for this workload: the contract's keys with the unmeasured ones null
            roof = {"bound": "valu", "achieved": None, "peak": rl.N_CUS * rl.SIMDS_PER_CU * 64 * rl.NOMINAL_CLOCK_HZ / rl.ISSUE_CYCLES / 1e9,
                    "unit": "G lane-instructions/s", "frac": None, "traffic": None}
        # `frac` credits every issued instruction (services, list bookkeeping, spill code): kept, and named for what it is;
        # `algorithmic_valu` prices the reference's own work only (roofline.py algorithmic_valu, tools/algorithmic_valu.py)
        roof["valu_lane_utilisation"] = roof.get("frac")
        roof["algorithmic_valu"] = rl.algorithmic_valu(rl.load_valu_costs(ROOT)[0], cst, rank_samples, kernel_s, root=ROOT)
        roof.update({
            "kernel": kernel_name + " (+ rtg::fold_samples_kernel, ~1%): HIP events around both on the launch stream",
            "kernel_ms_avg": avg_kernel_ms,
            # SURVEY.md 8(d)'s model figure, kept as a LABELLED secondary: these bytes are served by the LDS image of
            # the scene, not by HBM, so their rate may exceed the HBM peak and is no fraction of anything
            "algorithmic": {"bytes_per_launch": algo_bytes, "rate_GBps": algo_bytes / kernel_s / 1e9, "lds_served": True,
                            "formula": "32*aabb_tests + 32*prim_tests + 32*shaded_hits + 12*pixels (SURVEY.md 8d)"},
            "counters_per_launch": {k: cst[k] for k in ("aabb_tests", "prim_tests", "shaded_hits", "rays", "draws")},
        })
        if world == 1:
            shard_txt = "none"
        elif args.scaling == "strong" or args.spp:
            shard_txt = ("the FIXED %dx%dx%d frame, interleaved %dx%d pixel tiles (tile %% %d == rank): strong scaling; ONE RCCL "
                         "%s to rank 0 per frame" % (nx, ny, spp, frame.tile[0], frame.tile[1], world,
                                                     "reduce(sum) of the float3 framebuffer" if frame.mode == "reduce" else "gather of the ranks' packed tiles"))
        else:
            shard_txt = ("interleaved %dx%d pixel tiles (tile %% %d == rank), spp = %d*N: weak scaling; ONE RCCL reduce(sum) "
                         "of the float3 framebuffer to rank 0 per frame" % (frame.tile[0], frame.tile[1], world, wspp))
        line = {
            "metric": "Msamples/s (pixels*spp/s), %s %dx%d" % (
                {"book1": "book-1 random-spheres", "book2": "book-2 final scene", "cornell": "Cornell box"}[args.workload], nx, ny),
            "value": total_samples / (elapsed / args.steps) / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak" if (world > 1 and args.scaling == "weak" and not args.spp) else "strong",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "%s_%dx%dx%dspp" % ({"book1": "book1_random_spheres", "book2": "book2_final_scene",
                                                 "cornell": "cornell_box_with_boxes"}[args.workload], nx, ny, spp),
                "baseline_config": {("book1", 50): "configs[1]", ("book1", 500): "configs[2]", ("book2", 1000): "configs[3]",
                                    ("book2", 5000): "configs[4]", ("cornell", 100): "configs[0]"}.get((args.workload, spp)),
                "scene": "%s, %d flat-program instructions, %d materials, %d B in HBM"
                         % (wdesc, info["instructions"], info["materials"], info["hbm_bytes"]),
                "max_bounces": 50, "seed": hex(args.seed),
                "sharding": shard_txt,
            },
            "roofline": roof,
        }
        if per_rank is not None:
            line["per_rank"] = per_rank
            # rank 0 receives the frame: its reduce time with the slowest rank's kernel taken out is the transfer itself
            line["reduce_ms_avg"] = per_rank[0]["reduce_ms_avg"]
            line["reduce_mode"] = frame.mode                      # "reduce": full frames summed; "gather": packed tiles collected
            line["reduce_bytes_per_rank"] = frame.bytes_per_rank()
            line["slowest_rank_kernel_ms"] = max(r["kernel_ms_avg"] for r in per_rank)
        if also is not None:
            line["also"] = also
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
