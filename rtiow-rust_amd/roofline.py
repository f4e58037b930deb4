"""Roofline arithmetic for the render kernels -- shared by bench.py (live kernel time) and tools/roofline.py
(recomputes the same numbers from the committed profiles/<tag>/pmc_summary.json + kernel_stats.csv).

The path tracer's scene lives in LDS and its path state in LDS / L2, so neither HBM nor MFMA bounds it (SURVEY.md
8d "reality check"); the binding resource is VALU ISSUE: every SIMD of a CU can start one wave64 VALU instruction
every `issue_cycles` = 2 shader cycles (MI355X_MICROARCH.md "Wave scheduling"; tools/ubench/issue_rate.hip, timed with
s_memtime inside the kernel, reaches 2.17 cycles per instruction on the box step's own instruction mix from 2 waves
per SIMD on, 2.7 on a single repeated add / mul / fma, ~4.5 on max3 / cmp / v_pk_* / integer multiplies).  Hence

    lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)    -- how many of the 64 lanes do work
    peak      = CUs x 4 SIMDs x 64 lanes x 2.4 GHz (max clock) / issue_cycles        [lane-instructions / s]
    achieved  = SQ_INSTS_VALU per launch x 64 x lane_util / kernel seconds           [lane-instructions / s]
    frac      = achieved / peak       -- the fraction of the f32 lane peak doing work (divergence included): THE figure of merit
    valu_issue.frac = SQ_INSTS_VALU / kernel seconds / (CUs x 4 SIMDs x clock / issue_cycles) -- how busy the issue ports are
                      (a secondary: removing instructions lowers it while the kernel gets faster)
    extrapolated = the timed launch renders another number of samples than the profiled one: the counters are scaled per
                   sample, i.e. the fixed part of a launch is mis-weighted -- collect counters at the named config instead

and next to it the measured HBM side (rocprofv3 FETCH_SIZE / WRITE_SIZE, corrected as the guide prescribes) against
the 8 TB/s peak, and the LDS array duty (SQ_LDS_IDX_ACTIVE / CU-cycles).  The SURVEY 8(d) "algorithmic bytes" figure
stays as a labelled model number: those bytes are served by LDS, not by HBM.
"""
import json
import os

N_CUS = 256
SIMDS_PER_CU = 4
NOMINAL_CLOCK_HZ = 2.4e9   # MI355X_MICROARCH.md chip table
ISSUE_CYCLES = 2.0         # shader cycles per wave64 VALU instruction per SIMD (guide; issue_rate.hip confirms/corrects)
HBM_PEAK_GBS = 8000.0
XCDS = 8


def source_stamp(root):
    """Identity of the kernel BUILD the counters belong to: the git blob hash (sha1 of "blob <len>\\0" + bytes -- what
    `git hash-object` prints, computable without a .git directory) of every source the HIP library is compiled from,
    and one digest over them.  tools/summarize_pmc.py stores it in pmc_summary.json next to the demangled kernel name;
    bench.py recomputes it from the tree it runs from and refuses to quote counters of another build."""
    import hashlib
    csrc = os.path.join(root, "rtiow-rust_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith((".h", ".inc", ".hip", ".cpp")) or f == "Makefile")
    files = [os.path.join("rtiow-rust_amd", "csrc", f) for f in files] + [os.path.join("include", "rtiow_gpu.h")]
    blobs = {}
    for rel in files:
        data = open(os.path.join(root, rel), "rb").read()
        blobs[rel] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    digest = hashlib.sha1("".join("%s %s\n" % (k, v) for k, v in sorted(blobs.items())).encode()).hexdigest()
    return {"digest": digest, "blobs": blobs}


def profile_staleness(pmc, root, lib_override=None, knobs=None):
    """None when `pmc` was collected on the build in `root`, else the reason it must not be quoted."""
    st = pmc.get("build")
    if not st:
        return "the profile carries no build stamp (collected before round 3)"
    if lib_override:
        return "RTIOW_GPU_LIB points at another build of the library (%s)" % lib_override
    now = source_stamp(root)
    if st.get("digest") != now["digest"]:
        changed = sorted(k for k in set(now["blobs"]) | set(st.get("blobs", {})) if now["blobs"].get(k) != st.get("blobs", {}).get(k))
        return "kernel sources changed since the counters were collected: " + ", ".join(os.path.basename(c) for c in changed)
    if knobs:
        return "schedule options differ from the profiled run: " + ", ".join(knobs)
    return None


def load_pmc(path):
    with open(path) as f:
        return json.load(f)


def shader_clock_hz(pmc, kernel_s_profiled=None):
    """GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / kernel seconds = the clock the chip really ran at."""
    c = pmc.get("counters_avg_per_launch", {})
    if kernel_s_profiled and c.get("GRBM_GUI_ACTIVE"):
        return c["GRBM_GUI_ACTIVE"] / XCDS / kernel_s_profiled
    return None


def valu_roofline(pmc, kernel_s, samples=None, clock_hz=None, issue_cycles=None, stale=None):
    """pmc: a pmc_summary.json dict; kernel_s: seconds per launch (live HIP events or kernel_stats.csv);
    samples: samples rendered by the timed launch when it differs from the profiled one (counts scale per sample);
    stale: profile_staleness()'s verdict -- counters of another build give achieved / frac = null and the reason."""
    if stale:
        clock = clock_hz or NOMINAL_CLOCK_HZ
        return {"bound": "valu", "achieved": None, "peak": N_CUS * SIMDS_PER_CU * 64 * clock / (issue_cycles or ISSUE_CYCLES) / 1e9,
                "unit": "G lane-instructions/s", "frac": None, "traffic": None, "stale_profile": stale}
    c = pmc["counters_avg_per_launch"]
    scale = 1.0
    if samples is not None and pmc.get("samples_per_launch"):
        scale = samples / float(pmc["samples_per_launch"])
    insts = c["SQ_INSTS_VALU"] * scale
    clock = clock_hz or NOMINAL_CLOCK_HZ   # the chip's maximum clock: the profiled launch itself ran at pmc["shader_clock_hz"]
    cyc = issue_cycles or pmc.get("issue_cycles") or ISSUE_CYCLES
    issue_peak = N_CUS * SIMDS_PER_CU * clock / cyc
    issued = insts / kernel_s
    lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_ACTIVE_INST_VALU") else None
    out = {
        "bound": "valu",
        # THE fraction: lanes doing work / the f32 lane peak (issue fraction x lane utilisation)
        "achieved": issued * 64 * lane_util / 1e9 if lane_util else None, "peak": issue_peak * 64 / 1e9, "unit": "G lane-instructions/s",
        "frac": (issued / issue_peak) * lane_util if lane_util else None,
        "frac_definition": "active f32 lanes of issued VALU instructions / (256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 2 cycles per wave64 instruction)",
        "valu_issue": {"achieved": issued / 1e9, "peak": issue_peak / 1e9, "unit": "G wave-instructions/s", "frac": issued / issue_peak},
        "valu_wave_instructions_per_launch": insts,
        "lane_utilization": lane_util,
        "extrapolated": bool(abs(scale - 1.0) > 1e-9),
        "profiled_samples_per_launch": pmc.get("samples_per_launch"),
        "shader_clock_hz": clock, "shader_clock_measured_hz": pmc.get("shader_clock_hz"),
        "issue_cycles_per_wave_instruction": cyc,
    }
    if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_BUSY_CU_CYCLES"):
        # LDS-array cycles over the cycles CUs were busy (both summed over the chip)
        out["lds_array_duty"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CU_CYCLES"]
    if "hbm_bytes_per_launch" in pmc:
        traffic = pmc["hbm_bytes_per_launch"] * scale
        out["traffic"] = traffic
        out["hbm"] = {"achieved": traffic / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": traffic / kernel_s / 1e9 / HBM_PEAK_GBS,
                      "fetch_bytes": pmc.get("fetch_bytes_corrected", 0) * scale, "write_bytes": pmc.get("write_bytes", 0) * scale}
    else:
        out["traffic"] = None
    return out


def load_valu_costs(root):
    """The per-operation VALU costs tools/algorithmic_valu.py read from the production kernels' ISA (profiles/current.json
    "valu_costs"), with the same build-stamp rule as the counter profiles.  Returns (dict or None, path or None)."""
    idx = os.path.join(root, "profiles", "current.json")
    try:
        rel = json.load(open(idx)).get("valu_costs")
        if not rel:
            return None, None
        return json.load(open(os.path.join(root, rel))), rel
    except Exception:
        return None, None


def algorithmic_valu(costs, counters, samples, kernel_s, root=None, clock_hz=None, issue_cycles=None):
    """VALU lane-instructions of the REFERENCE's own work per launch / kernel seconds / the f32 lane peak:

        lanes = c_box * N + c_prim * P + c_shade * H + c_cam * samples

    N / P / H = Aabb::hit calls, primitive tests and shaded hits of the reference walk (the oracle's counters; the
    instrumented kernel's are tested equal to them at the named sizes), c_* = VALU instructions per lane and operation in the ISA
    of the lean production kernel (tools/algorithmic_valu.py: static counts, loops once -- a lower bound).  ONE set of costs prices
    every workload, whichever kernel renders it: the cheapest implementation of each reference operation the build has, so the
    figure falls when a kernel spends instructions on anything else (services, list bookkeeping, spill code, work done twice)
    and rises only when the same reference work takes less time."""
    clock = clock_hz or NOMINAL_CLOCK_HZ
    peak = N_CUS * SIMDS_PER_CU * 64 * clock / (issue_cycles or ISSUE_CYCLES)
    out = {"unit": "G lane-instructions/s", "peak": peak / 1e9, "achieved": None, "frac": None,
           "definition": "(c_box*aabb_tests + c_prim*prim_tests + c_shade*shaded_hits + c_cam*samples) / kernel seconds / "
                         "(256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 2 cycles per wave64 instruction)"}
    if not costs:
        out["stale_costs"] = "no cost file (tools/algorithmic_valu.py)"
        return out
    if root is not None:
        st = costs.get("build", {})
        now = source_stamp(root)
        if st.get("digest") != now["digest"]:
            out["stale_costs"] = "kernel sources changed since tools/algorithmic_valu.py read the ISA"
            return out
    k = costs["kernels"]["render_lean_pool"]
    c = {n: k["c_" + n] for n in ("box", "prim", "shade", "cam")}
    lanes = c["box"] * counters["aabb_tests"] + c["prim"] * counters["prim_tests"] + c["shade"] * counters["shaded_hits"] + c["cam"] * samples
    out.update({"c_box": c["box"], "c_prim": c["prim"], "c_shade": c["shade"], "c_cam": c["cam"], "costs_from": k["kernel"],
                "lane_instructions_per_launch": lanes, "achieved": lanes / kernel_s / 1e9, "frac": lanes / kernel_s / peak})
    return out


def find_profile(root, workload_key, spp=None, frame=None):
    """profiles/current.json maps a workload key ("book1", "book2", "cornell", "<workload>@<spp>" for counters collected
    at another named config: "book1@500" = C3's frame, "book2@1000" = C4 -- and "<workload>@<spp>@<nx>x<ny>" for another frame
    size: "book2@100@300x300" = the reference's shipped main(), main.rs:323-338) to the pmc_summary.json of the kernel build
    that is checked in (written by tools/collect_profiles.sh).  The entry of the timed launch's own frame and spp wins; the
    workload's base entry is the fallback, and valu_roofline then marks the object `extrapolated`."""
    idx = os.path.join(root, "profiles", "current.json")
    if not os.path.exists(idx):
        return None, None
    try:
        cur = json.load(open(idx))
        rel = None
        if spp and frame:
            rel = cur.get("%s@%d@%dx%d" % (workload_key, spp, frame[0], frame[1]))
        rel = rel or (cur.get("%s@%d" % (workload_key, spp)) if spp else None) or cur.get(workload_key)
        if not rel:
            return None, None
        path = os.path.join(root, rel)
        return load_pmc(path), rel
    except Exception:
        return None, None


def knob_differences(pmc, environ):
    """RTG_* schedule options that differ between the profiled run (pmc["env_options"]) and the current environment, in
    EITHER direction: set now but not then, set then but not now, or set to another value."""
    then = {k: v for k, v in pmc.get("env_options", {}).items() if k.startswith("RTG_") and k != "RTG_BENCH_BACKEND"}
    now = {k: v for k, v in environ.items() if k.startswith("RTG_") and k != "RTG_BENCH_BACKEND"}
    return sorted(k for k in set(then) | set(now) if then.get(k) != now.get(k))
