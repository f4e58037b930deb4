#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for the render kernel.

usage: summarize_pmc.py <out.json> <workload> <kernel_substr> <samples_per_launch> <dir-with-*_counter_collection.csv>...
Sums each counter over the dispatch's rows (rocprofv3 emits one row per counter instance) and averages
over the dispatches of the matching kernel (the instrumented COUNT variant `<..., true, ...>` of bench.py's untimed
counting pass is excluded by kernel_substr, which should name the timed instantiation).  Applies the gfx950
corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide
reads by 2x -> doubled.  The shader clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the duration of the same
dispatch in the kernel trace of the GRBM pass.
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    out, workload, kernel, samples = sys.argv[1:5]
    per_counter = collections.defaultdict(lambda: collections.defaultdict(float))
    durations = collections.defaultdict(dict)   # dir -> dispatch id -> seconds
    names = set()
    for d in sys.argv[5:]:
        for f in glob.glob(d + '/**/*_kernel_trace.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r['Kernel_Name']:
                    durations[d][r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r['Kernel_Name']:
                    names.add(r['Kernel_Name'])
                    per_counter[r['Counter_Name']][(d, r['Dispatch_Id'])] += float(r['Counter_Value'])
    summary = {c: sum(v.values()) / len(v) for c, v in per_counter.items()}
    res = {"workload": workload, "kernel": kernel, "samples_per_launch": int(samples), "counters_avg_per_launch": summary}
    # the BUILD these counters belong to (bench.py / roofline.py refuse to quote them for another one)
    graft.load_package()
    from rtiow_rust_amd import roofline as rl
    res["build"] = rl.source_stamp(ROOT)
    res["kernel_names"] = sorted(names)
    res["env_options"] = {k: v for k, v in os.environ.items() if k.startswith("RTG_") or k == "RTIOW_GPU_LIB"}
    if os.environ.get("PROFILE_FRAME"):   # "nx ny spp" of the profiled launch (tools/profile_kernel.sh): bench.py quotes the counters for that frame only
        nx, ny, spp = (int(v) for v in os.environ["PROFILE_FRAME"].split())
        res["frame"] = {"nx": nx, "ny": ny, "spp": spp}
    if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
        fetch = summary['FETCH_SIZE'] * 1024 * 2   # KiB -> B, gfx950 x2 correction
        write = summary['WRITE_SIZE'] * 1024
        res.update({"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                    "note": "FETCH_SIZE*1024*2 (gfx950 half-count correction, calibrated for wide coalesced reads only) + WRITE_SIZE*1024"})
    if 'SQ_THREAD_CYCLES_VALU' in summary and 'SQ_ACTIVE_INST_VALU' in summary:
        res["valu_lane_utilization"] = summary['SQ_THREAD_CYCLES_VALU'] / (summary['SQ_ACTIVE_INST_VALU'] * 64)
    clocks = []
    for (d, disp), v in per_counter.get('GRBM_GUI_ACTIVE', {}).items():
        if disp in durations.get(d, {}):
            clocks.append(v / 8.0 / durations[d][disp])
    if clocks:
        res["shader_clock_hz"] = sum(clocks) / len(clocks)
        res["shader_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, same pass (profiled runs clock a few % lower than un-profiled ones)"
    all_d = [t for d in durations.values() for t in d.values()]
    if all_d:
        res["kernel_ms_profiled_avg"] = 1e3 * sum(all_d) / len(all_d)
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


main()
