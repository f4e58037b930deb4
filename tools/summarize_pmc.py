#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for the render kernel.

usage: summarize_pmc.py <out.json> <workload> <kernel_substr> <dir-with-*_counter_collection.csv>...
Sums each counter over the dispatch's rows (rocprofv3 emits one row per counter instance) and averages
over the dispatches of the matching kernel.  Applies the gfx950 corrections of MI355X_MICROARCH.md
(HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide reads by 2x -> doubled.
"""
import collections, csv, glob, json, sys

def main():
    out, workload, kernel = sys.argv[1:4]
    per_counter = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in sys.argv[4:]:
        for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r['Kernel_Name']:
                    per_counter[r['Counter_Name']][(f, r['Dispatch_Id'])] += float(r['Counter_Value'])
    summary = {c: sum(v.values()) / len(v) for c, v in per_counter.items()}
    res = {"workload": workload, "kernel": kernel, "counters_avg_per_launch": summary}
    if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
        fetch = summary['FETCH_SIZE'] * 1024 * 2   # KiB -> B, gfx950 x2 correction
        write = summary['WRITE_SIZE'] * 1024
        res.update({"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                    "note": "FETCH_SIZE*1024*2 (gfx950 half-count correction, calibrated for wide coalesced reads only) + WRITE_SIZE*1024"})
    if 'SQ_THREAD_CYCLES_VALU' in summary and 'SQ_ACTIVE_INST_VALU' in summary:
        res["valu_lane_utilization"] = summary['SQ_THREAD_CYCLES_VALU'] / (summary['SQ_ACTIVE_INST_VALU'] * 64)
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))

main()
