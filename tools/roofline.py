#!/usr/bin/env python3
"""Recompute bench.py's `roofline` object from committed profile files alone.

usage: tools/roofline.py profiles/<tag>/pmc_summary.json profiles/<tag>/kernel_stats.csv [kernel_substr [bench.json [anchor]]]

With a bench line (profiles/<round>/bench*.json) the `algorithmic_valu` object is recomputed too: the reference work counters of
that line (`roofline.counters_per_launch`, or those of `also.<anchor>`) priced with profiles/<round>/algorithmic_valu_costs.json (next
to the bench line, else profiles/current.json "valu_costs") over the kernel time of kernel_stats.csv.

kernel time = the AverageNs of the timed render kernel in rocprofv3's --kernel-trace --stats summary (un-profiled
by counters); instruction counts, lane utilisation, LDS duty and HBM bytes from the PMC passes of the same command.
The arithmetic is rtiow_rust_amd.roofline.valu_roofline -- the same function bench.py calls with its live HIP-event time.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    pmc_path, stats_path = sys.argv[1:3]
    graft.load_package()
    from rtiow_rust_amd import roofline as rl
    pmc = rl.load_pmc(pmc_path)
    want = sys.argv[3] if len(sys.argv) > 3 else pmc["kernel"]
    ns = None
    for r in csv.DictReader(open(stats_path)):
        if want in r["Name"]:
            ns = float(r["AverageNs"])
            break
    if ns is None:
        raise SystemExit("kernel %r not in %s" % (want, stats_path))
    roof = rl.valu_roofline(pmc, ns * 1e-9)
    roof["kernel_ms_avg"] = ns * 1e-6
    if len(sys.argv) > 4:
        line = json.load(open(sys.argv[4]))
        r = line["also"][sys.argv[5]]["roofline"] if len(sys.argv) > 5 else line["roofline"]
        local = os.path.join(os.path.dirname(os.path.abspath(sys.argv[4])), "algorithmic_valu_costs.json")
        costs = json.load(open(local)) if os.path.exists(local) else rl.load_valu_costs(ROOT)[0]
        roof["valu_lane_utilisation"] = roof.get("frac")
        roof["algorithmic_valu"] = rl.algorithmic_valu(costs, r["counters_per_launch"], pmc["samples_per_launch"], ns * 1e-9)
    print(json.dumps(roof, indent=1))


main()
