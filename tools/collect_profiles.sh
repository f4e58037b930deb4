#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <tag>   -- everything profiles/<round>_<tag>/ holds, written to gpurun_out/<tag>/
# Counters are collected AT every named config (no per-sample extrapolation of a fraction): C2 (book-1 50 spp), C3's
# frame on one GPU (500 spp), book-2 at 100 spp, C4 (1000 spp), the reference's shipped main() (book-2 300x300x100, main.rs:323-338),
# C1 (Cornell).  profiles/current.json then maps "<workload>[@<spp>[@<nx>x<ny>]]" to the pmc_summary.json of that launch.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
LEAN="render_lean_pool<true, false"; FULL="render_full_pool<1, true, false"; FULL2="render_full_pool2<true, false"  # (book-2 frames of >= 32 M samples: the pool-2 kernel)
PROFILE_FRAME="1200 800 50"  tools/profile_kernel.sh ${tag}_book1 book1 "$LEAN" 48000000 > $O/profile_book1.log 2>&1
PROFILE_FRAME="1200 800 500" tools/profile_kernel.sh ${tag}_book1_c3 book1 "$LEAN" 480000000 --spp 500 > $O/profile_book1_c3.log 2>&1
PROFILE_FRAME="800 800 100"  tools/profile_kernel.sh ${tag}_book2 book2 "$FULL2" 64000000 --spp 100 > $O/profile_book2.log 2>&1
PROFILE_FRAME="800 800 1000" tools/profile_kernel.sh ${tag}_book2_c4 book2 "$FULL2" 640000000 > $O/profile_book2_c4.log 2>&1
PROFILE_FRAME="300 300 100"  tools/profile_kernel.sh ${tag}_book2_readme book2 "$FULL" 9000000 --nx 300 --ny 300 --spp 100 > $O/profile_book2_readme.log 2>&1
PROFILE_FRAME="300 300 100"  tools/profile_kernel.sh ${tag}_cornell cornell "render_full_sync<1, false, false" 9000000 > $O/profile_cornell.log 2>&1
for k in book1 book1_c3 book2 book2_c4 book2_readme cornell; do mkdir -p $O/$k; cp gpurun_out/${tag}_$k/* $O/$k/; done
# the bench lines below take their instruction counts from THIS build's counters (bench.py reads profiles/current.json)
python3 - $tag <<'PY'
import json, sys
tag = sys.argv[1]
cur = json.load(open("profiles/current.json"))
for key, d in (("book1", "book1"), ("book1@500", "book1_c3"), ("book2", "book2"), ("book2@1000", "book2_c4"), ("book2@100@300x300", "book2_readme"), ("cornell", "cornell")):
    cur[key] = "gpurun_out/%s/%s/pmc_summary.json" % (tag, d)
json.dump(cur, open("profiles/current.json", "w"), indent=1)   # ("valu_costs": tools/algorithmic_valu.py's file, made where hipcc is -- kept)
PY
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --spp 500 --no-cpu-baseline > $O/bench_500spp.json 2>/dev/null
python bench.py --bvh sah --no-cpu-baseline > $O/bench_sah.json 2>/dev/null
python bench.py --workload cornell > $O/bench_cornell_c1.json 2>/dev/null
python bench.py --workload book2 --steps 3 > $O/bench_book2_c4.json 2>/dev/null
python bench.py --workload book2 --spp 100 --no-cpu-baseline > $O/bench_book2_100spp.json 2>/dev/null
python bench.py --workload book2 --nx 300 --ny 300 --spp 100 --steps 20 > $O/bench_book2_readme.json 2>/dev/null
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched" | sort -u > $O/schedule_book1.txt
RTG_VERBOSE=1 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched\|^\[rtg\] services" | sort -u > $O/schedule_book2.txt
python tools/time_scenes.py cornell 300 300 100 cornell_smoke 300 300 100 book2 800 800 100 book2 300 300 100 book2_bvh 800 800 100 volume 300 300 100 simple_light 300 300 20 simple_light_1000 300 300 20 book1 1200 800 50 2>&1 | grep -v "^\[rtg\]" > $O/time_scenes.txt
python tools/verify_full.py > $O/verify_full.txt 2>&1
python tools/tail_probe.py > $O/tail_probe.txt 2>&1
python tools/tail_probe.py book2 800 800 > $O/tail_probe_book2.txt 2>&1
python tools/criterion_scene.py > $O/criterion_scene.txt 2>&1
python tools/latency_probe.py > $O/latency_probe.txt 2>&1
python tools/shard_time.py > $O/shard_time.txt 2>&1
python tools/probe_book2_levers.py > $O/book2_levers.txt 2>&1
python tools/stress_pool2.py 100000 300 > $O/stress_pool2.txt 2>&1
python tools/probe_pool2.py 800 800 100 > $O/pool_vs_pool2.txt 2>&1
python tools/probe_pool2.py 800 800 1000 --only all >> $O/pool_vs_pool2.txt 2>&1
RTG_POOL2=0 python tools/tail_probe.py book2 800 800 > $O/tail_probe_book2_first_kernel.txt 2>&1
python tools/time_deep_fuzz.py > $O/deep_fuzz.txt 2>&1
python -m pytest tests -m gpu -q --timeout=300 > $O/pytest_gpu.log 2>&1
ls -la $O
# where the waves wait (tools/pmc_wait.sh): issue / wait-to-issue / s_waitcnt shares of the wave-cycles, per instruction class
tools/pmc_wait.sh gpurun_out/$tag/wait_book1 > $O/wait_attribution_book1.txt 2>&1
tools/pmc_wait.sh gpurun_out/$tag/wait_book2 --workload book2 --spp 100 > $O/wait_attribution_book2.txt 2>&1
rm -rf $O/wait_book1 $O/wait_book2
