#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <tag>   -- everything profiles/<round>_<tag>/ holds, written to gpurun_out/<tag>/
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
# counters + kernel stats of the two production kernels (separate --pmc passes, tools/pmc.sh)
tools/profile_kernel.sh ${tag}_book1 book1 "render_lean_pool<true, false" 48000000 > $O/profile_book1.log 2>&1
tools/profile_kernel.sh ${tag}_book2 book2 "render_full_pool<1, true, false" 64000000 --spp 100 > $O/profile_book2.log 2>&1
tools/profile_kernel.sh ${tag}_cornell cornell "render_full_sync<1, false, false" 9000000 > $O/profile_cornell.log 2>&1
for k in book1 book2 cornell; do mkdir -p $O/$k; cp gpurun_out/${tag}_$k/* $O/$k/; done
# the bench lines below take their instruction counts from THIS build's counters (bench.py reads profiles/current.json)
printf '{\n "book1": "gpurun_out/%s/book1/pmc_summary.json",\n "book2": "gpurun_out/%s/book2/pmc_summary.json",\n "cornell": "gpurun_out/%s/cornell/pmc_summary.json"\n}\n' $tag $tag $tag > profiles/current.json
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --spp 500 --no-cpu-baseline > $O/bench_500spp.json 2>/dev/null
python bench.py --bvh sah --no-cpu-baseline > $O/bench_sah.json 2>/dev/null
python bench.py --bvh4 --no-cpu-baseline > $O/bench_bvh4.json 2>/dev/null
python bench.py --workload cornell > $O/bench_cornell_c1.json 2>/dev/null
python bench.py --workload book2 --steps 3 > $O/bench_book2_c4.json 2>/dev/null
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched" | sort -u > $O/schedule_book1.txt
RTG_VERBOSE=1 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched\|^\[rtg\] services" | sort -u > $O/schedule_book2.txt
python tools/time_scenes.py cornell 300 300 100 cornell_smoke 300 300 100 book2 800 800 100 book2_bvh 800 800 100 volume 300 300 100 simple_light 300 300 20 simple_light_1000 300 300 20 book1 1200 800 50 2>&1 | grep -v "^\[rtg\]" > $O/time_scenes.txt
python tools/verify_full.py > $O/verify_full.txt 2>&1
python tools/tail_probe.py > $O/tail_probe.txt 2>&1
python tools/criterion_scene.py > $O/criterion_scene.txt 2>&1
tools/ubench/issue_rate > $O/issue_rate.txt 2>&1
tools/ubench/mem_latency > $O/mem_latency.txt 2>&1
python tools/libm_exhaustive_gpu.py > $O/libm_exhaustive_gpu.txt 2>&1
python -m pytest tests -m gpu -q --timeout=300 > $O/pytest_gpu.log 2>&1
ls -la $O
