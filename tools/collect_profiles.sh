#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <tag>   -- everything profiles/<round>_<tag>/ holds, written to gpurun_out/<tag>/
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
tools/pmc.sh $tag > /dev/null 2>&1
python tools/summarize_pmc.py $O/pmc_summary.json book1_1200x800x50 render_lean_pool gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 gpurun_out/${tag}_grbm gpurun_out/${tag}_fetch gpurun_out/${tag}_write > /dev/null
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --spp 500 --no-cpu-baseline > $O/bench_500spp.json 2>/dev/null
python bench.py --bvh sah --no-cpu-baseline > $O/bench_sah.json 2>/dev/null
python bench.py --workload cornell > $O/bench_cornell_c1.json 2>/dev/null
python bench.py --workload book2 > $O/bench_book2_c4.json 2>/dev/null
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched" | sort -u > $O/schedule.txt
python tools/time_scenes.py 2>&1 | grep -v "^\[rtg\]" > $O/time_scenes.txt
python tools/verify_full.py > $O/verify_full.txt 2>&1
python tools/tail_probe.py > $O/tail_probe.txt 2>&1
# book-2 kernel stats + counters (full-feature kernel)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_b2stats -- python $R/bench.py --workload book2 --spp 100 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find $R/gpurun_out/${tag}_b2stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/book2_kernel_stats.csv
ls -la $O
