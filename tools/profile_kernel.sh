#!/bin/bash
# usage (GPU box): [PROFILE_FRAME="nx ny spp"] tools/profile_kernel.sh <tag> <workload> <kernel_substr> <samples_per_launch> [bench args...]
# -> gpurun_out/<tag>/{pmc_summary.json, kernel_stats.csv, roofline.json}: what profiles/<round>_<tag>/ holds for one kernel
tag=$1; wl=$2; kern=$3; samples=$4; shift 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
tools/pmc.sh $tag --workload $wl "$@" > /dev/null 2>&1
python tools/summarize_pmc.py $O/pmc_summary.json ${wl} "$kern" $samples gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 gpurun_out/${tag}_grbm gpurun_out/${tag}_fetch gpurun_out/${tag}_write > /dev/null
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
python tools/roofline.py $O/pmc_summary.json $O/kernel_stats.csv "$kern" > $O/roofline.json 2>&1
# the raw per-pass directories are bulky: keep only the summaries
rm -rf gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_sq3 gpurun_out/${tag}_grbm gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_stats
ls -la $O
