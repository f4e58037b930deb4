#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> <workload key> <kernel substring> <samples per launch> <bench args...>
# -- only the two fabric-traffic passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 --pmc runs with --kernel-trace) of a kernel
tag=$1; wl=$2; kern=$3; samples=$4; shift 4
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$c -- python $R/bench.py "$@" --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${tag}_$c.log 2>&1
done
cd $R; python tools/summarize_pmc.py gpurun_out/${tag}_fw.json $wl "$kern" $samples gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE | grep -E "bytes|ms"
