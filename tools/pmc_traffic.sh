tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$c -- python $R/bench.py "$@" --steps 2 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${tag}_$c.log 2>&1
done
cd $R; python tools/summarize_pmc.py gpurun_out/${tag}_fw.json book2 "render_full_pool<1, true, false" 64000000 gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE | grep -E "bytes|ms"
