tag=r02w; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
tools/profile_kernel.sh ${tag}_book1 book1 "render_lean_pool<true, false" 48000000 > $O/profile_book1.log 2>&1
mkdir -p $O/book1; cp gpurun_out/${tag}_book1/* $O/book1/
python - <<PY
import json
d=json.load(open("profiles/current.json")); d["book1"]="gpurun_out/$tag/book1/pmc_summary.json"; json.dump(d, open("profiles/current.json","w"), indent=1)
PY
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --spp 500 --no-cpu-baseline > $O/bench_500spp.json 2>/dev/null
python bench.py --bvh sah --no-cpu-baseline > $O/bench_sah.json 2>/dev/null
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 2>&1 | grep "^\[rtg\] wave\|^\[rtg\] pool sched" | sort -u > $O/schedule_book1.txt
python tools/tail_probe.py > $O/tail_probe.txt 2>&1
python tools/verify_full.py > $O/verify_full.txt 2>&1
python -m pytest tests -m gpu -q --timeout=120 > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log; tail -3 $O/tail_probe.txt; cat $O/verify_full.txt | tail -4
