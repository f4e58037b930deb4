O=gpurun_out/r02b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python tools/criterion_scene.py > $O/criterion.txt 2>&1
for blk in 1024 768 512; do RTG_BLOCK=$blk python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('block $blk', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/sweep.txt; done
for rm in 12 20 28 36 44; do RTG_REFILL_MIN=$rm python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('refill_min $rm', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/sweep.txt; done
for sm in 8 16 24 32; do RTG_SPHERE_MIN=$sm python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sphere_min $sm', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/sweep.txt; done
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 > $O/schedule.txt 2>&1
python bench.py --spp 500 --no-cpu-baseline --steps 3 > $O/bench_500.json 2>/dev/null
python bench.py --workload book2 --no-cpu-baseline --steps 2 > $O/bench_book2.json 2>/dev/null
tail -3 $O/pytest.log; cat $O/criterion.txt $O/sweep.txt; grep "^\[rtg\]" $O/schedule.txt | sort -u | head
