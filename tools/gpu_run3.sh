O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in "" rtiow-rust_amd/csrc/variants/base.so; do
  for i in 1 2; do env ${v:+RTIOW_GPU_LIB=$v} python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib [$v]', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/ab.txt; done
done
python bench.py --no-cpu-baseline --steps 5 --spp 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('500spp', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/ab.txt
for rm in 16 24 32 40; do RTG_REFILL_MIN=$rm python bench.py --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('refill_min $rm', d['value'], d['roofline']['kernel_ms_avg'])" >> $O/ab.txt; done
RTG_VERBOSE=1 python tools/time_scenes.py book1 1200 800 50 2>&1 | grep "^\[rtg\]" | sort -u > $O/schedule.txt
cat $O/ab.txt $O/schedule.txt
