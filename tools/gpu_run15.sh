O=gpurun_out/r02o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/libm_exhaustive_gpu.py > $O/libm_exhaustive_gpu.txt 2>&1; cat $O/libm_exhaustive_gpu.txt
for i in 1 2; do timeout 100 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('book1', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))" >> $O/t.txt; done
timeout 300 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 checker_scale 300 300 100 2>&1 | grep -v "^\[" >> $O/t.txt
cat $O/t.txt
