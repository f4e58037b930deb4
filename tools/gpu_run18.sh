O=gpurun_out/r02r; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 cornell_smoke 300 300 100 volume 300 300 100 simple_light 300 300 20 simple_light_1000 300 300 20 motion 300 300 100 checker_scale 300 300 100 2>&1 | grep -v "^\[" >> $O/t.txt; cat $O/t.txt
timeout 100 python bench.py --workload cornell --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C1', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))"
