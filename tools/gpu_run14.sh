O=gpurun_out/r02n; mkdir -p $O
timeout 300 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 cornell_smoke 300 300 100 volume 300 300 100 book2_bvh 800 800 100 simple_light 300 300 20 2>&1 | grep -v "^\[" >> $O/t.txt
for i in 1 2; do timeout 100 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('book1', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))" >> $O/t.txt; done
timeout 100 python bench.py --no-cpu-baseline --steps 5 --spp 500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('book1 500spp', round(d['value'],1), round(d['roofline']['kernel_ms_avg'],3))" >> $O/t.txt
cat $O/t.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
