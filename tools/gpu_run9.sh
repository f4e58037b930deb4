O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
timeout 200 python tools/time_scenes.py simple_light_1000 300 300 20 simple_light 300 300 20 > $O/time_scenes.txt 2>&1; cat $O/time_scenes.txt
timeout 100 python bench.py --no-cpu-baseline --steps 10 > $O/bench.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['kernel_ms_avg'])"
