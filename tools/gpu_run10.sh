O=gpurun_out/r02j; mkdir -p $O
for ss in 1 0; do RTG_SLOW_SELECT=$ss timeout 200 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 cornell_smoke 300 300 100 volume 300 300 100 book2_bvh 800 800 100 simple_light 300 300 20 2>&1 | grep -v "^\[" | sed "s/^/slow_select $ss /" >> $O/t.txt; done
for sm in 8 24 32; do RTG_SPHERE_MIN=$sm timeout 100 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^book2" | sed "s/^/sphere_min $sm /" >> $O/t.txt; done
for ra in 4 8 32; do RTG_RUN_AHEAD=$ra timeout 100 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^book2" | sed "s/^/run_ahead $ra /" >> $O/t.txt; done
RTG_VERBOSE=1 timeout 100 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\] pool sched\|wave-time" | sort -u >> $O/t.txt
cat $O/t.txt
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 -k "not book1 and not lean and not c2 and not c3" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
