O=gpurun_out/r02g; mkdir -p $O
for c in "fuzzb:0 stats" "cornell_smoke stats"; do timeout 60 python tools/try_case.py $c >> $O/try.txt 2>&1; echo "rc=$?" >> $O/try.txt; done
tail -4 $O/try.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout=120 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/time_scenes.py cornell_smoke 300 300 100 cornell 300 300 100 book2 800 800 100 book2_bvh 800 800 100 volume 300 300 100 > $O/time_scenes.txt 2>&1
RTG_VERBOSE=1 timeout 120 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\]" | sort -u > $O/schedule_book2.txt
RTG_VERBOSE=1 timeout 120 python tools/time_scenes.py cornell 300 300 100 2>&1 | grep "^\[rtg\]" | sort -u > $O/schedule_cornell.txt
for gm in 16 24 32 48; do RTG_GATHER_MIN=$gm timeout 120 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^book2" | sed "s/^/gather_min $gm /" >> $O/sweep_book2.txt; done
for rm in 12 28; do RTG_REFILL_MIN=$rm timeout 120 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^book2" | sed "s/^/refill_min $rm /" >> $O/sweep_book2.txt; done
cat $O/time_scenes.txt $O/schedule_book2.txt $O/schedule_cornell.txt $O/sweep_book2.txt
