O=gpurun_out/r02p; mkdir -p $O
for c in "cornell stats" "book2 stats" "volume_bvh stats" "cornell_smoke stats" "checker_scale"; do RTG_SYNC=1 timeout 60 python tools/try_case.py $c >> $O/try.txt 2>&1; echo "rc=$?" >> $O/try.txt; done
grep "bit-equal\|rc=" $O/try.txt
for sy in 1 0; do RTG_SYNC=$sy timeout 300 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 cornell_smoke 300 300 100 volume 300 300 100 book2_bvh 800 800 100 simple_light 300 300 20 2>&1 | grep -v "^\[" | sed "s/^/sync $sy /" >> $O/t.txt; done
RTG_SYNC=1 RTG_VERBOSE=1 timeout 100 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\] pool sched\|wave-time" | sort -u >> $O/t.txt
cat $O/t.txt
