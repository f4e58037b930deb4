O=gpurun_out/r02q; mkdir -p $O
for cfg in "RTG_SPHERE_MIN=64 RTG_BOX_LEAVE=64 RTG_GATHER_MIN=64" "RTG_SPHERE_MIN=48 RTG_BOX_LEAVE=64 RTG_GATHER_MIN=64" "RTG_SPHERE_MIN=32 RTG_BOX_LEAVE=64 RTG_GATHER_MIN=56" "RTG_SPHERE_MIN=64 RTG_BOX_LEAVE=64 RTG_GATHER_MIN=64 RTG_RUN_AHEAD=64" "RTG_SPHERE_MIN=24 RTG_BOX_LEAVE=48 RTG_GATHER_MIN=48"; do
  env RTG_SYNC=1 $cfg timeout 200 python tools/time_scenes.py book2 800 800 100 cornell 300 300 100 volume 300 300 100 2>&1 | grep -v "^\[" | sed "s/^/[$cfg] /" >> $O/t.txt
done
env RTG_SYNC=1 RTG_SPHERE_MIN=64 RTG_BOX_LEAVE=64 RTG_GATHER_MIN=64 RTG_VERBOSE=1 timeout 100 python tools/time_scenes.py book2 800 800 100 2>&1 | grep "^\[rtg\] pool sched\|wave-time" | sort -u >> $O/t.txt
cat $O/t.txt
