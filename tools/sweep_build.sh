#!/bin/bash
# usage (GPU box): tools/sweep_build.sh [-s] "<EXTRA defines>" ...   rebuilds the library per setting, then prints
# the C2 bench time (default) or the full-feature scene timings (-s)
mode=bench; [ "$1" = "-s" ] && { mode=scenes; shift; }
for ex in "$@"; do
  (cd rtiow-rust_amd/csrc && make -B EXTRA="$ex" librtiow_gpu.so > /dev/null 2>&1) || { echo "$ex => build failed"; continue; }
  if [ $mode = bench ]; then tools/sweep.sh "RTG_X=1" | sed "s|RTG_X=1|$ex|"; else tools/sweep_scenes.sh "RTG_X=1" | sed "s|RTG_X=1|$ex|"; fi
done
