#!/bin/bash
# usage (GPU box): tools/sweep_build.sh [-s] [-e "ENV=.. ENV=.."]... "<EXTRA defines>" ...   rebuilds the library per
# setting, then prints the C2 bench time (default) or the full-feature scene timings (-s) for every -e environment
mode=bench; [ "$1" = "-s" ] && { mode=scenes; shift; }
envs=()
while [ "$1" = "-e" ]; do envs+=("$2"); shift 2; done
[ ${#envs[@]} = 0 ] && envs=("RTG_X=1")
for ex in "$@"; do
  (cd rtiow-rust_amd/csrc && make -B EXTRA="$ex" librtiow_gpu.so > /dev/null 2>&1) || { echo "$ex => build failed"; continue; }
  if [ $mode = bench ]; then tools/sweep.sh "${envs[@]}" | sed "s|^|$ex |"; else tools/sweep_scenes.sh "${envs[@]}" | sed "s|^|$ex |"; fi
done
