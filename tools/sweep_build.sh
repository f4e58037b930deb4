#!/bin/bash
# usage (GPU box): tools/sweep_build.sh "<EXTRA defines>" ...   rebuilds the library per setting, prints bench ms
for ex in "$@"; do
  (cd rtiow-rust_amd/csrc && make -B EXTRA="$ex" librtiow_gpu.so > /dev/null 2>&1) || { echo "$ex => build failed"; continue; }
  tools/sweep.sh "RTG_X=1" | sed "s|RTG_X=1|$ex|"
done
