#!/bin/bash
# usage (GPU box): tools/ab.sh "<time_scenes.py args>" [variant ...]   -- time the scenes on the checked-in library ("head" = csrc/librtiow_gpu.so)
# and on every named build under csrc/variants/ (default: all), interleaved twice so that clock / thermal drift shows up.
# The crc column is the framebuffer's: every variant must print the same one.
SC=$1; shift
V=("$@"); [ ${#V[@]} = 0 ] && V=($(cd rtiow-rust_amd/csrc/variants && ls *.so | sed 's/\.so$//'))
for rep in 1 2; do
  echo "== checked-in"; timeout 300 python tools/time_scenes.py $SC 2>&1 | grep -v "^\[rtg\]" | cut -c1-130
  for v in "${V[@]}"; do
    echo "== $v"; RTIOW_GPU_LIB=$PWD/rtiow-rust_amd/csrc/variants/$v.so timeout 300 python tools/time_scenes.py $SC 2>&1 | grep -v "^\[rtg\]" | cut -c1-130
  done
done
