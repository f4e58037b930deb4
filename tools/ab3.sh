#!/bin/bash
# usage (GPU box): tools/ab3.sh <variant ...>  -- pool-2 kernel (pool2 = 2) of "head" and of the named builds on book-2 at three regimes:
# 800x800x3 (thin: the drain), 300x300x100 (the reference's shipped frame), 800x800x100, 800x800x300; interleaved twice; ms wall.
V=(head "$@")
for rep in 1 2; do
  for v in "${V[@]}"; do
    L=$PWD/rtiow-rust_amd/csrc/variants/$v.so; [ $v = head ] && L=$PWD/rtiow-rust_amd/csrc/librtiow_gpu.so
    printf "%-10s" $v
    RTG_POOL2=2 RTIOW_GPU_LIB=$L timeout 200 python tools/time_scenes.py book2 800 800 3 book2 300 300 100 book2 800 800 100 book2 800 800 300 2>&1 | grep "^book2" | cut -c30-41 | tr "\n" " "; echo
  done
done
