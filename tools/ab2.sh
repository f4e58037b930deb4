#!/bin/bash
# usage (GPU box): tools/ab2.sh "<probe_pool2.py args>" [variant ...] -- pool-2 kernel of the checked-in library ("head") and of the named builds
# under csrc/variants/ (default: all), interleaved twice; columns: first kernel ms, pool-2 ms, ratio
A=$1; shift
V=("$@"); [ ${#V[@]} = 0 ] && V=($(cd rtiow-rust_amd/csrc/variants && ls *.so | sed 's/\.so$//'))
for rep in 1 2; do
  printf "%-14s" head; timeout 200 python tools/probe_pool2.py $A 2>&1 | tail -1 | cut -c27-60
  for v in "${V[@]}"; do
    printf "%-14s" $v; RTIOW_GPU_LIB=$PWD/rtiow-rust_amd/csrc/variants/$v.so timeout 200 python tools/probe_pool2.py $A 2>&1 | tail -1 | cut -c27-60
  done
done
