#!/bin/bash
# usage (GPU box): tools/exp.sh <out-tag> [variant ...]   -- one A/B round of a schedule experiment: the GPU test-suite on the
# checked-in library, then kernel times (tools/ab.sh), the fixed cost per launch (tools/tail_probe.py) and, for builds with
# -DRT_TIMELINE (named tl*), the drain timeline of every named build under csrc/variants/.  Output: gpurun_out/<tag>/.
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
python -m pytest tests -m gpu -x -q --timeout=600 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
SC="book1 1200 800 50 book2 800 800 100 book2 300 300 100 book2_bvh 800 800 100 cornell 300 300 100"
V=(); for v in "$@"; do case $v in tl*) ;; *) V+=($v);; esac; done
tools/ab.sh "$SC" "${V[@]}" > $O/ab.txt 2>&1
for v in head "${V[@]}"; do
  [ $v = head ] && unset RTIOW_GPU_LIB || export RTIOW_GPU_LIB=$PWD/rtiow-rust_amd/csrc/variants/$v.so
  python tools/tail_probe.py > $O/tail_book1_$v.txt 2>&1
  python tools/tail_probe.py book2 800 800 > $O/tail_book2_$v.txt 2>&1
done
for v in "$@"; do case $v in tl*)
  export RTIOW_GPU_LIB=$PWD/rtiow-rust_amd/csrc/variants/$v.so
  python tools/timeline.py book1 1200 800 50 > $O/${v}_book1_50.txt 2>&1
  python tools/timeline.py book2 800 800 100 > $O/${v}_book2_100.txt 2>&1
  python tools/timeline.py book2 300 300 100 > $O/${v}_book2_300.txt 2>&1;; esac; done
unset RTIOW_GPU_LIB
cat $O/ab.txt; grep -h "max_bounces 50" $O/tail_*.txt
