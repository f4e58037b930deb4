#!/bin/bash
# usage (GPU box): tools/sweep_scenes.sh "ENV=.." ...  -- time the full-feature scenes under each env setting
SC=${SCENES:-"book2 800 800 100 book2_bvh 800 800 100 cornell 300 300 100 volume 300 300 100"}
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 200 python tools/time_scenes.py $SC 2>&1 | grep -v "^\[rtg\]" | cut -c1-75
done
