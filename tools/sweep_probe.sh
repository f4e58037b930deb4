#!/bin/bash
# usage: tools/sweep_probe.sh "ENV1=a ENV2=b" ... ; tail_probe.py (first line: bounce cap 50) for each env setting
for cfg in "$@"; do
  echo "$cfg => $(env $cfg python tools/tail_probe.py 2>/dev/null | head -1)"
done
