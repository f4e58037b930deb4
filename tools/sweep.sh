#!/bin/bash
# usage: tools/sweep.sh [-a "bench args"] "ENV1=a ENV2=b" ... ; prints kernel ms for each env setting (run on the GPU box)
ARGS=""
if [ "$1" = "-a" ]; then ARGS="$2"; shift 2; fi
for cfg in "$@"; do
  ms=$(env $cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.2f ms  %.0f Ms/s' % (j['roofline']['kernel_ms_avg'], j['value']))")
  echo "$ARGS $cfg => $ms"
done
