#!/bin/bash
# usage: tools/sweep.sh "ENV1=a ENV2=b" ... ; prints kernel ms for each env setting (run on the GPU box)
for cfg in "$@"; do
  ms=$(env $cfg python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.2f ms  %.0f Ms/s' % (j['roofline']['kernel_ms_avg'], j['value']))")
  echo "$cfg => $ms"
done
