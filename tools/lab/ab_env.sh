#!/bin/bash
# lab: bench.py with an environment switch on / off, alternating on one box: tools/lab/ab_env.sh VAR A B [bench args]
VAR=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 20 --no-cpu-baseline --no-dense-reference "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', round(d['value'],2), 'clouds/s', round(d['ms_per_step'],3), 'ms')"
  done
done
