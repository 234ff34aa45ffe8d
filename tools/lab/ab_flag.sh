#!/bin/bash
# lab: bench.py with the library built with / without a compile flag, alternating on one box: tools/lab/ab_flag.sh "<flags>" [bench args]
FLAGS=$1; shift
for rep in 1 2 3; do
  for v in "$FLAGS" ""; do
    GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
    python bench.py --steps 20 --no-cpu-baseline --no-dense-reference "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('flags: ${v:-(default)}', round(d['value'],2), 'clouds/s', round(d['ms_per_step'],3), 'ms')"
  done
done
python -m geot_amd.build --force > /dev/null 2>&1
