#!/bin/bash
# lab: bench.py (model workload) with the EdgeConv kernels of the start of the round against the current ones, alternating on one box
cp geot_amd/csrc/edgeconv.hip /tmp/edgeconv_new.hip
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp tools/_lab/edgeconv_old.hip geot_amd/csrc/edgeconv.hip; else cp /tmp/edgeconv_new.hip geot_amd/csrc/edgeconv.hip; fi
    python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
    python bench.py --steps 20 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', round(d['value'],2), 'clouds/s', round(d['ms_per_step'],3), 'ms;', 'edge grad', round(d['hot_path']['top_entry_points_ms'].get('geot_edgeconv_gn_max_grad_rix',0),3))"
  done
done
cp /tmp/edgeconv_new.hip geot_amd/csrc/edgeconv.hip
python -m geot_amd.build --force > /dev/null 2>&1
