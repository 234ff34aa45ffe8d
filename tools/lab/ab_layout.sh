#!/bin/bash
# lab: A/B on one box, alternating runs: where the look-ahead is queued (GEOT_LOOKAHEAD_AT=blocks|forward); WL=model|fixmatch
for rep in 1 2 3; do
  for at in blocks forward; do
    GEOT_LOOKAHEAD_AT=$at python bench.py --workload ${WL:-model} --steps 40 --warmup 5 --no-cpu-baseline --no-dense-reference 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('$at', round(r['value'],2), round(r['ms_per_step'],3), 'fps in-step ms', round(r['roofline']['avg_launch_ms'],3))"
  done
done
