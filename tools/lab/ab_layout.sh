#!/bin/bash
# lab: A/B on one box, alternating runs: priority of the model's side stream (GEOT_SIDE_PRIORITY=high|normal)
for rep in 1 2 3; do
  for prio in high normal; do
    GEOT_SIDE_PRIORITY=$prio python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dense-reference --no-saturated 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('$prio', round(r['value'],2), round(r['ms_per_step'],3), 'fps launch ms', round(r['roofline']['avg_launch_ms'],3))"
  done
done
