#!/bin/bash
# lab: A/B on one box, alternating runs: what the look-ahead queues (GEOT_LOOKAHEAD=group|all) and none (--no-lookahead)
for rep in 1 2 3; do
  for mode in group all none; do
    flag=""; [ $mode = none ] && flag="--no-lookahead"
    GEOT_LOOKAHEAD=$mode python bench.py $flag --steps 40 --warmup 5 --no-cpu-baseline --no-dense-reference --no-saturated 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('$mode', round(r['value'],2), round(r['ms_per_step'],3), 'fps in-step / alone ms', round(r['roofline']['avg_launch_ms'],3), round(r['roofline']['alone_launch_ms'],3))"
  done
done
