#!/bin/bash
# lab: A/B on one box, alternating runs: look-ahead on / off (bench.py --no-lookahead)
for rep in 1 2 3; do
  for flag in "" "--no-lookahead"; do
    python bench.py $flag --steps 40 --warmup 5 --no-cpu-baseline --no-dense-reference --no-saturated 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('lookahead' if '$flag' == '' else 'plain    ', round(r['value'],2), round(r['ms_per_step'],3), round(r.get('host_issue_ms_per_step'),2))"
  done
done
