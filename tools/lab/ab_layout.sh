#!/bin/bash
# lab: A/B on one box, alternating runs, fixmatch workload: look-ahead on / off
for rep in 1 2 3; do
  for flag in "" "--no-lookahead"; do
    python bench.py --workload fixmatch $flag --steps 40 --warmup 5 --no-cpu-baseline --no-dense-reference 2>/tmp/ab.err | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('lookahead' if '$flag' == '' else 'plain    ', round(r['value'],2), round(r['ms_per_step'],3), round(r.get('host_issue_ms_per_step'),2))" || tail -5 /tmp/ab.err
  done
done
