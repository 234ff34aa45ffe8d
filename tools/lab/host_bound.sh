#!/bin/bash
# lab: wall and host-issue time per step against the number of clouds (is the step host- or GPU-bound?)
for b in 1 2 4 8; do
    python bench.py --clouds $b --steps 20 --warmup 5 --no-cpu-baseline --no-dense-reference --no-saturated 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('clouds $b', round(r['value'],1), round(r['ms_per_step'],2), round(r.get('host_issue_ms_per_step'),2))"
done
