#!/bin/bash
# lab: A/B of the FP layout (GEOT_FP_LAYOUT) on one box, alternating runs, the recorded GEMM selection
for rep in 1 2 3; do
  for lay in cf cl; do
    GEOT_FP_LAYOUT=$lay python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-dense-reference --no-saturated 2>/dev/null | tail -1 > /tmp/ab.json
    python -c "import json; r=json.load(open('/tmp/ab.json')); print('$lay', round(r['value'],2), round(r['ms_per_step'],3), round(r.get('host_issue_ms_per_step'),2), r['config'].get('gemm_selection','')[:40] if isinstance(r.get('config'),dict) else '')"
  done
done
