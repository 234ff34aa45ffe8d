#!/bin/bash
# lab: bench.py (model workload) with the EdgeConv kernels of an earlier revision against the current ones, alternating on one box.
# usage: OLD_REV=<git rev that holds the baseline kernel> tools/lab/ab_edge.sh      (run where .git is available, i.e. the build container;
# on the GPU box pass OLD_SRC=<path of a copy of that revision's geot_amd/csrc/edgeconv.hip> made before the snapshot)
set -eu
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
CUR=geot_amd/csrc/edgeconv.hip
NEW=$(mktemp /tmp/edgeconv_new.XXXXXX.hip); OLD=$(mktemp /tmp/edgeconv_old.XXXXXX.hip)
cp "$CUR" "$NEW"
restore() { cp "$NEW" "$CUR"; python -m geot_amd.build --force > /dev/null 2>&1 || true; rm -f "$NEW" "$OLD"; }
trap restore EXIT
if [ -n "${OLD_SRC:-}" ]; then cp "$OLD_SRC" "$OLD"; else git show "${OLD_REV:?set OLD_REV or OLD_SRC}:$CUR" > "$OLD"; fi
[ -s "$OLD" ] || { echo "no baseline kernel"; exit 1; }
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp "$OLD" "$CUR"; else cp "$NEW" "$CUR"; fi
    python -m geot_amd.build --force > /dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
    python bench.py --steps 20 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$which', round(d['value'],2), 'clouds/s', round(d['ms_per_step'],3), 'ms;', 'edge grad', round(d['hot_path']['top_entry_points_ms'].get('geot_edgeconv_gn_max_grad_rix',0),3))"
  done
done
