#!/bin/bash
# per-kernel times of the reverse-index build + gradient walk at the group_points shape (kernel trace of group_grad_transposed.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rix -o kt -- python3 $ROOT/tools/lab/group_grad_transposed.py > $OUT/rix_kt.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof_rix/kt_kernel_stats.csv")))
for r in rows[:28]:
    print("%-70s calls %6s  avg %9.1f us  total %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
rm -rf $OUT/prof_rix
