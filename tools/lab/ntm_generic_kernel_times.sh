#!/bin/bash
# lab: kernel-only times (rocprofv3 --kernel-trace --stats) of sig_t_mean forward / d raw per class count, 8 x 24000 points:
# the HIP-event figures of ntm_generic_time.py include ~15 us of host time per call, which hides the small counts
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/ntm_kt
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$C -o kt -- python3 $ROOT/tools/lab/ntm_generic_time.py $C > $OUT/kt_$C.log 2>&1
  python3 - <<PY
import csv, glob
c = $C
fb, bb = 4.0 * 8 * 24000 * (c + c * c), 4.0 * 8 * 24000 * (c + 2 * c * c)
for f in glob.glob("$OUT/kt_$C/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "sig_t_mean" not in n:
            continue
        us = float(r["AverageNs"]) / 1e3
        kind = "mfma" if "mfma_kernel<%d" % (8 if c <= 8 else 16 if c <= 16 else 32) in n else ("rows" if "rows_kernel" in n else "specialised")
        bwd = n.split("(")[0].split("<")[1].split(",")[1].strip().startswith("true")   # second template argument: BACKWARD
        print("C = %2d  %-11s %-7s %8.1f us  %5.2f TB/s   (%s calls)" % (c, kind, "d raw" if bwd else "forward", us, (bb if bwd else fb) / us / 1e6, r["Calls"]))
PY
  rm -rf $OUT/kt_$C
done
