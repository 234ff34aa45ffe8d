#!/bin/bash
# lab: the BatchNorm row gather at the two 512-target stages (8 clouds, C = 1536) -- the few-target form (sources streamed once,
# sums in registers) against the list walk: HIP-event time, then FETCH_SIZE / WRITE_SIZE per launch in separate rocprofv3 passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/gr_small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SHAPE in "8192 512" "4096 512"; do
  set -- $SHAPE
  for FORM in default list; do
    echo "== $1 <- $2, GEOT_GR_FORM=$FORM"
    GEOT_GR_FORM=$FORM python3 $ROOT/tools/lab/gr_small_time.py $1 $2 2>&1 | grep -v amdgpu.ids
    for CTR in FETCH_SIZE WRITE_SIZE; do
      ITER=3 GEOT_GR_FORM=$FORM rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc_$1_${FORM}_$CTR -o pmc -- python3 $ROOT/tools/lab/gr_small_time.py $1 $2 > $OUT/pmc_$1_${FORM}_$CTR.log 2>&1
    done
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_$1_${FORM}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather_rows" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    alg = 4.0 * 8 * 1536 * (2 * $1 + $2)
    print("   %s: FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB per launch -> (2 x fetch + write) %.0f MB = %.2f x the algorithmic %.0f MB" % (n, fe, wr, (2 * fe + wr) * 1024 / 1e6, (2 * fe + wr) * 1024 / alg, alg / 1e6))
PY
    rm -rf $OUT/pmc_$1_${FORM}_*
  done
done
