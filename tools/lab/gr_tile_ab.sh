#!/bin/bash
# lab: the BatchNorm row gather at 8 clouds, 24000 <- 8192 -- tile form against the list walk: HIP-event time (tools/hbm_time.py),
# then FETCH_SIZE / WRITE_SIZE per launch in separate rocprofv3 passes.   Usage: bash tools/lab/gr_tile_ab.sh [CI ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/gr_tile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CI in ${@:-1536 384}; do
  for FORM in tile list; do
    echo "== C = $CI, GEOT_GR_FORM=$FORM"
    CI=$CI ONLY="gather_rows_csr_bn_cl" GEOT_GR_FORM=$FORM python3 $ROOT/tools/hbm_time.py 2>&1 | grep -v amdgpu.ids | grep "gather_rows"
    for CTR in FETCH_SIZE WRITE_SIZE; do
      CI=$CI ONLY="gather_rows_csr_bn_cl" ITER=3 GEOT_GR_FORM=$FORM rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc_${CI}_${FORM}_$CTR -o pmc -- python3 $ROOT/tools/hbm_time.py > $OUT/pmc_${CI}_${FORM}_$CTR.log 2>&1
    done
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_${CI}_${FORM}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gather_rows" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    alg = 4.0 * 8 * $CI * (2 * 24000 + 8192)
    print("   %s: FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB per launch -> (2 x fetch + write) %.0f MB = %.2f x the algorithmic %.0f MB" % (n, fe, wr, (2 * fe + wr) * 1024 / 1e6, (2 * fe + wr) * 1024 / alg, alg / 1e6))
PY
    rm -rf $OUT/pmc_${CI}_${FORM}_*
  done
done
