#!/bin/bash
# lab: fabric-side read traffic (FETCH_SIZE) and L2 hits / misses of the row gathers per channel-slab setting
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "1 1" "6 2" "4 2" "6 4"; do
  set -- $cfg
  echo "== GEOT_GR_SLABS=$1 GEOT_CL_TILES_MULT=$2"
  for CTR in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    TAG=$(echo $CTR | tr ' ' '_')
    GEOT_GR_SLABS=$1 GEOT_CL_TILES_MULT=$2 CI=1536 ONLY=gather_rows ITER=3 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/grpmc_$TAG -o pmc -- python3 $GRAFT_REPO_ROOT/tools/hbm_time.py > $OUT/grpmc.log 2>&1
    python3 - "$OUT/grpmc_$TAG" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "gather_rows_csr" in k:
        acc[k.split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        m = sum(v) / len(v)
        print("%-32s %-14s %12.4g%s" % (k, c, m, "  (= %.0f MB fetched, x2 rule applied)" % (2 * m * 1024 / 1e6) if c == "FETCH_SIZE" else ""))
PY
    rm -rf $OUT/grpmc_$TAG
  done
done
