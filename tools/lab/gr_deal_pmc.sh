#!/bin/bash
# lab: fabric-side read traffic (FETCH_SIZE) and L2 hits / misses of the row gathers, contiguous shares against the XCD-aware deal
OUT=$GRAFT_REPO_ROOT/gpurun_out
for v in "-DGEOT_GR_LAB_SHARES" ""; do
  cd $GRAFT_REPO_ROOT
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== ${v:-default: targets dealt round-robin inside an XCD}"
  cd /tmp && export TMPDIR=/tmp
  for CTR in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    TAG=$(echo $CTR | tr ' ' '_')
    for c in 384 1536; do
      CI=$c ONLY=gather_rows ITER=3 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/grpmc_$TAG -o pmc -- python3 $GRAFT_REPO_ROOT/tools/hbm_time.py > $OUT/grpmc.log 2>&1
      python3 - "$OUT/grpmc_$TAG" $c <<'PY'
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
        print("C=%-5s %-32s %-14s %12.4g%s" % (sys.argv[2], k, c, m, "  (= %.0f MB fetched, x2 rule applied)" % (2 * m * 1024 / 1e6) if c == "FETCH_SIZE" else ""))
PY
      rm -rf $OUT/grpmc_$TAG
    done
  done
done
cd $GRAFT_REPO_ROOT; python -m geot_amd.build --force > /dev/null 2>&1
