#!/bin/bash
# lab: L2 hit / miss and HBM-side request counters of the point-major kernels (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for CTR in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  TAG=$(echo $CTR | tr ' ' '_')
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/clpmc_$TAG -o pmc -- python3 $GRAFT_REPO_ROOT/tools/fp_cl_lab.py > $OUT/clpmc_$TAG.log 2>&1
  python3 - "$OUT/clpmc_$TAG" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "fp_front_cl_kernel<5>" in k or "gather_rows_csr_cl" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        # calls in launch order: first third of the calls = prop0
        runs = []          # run-length encoding of the per-call values (rounded to 3 digits)
        for x in v:
            t = float("%.3g" % x)
            if runs and runs[-1][0] == t: runs[-1][1] += 1
            else: runs.append([t, 1])
        print("%-62s %-18s %s" % (k, c, " ".join("%gx%d" % (a, n) for a, n in runs[:12])))
PY
  rm -rf $OUT/clpmc_$TAG
done
