#!/bin/bash
# Lab: the run-time-C sig_t_mean kernels (csrc/ntm_generic.hip) under rocprofv3 -- kernel times, then SQ / MFMA / traffic
# counters in separate passes.  Usage (GPU box): bash tools/lab/ntm_generic_lab.sh C [C ...]   -> gpurun_out/ntm_lab/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/ntm_lab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=$(echo "$*" | tr ' ' '_')
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$TAG -o kt -- python3 $ROOT/tools/lab/ntm_generic_time.py "$@" > $OUT/kt_$TAG.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/kt_$TAG/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:10]:
        print("%-110s calls %5s avg %10.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$T -o pmc -- python3 $ROOT/tools/lab/ntm_generic_time.py "$@" > $OUT/pmc_${TAG}_$T.log 2>&1 || tail -3 $OUT/pmc_${TAG}_$T.log
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_${TAG}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gen_sig" in n or "sig_t_mean" in n:
            acc[n[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n)
    for k, v in sorted(d.items()):
        print("    %-28s avg per launch %16.1f  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
