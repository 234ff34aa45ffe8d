#!/bin/bash
# Lab: the channels-first gradient kernels (csrc/tile_scatter.hip) under rocprofv3 -- kernel times, then SQ / traffic
# counters in separate passes.  Usage (GPU box): bash tools/ts_lab.sh [CI]   -> gpurun_out/ts_lab/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ts_lab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CI=${1:-1536} ONLY="bwd C" ITER=${ITER:-10}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$CI -o kt -- python3 $ROOT/tools/hbm_time.py > $OUT/kt_$CI.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/kt_$CI/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:8]:
        print("%-90s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  TAG=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_${CI}_$TAG -o pmc -- python3 $ROOT/tools/hbm_time.py > $OUT/pmc_${CI}_$TAG.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_${CI}_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ts_" in n or "table_gather" in n or "scatter_rows" in n or "transpose_add" in n or "rix_" in n:
            acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n)
    for k, v in sorted(d.items()):
        print("    %-24s avg per launch %16.1f  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
