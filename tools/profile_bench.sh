#!/bin/bash
# Profile one bench.py workload on the GPU box: bench JSON, rocprofv3 kernel trace (whole-process --stats + the
# steady-state window of tools/trace_window.py) and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separately, with
# --kernel-trace only, as gpurun requires).  Usage: tools/profile_bench.sh <tag> [bench.py args...]
# Writes gpurun_out/<tag>.json, <tag>_kernel_stats.csv, <tag>_window.csv, <tag>_pmc_traffic.json
# The profiled passes run the EAGER step (--no-graph): the same kernels as the replay, launch by launch -- under the profiler a
# graph launch is serialised node by node and the window would measure the tool.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
NOGRAPH=""
case " $* " in *" sa "*|*" ntm "*|*" ops_only "*|*" backbone_ops "*) ;; *) NOGRAPH="--no-graph";; esac
cd $ROOT
python3 bench.py "$@" > $OUT/$TAG.json 2> $OUT/$TAG.err || { echo "bench failed"; tail -5 $OUT/$TAG.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $ROOT/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline $NOGRAPH > $OUT/${TAG}_kt.log 2>&1
cp $OUT/prof_$TAG/kt_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 $ROOT/tools/trace_window.py $OUT/prof_$TAG/kt_kernel_trace.csv --skip 2 --steps 3 --anchors-per-step ${ANCHORS_PER_STEP:-1} -o $OUT/${TAG}_window.csv --per-launch $OUT/${TAG}_launches.csv > $OUT/${TAG}_window.txt 2>&1
rm -rf $OUT/prof_$TAG
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$CTR -o pmc -- python3 $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline $NOGRAPH > $OUT/${TAG}_pmc_$CTR.log 2>&1
done
cd $ROOT
GEOT_COMMIT="${GEOT_COMMIT:-unknown}" python3 tools/pmc_summary.py traffic $OUT/pmc_${TAG}_FETCH_SIZE/pmc_counter_collection.csv $OUT/pmc_${TAG}_WRITE_SIZE/pmc_counter_collection.csv $OUT/${TAG}_pmc_traffic.json "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py $* --steps 2 --warmup 1 --no-cpu-baseline (two separate passes)" > /dev/null 2>$OUT/${TAG}_pmc.err
rm -rf $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
# SQ_KERNEL=<substring>: issue-side counters of that kernel (what the CUs it occupies actually do), two passes
if [ -n "${SQ_KERNEL:-}" ]; then
  cd /tmp
  P=1
  for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/sq_${TAG}_$P -o pmc -- python3 $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline $NOGRAPH > $OUT/${TAG}_sq_$P.log 2>&1
    P=$((P+1))
  done
  cd $ROOT
  python3 tools/pmc_summary.py sq $OUT/${TAG}_sq_counters.json "rocprofv3 --pmc {SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES | SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS} --kernel-trace -- python3 bench.py $* --steps 2 --warmup 1 --no-cpu-baseline" "$SQ_KERNEL" $OUT/sq_${TAG}_1/pmc_counter_collection.csv $OUT/sq_${TAG}_2/pmc_counter_collection.csv > /dev/null 2>>$OUT/${TAG}_pmc.err
  rm -rf $OUT/sq_${TAG}_1 $OUT/sq_${TAG}_2
fi
head -3 $OUT/${TAG}_window.txt
