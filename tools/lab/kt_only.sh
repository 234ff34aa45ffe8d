#!/bin/bash
# lab: kernel-trace window only (no PMC) for a bench configuration: tools/lab/kt_only.sh <tag> [bench args]
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $ROOT/bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-dense-reference --no-saturated > $OUT/${TAG}_kt.log 2>&1
python3 $ROOT/tools/trace_window.py $OUT/prof_$TAG/kt_kernel_trace.csv --skip 2 --steps 3 -o $OUT/${TAG}_window.csv --per-launch $OUT/${TAG}_launches.csv > $OUT/${TAG}_window.txt 2>&1
rm -rf $OUT/prof_$TAG
head -3 $OUT/${TAG}_window.txt
