#!/bin/bash
# kernel trace of one bench.py command -> tools/trace_gaps.py (where the non-kernel time of a step is).  usage: gaps.sh <tag> <anchors/step> bench args...
TAG=$1; APS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $ROOT/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-dense-reference > $OUT/${TAG}_kt.log 2>&1
python3 $ROOT/tools/trace_gaps.py $OUT/prof_$TAG/kt_kernel_trace.csv --skip 3 --steps 3 --anchors-per-step $APS > $OUT/${TAG}_gaps.txt 2>&1
python3 $ROOT/tools/trace_window.py $OUT/prof_$TAG/kt_kernel_trace.csv --skip 3 --steps 3 --anchors-per-step $APS --top 40 -o $OUT/${TAG}_window.csv > $OUT/${TAG}_window.txt 2>&1
rm -rf $OUT/prof_$TAG
echo "== $TAG: $*"; cat $OUT/${TAG}_gaps.txt; head -1 $OUT/${TAG}_window.txt
