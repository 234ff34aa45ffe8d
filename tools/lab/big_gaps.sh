#!/bin/bash
# usage: big_gaps.sh <tag> bench args...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $ROOT/bench.py "$@" --steps 8 --warmup 2 --no-cpu-baseline --no-dense-reference > $OUT/${TAG}_kt.log 2>&1
python3 $ROOT/tools/lab/big_gaps.py $OUT/prof_$TAG/kt_kernel_trace.csv 14 > $OUT/${TAG}_biggaps.txt 2>&1
rm -rf $OUT/prof_$TAG
echo "== $TAG: $*"; cat $OUT/${TAG}_biggaps.txt
