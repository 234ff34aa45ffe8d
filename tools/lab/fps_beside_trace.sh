#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_fb -o kt -- python3 $ROOT/tools/lab/fps_beside_trace.py run ${1:-8} > $OUT/fb_kt.log 2>&1
python3 $ROOT/tools/lab/fps_beside_trace.py report $OUT/prof_fb/kt_kernel_trace.csv > $OUT/fps_beside.txt 2>&1
head -3 $OUT/prof_fb/kt_kernel_trace.csv > $OUT/fb_head.txt
rm -rf $OUT/prof_fb
cat $OUT/fps_beside.txt
