#!/bin/bash
# each sequence in its own process: a capture fault kills the process
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
out=gpurun_out/graph_bisect.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 300 python tools/lab/graph_bisect.py "$@" 2>&1 | grep -v "^  File\|amdgpu.ids\|Extension modules" | tail -8 >> $out; }
run g0
run g0
run g1,g0
run g0,g0
run e0,g0
run e1,g1,e0,g0
run g1,g0 keep
cat $out
