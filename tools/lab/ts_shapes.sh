#!/bin/bash
# Lab: the channels-first interpolation gradient at the model's FP shapes, new form (tiles) against the older ones.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for shape in "1536 24000 8192" "384 24000 8192" "1536 8192 512" "384 8192 512" "1536 4096 512" "384 4096 512" "384 8192 4096" "64 24000 6000"; do
  set -- $shape
  for impl in tiles csr; do
    printf "%-6s " $impl
    GEOT_GATHER_IMPL=$impl C=$1 N=$2 M=$3 FPS=1 python3 $ROOT/tools/lab/gather_grad_time.py 2>/dev/null
  done
done
