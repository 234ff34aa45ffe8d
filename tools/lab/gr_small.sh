#!/bin/bash
# lab: the row gathers at the two small FP stages (8192 <- 512 and 4096 <- 512 points, C = 1536) against slabs / rounds
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for shape in "8192 512" "4096 512"; do set -- $shape
  for sl in "" 1 2 3 6; do for mult in "" 1 2 8; do
    echo "== NU=$1 MK=$2 GEOT_GR_SLABS=${sl:-default} GEOT_CL_TILES_MULT=${mult:-default}"
    ( [ -n "$sl" ] && export GEOT_GR_SLABS=$sl; [ -n "$mult" ] && export GEOT_CL_TILES_MULT=$mult; NU=$1 MK=$2 CI=1536 ONLY=gather_rows timeout -k 10 300 python tools/hbm_time.py 2>&1 | grep -v amdgpu )
  done; done
done
