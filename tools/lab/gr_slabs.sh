#!/bin/bash
# lab: the row gathers with the rows cut into channel slabs (grid y), slab after slab
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
for sl in 1 2 3 4 6; do for mult in 1 2; do
  echo "== GEOT_GR_SLABS=$sl GEOT_CL_TILES_MULT=$mult"
  for c in 384 1536; do GEOT_GR_SLABS=$sl GEOT_CL_TILES_MULT=$mult CI=$c ONLY=gather_rows timeout -k 10 300 python tools/hbm_time.py 2>&1 | grep -v amdgpu; done
done; done
