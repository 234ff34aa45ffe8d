#!/bin/bash
# lab: the row gathers with the XCD-aware deal of the targets (default) against contiguous shares of the pair stream (-DGEOT_GR_LAB_SHARES)
for v in "-DGEOT_GR_LAB_SHARES" "" "-DGEOT_GR_LAB_GT=2" "-DGEOT_GR_LAB_GT=4" "-DGEOT_GR_LAB_GT=16"; do
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== ${v:-default: targets dealt round-robin inside an XCD}"
  for c in 384 1536; do CI=$c ONLY=gather_rows timeout -k 10 300 python tools/hbm_time.py 2>&1 | grep -v amdgpu; done
done
python -m geot_amd.build --force > /dev/null 2>&1
