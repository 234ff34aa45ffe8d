#!/bin/bash
# lab: pairs in flight per pipeline half of the BatchNorm-fused gather (2 row loads per pair)
for u in 2 3 4; do
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="-DGEOT_GRB_LAB_U=$u" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== GRB_U $u"
  timeout -k 10 300 python tools/fp_cl_lab.py 2>&1 | grep "fused backward"
done
python -m geot_amd.build --force > /dev/null 2>&1
