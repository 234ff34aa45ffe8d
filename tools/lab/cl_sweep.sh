#!/bin/bash
# lab: rows per group / stages of the forward's software pipeline
for cfg in "2 2" "2 3" "1 4" "1 3" "4 2" "2 4"; do
  set -- $cfg
  GEOT_EXTRA_HIPCC_FLAGS="-DGEOT_CL_LAB_U=$1 -DGEOT_CL_LAB_STAGES=$2" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== U $1 stages $2"
  timeout -k 10 300 python tools/lab/fp_cl_scaling.py 2>&1 | grep -E "n=24000 m= 8192|n= 8192 m=  512"
done
python -m geot_amd.build --force > /dev/null 2>&1
