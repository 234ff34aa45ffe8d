#!/bin/bash
# lab: the dP kernel of the EdgeConv gradient -- the round-2 kernel against the pipelined walk, group sizes and channels per workgroup
set -o pipefail
run() { timeout -k 10 300 python tools/lab/edge_time.py 2>&1 | grep -v amdgpu.ids; }
if [ -f tools/_lab/edgeconv_old.hip ]; then
  cp geot_amd/csrc/edgeconv.hip /tmp/edgeconv_new.hip
  cp tools/_lab/edgeconv_old.hip geot_amd/csrc/edgeconv.hip
  python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== round-2 kernel"; run
  cp /tmp/edgeconv_new.hip geot_amd/csrc/edgeconv.hip
fi
for v in "4 4 4" "2 4 4" "8 4 4" "4 3 4" "4 6 4" "4 4 2" "4 4 1"; do
  set -- $v
  GEOT_EXTRA_HIPCC_FLAGS="-DGEOT_EC_LAB_TG=$1 -DGEOT_EC_LAB_E=$2 -DGEOT_EC_LAB_CH=$3" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== TG $1 E $2 CH<= $3"; run
done
python -m geot_amd.build --force > /dev/null 2>&1
