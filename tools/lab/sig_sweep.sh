#!/bin/bash
# lab: sig_t_mean weight gradient (ntm.hip, fp32-MFMA kernel): threads per block, waves per SIMD the compiler is held to
set -o pipefail
for v in "-DGEOT_SIG_LAB_BWD_THREADS=256" "-DGEOT_SIG_LAB_BWD_THREADS=512" "-DGEOT_SIG_LAB_BWD_THREADS=512 -DGEOT_SIG_LAB_BWD_WPS=4" "-DGEOT_SIG_LAB_BWD_THREADS=256 -DGEOT_SIG_LAB_NOPREFETCH"; do
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== flags: $v"
  ONLY=sig_t_mean timeout -k 10 300 python tools/hbm_time.py 2>&1 | grep -v amdgpu.ids
  timeout -k 10 300 python -m pytest tests/test_ntm_gpu.py tests/test_ref_fixtures_gpu.py -q -x -k "sig_t_mean" 2>&1 | tail -1
done
python -m geot_amd.build --force > /dev/null 2>&1
