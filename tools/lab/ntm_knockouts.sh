#!/bin/bash
# lab: knock-outs of the run-time-C MFMA sig_t_mean kernel (tools/lab/kernels/ntm_generic.hip): which part of a row tile costs what
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for v in "" "-DGEN_KO_STORE" "-DGEN_KO_MFMA" "-DGEN_KO_POST" "-DGEN_KO_STORE -DGEN_KO_MFMA" "-DGEN_KO_STORE -DGEN_KO_POST" "-DGEN_KO_STORE -DGEN_KO_MFMA -DGEN_KO_POST"; do
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== flags: ${v:-none}"
  GEOT_NTM_GENERIC= timeout -k 10 120 python tools/lab/ntm_generic_time.py "$@" 2>&1 | grep "C =" | sed 's/   rows:.*//'
done
python -m geot_amd.build --force > /dev/null 2>&1
