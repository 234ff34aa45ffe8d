#!/bin/bash
# lab: the dP kernel of the EdgeConv gradient -- the round-2 kernel against the pipelined walk, group sizes and channels per workgroup
set -o pipefail
run() { timeout -k 10 300 python tools/lab/edge_time.py 2>&1 | grep -v amdgpu.ids; }
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
if [ -n "${OLD_SRC:-}" ] && [ -s "$OLD_SRC" ]; then      # OLD_SRC: a copy of an earlier revision's geot_amd/csrc/edgeconv.hip (git show <rev>:...)
  NEW=$(mktemp /tmp/edgeconv_new.XXXXXX.hip)
  cp geot_amd/csrc/edgeconv.hip "$NEW"
  trap 'cp "$NEW" geot_amd/csrc/edgeconv.hip; python -m geot_amd.build --force > /dev/null 2>&1; rm -f "$NEW"' EXIT
  cp "$OLD_SRC" geot_amd/csrc/edgeconv.hip
  python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== baseline kernel ($OLD_SRC)"; run
  cp "$NEW" geot_amd/csrc/edgeconv.hip
fi
for v in "" "-DGEOT_EC_LAB_LG=0" "-DGEOT_EC_LAB_TG=8" "-DGEOT_EC_LAB_E=8 -DGEOT_EC_LAB_TG=2" "-DGEOT_EC_LAB_NOWALK" "-DGEOT_EC_LAB_NOSTAGE" "-DGEOT_EC_LAB_NOLDSREAD" "-DGEOT_EC_LAB_NOREV"; do
  GEOT_LAB_KERNELS=tools/lab/kernels GEOT_EXTRA_HIPCC_FLAGS="$v" python -m geot_amd.build --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== flags: ${v:-default (TG 4, E 4, lanes per target from the mean list length)}"; run
done
python -m geot_amd.build --force > /dev/null 2>&1
