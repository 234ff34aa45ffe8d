#!/bin/bash
# every randomised checker of tools/ in one go (last lines of each); SEED shifts all of them
S=${SEED:-3}
for t in fuzz_parity grad_fuzz ntm_fuzz misc_fuzz dense_fuzz sa_fuzz cl_fuzz edge_fuzz; do
  echo "== $t"
  SEED=$S timeout -k 10 600 python tools/$t.py 2>&1 | grep -v amdgpu.ids | tail -2
done
