#!/bin/bash
# lab: every point-major measurement in one go -> gpurun_out/r03_cl_*.txt
O=gpurun_out
timeout -k 10 300 python tools/fp_cl_lab.py 2>&1 | grep -v amdgpu.ids > $O/r03_cl_kernels.txt
timeout -k 10 300 python tools/lab/fp_cl_locality.py 2>&1 | grep -v amdgpu.ids > $O/r03_cl_locality.txt
timeout -k 10 300 python tools/lab/fp_cl_scaling.py 2>&1 | grep -v amdgpu.ids > $O/r03_cl_scaling.txt
timeout -k 10 300 python tools/fp_stage_time.py 2>&1 | grep -v amdgpu.ids > $O/r03_cl_stage_time.txt
timeout -k 10 300 python tools/lab/fp_gemms.py 2>&1 | grep -v amdgpu.ids > $O/r03_cl_gemms.txt
bash tools/lab/ab_layout.sh > $O/r03_cl_ab_model.txt 2>&1
bash tools/lab/cl_pmc.sh > $O/r03_cl_pmc.txt 2>&1
tail -4 $O/r03_cl_kernels.txt; cat $O/r03_cl_ab_model.txt
