for run in 1 8; do
  echo "run=$run" >> gpurun_out/sa_grid.log
  GEOT_SA_RUN=$run timeout -k 10 120 python tools/sa_lab.py run 2>&1 | grep base >> gpurun_out/sa_grid.log || exit 1
done
