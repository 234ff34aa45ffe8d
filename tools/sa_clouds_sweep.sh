rm -f gpurun_out/sa_sweep.txt
for n in 1 2 4 8 16 32 64 128 256; do
  steps=10; [ $n -ge 64 ] && steps=4
  python bench.py --no-cpu-baseline --clouds $n --steps $steps --warmup 2 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('clouds_per_step=%d  %.1f clouds/s  %.2f ms/step  fps_launch %.2f ms  sa_body %.3f ms (%.1f %% of the fp32 MFMA peak)' % ($n, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_secondary']['avg_launch_ms'], 100*d['roofline_secondary']['frac']))" >> gpurun_out/sa_sweep.txt || exit 1
done
