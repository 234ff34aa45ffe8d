#!/bin/bash
# lab: kernel trace of tools/fp_stage_time.py -> per-kernel average duration, in launch order of one pass, per layout
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $OUT/st_prof -o kt -- python3 $GRAFT_REPO_ROOT/tools/fp_stage_time.py > $OUT/st.log 2>&1
python3 - <<'PY'
import csv, os, glob, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
f = glob.glob(out + "/st_prof/**/kt_kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# first 13 passes (3 warm + 10) of prop0 cf start at the first fp_front_kernel<4>; print avg per kernel name within prop0 cf and prop0 cl
def section(start_pred, stop_pred):
    on = False; acc = collections.OrderedDict()
    for s, e, k in rows:
        if not on and start_pred(k): on = True
        if on and stop_pred(k): break
        if on:
            a = acc.setdefault(k[:100], [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    return acc
cf = section(lambda k: "fp_front_kernel<4>" in k, lambda k: "fp_front_cl_kernel<5>" in k or "spatial" in k or "kg_" in k)
cl = section(lambda k: "fp_front_cl_kernel<5>" in k, lambda k: "fp_front_kernel<8>" in k)
for name, acc in (("prop0 cf", cf), ("prop0 cl", cl)):
    tot = sum(v[1] for v in acc.values())
    print("==", name, "total kernel time %.1f us over %s passes" % (tot, "?"))
    for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:22]:
        print("   %4d calls  avg %8.1f us  total %9.1f  %s" % (c, t / c, t, k[:80]))
PY
rm -rf $OUT/st_prof
