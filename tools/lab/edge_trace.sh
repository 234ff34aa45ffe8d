#!/bin/bash
# lab: per-kernel times (rocprofv3 kernel trace) and HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of tools/lab/edge_time.py
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/edge_prof -o kt -- python3 $GRAFT_REPO_ROOT/tools/lab/edge_time.py > $OUT/edge_kt.log 2>&1
for CTR in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/edge_pmc_$CTR -o pmc -- python3 $GRAFT_REPO_ROOT/tools/lab/edge_time.py > $OUT/edge_pmc_$CTR.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/edge_prof/**/kt_kernel_trace.csv", recursive=True)[0])))
by = collections.defaultdict(list)
for r in rows:
    if "edge_" in r["Kernel_Name"]:
        by[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pm = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/edge_pmc_%s/**/pmc_counter_collection.csv" % ctr, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "edge_" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
            pm.setdefault(key, {}).setdefault(ctr, []).append(float(r["Counter_Value"]))
print("%-42s %-18s %5s %9s %10s %10s" % ("kernel", "grid", "calls", "avg us", "fetch MB", "write MB"))
for key, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    f = pm.get(key, {}).get("FETCH_SIZE"); w = pm.get(key, {}).get("WRITE_SIZE")
    # counters in KB per launch, FETCH doubled on gfx950 (MI355X_MICROARCH.md, HBM section; as tools/pmc_summary.py)
    print("%-42s %-18s %5d %9.1f %10s %10s" % (key[0], "x".join(key[1:]), len(v), sum(v) / len(v),
          "%.1f" % (2 * sum(f) / len(f) * 1024 / 1e6) if f else "-", "%.1f" % (sum(w) / len(w) * 1024 / 1e6) if w else "-"))
PY
