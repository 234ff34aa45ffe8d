"""Turn rocprofv3 --pmc CSV outputs (separate FETCH_SIZE / WRITE_SIZE passes, optional MFMA pass) of one
command into the per-kernel JSON summaries kept under profiles/.

    python tools/pmc_summary.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>"
    python tools/pmc_summary.py mfma <counter_collection.csv> <out.json> "<command>" <kernel substring>
    python tools/pmc_summary.py sq <out.json> "<command>" <kernel substring> <counter_collection.csv> [more csv ...]

traffic_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch: FETCH doubled per the gfx950 rule of
MI355X_MICROARCH.md (HBM section).  Warm-up launches are included in the averages (they move the same bytes).
"""
import csv
import json
import os
import sys
from collections import defaultdict


def per_kernel(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    if sys.argv[1] == "traffic":
        fetch, write, out, cmd = sys.argv[2:6]
        F, W = per_kernel(fetch), per_kernel(write)
        kernels = {}
        for k in F:
            if not k.startswith(("void geot", "geot")):
                continue
            f = sum(F[k]["FETCH_SIZE"]) / len(F[k]["FETCH_SIZE"])
            w = sum(W[k]["WRITE_SIZE"]) / len(W[k]["WRITE_SIZE"]) if k in W else 0.0
            kernels[short(k)] = {"fetch_kb": f, "write_kb": w, "traffic_bytes": int(round((2 * f + w) * 1024)),
                                 "launches": len(F[k]["FETCH_SIZE"])}
        import subprocess
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            commit = None
        commit = os.environ.get("GEOT_COMMIT", commit)
        # sha256 (16 hex digits) of every kernel source at the time of the profile: bench.py compares them with the tree it
        # runs from, so a figure whose kernel has changed since is flagged instead of silently quoted
        import glob
        import hashlib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        src = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
               for f in sorted(glob.glob(os.path.join(root, "geot_amd", "csrc", "*.h*")))}
        json.dump({"command": cmd, "commit": commit, "csrc_sha16": src, "unit": "KB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes); traffic_bytes = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) KB: FETCH doubled per the gfx950 rule in MI355X_MICROARCH.md (HBM section)",
                   "kernels": kernels}, open(out, "w"), indent=1)
    elif sys.argv[1] == "sq":
        # issue-side counters of the kernels matching a substring (one or more --pmc passes): what the occupied CUs do
        out, cmd, sub = sys.argv[2:5]
        kernels = {}
        for path in sys.argv[5:]:
            for k, ctrs in per_kernel(path).items():
                if sub in k:
                    rec = kernels.setdefault(short(k), {"launches": 0})
                    for n, v in ctrs.items():
                        rec[n] = sum(v) / len(v)
                        rec["launches"] = max(rec["launches"], len(v))
        for rec in kernels.values():
            if rec.get("SQ_WAVES") and rec.get("SQ_WAVE_CYCLES"):
                rec["cycles_per_wave"] = rec["SQ_WAVE_CYCLES"] / rec["SQ_WAVES"]
            if rec.get("SQ_INSTS_VALU") and rec.get("SQ_WAVE_CYCLES"):
                rec["valu_insts_per_wave_cycle"] = rec["SQ_INSTS_VALU"] / rec["SQ_WAVE_CYCLES"]
            if rec.get("SQ_ACTIVE_INST_VALU") and rec.get("SQ_BUSY_CU_CYCLES"):
                rec["valu_active_share_of_busy_cu_cycles"] = rec["SQ_ACTIVE_INST_VALU"] / rec["SQ_BUSY_CU_CYCLES"]
        json.dump({"command": cmd, "unit": "per-launch averages of rocprofv3 --pmc counters (summed over XCDs / SEs as rocprofv3 "
                   "reports them)", "kernels": kernels}, open(out, "w"), indent=1)
    else:
        path, out, cmd, sub = sys.argv[2:6]
        M = per_kernel(path)
        k = [x for x in M if sub in x][0]
        c = {n: sum(v) / len(v) for n, v in M[k].items()}
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8) * 1024) * 100
        json.dump({"command": cmd, "kernel": short(k), "counters": c, "MfmaUtil_percent": util,
                   "note": "rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / "
                           "((GRBM_GUI_ACTIVE / 8) * 1024 SIMDs)"}, open(out, "w"), indent=1)
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
