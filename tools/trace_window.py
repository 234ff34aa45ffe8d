#!/usr/bin/env python
"""Steady-state per-kernel summary of a rocprofv3 --kernel-trace CSV of bench.py.

rocprofv3 --stats aggregates the whole process, warm-up included (library auto-tuning, first-touch allocation).
The 8192-sample FPS launch opens every model step (it is queued on the side stream at the top of forward), so the
window [start of launch #skip+1, start of launch #skip+steps+1) of the kernel named by --anchor covers exactly
`steps` steady-state steps.  Prints / writes a CSV: kernel, calls per step, average us, total ms per step, share.

    python tools/trace_window.py kernel_trace.csv --anchor fps_pruned_kernel --min-us 1000 --skip 2 --steps 3 -o out.csv
"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="fps_pruned_kernel")
    ap.add_argument("--min-us", type=float, default=1000.0, help="anchor launches shorter than this are ignored")
    ap.add_argument("--skip", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--anchors-per-step", type=int, default=1,
                    help="anchor launches per step (the FixMatch iteration runs the 8192-sample FPS twice: teacher + student)")
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--per-launch", default=None, help="also write every launch of the first window step, in start order")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                         r.get("LDS_Block_Size", ""), r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    full = rows
    rows = [r[:3] for r in rows]
    anchors = [s for s, e, n in rows if a.anchor in n and (e - s) >= a.min_us * 1e3]
    k = a.anchors_per_step
    assert len(anchors) > (a.skip + a.steps) * k, "only %d anchor launches" % len(anchors)
    t0, t1 = anchors[a.skip * k], anchors[(a.skip + a.steps) * k]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        if t0 <= s < t1:
            agg[n][0] += 1
            agg[n][1] += e - s
    busy = sum(v[1] for v in agg.values())
    wall = (t1 - t0) / a.steps / 1e6
    out = [("kernel", "calls_per_step", "avg_us", "ms_per_step", "share_of_kernel_time")]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((n, "%.2f" % (c / a.steps), "%.1f" % (t / c / 1e3), "%.3f" % (t / a.steps / 1e6), "%.4f" % (t / busy)))
    print("# window: %d steps, wall %.3f ms/step, summed kernel time %.3f ms/step, %d launches/step" %
          (a.steps, wall, busy / a.steps / 1e6, sum(v[0] for v in agg.values()) / a.steps))
    for r in out[:a.top + 1]:
        print("%-100s %8s %10s %10s %8s" % (r[0][:100], r[1], r[2], r[3], r[4]))
    if a.per_launch:
        with open(a.per_launch, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(("start_us", "dur_us", "grid_x", "wg_x", "lds", "stream", "kernel"))
            for s, e, n, g, wg, lds, st in full:
                if t0 <= s < anchors[(a.skip + 1) * k]:
                    w.writerow(("%.1f" % ((s - t0) / 1e3), "%.1f" % ((e - s) / 1e3), g, wg, lds, st, n[:110]))
    if a.out:
        with open(a.out, "w", newline="") as f:
            f.write("# steady-state window of %d steps: wall %.3f ms/step, summed kernel time %.3f ms/step\n" %
                    (a.steps, wall, busy / a.steps / 1e6))
            csv.writer(f).writerows(out)


if __name__ == "__main__":
    main()
