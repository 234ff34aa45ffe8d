#!/usr/bin/env python
"""Where a step's wall time goes that is NOT kernel execution: from a rocprofv3 --kernel-trace CSV of bench.py, over the
steady-state window tools/trace_window.py uses (anchor = the 8192-sample FPS that opens every step):

* wall per step, time covered by at least one kernel (union over all queues), time with NO kernel running (idle);
* per hardware queue: launches, busy time;
* on the busiest queue: the gaps between one kernel's end and the next one's start (count, sum, median), and how much of
  its busy time is spent in kernels shorter than 10 / 20 us (latency-bound launches).

    python tools/trace_gaps.py kernel_trace.csv --skip 2 --steps 3 [--anchors-per-step 2]
"""
import argparse
import collections
import csv
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="fps_pruned_kernel")
    ap.add_argument("--min-us", type=float, default=1000.0)
    ap.add_argument("--skip", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--anchors-per-step", type=int, default=1)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    anchors = [s for s, e, n, q in rows if a.anchor in n and (e - s) >= a.min_us * 1e3]
    k = a.anchors_per_step
    assert len(anchors) > (a.skip + a.steps) * k, "only %d anchor launches" % len(anchors)
    t0, t1 = anchors[a.skip * k], anchors[(a.skip + a.steps) * k]
    win = [r for r in rows if t0 <= r[0] < t1]
    ms = lambda ns: ns / a.steps / 1e6          # noqa: E731
    # union of busy intervals
    covered, cur_s, cur_e = 0, None, None
    for s, e, _, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                covered += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    covered += (cur_e - cur_s) if cur_e is not None else 0
    wall = t1 - t0
    print("window: %d steps | wall %.3f ms/step | some kernel running %.3f ms/step | NO kernel running %.3f ms/step (%.1f %%)"
          % (a.steps, ms(wall), ms(covered), ms(wall - covered), 100.0 * (wall - covered) / wall))
    per_q = collections.defaultdict(list)
    for r in win:
        per_q[r[3]].append(r)
    print("%-8s %10s %12s" % ("queue", "launches", "busy ms/step"))
    for q, rs in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
        print("%-8s %10.1f %12.3f" % (q, len(rs) / a.steps, ms(sum(e - s for s, e, _, _ in rs))))
    q, rs = max(per_q.items(), key=lambda kv: len(kv[1]))
    gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
    durs = [e - s for s, e, _, _ in rs]
    print("busiest queue %s: %d gaps/step, sum %.3f ms/step, median %.2f us, p90 %.2f us, > 20 us: %d/step (%.3f ms/step)"
          % (q, len(gaps) // a.steps, ms(sum(gaps)), statistics.median(gaps) / 1e3, sorted(gaps)[int(0.9 * len(gaps))] / 1e3,
             sum(g > 20e3 for g in gaps) // a.steps, ms(sum(g for g in gaps if g > 20e3))))
    for lim in (5, 10, 20, 50):
        sel = [d for d in durs if d < lim * 1e3]
        print("  kernels shorter than %3d us: %6.1f per step, %.3f ms/step of its busy time" % (lim, len(sel) / a.steps, ms(sum(sel))))


if __name__ == "__main__":
    main()
