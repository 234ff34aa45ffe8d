"""The largest idle gaps on the busiest queue of a rocprofv3 kernel trace: which kernels sit on either side, and what the other
queues ran meanwhile.  usage: big_gaps.py kernel_trace.csv [n]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
per_q = collections.defaultdict(list)
for r in rows:
    per_q[r[3]].append(r)
q, rs = max(per_q.items(), key=lambda kv: len(kv[1]))
rs = rs[len(rs) // 2:]                      # steady state: the second half of the run
gaps = sorted(((rs[i + 1][0] - rs[i][1], i) for i in range(len(rs) - 1)), reverse=True)[:n]
for g, i in sorted(gaps, key=lambda t: t[1]):
    a, b = rs[i], rs[i + 1]
    others = [r for r in rows if r[3] != q and r[1] > a[1] and r[0] < b[0]]
    print("gap %8.1f us  after %-50s before %-50s | other queues: %d kernels %s" % (
        g / 1e3, a[2], b[2], len(others), sorted(set(o[2][:28] for o in others))[:3]))
