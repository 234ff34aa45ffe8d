#!/usr/bin/env python
"""What is left outside the GEMMs and the hand-written kernels: reads the per-launch table of one steady-state step
(tools/trace_window.py --per-launch) and prints the time per class, the torch kernels by total time, and the largest
single torch launches with the launch in front of them (to find the op chain they belong to).

    python tools/glue_report.py profiles/r02_bench_model_v10_launches.csv [--top 20]
"""
import argparse
import collections
import csv


def short(n, w=110):
    n = n.replace("void ", "")
    for k in ("at::native::", "(anonymous namespace)::"):
        n = n.replace(k, "")
    return n[:w]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("launches")
    ap.add_argument("--top", type=int, default=20)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.launches)))
    tot, cnt = collections.Counter(), collections.Counter()
    for r in rows:
        n = r["kernel"]
        c = "gemm" if n.startswith("Cijk") else ("geot" if "geot::" in n else "glue")
        r["cls"] = c
        tot[c] += float(r["dur_us"])
        cnt[c] += 1
    print("%d launches: " % len(rows) + ", ".join("%s %.2f ms (%d)" % (c, tot[c] / 1e3, cnt[c]) for c in ("gemm", "geot", "glue")))
    g, gc = collections.Counter(), collections.Counter()
    for r in rows:
        if r["cls"] == "glue":
            g[short(r["kernel"])] += float(r["dur_us"])
            gc[short(r["kernel"])] += 1
    print("\ntorch kernels by total time:")
    for k, v in g.most_common(a.top):
        print("%8.0f us %4d  %s" % (v, gc[k], k))
    big = sorted(((float(r["dur_us"]), i) for i, r in enumerate(rows) if r["cls"] == "glue"), reverse=True)
    print("\nlargest single torch launches (and the launch before):")
    for d, i in big[:a.top]:
        print("%8.1f us  %-72s | after %s" % (d, short(rows[i]["kernel"], 72), short(rows[i - 1]["kernel"], 48)))


if __name__ == "__main__":
    main()
