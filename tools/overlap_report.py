"""Concurrency report from a rocprofv3 --kernel-trace CSV: how much of the GPU timeline has 0 / 1 / >= 2 kernels in flight,
and which kernels run beside others:  python tools/overlap_report.py <kernel_trace.csv> [small_us=60]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    small_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    for ts, d in ev:
        hist[min(depth, 3)] += ts - last
        last, depth = ts, depth + d
    span = t1 - t0
    total = sum(e - s for s, e, _ in rows)
    small = [(s, e) for s, e, _ in rows if (e - s) < small_us * 1e3]
    print("kernels %d, span %.1f ms, sum of durations %.1f ms (%.3f of span)" % (len(rows), span / 1e6, total / 1e6, total / span))
    for k in sorted(hist):
        print("  %s kernels in flight: %8.1f ms (%.1f %%)" % (">=3" if k == 3 else str(k), hist[k] / 1e6, 100.0 * hist[k] / span))
    print("launches under %.0f us: %d (%.1f %% of launches), %.1f ms (%.1f %% of the sum of durations)" % (
        small_us, len(small), 100.0 * len(small) / len(rows), sum(e - s for s, e in small) / 1e6,
        100.0 * sum(e - s for s, e in small) / total))
    # time with ONLY small kernels in flight (what the side streams are meant to fill)
    ev = []
    for s, e, _ in rows:
        big = (e - s) >= small_us * 1e3
        ev.append((s, 1, big))
        ev.append((e, -1, big))
    ev.sort(key=lambda x: (x[0], x[1]))
    nb = ns = 0
    last, only_small = t0, 0
    for ts, d, big in ev:
        if nb == 0 and ns > 0:
            only_small += ts - last
        last = ts
        if big:
            nb += d
        else:
            ns += d
    print("timeline with only sub-%.0f-us kernels in flight: %.1f ms (%.1f %% of span)" % (small_us, only_small / 1e6, 100.0 * only_small / span))


if __name__ == "__main__":
    main()
