"""GPU idle time inside the steady state of a traced run: from a rocprofv3 --kernel-trace CSV, the share of the timeline with NO kernel in
flight, in buckets over the last seconds of the trace (the bench's timed GOPs sit there; the set-up phases in front of them idle for host
reasons and say nothing):  python tools/timeline_idle.py <kernel_trace.csv> [last_ms=4000] [bucket_ms=250]"""
import csv
import sys


def main():
    path = sys.argv[1]
    last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4000.0
    bucket_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    t1 = max(e for _, e in rows)
    t0 = t1 - int(last_ms * 1e6)
    ev = []
    for s, e in rows:
        if e <= t0:
            continue
        ev.append((max(s, t0), 1))
        ev.append((e, -1))
    ev.sort()
    nb = int(last_ms / bucket_ms)
    idle, one, launches, gaps = [0] * nb, [0] * nb, [0] * nb, [[] for _ in range(nb)]
    depth, last = 0, t0

    def add(a, b, d):
        while a < b:
            k = min(nb - 1, int((a - t0) / (bucket_ms * 1e6)))
            end = min(b, t0 + int((k + 1) * bucket_ms * 1e6))
            if d == 0:
                idle[k] += end - a
            elif d == 1:
                one[k] += end - a
            a = end

    for ts, d in ev:
        if depth == 0 and ts > last:
            gaps[min(nb - 1, int((last - t0) / (bucket_ms * 1e6)))].append(ts - last)
        add(last, ts, depth)
        last, depth = ts, depth + d
        if d == 1:
            launches[min(nb - 1, int((ts - t0) / (bucket_ms * 1e6)))] += 1
    print("last %.0f ms of the trace in buckets of %.0f ms: share of the timeline with 0 kernels / exactly 1 kernel in flight, launches, idle gaps (count, median us)" % (last_ms, bucket_ms))
    for k in range(nb):
        g = sorted(gaps[k])
        print("  -%5.0f ms: idle %5.1f %%   one kernel %5.1f %%   %5d launches   %5d gaps, median %.1f us, sum of gaps > 20 us: %.1f ms" % (
            last_ms - k * bucket_ms, 100.0 * idle[k] / (bucket_ms * 1e6), 100.0 * one[k] / (bucket_ms * 1e6), launches[k], len(g),
            (g[len(g) // 2] / 1e3) if g else 0.0, sum(x for x in g if x > 20e3) / 1e6))


if __name__ == "__main__":
    main()
