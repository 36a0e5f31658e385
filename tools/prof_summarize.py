"""Summarise rocprofv3 output directories of ONE bench command into the tracked files under profiles/ (round tag given):
    python tools/prof_summarize.py r02 --stats DIR --fetch DIR --write DIR --sq DIR --cmd "python3 bench.py ..."
  profiles/<tag>_bench_kernel_stats.csv  per-kernel launches / total / average duration (from --kernel-trace)
  profiles/<tag>_pmc_traffic.json        HBM bytes per launch per kernel: FETCH_SIZE and WRITE_SIZE from SEPARATE --pmc passes,
                                         both in KB; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md)
  profiles/<tag>_mfma_busy.json          per kernel: matrix-pipe busy fraction and the shader clock it implies, from
                                         SQ_VALU_MFMA_BUSY_CYCLES (busy cycles summed over all 1024 SIMDs), SQ_BUSY_CYCLES
                                         (summed over the 32 shader engines) and GRBM_GUI_ACTIVE (summed over the 8 XCDs)
Each DIR is searched recursively for *_kernel_trace.csv / *_counter_collection.csv."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_SE, N_XCD = 1024, 32, 8


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("lssvc::", "")


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no %s under %s" % (suffix, d))
    return hits[0]


def kernel_durations(d):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(find(d, "_kernel_trace.csv"))):
        a = acc.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc


def counters(d, names):
    """{kernel: {counter: [sum, n]}} plus per-kernel summed duration of the same dispatches."""
    acc, dur = {}, collections.defaultdict(lambda: [0, 0])
    seen = set()
    for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
        if r["Counter_Name"] not in names:
            continue
        k = short(r["Kernel_Name"])
        c = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"])
        c[1] += 1
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur[k][1] += 1
    return acc, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--cmd", default="")
    ap.add_argument("--only", default="", help="keep kernels whose name contains this (default: all lssvc kernels)")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    if args.stats:
        d = kernel_durations(args.stats)
        tot = sum(v[1] for v in d.values())
        with open(os.path.join(out_dir, "%s_bench_kernel_stats.csv" % args.tag), "w") as f:
            f.write("# %s\n" % args.cmd)
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
            for k, (n, ns) in sorted(d.items(), key=lambda kv: -kv[1][1]):
                f.write('"%s",%d,%d,%.1f,%.2f\n' % (k, n, ns, ns / n, 100.0 * ns / tot))
        print("kernel stats: %d kernels, %.1f ms" % (len(d), tot / 1e6))
    if args.fetch and args.write:
        fe, _ = counters(args.fetch, {"FETCH_SIZE"})
        wr, _ = counters(args.write, {"WRITE_SIZE"})
        recs = []
        for k, c in fe.items():
            if "conv" not in k and "ffn" not in k and "dwpre" not in k:
                continue
            f_sum, n = c["FETCH_SIZE"]
            w_sum, wn = wr.get(k, {}).get("WRITE_SIZE", [0.0, 0])
            f_kb, w_kb = f_sum / n, (w_sum / wn if wn else 0.0)
            recs.append({"kernel": k, "launches": n, "fetch_kb_per_launch_raw": f_kb, "write_kb_per_launch": w_kb,
                         "hbm_bytes_per_launch_corrected": (2.0 * f_kb + w_kb) * 1024.0,
                         "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                                 "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); counters are in KB",
                         "command": args.cmd})
        recs.sort(key=lambda r: -r["hbm_bytes_per_launch_corrected"] * r["launches"])
        json.dump(recs[:30], open(os.path.join(out_dir, "%s_pmc_traffic.json" % args.tag), "w"), indent=1)
        print("traffic: %d kernels" % len(recs))
    if args.sq:
        names = {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"}
        sq, dur = counters(args.sq, names)
        recs = []
        for k, c in sq.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"][0] <= 0:
                continue
            n = c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
            mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n
            busy = c.get("SQ_BUSY_CYCLES", [0.0, 1])[0] / n
            grbm = c.get("GRBM_GUI_ACTIVE", [0.0, 1])[0] / n
            ns = dur[k][0] / max(dur[k][1], 1)
            rec = {"kernel": k, "launches": n, "avg_duration_us": ns / 1e3,
                   "mfma_busy_cycles_per_launch": mfma, "sq_busy_cycles_per_launch": busy, "grbm_gui_active_per_launch": grbm}
            if busy > 0:
                rec["mfma_busy_frac_of_sq_busy"] = mfma / (N_SIMD * busy / N_SE)          # per-SIMD busy / per-SE busy cycles
                rec["clock_ghz_from_sq_busy"] = busy / N_SE / ns
            if grbm > 0:
                rec["clock_ghz_from_grbm"] = grbm / N_XCD / ns
                rec["mfma_busy_frac_of_wall"] = mfma / (N_SIMD * grbm / N_XCD)
            recs.append(rec)
        recs.sort(key=lambda r: -r["mfma_busy_cycles_per_launch"] * r["launches"])
        json.dump({"command": args.cmd,
                   "how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace "
                          "(one pass, program directly after --); MFMA busy cycles are summed over the 1024 SIMDs, SQ_BUSY_CYCLES over "
                          "the 32 shader engines, GRBM_GUI_ACTIVE over the 8 XCDs; durations are those of the profiled pass",
                   "kernels": recs[:24]}, open(os.path.join(out_dir, "%s_mfma_busy.json" % args.tag), "w"), indent=1)
        print("mfma busy: %d kernels" % len(recs))


if __name__ == "__main__":
    main()
