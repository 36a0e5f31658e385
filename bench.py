#!/usr/bin/env python3
"""Benchmark of the LSSVC hot path on MI355X: encoded frames/s on BASELINE.json configs[1]
(two-layer x2, EL 1080p padded to 1152x1920, BL 576x960, 32-frame GOP = 1 I + 31 P, estimate mode).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one GOP (32 frames) per rank. GOPs are independent (each restarts from an I-frame), so
ranks shard GOPs with no data-path collective ("weak" scaling: one GOP per GPU per step); the only
communication is the barrier / max-reduce of the timing and the gather of the per-rank records.

Round 6. The workload IS the reference fixture's: the clip, the weights and the frame loop of tests/golden/x2_1080p_gop32.npz (the
reference itself run on configs[1]'s GOP: synth_clip_exact seed 4, synthetic weights seed 4 / gain 0.55), so the line carries a
`parity` record -- the timed GOPs' bit counts against the reference's stored ones, frame by frame -- beside the throughput: same
inputs for both. `value` is measured as BASELINE.md section 3 defines the GPU side: per frame INSIDE the clock the H2D of the 8-bit
frame from pinned host memory, u8 -> fp32, zero padding, the bicubic base layer, the encode and the D2H of the bit counts; the upload
and pre-processing of frame t+2 run on a copy stream beside frame t. The loop with the inputs resident in HBM (the headline of rounds
1-5) is timed beside it as `resident`. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (lssvc_amd/__init__.py: before the HIP runtime initialises; +0.6 ... 1.8 %, profiles/r06_hw_queues_ab.txt)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOP = 32
HEIGHT, WIDTH, RATIO = 1080, 1920, 2.0
GAIN = 0.55                       # synthetic-weight gain at which a 32-frame GOP stays numerically stable
FIXTURE = "x2_1080p_gop32"        # tests/golden/<FIXTURE>.npz: the reference's own run of this workload (make_golden_full.py)
FIXTURE_SEED = 4                  # its clip (synth_clip_exact) and weights (synth_state_dict) seed; gain = GAIN
PEAK_FP16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured achievable)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 = 64 FLOP/clk/SIMD


DTYPE = {"f32": "f32", "f16x3": "f16x3 (convolutions: fp16 MFMA on hi/lo-split operands, fp32 accumulate, fp32-class results; GDN and pointwise/entropy kernels f32)"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(device, seed, frames, exact=False, hw=None, ratio=None):
    """-> (BL frames, EL frames, padding info, the 8-bit host clip in pinned memory). exact: the integer-exact clip of the reference
    fixtures (synth_clip_exact) instead of the float-noise one; hw / ratio: another picture size / scale factor than the headline's."""
    from lssvc_amd.synth import synth_clip, synth_clip_exact
    from lssvc_amd.prepost import FramePrep
    height, width = hw if hw is not None else (HEIGHT, WIDTH)
    clip = (synth_clip_exact if exact else synth_clip)(frames, height, width, seed=seed).pin_memory()
    prep = FramePrep(device)
    x_els, x_bls = [], []
    for t in range(frames):
        x_bl, x_el, pad = prep.make_layers_rgb8(clip[t].to(device), RATIO if ratio is None else ratio)
        x_els.append(x_el)
        x_bls.append(x_bl)
    return x_bls, x_els, pad, clip


class HostFrames:
    """Frame source of the headline loop: frame t is copied up from pinned host memory (8-bit RGB, 6.2 MB), converted to fp32, zero-
    padded and resampled to the base layer on the device EVERY time it is asked for (test.py:185-199; csrc/prepost.hip). Round 6: that
    work runs on a COPY STREAM up to `depth` frames ahead of the frame being coded -- the caller's stream waits for the frame's event,
    not for the upload -- so the per-frame H2D + pre-processing is inside the clock and off the critical path (it cost 0.8-3 % of the
    frame rate when it sat in front of every frame on the coding stream)."""

    def __init__(self, clip_u8, device, depth=2, side_stream=True):
        from lssvc_amd.prepost import FramePrep
        self.clip, self.device = clip_u8, device
        self.prep = FramePrep(device)                    # csrc/prepost.hip: u8 -> fp32 + padding, bicubic base layer
        self.depth = depth
        self.stream = torch.cuda.Stream(device=device) if side_stream else None
        self.ready = {}                                  # frame index -> (x_bl, x_el, event)
        self.uploads = 0

    def __len__(self):
        return self.clip.shape[0]

    def _issue(self, t):
        main = torch.cuda.current_stream(self.device)
        if self.stream is None:
            x_bl, x_el, _ = self.prep.make_layers_rgb8(self.clip[t].to(self.device, non_blocking=True), RATIO)
            self.ready[t] = (x_bl, x_el, None)
        else:
            with torch.cuda.stream(self.stream):
                u8 = self.clip[t].to(self.device, non_blocking=True)
                x_bl, x_el, _ = self.prep.make_layers_rgb8(u8, RATIO)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            for x in (u8, x_bl, x_el):
                x.record_stream(main)                    # allocated on the copy stream, read on the coding stream
            self.ready[t] = (x_bl, x_el, ev)
        self.uploads += 1

    def prefetch(self, t):
        """Make sure frames t .. t + depth (cyclically: the next GOP starts at frame 0 again) are uploaded or on their way."""
        n = len(self)
        for d in range(self.depth + 1):
            k = (t + d) % n
            if k not in self.ready:
                self._issue(k)

    def layers(self, t, keep=False):
        """(x_bl, x_el) of frame t on the current stream. keep: leave it in the table (the look-ahead reads frame t+1's base layer a
        frame early; the frame itself is taken, and dropped, by the next call)."""
        if t not in self.ready:
            self._issue(t)
        x_bl, x_el, ev = self.ready[t] if keep else self.ready.pop(t)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return x_bl, x_el


EVENT_FRAMES = 8      # per-launch HIP events are sampled on the last 8 P-frames of the last timed GOP only: the event markers cost
#                       ~2.5 % of stream time when put around every launch, and 8 P-frames (31 of a GOP's 32 frames are
#                       P-frames, so this is the GOP's launch mix) already hold >650 launches of the dominant kernel


LOOKAHEAD = True      # P-frames name the next frame's base-layer input: BL(t+1) is coded beside EL(t) (LSSVC_extend.forward_one_frame)


def encode_gop(inet, pnet, x_bls, x_els, shape_hr, op_log=None, host_frames=None, lookahead=None):
    """test.py's frame loop (test.py:182-250) for one GOP: I-frame, then P-frames chained through the DPB.
    op_log: list that receives the per-launch records (with HIP events) of the last EVENT_FRAMES P-frames.
    host_frames: a HostFrames -- take every frame from host memory instead of the resident x_bls / x_els.
    lookahead (default: LOOKAHEAD): the loop knows the next frame, so it hands its base-layer input to the P-frame call."""
    from lssvc_amd import hip_ops
    lookahead = LOOKAHEAD if lookahead is None else lookahead
    bits = []
    dpb = None
    n = len(host_frames) if host_frames is not None else len(x_els)
    graph, streams = pnet.graph_mode, hip_ops.MULTI_STREAM
    for t in range(n):
        first_logged = max(1, n - EVENT_FRAMES)           # the LAST EVENT_FRAMES P-frames: behind them nothing needs a plan the
        logging = op_log is not None and t >= first_logged      # look-ahead GOPs never use (the whole-frame steady-P plan)
        hip_ops.OP_LOG = op_log if logging else None
        pnet.graph_mode = graph and not logging        # per-launch events need the eager path for these frames ...
        hip_ops.MULTI_STREAM = streams and not logging # ... and one stream: a launch timed beside another stream's kernels
        #                                                measures the contention, not the kernel
        inet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
        pnet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
        if host_frames is not None:
            host_frames.prefetch(t)                         # frames t .. t+2 are uploaded / being uploaded on the copy stream
            x_bl, x_el = host_frames.layers(t)
            next_bl = host_frames.layers(t + 1, keep=True)[0] if (lookahead and t >= 1 and t + 1 < n) else None
        else:
            x_bl, x_el = x_bls[t], x_els[t]
            next_bl = x_bls[t + 1] if t + 1 < n else None
        if op_log is not None and t + 1 >= max(1, n - EVENT_FRAMES):
            next_bl = None                                  # the next frame is coded eagerly with events: no base layer ahead of it
        if t == 0:
            r = inet.encode_decode(x_bl, x_el, None, None)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None,
                   "ref_feature_el": r["feature_el"]}
        else:
            r = pnet.encode_decode(x_bl, x_el, dpb, **(dict(next_x_bl=next_bl, frame_id=t) if lookahead else {}))
            dpb = r["dpb"]
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        bits.append((r["bit_bl"], r["bit_el"]))
    hip_ops.OP_LOG = None
    pnet.graph_mode = graph
    hip_ops.MULTI_STREAM = streams
    return bits, dpb


def roofline_from_log(op_log):
    """Group the conv launches of one GOP by the kernel instantiation that ran (name reported by the library);
    dominant = largest total time. MFMA-bound kernels (3x3 / 7x7 ...) are priced in algorithmic TFLOP/s, the
    1x1 kernels -- no spatial reuse, a few dozen FLOP per byte -- against HBM bandwidth with their algorithmic
    bytes (input + output + residual read, fp32)."""
    groups = {}
    for e in op_log:
        ms = e["events"][0].elapsed_time(e["events"][1])
        g = groups.setdefault(e["kernel"], {"ms": 0.0, "macs": 0, "bytes": 0, "launches": 0, "ks": e["ks"]})
        g["ms"] += ms
        g["macs"] += e["macs"]
        g["bytes"] += e["bytes"]
        g["launches"] += 1
    table = []
    for v, g in groups.items():
        table.append({"kernel": v, "launches": g["launches"], "ks": g["ks"],
                      "total_ms": round(g["ms"], 3), "avg_us": round(1e3 * g["ms"] / g["launches"], 2),
                      "gflop_per_launch": round(2e-9 * g["macs"] / g["launches"], 3),
                      "mbytes_per_launch": round(1e-6 * g["bytes"] / g["launches"], 2),
                      "tflops": round(2e-9 * g["macs"] / g["ms"], 2) if g["ms"] > 0 else 0.0,
                      "gbps": round(1e-6 * g["bytes"] / g["ms"], 1) if g["ms"] > 0 else 0.0})
    table.sort(key=lambda r: -r["total_ms"])
    if os.environ.get("LSSVC_BENCH_SIGNATURES"):      # per-signature breakdown for kernel work (not part of the JSON line)
        sig = {}
        for e in op_log:
            k = (e["kind"], e["cin"], e["cout"], e["hout"], e["wout"], e["kernel"])
            g = sig.setdefault(k, [0.0, 0, 0, 0])
            g[0] += e["events"][0].elapsed_time(e["events"][1])
            g[1] += e["macs"]
            g[2] += 1
            g[3] += e["bytes"]
        with open(os.environ["LSSVC_BENCH_SIGNATURES"], "w") as f:
            tot = sum(g[0] for g in sig.values())
            for k, g in sorted(sig.items(), key=lambda kv: -kv[1][0]):
                f.write("%-10s cin %4d cout %4d @%4dx%-4d %-40s n=%4d  %8.2f ms (%4.1f%%)  %7.1f us  %6.1f TF  %5.2f TB/s\n" % (
                    k[0], k[1], k[2], k[3], k[4], k[5], g[2], g[0], 100 * g[0] / tot, 1e3 * g[0] / g[2], 2e-9 * g[1] / g[0], 1e-9 * g[3] / g[0]))
    dom = table[0]
    common = {"kernel": dom["kernel"], "launches": dom["launches"], "avg_launch_us": dom["avg_us"],
              "gflop_per_launch": dom["gflop_per_launch"], "mbytes_per_launch": dom["mbytes_per_launch"],
              "traffic": pmc_traffic(dom["kernel"]), "mfma_busy": pmc_mfma_busy(dom["kernel"]), "clock": stamped_clock(dom["kernel"]), "chip_sustains": clock_probe(),
              "sampled_frames": EVENT_FRAMES,
              "conv_time_ms_sampled": round(sum(r["total_ms"] for r in table), 2),
              "conv_tflop_sampled": round(sum(r["gflop_per_launch"] * r["launches"] for r in table) * 1e-3, 3)}
    if dom["ks"] == 1:
        roof = {"bound": "hbm", "achieved": dom["gbps"], "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(dom["gbps"] / PEAK_HBM_GBPS, 4)}
    else:
        # `frac` = ALGORITHMIC flops / time / the guide's dense peak of the pipe the kernel runs on: 2.5 PFLOP/s fp16 for the
        # f16x3 kernels (which ISSUE 3 fp16-MFMA flops per algorithmic flop: hi*hi + hi*lo + lo*hi, so their `frac` cannot
        # exceed 1/3; the issued-flop utilisation is printed beside it), 157.3 TFLOP/s for the exact-fp32 kernel
        f16 = "f16x3" in dom["kernel"]
        peak = PEAK_FP16_MFMA_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": dom["tflops"], "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(dom["tflops"] / peak, 4)}
        if f16:
            roof["peak_note"] = ("peak = dense fp16 MFMA (MI355X_MICROARCH.md); achieved = algorithmic (fp32-class) flops; the f16x3 "
                                 "kernels issue 3 fp16-MFMA flops per algorithmic flop, so frac <= 0.333 by construction")
            roof["frac_issued_fp16"] = round(3.0 * dom["tflops"] / PEAK_FP16_MFMA_TFLOPS, 4)
    roof.update(common)
    for r in table:
        r.pop("ks")
    return roof, table


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present (the PMC passes cannot run inside bench.py: rocprofv3 wraps it)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    legacy = os.path.join(ROOT, "profiles", suffix)
    return hits[-1] if hits else (legacy if os.path.exists(legacy) else None)


def _strip_tmpl(name):
    return name.replace("lssvc::", "").replace(" ", "")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 FETCH correction applied) -- bench.py cannot run the profiler on itself."""
    path = _latest_profile("pmc_traffic.json")
    try:
        with open(path) as f:
            for rec in json.load(f):
                if _strip_tmpl(rec["kernel"]).startswith(_strip_tmpl(kernel).rstrip(">")):
                    return {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch_corrected"], "kernel_in_profile": rec["kernel"],
                            "source": os.path.relpath(path, ROOT), "collected_on": rec.get("command")}
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def stamped_clock(kernel):
    """The shader clock the chip held inside `kernel` (delta s_memtime / delta s_memrealtime of its diagnostic stamp build after 200
    back-to-back launches, tools/p3_stamps.py --json; DESIGN section 8: the dominant conv kernel runs at 1.56-1.69 GHz, not 2.4) and the
    MFMA-issue share of its consumer waves' cycles -- committed per round, like the PMC passes."""
    path = _latest_profile("p3_stamps.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        for rec in doc["kernels"]:
            if _strip_tmpl(kernel).startswith(_strip_tmpl(rec["kernel"]).rstrip(">")):
                out = dict(rec)
                out.update(source=os.path.relpath(path, ROOT), nominal_ghz=2.4, collected_on=doc.get("command"))
                return out
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def clock_probe():
    """What the chip sustains for a bare v_mfma_f32_16x16x32_f16 loop (tools/probes/clock_probe.hip, committed per round): issued TFLOP/s and
    in-kernel clock on zeros, on random operands, and on random operands re-read from LDS at the conv kernel's ratio -- the ceiling the
    2.5 PFLOP/s `peak` stands for on real data."""
    path = _latest_profile("clock_probe.txt")
    try:
        out = {}
        with open(path) as f:
            for ln in f:
                t = ln.split()
                if len(t) > 8 and t[2] == "issued":
                    out[t[0]] = {"issued_tflops": float(t[1]), "clock_mhz": float(t[t.index("clock") + 1]), "mfma_duty": float(t[t.index("duty") + 1])}
        if out:
            out["source"] = os.path.relpath(path, ROOT)
            return out
    except (OSError, ValueError, IndexError):
        pass
    return None


def pmc_mfma_busy(kernel):
    """Matrix-pipe busy fraction and implied shader clock of `kernel` from the committed SQ counter pass."""
    path = _latest_profile("mfma_busy.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        for rec in doc["kernels"]:
            if _strip_tmpl(rec["kernel"]).startswith(_strip_tmpl(kernel).rstrip(">")):
                out = {k: rec[k] for k in ("mfma_busy_frac_of_wall", "clock_ghz_from_grbm", "mfma_busy_frac_of_sq_busy",
                                           "clock_ghz_from_sq_busy") if k in rec}
                out.update(kernel_in_profile=rec["kernel"], source=os.path.relpath(path, ROOT), collected_on=doc.get("command"))
                return out
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None


def _oracle_frames(H, W, threads, n_p=1, warm=None):
    """Seconds the CPU oracle takes for one I-frame and the first n_p P-frames (the second is a steady-state P-frame: it
    runs the 48-channel feature adaptors) at EL HxW / BL (H/2)x(W/2): (t_i, t_p of the LAST P-frame timed, [all t_p]).
    warm (round 6): {"dpb": the DPB after frame 1 as the GPU path produced it (CPU tensors), "x_bl", "x_el": frames 0..2 as (1,3,h,w)
    CPU tensors, "seed"} -- the I-frame is timed on frame 0 and ONE steady-state P-frame on frame 2 from that DPB, so the sample holds a
    steady P-frame without paying for the first one (62 s of CPU)."""
    from lssvc_oracle.intra import intra_forward
    from lssvc_oracle.inter import inter_forward
    from lssvc_amd.synth import synth_state_dict, synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    torch.set_num_threads(threads)
    seed = warm["seed"] if warm is not None else 0
    sd_i, sd_p = synth_state_dict("intra_ss", seed, GAIN), synth_state_dict("lssvc_extend", seed, GAIN)
    if warm is not None:
        with torch.no_grad():
            t0 = time.time()
            o = intra_forward(sd_i, warm["x_bl"][0], warm["x_el"][0], (H, W))
            t_i = time.time() - t0
            del o
            t0 = time.time()
            r = inter_forward(sd_p, warm["x_bl"][2], warm["x_el"][2], warm["dpb"], (H, W), RATIO)
            t_p = time.time() - t0
            del r
        return t_i, t_p, [t_p]
    clip = synth_clip(1 + n_p, H, W, seed=0).float() / 255.0
    x_bl = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1)
    with torch.no_grad():
        t0 = time.time()
        o = intra_forward(sd_i, x_bl[0:1], clip[0:1], (H, W))
        t_i = time.time() - t0
        dpb = {"ref_frame_bl": o["x_hat_bl"].clamp_(0, 1), "ref_frame_el": o["x_hat_el"].clamp_(0, 1),
               "ref_feature_bl": None, "ref_feature_el": o["feature_el"]}
        del o
        t_ps = []
        for t in range(1, 1 + n_p):
            t0 = time.time()
            r = inter_forward(sd_p, x_bl[t:t + 1], clip[t:t + 1], dpb, (H, W), RATIO)
            t_ps.append(time.time() - t0)
            dpb = r["dpb"]
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
            del r
    return t_i, t_ps[-1], t_ps


def config0_latency(device, graph=True, reps=30):
    """BASELINE configs[0] (the reference's own CPU-runnable case): IntraSS, ONE 256x256 frame, x2 (BL 128x128), estimate
    mode. Launch-bound on a GPU, so what matters is the per-frame latency of the hipGraph frame plan: median wall time of
    encode_decode() + the D2H of its bit counts over `reps` frames, and the host time to issue one frame."""
    from lssvc_amd import IntraSS
    from lssvc_amd.prepost import FramePrep
    from lssvc_amd.synth import synth_clip, synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, GAIN)).to(device).eval()
    inet.set_graph_mode(graph, alias_outputs=True)
    prep = FramePrep(device)
    x_bl, x_el, pad = prep.make_layers_rgb8(synth_clip(1, 256, 256, seed=1)[0].to(device), 2.0)
    inet.set_scale_information(2.0, pad["HR_padded_size"], (0, 0, 0, 0))
    lat, issue = [], []
    with torch.no_grad():
        for i in range(reps + 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = inet.encode_decode(x_bl, x_el, None, None)
            float(r["bit_bl"]), float(r["bit_el"])
            torch.cuda.synchronize()
            if i >= 5:
                lat.append(time.perf_counter() - t0)
                issue.append(getattr(inet, "last_issue_s", 0.0) or 0.0)
    lat.sort()
    issue.sort()
    med = lat[len(lat) // 2]
    return {"workload": "configs[0]: IntraSS, one 256x256 frame, x2 (BL 128x128), estimate mode", "latency_ms": round(1e3 * med, 3),
            "frames_per_s": round(1.0 / med, 1), "host_issue_ms": round(1e3 * issue[len(issue) // 2], 3),
            "launch": "hipGraph frame plan" if graph else "eager", "reps": reps}


def _cached_cpu_baseline():
    """The full-size run (1 I + 2 P at 1152x1920, `python bench.py --cpu-baseline-full`, several minutes of CPU work) is made
    once per round on a GPU box and kept under profiles/ with the host description; the default run quotes it beside its
    own bounded sample."""
    path = _latest_profile("cpu_baseline_full.json")
    try:
        with open(path) as f:
            d = json.load(f)
        d["source"] = os.path.relpath(path, ROOT)
        return d
    except (OSError, ValueError, TypeError):
        return None


def config4_stream(device, frames=8):
    """BASELINE configs[4] at configs[1]'s size: write_stream=1 on 1 I + (frames - 1) P frames at 1152x1920 / 576x960, real
    rANS strings written to and read back from files, exactly what the reference times as `encoding_time` /
    `decoding_time` per P-frame (LSSVC_net_extend.py:158-171, dmc_net_extend.py:150-163; its published 1.441 s / 1.348 s,
    json_results/LSSVC/IP32/x2_FL.json, are on unstated hardware). Pass 1 gives those two numbers undisturbed; pass 2
    repeats the clip with the profiler on (it drains the device before every copy, so its total is larger) and splits a
    P-frame into GPU work waited for, D2H / H2D of the int16 planes, host rANS, file I/O."""
    import shutil
    import tempfile
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    from lssvc_amd.synth import synth_state_dict
    inet, pnet, t_update = _stream_models(device)
    x_bls, x_els, pad, _ = build_inputs(device, seed=3, frames=frames)
    shape_hr = pad["HR_padded_size"]
    tmp = tempfile.mkdtemp(prefix="lssvc_bench_")

    def run():
        rows, dpb = [], None
        for t in range(frames):
            inet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
            pnet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
            pb, pe = os.path.join(tmp, "bl_%d.bin" % t), os.path.join(tmp, "el_%d.bin" % t)
            if t == 0:
                r = inet.encode_decode(x_bls[t], x_els[t], pb, pe, shape_hr[0] // 2, shape_hr[1] // 2, shape_hr[0], shape_hr[1])
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bls[t], x_els[t], dpb, pb, pe)
                dpb = r["dpb"]
                rows.append(r)
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
        return rows

    try:
        with torch.no_grad():
            run()                                       # warm-up: weight layouts, LDS grants, allocator
            rows = run()
            n = len(rows)
            enc = sum(r["encoding_time_BL"] + r["encoding_time_EL"] for r in rows) / n
            dec = sum(r["decoding_time_BL"] + r["decoding_time_EL"] for r in rows) / n
            bits = sum(r["bit_bl"] + r["bit_el"] for r in rows) / n
            est = sum(r["bit_bl_estimate"] + r["bit_el_estimate"] for r in rows) / n
            prof = hip_ops.STREAM_PROF = {}
            try:
                rows2 = run()
            finally:
                hip_ops.STREAM_PROF = None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    n_all = frames                                       # the profiled pass includes the I-frame's (smaller) share
    enc2 = sum(r["encoding_time_BL"] + r["encoding_time_EL"] for r in rows2) / n
    dec2 = sum(r["decoding_time_BL"] + r["decoding_time_EL"] for r in rows2) / n
    per = lambda k: round(1e3 * prof.get(k, 0.0) / n_all, 2)
    return {"workload": "configs[4] at configs[1]'s size: write_stream=1, 1 I + %d P at EL 1152x1920 / BL 576x960, real rANS files" % (frames - 1),
            "encoding_time_s_per_p_frame": round(enc, 4), "decoding_time_s_per_p_frame": round(dec, 4),
            "reference_published": {"encoding_time": 1.441, "decoding_time": 1.348, "hardware": "unstated CUDA GPU",
                                    "source": "json_results/LSSVC/IP32/x2_FL.json (HEVC_B mean)"},
            "stream_bits_per_p_frame": round(bits, 1), "estimated_bits_per_p_frame": round(est, 1),
            "profiled_pass_ms_per_frame": {"gpu_work_waited_for": per("gpu_wait_s"), "d2h_int16_planes": per("d2h_s"),
                                           "h2d_int16_planes": per("h2d_s"), "host_rans_encode": per("rans_enc_s"),
                                           "host_rans_decode": per("rans_dec_s"), "file_io": per("io_s"),
                                           "encode_plus_decode_wall_p_frame": round(1e3 * (enc2 + dec2), 2)},
            "d2h_mb_per_frame": round(1e-6 * prof.get("d2h_bytes", 0) / n_all, 2), "h2d_mb_per_frame": round(1e-6 * prof.get("h2d_bytes", 0) / n_all, 2),
            "host_rans_msymbols_per_s": {"encode": round(1e-6 * prof.get("enc_symbols", 0) / max(prof.get("rans_enc_s", 0.0), 1e-9), 1),
                                         "decode": round(1e-6 * prof.get("dec_symbols", 0) / max(prof.get("rans_dec_s", 0.0), 1e-9), 1)},
            "symbols_per_frame": int(prof.get("enc_symbols", 0) / n_all), "cdf_table_build_s": round(t_update, 2)}


_STREAM_MODELS = {}


def _stream_models(device):
    """One pair of models with their CDF tables built (update(force=True): what write_stream=1 needs), shared by the side measurements."""
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict
    if "m" not in _STREAM_MODELS:
        inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, GAIN)).to(device).eval()
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", 0, GAIN))
        pnet.to(device).eval()
        t0 = time.time()
        inet.update(force=True)
        pnet.update(force=True)
        _STREAM_MODELS["m"] = (inet, pnet, time.time() - t0)
    return _STREAM_MODELS["m"]


def published_points(device, frames=4, est_frames=8):
    """The reference's OTHER published operating points (json_results/LSSVC/IP32/x1_5_FL.json, x2_FL.json -- encode / decode
    seconds per P-frame with write_stream=1, on unstated hardware), measured here at the same picture sizes on synthetic clips: real rANS
    files written and read back (1 I + 3 P, second pass timed), and the estimate-mode frame rate over 8 frames (eager launches). Context
    beside `config4_stream` (1080p x2), not a headline."""
    import shutil
    import tempfile
    inet, pnet, _ = _stream_models(device)
    points = (("1080p x1.5 (HEVC-B class; BL 720p)", 1080, 1920, 1.5, 1.541, 1.460, "x1_5_FL.json"),
              ("720p x2 (HEVC-E class)", 720, 1280, 2.0, 0.657, 0.639, "x2_FL.json"),
              ("480p x2 (HEVC-C class)", 480, 832, 2.0, 0.353, 0.343, "x2_FL.json"))
    out = []
    for name, ph, pw, ratio, ref_enc, ref_dec, src in points:
        x_bls, x_els, pad, _ = build_inputs(device, seed=7, frames=max(frames, est_frames), hw=(ph, pw), ratio=ratio)
        (H, W), (h, w) = pad["HR_padded_size"], pad["LR_padded_size"]
        tmp = tempfile.mkdtemp(prefix="lssvc_bench_")

        def run(n, stream):
            rows, dpb = [], None
            for t in range(n):
                inet.set_scale_information(ratio, (H, W), (0, 0, 0, 0))
                pnet.set_scale_information(ratio, (H, W), (0, 0, 0, 0))
                pb, pe = (os.path.join(tmp, "bl_%d.bin" % t), os.path.join(tmp, "el_%d.bin" % t)) if stream else (None, None)
                if t == 0:
                    r = inet.encode_decode(x_bls[t], x_els[t], pb, pe, h, w, H, W)
                    dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
                else:
                    r = pnet.encode_decode(x_bls[t], x_els[t], dpb, pb, pe)
                    dpb = r["dpb"]
                    rows.append(r)
                dpb["ref_frame_bl"].clamp_(0, 1)
                dpb["ref_frame_el"].clamp_(0, 1)
            return rows

        try:
            with torch.no_grad():
                run(frames, True)                           # warm-up: weight layouts, LDS grants, allocator at this size
                rows = run(frames, True)
                run(est_frames, False)
                torch.cuda.synchronize()
                t0 = time.time()
                est_rows = run(est_frames, False)
                torch.cuda.synchronize()
                dt = time.time() - t0
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        n = len(rows)
        out.append({"point": name, "el": "%dx%d" % (H, W), "bl": "%dx%d" % (h, w), "ratio": ratio,
                    "encoding_time_s_per_p_frame": round(sum(r["encoding_time_BL"] + r["encoding_time_EL"] for r in rows) / n, 4),
                    "decoding_time_s_per_p_frame": round(sum(r["decoding_time_BL"] + r["decoding_time_EL"] for r in rows) / n, 4),
                    "reference_published": {"encoding_time": ref_enc, "decoding_time": ref_dec, "hardware": "unstated CUDA GPU", "source": "json_results/LSSVC/IP32/" + src},
                    "estimate_mode_frames_per_s": round(est_frames / dt, 2), "estimate_mode_note": "1 I + %d P, eager launches (no frame plans at this size)" % (est_frames - 1),
                    "p_frame_bpp_el_mean": round(sum(r["bit_el"] for r in est_rows) / max(1, len(est_rows)) / (ph * pw), 5)})
        del x_bls, x_els
        torch.cuda.empty_cache()
    return {"workload": "the reference's other published points (write_stream=1 encode / decode per P-frame; estimate-mode frame rate), synthetic clips and weights", "points": out}


def config3_2160p(device, gop=12):
    """BASELINE configs[3]: EL 2176x3840 (2160p padded) / BL 1088x1920, IP12 (1 I + 11 P), estimate mode, the f16x3 conv path:
    frames/s of one GOP through hipGraph frame plans (one warm-up GOP captures them) and the dominant kernel's roofline
    fraction from per-launch HIP events on P-frames 1..4 of an eager pass."""
    global HEIGHT, WIDTH, EVENT_FRAMES
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict
    keep = (HEIGHT, WIDTH, EVENT_FRAMES)
    HEIGHT, WIDTH, EVENT_FRAMES = 2160, 3840, 4
    try:
        inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, GAIN)).to(device).eval()
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", 0, GAIN))
        pnet.to(device).eval()
        inet.set_graph_mode(True, alias_outputs=True)
        pnet.set_graph_mode(True, alias_outputs=True)
        x_bls, x_els, pad, _ = build_inputs(device, seed=5, frames=gop)
        shape_hr = pad["HR_padded_size"]
        with torch.no_grad():
            encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=False)                  # eager first calls of the three frame types; steady-P captured
            encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=False)                  # I and first-P captured (a plan is captured on its SECOND call)
            torch.cuda.synchronize()
            t0 = time.time()
            bits, _ = encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=False)
            torch.cuda.synchronize()
            dt = time.time() - t0
            op_log = []
            encode_gop(inet, pnet, x_bls[:1 + EVENT_FRAMES], x_els[:1 + EVENT_FRAMES], shape_hr, op_log, lookahead=False)
            torch.cuda.synchronize()
        roof, _ = roofline_from_log(op_log)
        for k in ("traffic", "mfma_busy"):
            roof.pop(k, None)                                                # the committed PMC passes are of the 1080p run
        inet.set_graph_mode(False)
        pnet.set_graph_mode(False)
        return {"workload": "configs[3]: LSSVC two-layer x2, EL %dx%d (2160p padded) / BL %dx%d, IP%d, write_stream=0" % (
                    shape_hr[0], shape_hr[1], shape_hr[0] // 2, shape_hr[1] // 2, gop),
                "value": round(gop / dt, 3), "unit": "frames/s", "ms_per_gop": round(1e3 * dt, 1),
                "p_frame_bpp_el_mean": round(sum(b[1] for b in bits[1:]) / (len(bits) - 1) / (HEIGHT * WIDTH), 5), "roofline": roof}
    finally:
        HEIGHT, WIDTH, EVENT_FRAMES = keep
        torch.cuda.empty_cache()


def _host_description():
    import platform
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"cpu": cpu, "os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "platform": platform.platform(),
            "torch": torch.__version__}


def cpu_baseline(mode="full", warm=None):
    """The CPU oracle (a port of the reference's PyTorch CPU path, pinned bit-exact to it on the golden fixtures) timed on
    this box's host cores on a BOUNDED sample of the benchmark's own workload, IN THIS RUN:
      mode "full" (default, round 5): 1 I-frame + 1 P-frame at the FULL EL 1152x1920 / BL 576x960 size on all cores the process
          may use (capped at the 16-core share of a 1-GPU box), no scaling: about 70 s of CPU work; frames/s of the GOP-32
          mix = 32 / (I + 31 P). `value` is this measurement; the once-per-round file under profiles/ (1 I + 2 P, the second
          P a steady-state one) is quoted beside it as a cross-check only.
      mode "steady": the same with a second, steady-state P-frame (`--cpu-baseline-full`, ~2 min).
      mode "small": the round-1..4 default, 1 I + 1 P at EL 384x640 (1/9 of the pixels) scaled by 1/9 (`--cpu-baseline-small`;
          it flatters the CPU by about a third: smaller working set).
    Beside it, always: torch.set_num_threads(1) -- what the reference pins per worker (test.py:642) -- on a 1/22.5-size sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    cores = max(1, min(usable, 16))       # a 1-GPU box owns a 16-core share of a 256-thread host: more threads than that
    #                                        oversubscribe the share and run an order of magnitude slower (measured)
    full_size = mode != "small"
    H, W = (1152, 1920) if full_size else (384, 640)
    n_p = 2 if mode == "steady" else 1
    if mode != "full":
        warm = None
    log("  cpu baseline: %d threads (os.cpu_count() = %s), 1 I + %s at EL %dx%d ..." % (cores, os.cpu_count(), "1 steady-state P (from the GPU path's DPB)" if warm else "%d P" % n_p, H, W))
    t_i, t_p, t_ps = _oracle_frames(H, W, cores, n_p=n_p, warm=warm)
    scale = (H * W) / (1152.0 * 1920.0)
    fps = GOP / (t_i + (GOP - 1) * t_p) * scale
    h1, w1 = 256, 384                     # smallest sample whose BL (128x192) is still a multiple of 64
    log("  cpu baseline: I %.2f s, P %.2f s; now 1 thread at EL %dx%d ..." % (t_i, t_p, h1, w1))
    s_i, s_p, _ = _oracle_frames(h1, w1, 1)
    fps1 = GOP / (s_i + (GOP - 1) * s_p) * (h1 * w1) / (1152.0 * 1920.0)
    torch.set_num_threads(cores)
    size = "the full EL 1152x1920 / BL 576x960 size, no scaling" if full_size else \
        "EL 384x640 / BL 192x320 (1/9 of the pixels), scaled by 1/9"
    sample = ("measured in this run: 1 I-frame (%.2f s) + %s (%.2f s) at %s, GOP-32 mix (1 I + 31 P); %d threads "
              "(os.cpu_count() = %s, affinity = %d, capped at the 16-core share of a 1-GPU box); torch %s CPU fp32" % (
                  t_i, "a steady-state P-frame (frame 2 of the workload's clip, started from the DPB the GPU path produced for frames 0-1: the CPU's own differs from it by ~1e-6, "
                       "which the time does not see)" if warm else ("the steady-state second P-frame" if n_p == 2 else "the first P-frame (10 % slower than a steady one)"),
                  t_p, size, cores, os.cpu_count(), usable, torch.__version__))
    single = {"value": round(fps1, 6), "unit": "frames/s", "cores": 1,
              "sample": "torch.set_num_threads(1) (test.py:642): 1 I (%.2f s) + 1 P (%.2f s) at EL %dx%d, scaled by "
                        "pixel count (1/22.5)" % (s_i, s_p, h1, w1)}
    out = {"value": round(fps, 6), "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
           "i_frame_seconds": round(t_i, 2), "p_frame_seconds": [round(t, 2) for t in t_ps], "host": _host_description(), "single_thread": single}
    cached = _cached_cpu_baseline()
    if cached and "value" in cached:
        out["cross_check_measured_once"] = {"value": cached["value"], "cores": cached.get("cores"), "p_frame_seconds": cached.get("p_frame_seconds"),
                                            "host": cached.get("host"), "source": cached.get("source")}
    return out


# ---- round 6: the reference fixture as the workload, the parity record, the per-rank records --------------------------------------
def load_fixture(frames):
    """tests/golden/<FIXTURE>.npz -- the reference's own run of configs[1]'s GOP (bits, PSNR, quantised latents of all 32 frames; inputs
    are re-drawn from the seed and checked by sha1) -- or None when the file is not there or the run codes another GOP length."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", FIXTURE + ".npz")
    if frames != GOP or not os.path.exists(path):
        return None
    z = np.load(path)
    m = z["meta"]
    if int(m[0]) != GOP or (int(m[1]), int(m[2])) != (HEIGHT, WIDTH) or int(m[7]) != FIXTURE_SEED:
        return None
    return {"z": z, "H": int(m[3]), "W": int(m[4]), "h": int(m[5]), "w": int(m[6]), "path": os.path.relpath(path, ROOT),
            "threads": int(z["reference_threads"]) if "reference_threads" in z.files else None}


def reference_against_itself():
    """What the reference does to ITSELF on this GOP between two runs on other thread counts (tests/golden/<FIXTURE>_ref_t2.npz, round 6;
    tests/helpers.reference_self_disagreement reads the same file): the context the parity record is to be read in."""
    import numpy as np
    a_path, b_path = (os.path.join(ROOT, "tests", "golden", FIXTURE + sfx + ".npz") for sfx in ("", "_ref_t2"))
    if not (os.path.exists(a_path) and os.path.exists(b_path)):
        return None
    a, b = np.load(a_path), np.load(b_path)
    H, W, h, w = (int(v) for v in a["meta"][3:7])
    px = np.array([h * w, H * W], dtype=np.float64)
    n = int(a["meta"][0])
    d_bpp = [np.abs(b["f%d_bits" % t] - a["f%d_bits" % t]) / px for t in range(n)]
    d_psnr = [np.abs(b["f%d_psnr" % t] - a["f%d_psnr" % t]) for t in range(n)]
    sym = [sum(len(b[k]) for k in b.files if k.startswith("f%d_symdiff_" % t) and k.endswith("_idx")) for t in range(n)]
    return {"runs": "the reference on %d threads against the reference on %d (free-running, like this loop)" % (int(b["reference_threads"]), int(a["reference_threads"])),
            "frames_inside_plain_bars": int(sum(1 for t in range(n) if d_bpp[t].max() <= 1e-5 and d_psnr[t].max() <= 1e-4)),
            "max_d_bpp": float(max(d.max() for d in d_bpp)), "max_d_psnr_db": float(max(d.max() for d in d_psnr)),
            "symbols_flipped": int(sum(sym)), "frames_with_flipped_symbols": int(sum(1 for v in sym if v)),
            "source": "tests/golden/%s_ref_t2.npz" % FIXTURE}


def input_check(fx, x_bls, x_els, clip_u8):
    """Are the frames the DEVICE pre-processing makes (csrc/prepost.hip: u8 -> fp32, zero padding, the bicubic base layer) the frames
    the reference was given? sha1 of every base-layer frame against the fixture's, and of the clip."""
    import hashlib
    z = fx["z"]
    clip_ok = hashlib.sha1(clip_u8.numpy().tobytes()).hexdigest() == str(z["clip_sha1"])
    bl_ok = sum(1 for t in range(len(x_bls)) if hashlib.sha1(x_bls[t].contiguous().cpu().numpy().tobytes()).hexdigest() == str(z["f%d_x_bl_sha1" % t]))
    return {"clip_sha1_equal": bool(clip_ok), "base_layer_frames_bit_equal": bl_ok, "of": len(x_bls)}


def parity_from_bits(fx, bits):
    """Per frame |d bpp| of both layers: the GOP's bit counts against the reference's stored ones (test.py:413,441: bits / pixels)."""
    z = fx["z"]
    rows = []
    for t, (b_bl, b_el) in enumerate(bits):
        ref = z["f%d_bits" % t]
        rows.append((abs(float(b_bl) - float(ref[0])) / (fx["h"] * fx["w"]), abs(float(b_el) - float(ref[1])) / (fx["H"] * fx["W"])))
    return rows


def parity_gop(fx, inet, pnet, x_bls, x_els, shape_hr):
    """ONE untimed GOP, eager, frame after frame (test.py:182-250), with the quantised latents tapped: per frame the bit counts, the PSNR
    of both layers against the inputs, and the number of symbols that differ from the reference's. Its bit counts are compared with the
    timed GOPs' by the caller: equal counts = the same computation, so its PSNRs and symbols are the timed GOPs' too."""
    import numpy as np
    from lssvc_amd.preprocess import psnr
    z = fx["z"]
    keep = (inet.graph_mode, pnet.graph_mode)
    inet.graph_mode = pnet.graph_mode = False
    rows, bits, dpb = [], [], None
    try:
        for t in range(len(x_els)):
            net = inet if t == 0 else pnet
            inet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
            pnet.set_scale_information(RATIO, shape_hr, (0, 0, 0, 0))
            taps = net.taps = {}
            if t == 0:
                r = inet.encode_decode(x_bls[t], x_els[t], None, None)
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bls[t], x_els[t], dpb)
                dpb = r["dpb"]
            net.taps = None
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
            bits.append((r["bit_bl"], r["bit_el"]))
            want = z["f%d_psnr" % t]
            d_psnr = (abs(psnr(x_bls[t], dpb["ref_frame_bl"]) - float(want[0])), abs(psnr(x_els[t], dpb["ref_frame_el"]) - float(want[1])))
            differing = 0
            for key in [k[len("f%d_sym_" % t):] for k in z.files if k.startswith("f%d_sym_" % t)]:
                if key in taps:
                    differing += int(np.count_nonzero(taps[key].reshape(-1).numpy().astype(np.int32) - z["f%d_sym_%s" % (t, key)].astype(np.int32)))
            rows.append({"d_psnr": d_psnr, "symbols": differing})
    finally:
        inet.taps = pnet.taps = None
        inet.graph_mode, pnet.graph_mode = keep
    return bits, rows


def pin_rank_affinity(local, nlocal):
    """Before any GPU call: give rank `local` of `nlocal` on this node its own slice of the CPUs this process may use (the host threads of
    N ranks otherwise share, and migrate over, the same cores). -> the CPU list this rank keeps."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if nlocal > 1 and len(cpus) >= nlocal:
        per = len(cpus) // nlocal
        cpus = cpus[local * per:(local + 1) * per]
        os.sched_setaffinity(0, cpus)
    return cpus


def _compact_cpus(cpus):
    if not cpus:
        return None
    out, a, b = [], cpus[0], cpus[0]
    for c in cpus[1:] + [None]:
        if c is not None and c == b + 1:
            b = c
            continue
        out.append(str(a) if a == b else "%d-%d" % (a, b))
        if c is not None:
            a = b = c
    return ",".join(out)


def gather_rank_records(dist, rec, world):
    """Every rank's record on rank 0 (pickled through the process group: RCCL in the real run, gloo in rehearsal / dry run)."""
    if dist is None or world == 1:
        return [rec]
    out = [None] * world
    dist.all_gather_object(out, rec)
    return out


def device_identity(device):
    p = torch.cuda.get_device_properties(device)
    ident = {"index": device.index, "name": p.name, "total_memory_gib": round(p.total_memory / 2 ** 30, 1)}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id", "gcnArchName", "multi_processor_count"):
        v = getattr(p, k, None)
        if v is not None:
            ident[k] = str(v) if k == "uuid" else v
    return ident


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher in front: start N fresh rank processes (one per GPU, as
    test.py:648-656,685-748 starts one worker per GPU) through torch.distributed.run as a CHILD process, relay rank 0's
    JSON line, return the launcher's exit code. Called before this process has touched HIP (nothing above imports
    lssvc_amd or calls torch.cuda), and it never replaces this process: the children are ordinary subprocesses."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    log("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (n, n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def launcher_dry_run(args):
    """LSSVC_BENCH_DRYRUN=1: the rank flow of this file with the codec left out, for hosts without a GPU (the CPU test of
    the launcher, tests/test_host_logic.py): rendezvous over gloo, rank 0's checkpoint broadcast (lssvc_amd.shard, the
    same call the real run makes), barriers around K empty steps, the max-over-ranks reduce, ONE line on rank 0 with
    `dry_run: true` and no value. Never a measurement."""
    import torch.distributed as dist
    from lssvc_amd.shard import broadcast_state_dicts
    from lssvc_amd.synth import synth_state_dict
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, (world, args.gpus)
    cpus = pin_rank_affinity(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))      # as the real run: before anything else
    if world > 1:
        dist.init_process_group(backend="gloo")
    sds = broadcast_state_dicts(["intra_ss"], dist if world > 1 else None, torch.device("cpu"), loader=lambda name: synth_state_dict(name, 0, GAIN))
    n_tensors = len(sds["intra_ss"])
    if os.environ.get("LSSVC_BENCH_DRYRUN_FAIL_RANK") == str(rank):      # the test of the failure path
        raise RuntimeError("rank %d fails on purpose" % rank)
    if world > 1:
        dist.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    my_dt = time.time() - t0                             # this rank's own steps; the line's clock stops after the barrier
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.time() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the per-rank records of the real run (same helper, same fields where they exist without a GPU)
    import hashlib
    ranks = gather_rank_records(dist if world > 1 else None, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": None, "cpu_affinity": _compact_cpus(cpus),
                                                              "frames_per_s": round(args.frames * args.steps / my_dt, 4), "ms_per_step": round(1e3 * my_dt / args.steps, 2),
                                                              "bits_sha1": hashlib.sha1(b"dry run: every rank the same").hexdigest(), "pid": os.getpid()}, world)
    if rank == 0:
        print(json.dumps({"metric": "launcher dry run (no codec, no GPU)", "value": None, "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * t.item() / args.steps, 2), "checkpoint_tensors_broadcast": n_tensors, "ranks": ranks,
                          "rank_check": {"frames_per_s_per_rank": [r["frames_per_s"] for r in ranks], "all_ranks_bits_equal_rank0": all(r["bits_sha1"] == ranks[0]["bits_sha1"] for r in ranks),
                                         "process_group": None if world == 1 else {"backend": dist.get_backend(), "world_size": dist.get_world_size()}}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2, help="untimed GOPs first (default 2: a frame plan is captured on the SECOND call of its frame type, "
                                                          "so the I and first-P plans of a GOP are captured during the second GOP)")
    ap.add_argument("--frames", type=int, default=GOP, help="frames per GOP (default 32 = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events in the last timed step")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python instead of replaying hipGraph frame plans")
    ap.add_argument("--no-streams", action="store_true", help="one stream: no parallel branches in the frame plans (same as LSSVC_STREAMS=0)")
    ap.add_argument("--no-lookahead", action="store_true", help="code BL(t+1) after EL(t), not beside it (the plain per-frame protocol of test.py)")
    ap.add_argument("--no-h2d-pass", action="store_true", help="skip the second timed loop (round 6: the resident-inputs loop beside the h2d-inclusive headline)")
    ap.add_argument("--resident-headline", action="store_true", help="`value` from the loop with the inputs resident in HBM (the headline of rounds 1-5); the h2d-inclusive loop beside it")
    ap.add_argument("--no-copy-stream", action="store_true", help="upload + pre-process every frame on the coding stream, in front of it (round 5's h2d_inclusive loop)")
    ap.add_argument("--no-parity-pass", action="store_true", help="skip the untimed eager GOP that supplies the parity record's PSNR and flipped-symbol counts")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="CPU baseline at the full 1152x1920 size with a steady-state second P-frame, 1 I + 2 P (~2 min; the default is 1 I + 1 P, ~70 s)")
    ap.add_argument("--cpu-baseline-small", action="store_true", help="CPU baseline on the 1/9-size sample of rounds 1-4 (EL 384x640, scaled; ~7 s)")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the configs[3] (2160p IP12) and configs[4] (write_stream=1) side measurements")
    ap.add_argument("--precision", choices=["f32", "f16x3"], default=None,
                    help="conv arithmetic (default: lssvc_amd's default, see hip_ops.CONV_PRECISION)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="time the CPU oracle at full size (1 I + 2 P at 1152x1920) on this host, "
                    "print its JSON with the host description and exit: the once-per-round run kept as profiles/rNN_cpu_baseline_full.json")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("LSSVC_BENCH_DRYRUN", "0") == "1":
        return launcher_dry_run(args)
    if args.cpu_baseline_only:
        d = cpu_baseline("steady")
        d.pop("cross_check_measured_once", None)
        print(json.dumps(d), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpus = pin_rank_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))      # BEFORE any GPU call
    if cpus and world > 1:
        torch.set_num_threads(max(1, min(len(cpus), 8)))
    # Rehearsal of the multi-rank flow on a box with ONE GPU (never a measurement: the ranks share the card):
    # LSSVC_BENCH_REHEARSAL=1 puts every rank on cuda:0 and runs the collectives over gloo on host tensors.
    rehearsal = os.environ.get("LSSVC_BENCH_REHEARSAL", "0") == "1"
    device = torch.device("cuda", 0 if rehearsal else local)
    torch.cuda.set_device(device)
    coll_device = torch.device("cpu") if rehearsal else device
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus (got world=%d, --gpus=%d)" % (world, args.gpus)

    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    if args.precision:
        hip_ops.set_conv_precision(args.precision)
    if args.no_lookahead:
        globals()["LOOKAHEAD"] = False
    if args.no_streams:
        hip_ops.MULTI_STREAM = False
    from lssvc_amd.synth import synth_state_dict
    from lssvc_amd.shard import broadcast_state_dicts
    # The workload: the reference fixture's clip and weights (every rank codes that GOP -- its own copy, uploaded and coded by itself; with
    # one content the ranks' bit counts can be held against each other and against the reference). Without the fixture file (or with
    # --frames != 32) the round-5 workload: seed 0 weights, a clip per rank, no parity record.
    fx = load_fixture(args.frames)
    seed = FIXTURE_SEED if fx is not None else 0
    # the deployment path (harness under torchrun): rank 0 alone has the checkpoints and broadcasts them over RCCL, ~245 MB
    # once; with one rank this is a plain local load
    t0 = time.time()
    sds = broadcast_state_dicts(["intra_ss", "lssvc_extend"], dist, coll_device, loader=lambda name: synth_state_dict(name, seed, GAIN))
    if rank == 0 and world > 1:
        log("checkpoints broadcast from rank 0 over %s in %.2f s" % ("gloo (rehearsal)" if rehearsal else "RCCL", time.time() - t0))
    inet = IntraSS.from_state_dict(sds["intra_ss"]).to(device).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(sds["lssvc_extend"])
    pnet.to(device).eval()
    del sds

    if not args.no_graph:
        inet.set_graph_mode(True, alias_outputs=True)                  # FramePlan: the per-frame launch sequence replayed as a hipGraph
        pnet.set_graph_mode(True, alias_outputs=True)
    hip_ops.reserve_device_memory(device)          # one hipMalloc up front instead of pool growth during the first GOPs
    t0 = time.time()
    x_bls, x_els, pad, clip_u8 = build_inputs(device, seed=seed if fx is not None else rank, frames=args.frames, exact=fx is not None)
    shape_hr = pad["HR_padded_size"]
    torch.cuda.synchronize()
    inputs = input_check(fx, x_bls, x_els, clip_u8) if fx is not None else None
    if rank == 0:
        log("inputs ready in %.1f s: EL %s BL %s, %d frames/GOP; %s" % (time.time() - t0, tuple(x_els[0].shape), tuple(x_bls[0].shape), args.frames,
                                                                      ("fixture %s, inputs %s" % (fx["path"], inputs)) if fx is not None else "no reference fixture for this run"))

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    host = HostFrames(clip_u8, device, side_stream=not args.no_copy_stream)

    def timed_loop(steps, source, events):
        """K GOPs between two barriers; `source`: "host" (per-frame H2D + pre-processing inside the clock) or "resident"."""
        sync_all()
        t_start = time.time()
        bits_, log_ = None, None
        for k in range(steps):
            if k == steps - 1 and events:
                log_ = []
            if source == "host":
                bits_, _ = encode_gop(inet, pnet, None, None, shape_hr, log_, host_frames=host)
            else:
                bits_, _ = encode_gop(inet, pnet, x_bls, x_els, shape_hr, log_)
        torch.cuda.synchronize()
        own = time.time() - t_start                     # this rank's own K steps (the per-rank record); the line's clock stops after the barrier
        sync_all()
        return time.time() - t_start, bits_, log_, own

    headline_src = "resident" if args.resident_headline else "host"
    with torch.no_grad():
        if LOOKAHEAD and hip_ops.MULTI_STREAM and not args.no_graph:
            # set-up, before the W warm-up steps: the frame-after-frame plans (two GOPs: eager first calls with their fp16 range audit,
            # then the captures) -- the event-sampled frames of the last timed step and the look-ahead self-check use them -- and the
            # eager first calls of the look-ahead plans, so that the first warm-up step captures those and the timed steps only replay
            t0 = time.time()
            for la in (False, False, True):
                encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=la)
            torch.cuda.synchronize()
            if rank == 0:
                log("frame plans primed in %.2f s (2 GOPs frame after frame, 1 with look-ahead)" % (time.time() - t0))
        for w in range(args.warmup):
            t0 = time.time()
            if headline_src == "host":
                encode_gop(inet, pnet, None, None, shape_hr, host_frames=host)
            else:
                encode_gop(inet, pnet, x_bls, x_els, shape_hr)
            torch.cuda.synchronize()
            if rank == 0:
                log("warmup GOP %d: %.2f s" % (w, time.time() - t0))
        events_on = rank == 0 and not args.no_events
        dt, bits, op_log, my_dt = timed_loop(args.steps, headline_src, events_on)
        # look-ahead self-check: the timed GOPs' bit counts against one GOP coded frame after frame (untimed). A mismatch voids the
        # look-ahead number: the timed loop is then repeated without it.
        lookahead_check = None
        if LOOKAHEAD:
            plain_bits, _ = encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=False)
            sync_all()
            lookahead_check = bool(plain_bits == bits) and os.environ.get("LSSVC_BENCH_FAIL_LOOKAHEAD_CHECK") != "1"      # (env: exercises the path below)
            if not lookahead_check:
                log("!! look-ahead bits differ from the plain protocol's: timing again without look-ahead")
                globals()["LOOKAHEAD"] = False
                for _ in range(2):
                    encode_gop(inet, pnet, x_bls, x_els, shape_hr)
                dt, bits, op_log, my_dt = timed_loop(args.steps, headline_src, events_on)
        # the other loop beside the headline (round 6: the resident loop; with --resident-headline the h2d-inclusive one), fewer steps
        dt_side, side_steps, bits_side = None, 0, None
        side_src = "resident" if headline_src == "host" else "host"
        if not args.no_h2d_pass:
            side_steps = max(1, min(args.steps, 5))
            if side_src == "host":
                encode_gop(inet, pnet, None, None, shape_hr, host_frames=host)          # untimed: allocator warm-up of this path
            else:
                encode_gop(inet, pnet, x_bls, x_els, shape_hr)
            dt_side, bits_side, _, _ = timed_loop(side_steps, side_src, False)
        # parity: the timed GOP's bit counts against the reference's; PSNR and symbols from one untimed eager GOP on rank 0
        parity, prow = None, None
        if fx is not None:
            d_bpp = parity_from_bits(fx, bits)
            parity = {"frames": len(d_bpp), "max_d_bpp": max(max(r) for r in d_bpp), "frames_inside_1e-5_bpp": sum(1 for r in d_bpp if max(r) <= 1e-5),
                      "gop_avg_d_bpp": max(abs(sum(float(b[l]) - float(fx["z"]["f%d_bits" % t][l]) for t, b in enumerate(bits))) / (len(bits) * px)
                                           for l, px in ((0, fx["h"] * fx["w"]), (1, fx["H"] * fx["W"])))}
            if rank == 0 and not args.no_parity_pass:
                t0 = time.time()
                pbits, prow = parity_gop(fx, inet, pnet, x_bls, x_els, shape_hr)
                torch.cuda.synchronize()
                log("parity GOP (eager, latents tapped, untimed): %.1f s" % (time.time() - t0))
                parity["eager_gop_bits_equal_timed_gop"] = bool(pbits == bits)
    if dist is not None:
        t = torch.tensor([dt, dt_side if dt_side is not None else 0.0], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t[0].item()
        dt_side = t[1].item() if dt_side is not None else None
    import hashlib
    rec = {"rank": rank, "local_rank": local, "device": device_identity(device), "cpu_affinity": _compact_cpus(cpus), "frames_per_s": round(args.frames * args.steps / my_dt, 4),
           "ms_per_step": round(1e3 * my_dt / args.steps, 2), "bits_sha1": hashlib.sha1(repr([(float(a), float(b)) for a, b in bits]).encode()).hexdigest(),
           "parity": parity and {k: parity[k] for k in ("max_d_bpp", "frames_inside_1e-5_bpp")}, "host_uploads": host.uploads, "pid": os.getpid()}
    ranks = gather_rank_records(dist, rec, world)

    if rank == 0:
        frames = world * args.frames * args.steps
        incl = "per-frame H2D of the 8-bit frame + pre-processing + encode + D2H of the bit counts inside the clock (BASELINE.md section 3 'GPU side')"
        res = "inputs resident in HBM"
        out = {
            "metric": "encoded frames/sec, LSSVC two-layer x2 (BL 540p + EL 1080p), GOP 32 (1 I + 31 P), estimate mode",
            "value": round(frames / dt, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[hip_ops.CONV_PRECISION], "data": "synthetic",
            "config": {"workload": "configs[1]: LSSVC two-layer x2, EL 1152x1920 (1080p padded) / BL 576x960, "
                                   "%d-frame GOP per GPU per step, write_stream=0; %s" % (args.frames, incl if headline_src == "host" else res + " for `value` (--resident-headline)"),
                       "frames_per_step_per_gpu": args.frames,
                       "weights": "seeded synthetic (lssvc_amd.synth, seed %d, gain %.2f)" % (seed, GAIN),
                       "clip": ("the reference fixture's: synth_clip_exact(seed %d), the GOP of %s; every rank codes it" % (FIXTURE_SEED, fx["path"])) if fx is not None
                               else "synth_clip(seed = rank): no reference fixture for this run",
                       "per_frame_pipeline": "frame t+2's upload (6.2 MB from pinned host memory) and pre-processing (u8 -> fp32, zero padding to 1152x1920, MATLAB-bicubic BL 576x960: "
                                             "csrc/prepost.hip) on a copy stream beside frame t; every frame is uploaded and pre-processed each time it is coded (%d uploads on rank 0)" % host.uploads
                                             if not args.no_copy_stream else "upload + pre-processing in front of every frame on the coding stream (--no-copy-stream)",
                       "parallelism": ("gop-shard x%d (no data-path collective; weights broadcast once from rank 0 over RCCL)" % world
                                       + (" -- REHEARSAL: all ranks on one GPU, gloo; not a measurement" if rehearsal else "")) if world > 1
                       else "gop-shard x1",
                       "launch": ("eager (ctypes per kernel)" if args.no_graph else "hipGraph frame plans (I / first-P / steady-P)")
                                 + (", independent chains of a frame as parallel branches (side streams)" if hip_ops.MULTI_STREAM else ", single stream")},
        }
        if dt_side is not None:
            out["resident" if side_src == "resident" else "h2d_inclusive"] = {
                "value": round(world * args.frames * side_steps / dt_side, 4), "unit": "frames/s", "steps": side_steps,
                "ms_per_step": round(1e3 * dt_side / side_steps, 2), "what": "same GOP, " + (res + " before the clock starts (the headline of rounds 1-5)" if side_src == "resident" else incl),
                "bits_equal_headline_run": bool(bits_side == bits)}
        # ---- parity: throughput and parity on the same inputs, in one record
        if parity is not None:
            bars = None
            if prow is not None:
                d_bpp = parity_from_bits(fx, bits)
                bars = sum(1 for t in range(len(bits)) if max(d_bpp[t]) <= 1e-5 and max(prow[t]["d_psnr"]) <= 1e-4)
                flipped = [r["symbols"] for r in prow]
                parity.update({"frames_inside_plain_bars": bars, "max_d_psnr_db": max(max(r["d_psnr"]) for r in prow), "symbols_flipped": sum(flipped),
                               "frames_with_flipped_symbols": sum(1 for v in flipped if v), "first_frame_with_a_flipped_symbol": next((t for t, v in enumerate(flipped) if v), None),
                               "worst_frames": sorted(({"frame": t, "d_bpp": max(d_bpp[t]), "d_psnr_db": max(prow[t]["d_psnr"]), "symbols_flipped": flipped[t]}
                                                       for t in range(len(bits))), key=lambda r: -r["d_bpp"])[:3]})
            second = os.path.join(ROOT, "tests", "golden", FIXTURE + "_ref_t2.npz")
            if os.path.exists(second):
                import numpy as np
                z2 = np.load(second)
                d2 = [max(abs(float(b[0]) - float(z2["f%d_bits" % t][0])) / (fx["h"] * fx["w"]), abs(float(b[1]) - float(z2["f%d_bits" % t][1])) / (fx["H"] * fx["W"])) for t, b in enumerate(bits)]
                parity["against_the_references_second_run"] = {"frames_inside_1e-5_bpp": sum(1 for v in d2 if v <= 1e-5), "max_d_bpp": max(d2),
                                                               "what": "the same timed bit counts against the reference's run on %d threads (tests/golden/%s_ref_t2.npz): "
                                                                       "the two reference runs differ from each other as `reference_against_itself` says" % (int(z2["reference_threads"]), FIXTURE)}
            parity.update({"fixture": fx["path"] + " (the reference itself on %s CPU threads; tests/golden/make_golden_full.py)" % fx["threads"],
                           "bars": "per frame and layer |d bpp| <= 1e-5 and |d PSNR| <= 1e-4 dB (BASELINE.json north_star), FREE-RUNNING: no re-alignment of the closed loop after a rounding tie",
                           "bits_from": "the last timed GOP of the headline loop", "psnr_and_symbols_from": "one untimed eager GOP whose bit counts are compared with the timed GOP's",
                           "inputs": inputs, "reference_against_itself": reference_against_itself()})
        out["parity"] = parity
        # ---- the ranks, one record each (round 6): what makes an N-GPU line checkable from the line itself
        out["ranks"] = ranks
        out["rank_check"] = {"frames_per_s_per_rank": [r["frames_per_s"] for r in ranks], "sum_of_per_rank_rates": round(sum(r["frames_per_s"] for r in ranks), 4),
                             "distinct_devices": len({json.dumps(r["device"], sort_keys=True) for r in ranks}),
                             "all_ranks_bits_equal_rank0": all(r["bits_sha1"] == ranks[0]["bits_sha1"] for r in ranks) if fx is not None else None,
                             "process_group": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                                         "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if not rehearsal else None}}
        out["hbm_reserved_gib"] = round(torch.cuda.max_memory_reserved(device) / 2 ** 30, 1)      # PEAK over the run, priming included (of 288 GB)
        out["hbm_reserved_steady_gib"] = round(torch.cuda.memory_reserved(device) / 2 ** 30, 1)  # now: the frame plans' graph pools + inputs
        out["lookahead"] = {"on": bool(LOOKAHEAD), "streams": "side streams" if hip_ops.MULTI_STREAM else "single stream", "bits_equal_frame_after_frame_run": lookahead_check,
                            "what": "the frame loop hands frame t+1's base-layer input to the call of frame t (LSSVC_extend.forward_one_frame, "
                                    "frame_id / next_x_bl): BL(t+1) is coded on a second stream beside EL(t); same launches, bit-identical results"}
        pel = HEIGHT * WIDTH
        out["bpp_check"] = {"i_frame_bpp_el": round(bits[0][1] / pel, 5),
                            "p_frame_bpp_el_mean": round(sum(b[1] for b in bits[1:]) / max(1, len(bits) - 1) / pel, 5)}
        if op_log:
            roof, table = roofline_from_log(op_log)
            out["roofline"] = roof
            out["roofline_by_kernel"] = table[:6]
            # progress outside the dominant kernel: ALGORITHMIC conv flops of the whole GOP over the GOP's wall time. The
            # per-P-frame figure is the sampled launches' (the GOP's last EVENT_FRAMES P-frames); the I-frame's share comes from
            # SURVEY section 8d (5.04 TFLOP).
            logged = max(1, min(args.frames - 1, EVENT_FRAMES))       # P-frames that carried events (encode_gop: the GOP's last min(n - 1, EVENT_FRAMES))
            p_tflop = roof["conv_tflop_sampled"] / logged
            # (the I-frame's 5.04 TFLOP is SURVEY section 8d's figure for EL 1152x1920 / BL 576x960; conv work is linear in pixels)
            gop_tflop = 5.04 * (shape_hr[0] * shape_hr[1]) / (1152.0 * 1920.0) + (args.frames - 1) * p_tflop
            ach = gop_tflop / (dt / args.steps)
            out["whole_frame"] = {"algorithmic_conv_tflop_per_gop": round(gop_tflop, 1), "achieved": round(ach, 1), "unit": "TFLOP/s",
                                  "peak": PEAK_FP16_MFMA_TFLOPS, "frac": round(ach / PEAK_FP16_MFMA_TFLOPS, 4),
                                  "frac_issued_fp16": round(3.0 * ach / PEAK_FP16_MFMA_TFLOPS, 4),
                                  }
            out["timed_region_note"] = ("the last timed step issues its last %d P-frames eagerly on one stream with a HIP event pair around "
                                        "every launch (the roofline's live durations); that costs the headline about 1 %%" % EVENT_FRAMES)
        else:
            out["roofline"] = None
        warm = None
        if world == 1 and not args.no_cpu_baseline and not (args.cpu_baseline_full or args.cpu_baseline_small):
            # frames 0-1 on the GPU (eager, untimed) -> the DPB a steady-state P-frame starts from, for the CPU baseline's sample
            with torch.no_grad():
                g = (inet.graph_mode, pnet.graph_mode)
                inet.graph_mode = pnet.graph_mode = False
                _, dpb_g = encode_gop(inet, pnet, x_bls[:2], x_els[:2], shape_hr, lookahead=False)
                inet.graph_mode, pnet.graph_mode = g
                warm = {"seed": seed, "dpb": {k: (v.detach().cpu().clone() if v is not None else None) for k, v in dpb_g.items()},
                        "x_bl": [x.cpu() for x in x_bls[:3]], "x_el": [x.cpu() for x in x_els[:3]]}
                del dpb_g
        if world == 1:
            try:
                out["config0_latency"] = config0_latency(device, graph=not args.no_graph)
            except Exception as e:                          # a side measurement must not take the headline line down
                out["config0_latency"] = {"error": repr(e)}
        if world == 1 and not args.no_side_configs:
            del x_bls, x_els
            host.ready.clear()
            inet.set_graph_mode(False)                      # drop the 1080p frame plans and their graph pools first
            pnet.set_graph_mode(False)
            torch.cuda.empty_cache()
            for name, fn in (("config4_stream", config4_stream), ("published_points", published_points), ("config3_2160p", config3_2160p)):
                try:
                    log("side measurement %s ..." % name)
                    out[name] = fn(device)
                except Exception as e:
                    out[name] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle on a bounded sample (1 I + 1 steady-state P at full size: about 70 s) ...")
            out["cpu_baseline"] = cpu_baseline("steady" if args.cpu_baseline_full else ("small" if args.cpu_baseline_small else "full"), warm=warm)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
