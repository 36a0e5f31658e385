"""In-kernel stamps of the persistent 3x3 kernel's consumer waves (diagnostic build, LSSVC_CONV_DEBUG=256):
where a consumer's time goes -- MFMA phases, barrier waits, epilogues -- and the shader clock it ran at.
    LSSVC_CONV_DEBUG=256 python tools/p3_stamps.py"""
import math
import os
import sys

import torch

os.environ.setdefault("LSSVC_CONV_DEBUG", "256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd import _lib  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    records = []
    ops.set_conv_precision("f16x3")
    g = torch.Generator().manual_seed(0)
    shapes = (("64->64 @1152x1920", 64, 64, 1152, 1920), ("128->64 @576x960", 128, 64, 576, 960), ("48->48 @1152x1920", 48, 48, 1152, 1920),
              ("96->48 @1152x1920", 96, 48, 1152, 1920))
    stride = int(os.environ.get("P3_STRIDE", "1"))            # 2: the stride-2 form (round 6; MF = 4 shapes only)
    if stride == 2:
        shapes = (("s2 48->64 in 1152x1920", 48, 64, 1152, 1920), ("s2 64->64 in 576x960", 64, 64, 576, 960), ("s2 128->128 in 288x480", 128, 128, 288, 480))
    if os.environ.get("P3_LATE", "0") == "1":                 # the late-loads schedule's stamp build (round 6; MF = 4 without an input activation only)
        _lib.check(_lib.lib.lssvc_set_option(b"p3_big_pair", 4))
        shapes = (("64->64 @1152x1920", 64, 64, 1152, 1920), ("128->64 @576x960", 128, 64, 576, 960), ("64->64 @576x960", 64, 64, 576, 960))
    for name, cin, cout, H, W in shapes:
        w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
        b = torch.randn(cout, generator=g)
        Wt = WeightStore({"c.weight": w, "c.bias": b}, dev)
        x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
        split = os.environ.get("P3_SPLIT", "0") == "1"       # pre-split input: the patch by LDS-DMA (round 5)
        if split:
            x = ops.presplit(x)
        stamps = torch.zeros(2 * 256 * 4 * 8, dtype=torch.int64, device=dev)      # consumer records, then producer records
        out = ops.T.empty(H // stride, W // stride, cout, dev)
        import ctypes as C
        w_dev, b_dev, co, m_pad, KH, KW = Wt.conv("c", [cin], False)
        w16 = Wt.conv_f16x3("c", [cin], False)
        d = _lib.ConvDesc()
        d.inp[0] = x.v
        d.n_in = 1
        d.weight, d.bias = w_dev.data_ptr(), b_dev.data_ptr()
        d.KH, d.KW, d.stride, d.pad_t, d.pad_l = 3, 3, stride, 1, 1
        d.Cout, d.M_pad = co, m_pad
        d.in_act, d.in_slope, d.epilogue = 0, 0.01, 0
        d.gdn_x = _lib.View(stamps.data_ptr(), 1, 1, 1, 1)
        d.act, d.slope, d.out_scale, d.pixel_shuffle = 0, 0.01, 1.0, 0
        d.residual = _lib.View(None, 0, 0, 0, 0)
        d.out = out.v
        d.precision, d.weight16, d.weight16_unscale = _lib.PREC_F16X3 | (_lib.PREC_SPLIT_IN if split else 0), w16[0].data_ptr(), w16[1]
        for _ in range(200):                      # warm the clock governor with back-to-back launches
            _lib.check(_lib.lib.lssvc_conv2d(C.byref(d), ops.stream_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            _lib.check(_lib.lib.lssvc_conv2d(C.byref(d), ops.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        wall_us = e0.elapsed_time(e1) / 30 * 1e3
        allrec = stamps.view(-1, 8).cpu().double()
        s = allrec[:1024]
        s = s[s[:, 5] > 0]
        pr = allrec[1024:2048]
        pr = pr[pr[:, 5] > 0]
        comp, bar, epi, cyc, real, phases, tiles = (s[:, i].median().item() for i in range(7))
        zero = s[:, 7].median().item()
        clock = cyc / real * 100.0                 # s_memrealtime ticks at 100 MHz
        print("%s (%s) LSSVC_CONV_DEBUG=%s: %.1f us per launch; per consumer wave, median over %d waves: %d phases / %d tiles; total %.0f cycles at %.0f MHz" %
              (name, _lib.lib.lssvc_conv2d_last_kernel().decode(), os.environ["LSSVC_CONV_DEBUG"], wall_us, s.shape[0], phases, tiles, cyc, clock))
        print("   compute %5.1f %% (%.0f cyc/phase; MF = 4: 336 MFMAs = 5376 issue cycles, MF = 3: 252 = 4032)   barrier wait %5.1f %% (%.0f cyc/phase)   "
              "epilogue %5.1f %% (%.0f cyc/tile, of which zeroing the accumulators %.0f)" % (100 * comp / cyc, comp / phases, 100 * bar / cyc, bar / phases,
                                                    100 * epi / cyc, epi / max(tiles, 1), zero / max(tiles, 1)))
        records.append({"kernel": _lib.lib.lssvc_conv2d_last_kernel().decode().replace(" split", ""), "shape": name, "in_kernel_clock_ghz": round(clock / 1e3, 3),
                        "mfma_issue_share_of_cycles": round(phases * (5376.0 if cout % 64 == 0 else 4032.0) / cyc, 3), "consumer_cycles": {"compute": round(comp / cyc, 3), "wait_for_fills": round(bar / cyc, 3), "epilogue": round(epi / cyc, 3)},
                        "us_per_launch": round(wall_us, 1)})
        if pr.shape[0]:
            dma, ld, wait, cvt, pbar, pph, geo = (pr[:, i].median().item() for i in range(7))
            print("   producer wave, cycles per phase: weight-DMA issue %.0f, patch-load issue %.0f, convert + LDS stores %.0f, "
                  "barrier wait %.0f, tile geometry %.0f; staged build: boundary fill (consumers done -> fill signalled) %.0f cycles per tile%s" % (
                      dma / pph, ld / pph, cvt / pph, pbar / pph, geo / pph, wait / max(tiles, 1),
                      "; late-loads build: wait for the weight DMA %.0f cycles per phase" % (wait / pph) if os.environ.get("P3_LATE", "0") == "1" else ""))
    if "--json" in sys.argv:
        _write_json(records)


def _unused():
    pass


def _write_json(records):
    import json
    path = sys.argv[sys.argv.index("--json") + 1]
    with open(path, "w") as f:
        json.dump({"command": "LSSVC_CONV_DEBUG=256 python tools/p3_stamps.py --json <path> (diagnostic stamp build; 200 warm-up launches, then 30 timed)",
                   "kernels": records[:1]}, f, indent=1)          # the first shape is the bench's dominant one (64->64 @1152x1920)


if __name__ == "__main__":
    main()
