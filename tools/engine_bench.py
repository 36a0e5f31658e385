"""Throughput of the C++ plan runtime at BASELINE configs[1]'s size: compile the I / first-P / steady-P plans of a 1152x1920 /
576x960 checkpoint, then code a GOP through lssvc_engine_iframe / lssvc_engine_pframe (ctypes, device pointers, the engine's
own stream and hipGraph) and through the Python frame plans, same inputs:  python tools/engine_bench.py [frames]"""
import ctypes as C
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.prepost import FramePrep  # noqa: E402
from lssvc_amd.synth import synth_clip, synth_state_dict  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.55)).to(dev).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, 0.55))
    pnet.to(dev).eval()
    prep = FramePrep(dev)
    clip = synth_clip(n, 1080, 1920, seed=0)
    layers = [prep.make_layers_rgb8(clip[t].to(dev), 2.0) for t in range(n)]
    x_bl = [l[0].contiguous() for l in layers]
    x_el = [l[1].contiguous() for l in layers]
    H, W = layers[0][2]["HR_padded_size"]
    h, w = H // 2, W // 2
    for net in (inet, pnet):
        net.set_scale_information(2.0, (H, W), (0, 0, 0, 0))

    def python_gop():
        dpb, bits = None, []
        for t in range(n):
            if t == 0:
                r = inet.encode_decode(x_bl[t], x_el[t], None, None)
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bl[t], x_el[t], dpb)
                dpb = r["dpb"]
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
            bits.append((r["bit_bl"], r["bit_el"]))
        return bits, dpb

    # DPBs to compile from (eager)
    r = inet.encode_decode(x_bl[0], x_el[0], None, None)
    d0 = {"ref_frame_bl": r["x_hat_bl"].contiguous().clamp(0, 1), "ref_frame_el": r["x_hat_el"].contiguous().clamp(0, 1), "ref_feature_bl": None,
          "ref_feature_el": r["feature_el"].contiguous()}
    r = pnet.encode_decode(x_bl[1], x_el[1], d0)
    d1 = {k: v.contiguous() for k, v in r["dpb"].items()}
    d1["ref_frame_bl"] = d1["ref_frame_bl"].clamp(0, 1)
    d1["ref_frame_el"] = d1["ref_frame_el"].clamp(0, 1)
    tmp = tempfile.mkdtemp(prefix="lssvc_plans_")
    paths = [os.path.join(tmp, p) for p in ("i.plan", "p1.plan", "p.plan")]
    t0 = time.time()
    print("iframe plan  ", plan_compiler.compile_iframe(inet, x_bl[0], x_el[0], paths[0])[0])
    print("first-P plan ", plan_compiler.compile_pframe(pnet, x_bl[1], x_el[1], d0, paths[1])[0])
    print("steady-P plan", plan_compiler.compile_pframe(pnet, x_bl[2], x_el[2], d1, paths[2])[0])
    # round 6: the P-frame as base-layer + enhancement-layer plans (the look-ahead entry point)
    lpaths = [os.path.join(tmp, p) for p in ("bl1.plan", "bl.plan", "el1.plan", "el.plan")]
    i1 = plan_compiler.compile_pframe_layers(pnet, x_bl[1], x_el[1], d0, lpaths[0], lpaths[2])
    i2 = plan_compiler.compile_pframe_layers(pnet, x_bl[2], x_el[2], d1, lpaths[1], lpaths[3])
    print("layer plans  ", i1[0], i1[1], i2[0], i2[1])
    print("compiled in %.1f s, files %.0f MB" % (time.time() - t0, sum(os.path.getsize(p) for p in paths + lpaths) / 1e6))

    inet.set_graph_mode(True, alias_outputs=True)
    pnet.set_graph_mode(True, alias_outputs=True)
    python_gop()
    python_gop()
    torch.cuda.synchronize()
    t0 = time.time()
    want, _ = python_gop()
    torch.cuda.synchronize()
    t_py = time.time() - t0
    inet.set_graph_mode(False)
    pnet.set_graph_mode(False)
    torch.cuda.empty_cache()

    eng = C.c_void_p(lib.lssvc_engine_create(0))
    for model, net in ((0, inet), (1, pnet)):                      # the raw checkpoints first: plans hold launches only
        table, n_tensors = net.W._ckpt()
        check(lib.lssvc_engine_load_checkpoint(eng, model, table, n_tensors))
    check(lib.lssvc_engine_load_intra(eng, paths[0].encode()))
    check(lib.lssvc_engine_load_inter(eng, paths[1].encode(), paths[2].encode()))
    check(lib.lssvc_engine_load_inter_layers(eng, *[p.encode() for p in lpaths]))
    check(lib.lssvc_engine_set_scale(eng, 2.0, H, W))
    P = lambda t: C.c_void_p(t.data_ptr())
    ref_bl, ref_el = torch.empty(1, 3, h, w, device=dev), torch.empty(1, 3, H, W, device=dev)
    feat_bl, feat_el64, feat_el48 = torch.empty(1, 64, h, w, device=dev), torch.empty(1, 64, H, W, device=dev), torch.empty(1, 48, H, W, device=dev)
    n_ref_bl, n_ref_el, n_feat_bl, n_feat_el = (torch.empty_like(x) for x in (ref_bl, ref_el, feat_bl, feat_el48))

    def engine_gop(lookahead=False):
        nonlocal ref_bl, ref_el, feat_bl, feat_el48, n_ref_bl, n_ref_el, n_feat_bl, n_feat_el
        bits_all, b = [], (C.c_double * 2)()
        for t in range(n):
            if t == 0:
                check(lib.lssvc_engine_iframe(eng, P(x_bl[t]), P(x_el[t]), b, P(ref_bl), P(ref_el), P(feat_el64), None))
                feat_el = feat_el64
            elif lookahead:
                check(lib.lssvc_engine_pframe_lookahead(eng, P(x_bl[t]), P(x_el[t]), P(x_bl[t + 1]) if t + 1 < n else None, P(ref_bl), P(ref_el),
                                                        P(feat_bl) if t > 1 else None, P(feat_el), b, P(n_ref_bl), P(n_feat_bl), P(n_ref_el), P(n_feat_el), None, None, None))
            else:
                check(lib.lssvc_engine_pframe(eng, P(x_bl[t]), P(x_el[t]), P(ref_bl), P(ref_el), P(feat_bl) if t > 1 else None, P(feat_el), b,
                                              P(n_ref_bl), P(n_feat_bl), P(n_ref_el), P(n_feat_el), None, None, None))
            if t > 0:
                ref_bl, n_ref_bl, ref_el, n_ref_el = n_ref_bl, ref_bl, n_ref_el, ref_el
                feat_bl, n_feat_bl = n_feat_bl, feat_bl
                feat_el48, n_feat_el = n_feat_el, feat_el48
                feat_el = feat_el48
            check(lib.lssvc_clamp_inplace(P(ref_bl), ref_bl.numel(), 0.0, 1.0, None))
            check(lib.lssvc_clamp_inplace(P(ref_el), ref_el.numel(), 0.0, 1.0, None))
            torch.cuda.synchronize()
            bits_all.append((b[0], b[1]))
        return bits_all

    engine_gop()
    engine_gop()
    torch.cuda.synchronize()
    t0 = time.time()
    got = engine_gop()
    torch.cuda.synchronize()
    t_eng = time.time() - t0
    print("python frame plans: %.2f frames/s; C++ plan runtime (NCHW in/out, copies in and out of the plan's buffers): %.2f frames/s" % (n / t_py, n / t_eng))
    print("bit counts equal:", got == want)
    times = {False: [], True: []}
    for rnd in range(3):                                           # interleaved: whole frames | base layer a frame ahead
        for la in (False, True):
            engine_gop(la)
            torch.cuda.synchronize()
            t0 = time.time()
            g = engine_gop(la)
            torch.cuda.synchronize()
            times[la].append(n / (time.time() - t0))
            assert g == want, "look-ahead %s: bit counts differ from the Python path's" % la
    med = lambda v: sorted(v)[len(v) // 2]
    print("C++ plan runtime, whole frames (lssvc_engine_pframe): %.2f frames/s; base layer a frame ahead (lssvc_engine_pframe_lookahead): %.2f frames/s  (x%.3f; bit counts equal)" % (
        med(times[False]), med(times[True]), med(times[True]) / med(times[False])))
    lib.lssvc_engine_destroy(eng)
    for p in paths + lpaths:
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
