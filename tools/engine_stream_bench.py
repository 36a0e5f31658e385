"""write_stream = 1 at BASELINE configs[1]'s size through the C++ plan runtime against the Python path, same box, same frames:
compile the six encoder / decoder plans of a 1152x1920 / 576x960 checkpoint, code 1 I + (n-1) P with real rANS layer files
through lssvc_engine_encode_* / lssvc_engine_decode_* (ctypes, device tensors, host byte buffers) and through
IntraSS / LSSVC_extend.encode_decode with bin paths; compares the files:  python tools/engine_stream_bench.py [frames]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler  # noqa: E402
from lssvc_amd._lib import lib, check  # noqa: E402
from lssvc_amd.prepost import FramePrep  # noqa: E402
from lssvc_amd.synth import synth_clip, synth_state_dict  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
        net.update(force=True)
    tmp = tempfile.mkdtemp(prefix="lssvc_splans_")

    # ---- the Python path (its own timers: LSSVC_net_extend.py:158-171), twice: the second pass is the warm one
    def python_pass(tag):
        dpb, enc, dec, dpbs = None, [], [], []
        for t in range(n):
            pb, pe = os.path.join(tmp, "%s_%d_BL.bin" % (tag, t)), os.path.join(tmp, "%s_%d_EL.bin" % (tag, t))
            if t == 0:
                r = inet.encode_decode(x_bl[t], x_el[t], pb, pe, h, w, H, W)
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            else:
                r = pnet.encode_decode(x_bl[t], x_el[t], dpb, pb, pe, W, H, w, h)
                dpb = r["dpb"]
                enc.append(r["encoding_time_BL"] + r["encoding_time_EL"])
                dec.append(r["decoding_time_BL"] + r["decoding_time_EL"])
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
            if t < 2:
                dpbs.append({k: (None if v is None else v.contiguous().clone()) for k, v in dpb.items()})
        return enc, dec, dpbs
    python_pass("py")
    enc_py, dec_py, dpbs = python_pass("py")

    # ---- compile
    names = ["i_enc", "i_dec", "p1_enc", "p1_dec", "p_enc", "p_dec"]
    paths = {k: os.path.join(tmp, k + ".plan") for k in names}
    t0 = time.time()
    print("I      ", plan_compiler.compile_iframe_stream(inet, x_bl[0], x_el[0], paths["i_enc"], paths["i_dec"])[:2])
    print("first P", plan_compiler.compile_pframe_stream(pnet, x_bl[1], x_el[1], dpbs[0], paths["p1_enc"], paths["p1_dec"])[:2])
    print("steady ", plan_compiler.compile_pframe_stream(pnet, x_bl[2], x_el[2], dpbs[1], paths["p_enc"], paths["p_dec"])[:2])
    print("compiled in %.1f s, files %.0f MB" % (time.time() - t0, sum(os.path.getsize(p) for p in paths.values()) / 1e6))
    del inet, pnet
    torch.cuda.empty_cache()

    eng = C.c_void_p(lib.lssvc_engine_create(0))
    for model, net in ((0, inet), (1, pnet)):                      # the raw checkpoints first: plans hold launches only
        table, n = net.W._ckpt()
        check(lib.lssvc_engine_load_checkpoint(eng, model, table, n))
    check(lib.lssvc_engine_load_stream(eng, *[paths[k].encode() for k in names]))
    check(lib.lssvc_engine_set_scale(eng, 2.0, H, W))
    P = lambda t: C.c_void_p(t.data_ptr())
    cap = 16 << 20
    f_bl, f_el = np.empty(cap, dtype=np.uint8), np.empty(cap, dtype=np.uint8)
    n_bl, n_el = C.c_int64(), C.c_int64()
    mk = lambda: {"ref_bl": torch.empty(1, 3, h, w, device=dev), "ref_el": torch.empty(1, 3, H, W, device=dev), "feat_bl": torch.empty(1, 64, h, w, device=dev),
                  "feat_el64": torch.empty(1, 64, H, W, device=dev), "feat_el48": torch.empty(1, 48, H, W, device=dev)}

    def engine_pass(tag, decode):
        cur, nxt, times = mk(), mk(), []
        for t in range(n):
            pb, pe = os.path.join(tmp, "eng_%d_BL.bin" % t), os.path.join(tmp, "eng_%d_EL.bin" % t)
            if decode:
                a, b = np.fromfile(pb, dtype=np.uint8), np.fromfile(pe, dtype=np.uint8)
                f_bl[:a.size], f_el[:b.size] = a, b
                n_bl.value, n_el.value = a.size, b.size
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if t == 0:
                if decode:
                    check(lib.lssvc_engine_decode_iframe(eng, f_bl.ctypes.data, n_bl, f_el.ctypes.data, n_el, P(nxt["ref_bl"]), P(nxt["ref_el"]), P(nxt["feat_el64"]), None))
                else:
                    check(lib.lssvc_engine_encode_iframe(eng, P(x_bl[t]), P(x_el[t]), f_bl.ctypes.data, cap, C.byref(n_bl), f_el.ctypes.data, cap, C.byref(n_el),
                                                         P(nxt["ref_bl"]), P(nxt["ref_el"]), P(nxt["feat_el64"]), None))
            else:
                fe = cur["feat_el64"] if t == 1 else cur["feat_el48"]
                fb = P(cur["feat_bl"]) if t > 1 else None
                if decode:
                    check(lib.lssvc_engine_decode_pframe(eng, f_bl.ctypes.data, n_bl, f_el.ctypes.data, n_el, P(cur["ref_bl"]), P(cur["ref_el"]), fb, P(fe),
                                                         P(nxt["ref_bl"]), P(nxt["feat_bl"]), P(nxt["ref_el"]), P(nxt["feat_el48"]), None))
                else:
                    check(lib.lssvc_engine_encode_pframe(eng, P(x_bl[t]), P(x_el[t]), P(cur["ref_bl"]), P(cur["ref_el"]), fb, P(fe), f_bl.ctypes.data, cap,
                                                         C.byref(n_bl), f_el.ctypes.data, cap, C.byref(n_el), P(nxt["ref_bl"]), P(nxt["feat_bl"]), P(nxt["ref_el"]),
                                                         P(nxt["feat_el48"]), None))
            dt = time.perf_counter() - t0                                   # (the entry points return after their stream has drained)
            if t > 0:
                times.append(dt)
            if not decode:
                f_bl[:n_bl.value].tofile(pb)
                f_el[:n_el.value].tofile(pe)
            cur, nxt = nxt, cur
            check(lib.lssvc_clamp_inplace(P(cur["ref_bl"]), cur["ref_bl"].numel(), 0.0, 1.0, None))
            check(lib.lssvc_clamp_inplace(P(cur["ref_el"]), cur["ref_el"].numel(), 0.0, 1.0, None))
        return times, cur
    engine_pass("eng", False)
    enc_eng, last_e = engine_pass("eng", False)
    engine_pass("eng", True)
    dec_eng, last_d = engine_pass("eng", True)
    same_files = all(open(os.path.join(tmp, "py_%d_%s.bin" % (t, g)), "rb").read() == open(os.path.join(tmp, "eng_%d_%s.bin" % (t, g)), "rb").read()
                     for t in range(n) for g in ("BL", "EL"))
    same_recon = torch.equal(last_e["ref_el"], last_d["ref_el"]) and torch.equal(last_e["feat_el48"], last_d["feat_el48"])
    ms = lambda v: 1e3 * sum(v) / len(v)
    print("per P-frame, write_stream=1 at 1152x1920 / 576x960 (%d P-frames): python path enc %.1f ms dec %.1f ms; C++ plan runtime enc %.1f ms dec %.1f ms"
          % (n - 1, ms(enc_py), ms(dec_py), ms(enc_eng), ms(dec_eng)))
    print("layer files equal the Python path's byte for byte:", same_files, " decoder-side DPB equals encoder-side:", same_recon)
    lib.lssvc_engine_destroy(eng)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
