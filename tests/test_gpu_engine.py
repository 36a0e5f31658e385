"""The engine entry points (include/lssvc_hip.h "engine", csrc/plan_runtime.cpp) from a caller WITHOUT Python: frame plans
compiled here by the Python front end (lssvc_amd/plan_compiler.py), then tests/engine_demo.c -- plain C, gcc, linked with
liblssvc_hip.so only -- runs as its own process, codes I + P + P + P + P + P and writes what the engine returned. Every
bit count and every tensor must equal the Python path's bit for bit (same kernels, same arguments, same order; the engine's
first call of a plan is eager, the second captures a hipGraph inside the library, later ones replay it), and the first
three frames are the golden case x2_128_ipp, held to the north-star bars against the REFERENCE's stored outputs."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from helpers import load_case, write_checkpoint_blob

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_codes_a_gop_through_compiled_plans(tmp_path):
    from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler
    from lssvc_amd.preprocess import psnr
    from lssvc_amd.synth import synth_state_dict
    z, m = load_case("x2_128_ipp")
    H, W, h, w = m["H"], m["W"], m["h"], m["w"]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
    pnet.to(DEV).eval()
    order = [0, 1, 2, 1, 2, 1]                                     # six frames: the golden's three, then three more P-frames
    x_el = [(torch.from_numpy(z["x_el_u8"][t:t + 1]).float() / 255.0).to(DEV) for t in order]
    x_bl = [torch.from_numpy(z["x_bl"][t:t + 1]).to(DEV) for t in order]
    for net in (inet, pnet):
        net.set_scale_information(m["scale"], (H, W), (0, 0, 0, 0))

    # ---- the Python path: expected results, and the DPBs the plans are compiled from
    want, dpbs, dpb = [], [], None
    for t in range(len(order)):
        if t == 0:
            r = inet.encode_decode(x_bl[t], x_el[t], None, None, h, w, H, W)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            extra = []
        else:
            r = pnet.encode_decode(x_bl[t], x_el[t], dpb, None, None, W, H, w, h)
            dpb = r["dpb"]
            extra = [dpb["ref_feature_bl"], r["mv_hat"], r["warp_frame"]]
        want.append((r["bit_bl"], r["bit_el"], [v.contiguous().clone() for v in (dpb["ref_frame_bl"], dpb["ref_frame_el"], dpb["ref_feature_el"])]
                     + [v.contiguous().clone() for v in extra]))
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        dpbs.append({k: (None if v is None else v.contiguous().clone()) for k, v in dpb.items()})

    # ---- compile: one plan per frame type
    plans = [str(tmp_path / n) for n in ("iframe.plan", "first_p.plan", "steady_p.plan")]
    info_i, _ = plan_compiler.compile_iframe(inet, x_bl[0], x_el[0], plans[0])
    info_1, _ = plan_compiler.compile_pframe(pnet, x_bl[1], x_el[1], dpbs[0], plans[1])
    info_2, _ = plan_compiler.compile_pframe(pnet, x_bl[2], x_el[2], dpbs[1], plans[2])
    print("plans:", info_i, info_1, info_2)
    assert info_1["streams"] >= 2 and info_2["launches"] > 300       # side-stream branches are part of the P plans
    # a plan holds launches only: every weight tensor is a recipe over the raw checkpoint, and the file is small
    for info, path in ((info_i, plans[0]), (info_1, plans[1]), (info_2, plans[2])):
        assert info["weight_recipes"] > 50 and info["embedded_weight_bytes"] == 0, info
        assert os.path.getsize(path) <= 5 * 2 ** 20, (path, os.path.getsize(path))
    ckpts = [str(tmp_path / "intra.ckpt"), str(tmp_path / "inter.ckpt")]
    write_checkpoint_blob(synth_state_dict("intra_ss", m["seed"], m["gain"]), ckpts[0])
    write_checkpoint_blob(synth_state_dict("lssvc_extend", m["seed"], m["gain"]), ckpts[1])

    # ---- the C program, in its own process
    case, outp, exe = str(tmp_path / "case.bin"), str(tmp_path / "out.bin"), str(tmp_path / "engine_demo")
    with open(case, "wb") as f:
        f.write(struct.pack("<5if", len(order), H, W, h, w, m["scale"]))
        for t in range(len(order)):
            f.write(x_bl[t].cpu().contiguous().numpy().tobytes())
            f.write(x_el[t].cpu().contiguous().numpy().tobytes())
    libdir = os.path.join(ROOT, "lssvc_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "engine_demo.c"),
                           "-L", libdir, "-llssvc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([exe] + ckpts + plans + [case, outp], capture_output=True, text=True, env=env, timeout=600)
    print(res.stdout, res.stderr)
    assert res.returncode == 0, res.stderr

    # ---- compare
    raw = np.fromfile(outp, dtype=np.uint8)
    pos = 0

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape)) * 4
        a = torch.from_numpy(raw[pos:pos + n].view(np.float32).reshape(shape).copy())
        pos += n
        return a

    for t in range(len(order)):
        bits = raw[pos:pos + 16].view(np.float64)
        pos += 16
        shapes = [(1, 3, h, w), (1, 3, H, W), (1, 64 if t == 0 else 48, H, W)] + ([(1, 64, h, w), (1, 2, H, W), (1, 3, H, W)] if t else [])
        got = [take(s) for s in shapes]
        wb, we, tens = want[t]
        assert (float(bits[0]), float(bits[1])) == (wb, we), (t, bits, wb, we)
        for g, x in zip(got, tens):
            assert torch.equal(g, x.cpu()), (t, tuple(g.shape), (g - x.cpu()).abs().max().item())
        if t < 3:                                                  # the golden frames: the reference's own numbers
            ref_bits = z["f%d_bits" % t]
            assert abs(bits[0] - ref_bits[0]) / (h * w) <= 1e-5 and abs(bits[1] - ref_bits[1]) / (H * W) <= 1e-5
            p_el = psnr(x_el[t].cpu(), got[1].clamp(0, 1))
            assert abs(p_el - z["f%d_psnr" % t][1]) <= 1e-4
    assert pos == raw.size


def test_c_program_codes_a_gop_with_the_base_layer_a_frame_ahead(tmp_path):
    """Round 6: the look-ahead of LSSVC_extend.forward_one_frame(next_x_bl=...) for a caller without Python. The
    P-frame is compiled as a base-layer plan and an enhancement-layer plan per frame type (plan_compiler.compile_pframe_layers);
    tests/engine_demo.c, given those four plans, codes I + 6 P through lssvc_engine_pframe_lookahead -- BL(t+1) on the engine's second
    stream beside EL(t) -- and a second run of the same program codes the same clip through lssvc_engine_pframe. Both outputs must
    equal the Python path's, bit for bit: every bit count, every DPB tensor, mv_hat and warp_frame of every frame (the first call of
    a plan is eager, the second captures, later ones replay: seven frames pass through all three)."""
    from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler
    from lssvc_amd.synth import synth_state_dict
    z, m = load_case("x2_128_ipp")
    H, W, h, w = m["H"], m["W"], m["h"], m["w"]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
    pnet.to(DEV).eval()
    order = [0, 1, 2, 1, 2, 1, 2]
    x_el = [(torch.from_numpy(z["x_el_u8"][t:t + 1]).float() / 255.0).to(DEV) for t in order]
    x_bl = [torch.from_numpy(z["x_bl"][t:t + 1]).to(DEV) for t in order]
    for net in (inet, pnet):
        net.set_scale_information(m["scale"], (H, W), (0, 0, 0, 0))
    want, dpbs, dpb = [], [], None
    for t in range(len(order)):
        if t == 0:
            r = inet.encode_decode(x_bl[t], x_el[t], None, None, h, w, H, W)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
            extra = []
        else:
            r = pnet.encode_decode(x_bl[t], x_el[t], dpb, None, None, W, H, w, h)
            dpb = r["dpb"]
            extra = [dpb["ref_feature_bl"], r["mv_hat"], r["warp_frame"]]
        want.append((r["bit_bl"], r["bit_el"], [v.contiguous().clone() for v in (dpb["ref_frame_bl"], dpb["ref_frame_el"], dpb["ref_feature_el"])]
                     + [v.contiguous().clone() for v in extra]))
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        dpbs.append({k: (None if v is None else v.contiguous().clone()) for k, v in dpb.items()})
    plans = [str(tmp_path / n) for n in ("iframe.plan", "first_p.plan", "steady_p.plan", "bl_first.plan", "bl_steady.plan", "el_first.plan", "el_steady.plan")]
    plan_compiler.compile_iframe(inet, x_bl[0], x_el[0], plans[0])
    plan_compiler.compile_pframe(pnet, x_bl[1], x_el[1], dpbs[0], plans[1])
    plan_compiler.compile_pframe(pnet, x_bl[2], x_el[2], dpbs[1], plans[2])
    i_bl1, i_el1, _, _ = plan_compiler.compile_pframe_layers(pnet, x_bl[1], x_el[1], dpbs[0], plans[3], plans[5])
    i_bl2, i_el2, _, _ = plan_compiler.compile_pframe_layers(pnet, x_bl[2], x_el[2], dpbs[1], plans[4], plans[6])
    print("layer plans:", i_bl1, i_el1, i_bl2, i_el2)
    assert i_bl2["launches"] + i_el2["launches"] > 300 and i_bl2["launches"] > 50
    ckpts = [str(tmp_path / "intra.ckpt"), str(tmp_path / "inter.ckpt")]
    write_checkpoint_blob(synth_state_dict("intra_ss", m["seed"], m["gain"]), ckpts[0])
    write_checkpoint_blob(synth_state_dict("lssvc_extend", m["seed"], m["gain"]), ckpts[1])
    case, exe = str(tmp_path / "case.bin"), str(tmp_path / "engine_demo")
    with open(case, "wb") as f:
        f.write(struct.pack("<5if", len(order), H, W, h, w, m["scale"]))
        for t in range(len(order)):
            f.write(x_bl[t].cpu().contiguous().numpy().tobytes())
            f.write(x_el[t].cpu().contiguous().numpy().tobytes())
    libdir = os.path.join(ROOT, "lssvc_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "engine_demo.c"),
                           "-L", libdir, "-llssvc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    outs = {}
    for mode, extra_plans in (("whole frames", []), ("base layer a frame ahead", plans[3:])):
        outp = str(tmp_path / ("out_%d.bin" % len(extra_plans)))
        res = subprocess.run([exe] + ckpts + plans[:3] + [case, outp] + extra_plans, capture_output=True, text=True, env=env, timeout=600)
        print(mode, res.stdout[-600:], res.stderr)
        assert res.returncode == 0, res.stderr
        outs[mode] = np.fromfile(outp, dtype=np.uint8)
    assert np.array_equal(outs["whole frames"], outs["base layer a frame ahead"]), "the look-ahead entry point does not reproduce lssvc_engine_pframe"
    raw, pos = outs["base layer a frame ahead"], 0
    for t in range(len(order)):
        bits = raw[pos:pos + 16].view(np.float64)
        pos += 16
        shapes = [(1, 3, h, w), (1, 3, H, W), (1, 64 if t == 0 else 48, H, W)] + ([(1, 64, h, w), (1, 2, H, W), (1, 3, H, W)] if t else [])
        wb, we, tens = want[t]
        assert (float(bits[0]), float(bits[1])) == (wb, we), (t, bits, wb, we)
        for shp, x in zip(shapes, tens):
            n = int(np.prod(shp)) * 4
            g = torch.from_numpy(raw[pos:pos + n].view(np.float32).reshape(shp).copy())
            pos += n
            assert torch.equal(g, x.cpu()), (t, shp, (g - x.cpu()).abs().max().item())
    assert pos == raw.size


def test_c_programs_write_and_read_real_bitstreams(tmp_path):
    """write_stream = 1 through the engine: encoder and decoder plans compiled by the front end
    (plan_compiler.compile_iframe_stream / compile_pframe_stream), then tests/engine_stream_demo.c -- plain C -- runs TWICE as
    separate processes: `enc` codes I + P + P + P + P into the reference's layer files, `dec` (a fresh process that is given only
    the decoder plans and those files) reconstructs them. The files must be byte for byte what the Python path writes for the
    same frames, and encoder-side, decoder-side and Python reconstructions must be identical bit for bit."""
    from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler
    from lssvc_amd.synth import synth_state_dict
    z, m = load_case("x2_128_ipp")
    H, W, h, w = m["H"], m["W"], m["h"], m["w"]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", m["seed"], m["gain"]))
    pnet.to(DEV).eval()
    order = [0, 1, 2, 1, 2]
    x_el = [(torch.from_numpy(z["x_el_u8"][t:t + 1]).float() / 255.0).to(DEV) for t in order]
    x_bl = [torch.from_numpy(z["x_bl"][t:t + 1]).to(DEV) for t in order]
    for net in (inet, pnet):
        net.set_scale_information(m["scale"], (H, W), (0, 0, 0, 0))
        net.update(force=True)

    # ---- the Python path with real bitstreams: expected files and reconstructions, and the DPBs the plans are compiled from
    py = tmp_path / "py"
    py.mkdir()
    want, dpbs, dpb = [], [], None
    for t in range(len(order)):
        pb, pe = str(py / ("%d_BL.bin" % t)), str(py / ("%d_EL.bin" % t))
        if t == 0:
            r = inet.encode_decode(x_bl[t], x_el[t], pb, pe, h, w, H, W)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
        else:
            r = pnet.encode_decode(x_bl[t], x_el[t], dpb, pb, pe, W, H, w, h)
            dpb = r["dpb"]
        want.append([v.contiguous().clone().cpu() for v in (dpb["ref_frame_bl"], dpb["ref_frame_el"], dpb["ref_feature_el"])]
                    + ([dpb["ref_feature_bl"].contiguous().clone().cpu()] if t else []))
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        dpbs.append({k: (None if v is None else v.contiguous().clone()) for k, v in dpb.items()})

    # ---- compile the six halves
    d = tmp_path / "eng"
    d.mkdir()
    ie, idc, s_i = plan_compiler.compile_iframe_stream(inet, x_bl[0], x_el[0], str(d / "i_enc.plan"), str(d / "i_dec.plan"))
    e1, d1, s_1 = plan_compiler.compile_pframe_stream(pnet, x_bl[1], x_el[1], dpbs[0], str(d / "p1_enc.plan"), str(d / "p1_dec.plan"))
    e2, d2, s_2 = plan_compiler.compile_pframe_stream(pnet, x_bl[2], x_el[2], dpbs[1], str(d / "p_enc.plan"), str(d / "p_dec.plan"))
    print("stream plans:", ie, idc, e1, d1, e2, d2)
    write_checkpoint_blob(synth_state_dict("intra_ss", m["seed"], m["gain"]), str(d / "intra.ckpt"))
    write_checkpoint_blob(synth_state_dict("lssvc_extend", m["seed"], m["gain"]), str(d / "inter.ckpt"))
    for info in (ie, idc, e1, d1, e2, d2):                        # what a stream plan still embeds: the bottleneck medians of update()'s tables
        assert info["weight_recipes"] > 20 and info["embedded_weight_bytes"] <= 4096, info
    assert ie["host_steps"] >= 8 and d2["host_steps"] >= 20 and e2["tables"] >= 3
    assert len(s_i) == 4 and len(s_1) == 2 and len(s_2) == 2

    # ---- the C program: encoder process, then decoder process
    case, exe = str(tmp_path / "case.bin"), str(tmp_path / "engine_stream_demo")
    with open(case, "wb") as f:
        f.write(struct.pack("<5if", len(order), H, W, h, w, m["scale"]))
        for t in range(len(order)):
            f.write(x_bl[t].cpu().contiguous().numpy().tobytes())
            f.write(x_el[t].cpu().contiguous().numpy().tobytes())
    libdir = os.path.join(ROOT, "lssvc_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "engine_stream_demo.c"),
                           "-L", libdir, "-llssvc_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    for role in ("enc", "dec"):
        res = subprocess.run([exe, role, str(d), case], capture_output=True, text=True, env=env, timeout=600)
        print(res.stdout, res.stderr)
        assert res.returncode == 0, res.stderr

    # ---- compare: files, then tensors
    for t in range(len(order)):
        for tag in ("BL", "EL"):
            a = open(str(py / ("%d_%s.bin" % (t, tag))), "rb").read()
            b = open(str(d / ("%d_%s.bin" % (t, tag))), "rb").read()
            assert a == b, (t, tag, len(a), len(b))
    outs = {}
    for role in ("enc", "dec"):
        raw = np.fromfile(str(d / (role + ".out")), dtype=np.float32)
        pos, frames = 0, []
        for t in range(len(order)):
            shapes = [(1, 3, h, w), (1, 3, H, W), (1, 64 if t == 0 else 48, H, W)] + ([(1, 64, h, w)] if t else [])
            got = []
            for sh in shapes:
                n = int(np.prod(sh))
                got.append(torch.from_numpy(raw[pos:pos + n].reshape(sh).copy()))
                pos += n
            frames.append(got)
        assert pos == raw.size
        outs[role] = frames
    for t in range(len(order)):
        for k, (e, dd, x) in enumerate(zip(outs["enc"][t], outs["dec"][t], want[t])):
            assert torch.equal(e, dd), (t, k, "encoder-side and decoder-side reconstructions differ")
            assert torch.equal(dd, x), (t, k, (dd - x).abs().max().item())


def test_plan_with_inter_layer_padding_replays_through_the_engine(tmp_path):
    """A plan compiled with a NON-ZERO pad_size (set_scale_information's inter-layer padding, get_depadded_feature:
    IntraSS.py:124-147): the crop / pad of the BL texture and latent is a library launch (lssvc_pad_crop), so the recorder
    sees it and the engine -- driven here through ctypes, no model object involved in the replay -- returns the Python
    path's I-frame bit for bit; the header stores the padding the plan was compiled for (lssvc_engine_plan_meta) and the fp32
    layer set of the range audit. Golden case x2_128_ip_depad holds the Python path itself to the reference."""
    import ctypes as C
    from lssvc_amd import IntraSS, plan_compiler, _lib
    from lssvc_amd.synth import synth_state_dict
    z, m = load_case("x2_128_ip_depad")
    assert any(m["pad"])
    H, W, h, w = m["H"], m["W"], m["h"], m["w"]
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", m["seed"], m["gain"])).to(DEV).eval()
    inet.set_scale_information(m["scale"], (H, W), m["pad"])
    x_el = (torch.from_numpy(z["x_el_u8"][0:1]).float() / 255.0).to(DEV)
    x_bl = torch.from_numpy(z["x_bl"][0:1]).to(DEV)
    r = inet.encode_decode(x_bl, x_el, None, None, h, w, H, W)
    path = str(tmp_path / "iframe_pad.plan")
    info, _ = plan_compiler.compile_iframe(inet, x_bl, x_el, path)
    lib = _lib.lib
    eng = lib.lssvc_engine_create(0)
    assert eng
    try:
        assert lib.lssvc_engine_load_intra(eng, path.encode()) != 0            # no checkpoint yet: a clean error
        assert b"lssvc_engine_load_checkpoint" in lib.lssvc_last_error()
        table, n = inet.W._ckpt()                                              # the raw tensors, as the library takes them
        _lib.check(lib.lssvc_engine_load_checkpoint(eng, 0, table, n))
        _lib.check(lib.lssvc_engine_load_intra(eng, path.encode()))
        _lib.check(lib.lssvc_engine_set_scale(eng, C.c_float(m["scale"]), H, W))
        v = C.c_int64()
        for name, want in zip(("pad_left", "pad_right", "pad_top", "pad_bottom"), m["pad"]):
            _lib.check(lib.lssvc_engine_plan_meta(eng, 0, name.encode(), C.byref(v)))
            assert v.value == want, (name, v.value, want)
        _lib.check(lib.lssvc_engine_plan_meta(eng, 0, b"f32_layers_n", C.byref(v)))
        assert v.value == 0
        assert lib.lssvc_engine_plan_meta(eng, 0, b"no_such_entry", C.byref(v)) != 0
        outs = [torch.empty(1, 3, h, w, device=DEV), torch.empty(1, 3, H, W, device=DEV), torch.empty(1, 64, H, W, device=DEV)]
        bits = (C.c_double * 2)()
        for _ in range(3):                                          # eager, capture, replay
            _lib.check(lib.lssvc_engine_iframe(eng, C.c_void_p(x_bl.data_ptr()), C.c_void_p(x_el.data_ptr()), bits,
                                               *[C.c_void_p(o.data_ptr()) for o in outs], None))
            torch.cuda.synchronize()
            assert (bits[0], bits[1]) == (r["bit_bl"], r["bit_el"])
            for o, k in zip(outs, ("x_hat_bl", "x_hat_el", "feature_el")):
                assert torch.equal(o, r[k].contiguous()), k
    finally:
        lib.lssvc_engine_destroy(eng)


def test_a_plan_serves_any_checkpoint_of_the_architecture(tmp_path):
    """A plan holds launches and weight RECIPES, no weights: compiled from one checkpoint it codes with another (handed to
    lssvc_engine_load_checkpoint as a raw state dict), and gives that checkpoint's Python-path result bit for bit -- including
    the per-layer power-of-two prescale of the fp16 weight planes, which is a function of the checkpoint and is patched into the
    launch descriptors when the plan is bound. I-frame and first P-frame at 128x128 / 64x64."""
    import ctypes as C
    from lssvc_amd import IntraSS, LSSVC_extend, plan_compiler, _lib
    from lssvc_amd.synth import synth_state_dict
    g = torch.Generator().manual_seed(12)
    H = W = 128
    x_el = [torch.rand(1, 3, H, W, generator=g).to(DEV) for _ in range(2)]
    x_bl = [torch.rand(1, 3, H // 2, W // 2, generator=g).to(DEV) for _ in range(2)]

    def nets(seed, gain):
        inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
        pnet.to(DEV).eval()
        for net in (inet, pnet):
            net.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        return inet, pnet

    def python_path(inet, pnet):
        r = inet.encode_decode(x_bl[0], x_el[0], None, None, H // 2, W // 2, H, W)
        dpb = {"ref_frame_bl": r["x_hat_bl"].clone().clamp_(0, 1), "ref_frame_el": r["x_hat_el"].clone().clamp_(0, 1), "ref_feature_bl": None,
               "ref_feature_el": r["feature_el"]}
        q = pnet.encode_decode(x_bl[1], x_el[1], dpb, None, None, W, H, W // 2, H // 2)
        return r, dpb, q

    a_i, a_p = nets(3, 0.6)                                       # the checkpoint the plans are compiled from
    _, dpb_a, _ = python_path(a_i, a_p)
    plans = [str(tmp_path / "i.plan"), str(tmp_path / "p1.plan")]
    plan_compiler.compile_iframe(a_i, x_bl[0], x_el[0], plans[0])
    plan_compiler.compile_pframe(a_p, x_bl[1], x_el[1], {k: (None if v is None else v.contiguous()) for k, v in dpb_a.items()}, plans[1])
    b_i, b_p = nets(4, 0.45)                                      # another checkpoint: other weights, other magnitudes
    want_i, dpb_b, want_p = python_path(b_i, b_p)
    assert want_i["bit_el"] != python_path(a_i, a_p)[0]["bit_el"]
    lib = _lib.lib
    eng = lib.lssvc_engine_create(0)
    try:
        for model, net in ((0, b_i), (1, b_p)):
            table, n = net.W._ckpt()
            _lib.check(lib.lssvc_engine_load_checkpoint(eng, model, table, n))
        _lib.check(lib.lssvc_engine_load_intra(eng, plans[0].encode()))
        P = lambda t: C.c_void_p(t.data_ptr())
        bits = (C.c_double * 2)()
        oi = [torch.empty(1, 3, H // 2, W // 2, device=DEV), torch.empty(1, 3, H, W, device=DEV), torch.empty(1, 64, H, W, device=DEV)]
        _lib.check(lib.lssvc_engine_iframe(eng, P(x_bl[0]), P(x_el[0]), bits, *[P(o) for o in oi], None))
        torch.cuda.synchronize()
        assert (bits[0], bits[1]) == (want_i["bit_bl"], want_i["bit_el"])
        for o, k in zip(oi, ("x_hat_bl", "x_hat_el", "feature_el")):
            assert torch.equal(o, want_i[k].contiguous()), k
    finally:
        lib.lssvc_engine_destroy(eng)
    # the P-frame plan through its own engine (load_inter wants both P plans; the first-P plan alone goes through the stream-less
    # loader of a second engine with the same checkpoint)
    eng = lib.lssvc_engine_create(0)
    try:
        table, n = b_p.W._ckpt()
        _lib.check(lib.lssvc_engine_load_checkpoint(eng, 1, table, n))
        steady = str(tmp_path / "p.plan")
        plan_compiler.compile_pframe(a_p, x_bl[1], x_el[1], {k: v.contiguous() for k, v in python_path(a_i, a_p)[2]["dpb"].items()}, steady)
        _lib.check(lib.lssvc_engine_load_inter(eng, plans[1].encode(), steady.encode()))
        op = [torch.empty(1, 3, H // 2, W // 2, device=DEV), torch.empty(1, 64, H // 2, W // 2, device=DEV), torch.empty(1, 3, H, W, device=DEV),
              torch.empty(1, 48, H, W, device=DEV), torch.empty(1, 2, H, W, device=DEV), torch.empty(1, 3, H, W, device=DEV)]
        d = {k: v.contiguous() for k, v in dpb_b.items() if v is not None}
        _lib.check(lib.lssvc_engine_pframe(eng, P(x_bl[1]), P(x_el[1]), P(d["ref_frame_bl"]), P(d["ref_frame_el"]), None, P(d["ref_feature_el"]), bits,
                                           *[P(o) for o in op], None))
        torch.cuda.synchronize()
        assert (bits[0], bits[1]) == (want_p["bit_bl"], want_p["bit_el"])
        for o, x in zip(op, (want_p["dpb"]["ref_frame_bl"], want_p["dpb"]["ref_feature_bl"], want_p["dpb"]["ref_frame_el"], want_p["dpb"]["ref_feature_el"],
                             want_p["mv_hat"], want_p["warp_frame"])):
            assert torch.equal(o, x.contiguous())
    finally:
        lib.lssvc_engine_destroy(eng)


def test_a_stream_plan_refuses_another_checkpoint(tmp_path):
    """a write_stream plan embeds the CDF tables (and bottleneck medians) that update() built from the checkpoint it was
    compiled with. Bound to another checkpoint it would code that checkpoint's latents against the wrong tables -- strings no other
    decoder could read. The plan carries a CRC of the entropy parameters (plan_compiler.entropy_params_crc) and the engine compares
    it with the checkpoint it was given: the same checkpoint loads, another one is an error that says so."""
    import ctypes as C
    from lssvc_amd import IntraSS, plan_compiler, _lib
    from lssvc_amd.synth import synth_state_dict
    g = torch.Generator().manual_seed(3)
    H = W = 128
    x_el, x_bl = torch.rand(1, 3, H, W, generator=g).to(DEV), torch.rand(1, 3, H // 2, W // 2, generator=g).to(DEV)

    def net(seed, gain):
        n = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
        n.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
        n.update(force=True)
        return n

    a, b = net(3, 0.6), net(4, 0.45)
    enc, dec = str(tmp_path / "i_enc.plan"), str(tmp_path / "i_dec.plan")
    plan_compiler.compile_iframe_stream(a, x_bl, x_el, enc, dec)
    lib = _lib.lib
    for other, ok in ((a, True), (b, False)):
        eng = lib.lssvc_engine_create(0)
        try:
            table, n = other.W._ckpt()
            _lib.check(lib.lssvc_engine_load_checkpoint(eng, 0, table, n))
            rc = lib.lssvc_engine_load_stream(eng, enc.encode(), dec.encode(), None, None, None, None)
            if ok:
                _lib.check(rc)
            else:
                assert rc != 0 and "another checkpoint's entropy parameters" in lib.lssvc_last_error().decode()
        finally:
            lib.lssvc_engine_destroy(eng)


def test_the_engine_audits_the_fp16_range_of_a_checkpoint_it_was_not_compiled_with(tmp_path):
    """a plan bakes in the kernel choice of the fp16 range audit of the checkpoint it was compiled from, and the engine binds
    it to any checkpoint of the architecture. The first frame a plan codes is therefore audited in the engine too (max |x| of every input
    of its f16x3 launches against the limits of hip_ops.RangeAudit): a checkpoint whose activations leave the range is an error that
    says what to do, not a silently saturated frame; the checkpoint the plan came from passes."""
    import ctypes as C
    from lssvc_amd import IntraSS, plan_compiler, _lib
    from lssvc_amd.synth import synth_state_dict
    g = torch.Generator().manual_seed(5)
    H = W = 128
    x_el, x_bl = torch.rand(1, 3, H, W, generator=g).to(DEV), torch.rand(1, 3, H // 2, W // 2, generator=g).to(DEV)
    sd = synth_state_dict("intra_ss", 3, 0.6)
    a = IntraSS.from_state_dict(sd).to(DEV).eval()
    a.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    plan = str(tmp_path / "i.plan")
    plan_compiler.compile_iframe(a, x_bl, x_el, plan)
    loud = dict(sd)
    loud["base_layer_model.g_a.0.conv1.weight"] = sd["base_layer_model.g_a.0.conv1.weight"] * 3.0e6      # its output feeds the next f16x3 conv
    b = IntraSS.from_state_dict(loud)                                                                      # (host side only: the table of raw tensors)
    from lssvc_amd.weights import WeightStore
    lib = _lib.lib
    P = lambda t: C.c_void_p(t.data_ptr())
    for name, store, ok in (("same", a.W, True), ("loud", WeightStore(b._sd, torch.device(DEV)), False)):
        eng = lib.lssvc_engine_create(0)
        try:
            table, n = store._ckpt()
            _lib.check(lib.lssvc_engine_load_checkpoint(eng, 0, table, n))
            _lib.check(lib.lssvc_engine_load_intra(eng, plan.encode()))
            bits = (C.c_double * 2)()
            oi = [torch.empty(1, 3, H // 2, W // 2, device=DEV), torch.empty(1, 3, H, W, device=DEV), torch.empty(1, 64, H, W, device=DEV)]
            rc = lib.lssvc_engine_iframe(eng, P(x_bl), P(x_el), bits, *[P(o) for o in oi], None)
            torch.cuda.synchronize()
            if ok:
                _lib.check(rc)
            else:
                msg = lib.lssvc_last_error().decode()
                assert rc != 0 and "outside what its fp16-split kernel can hold" in msg and "compile the plans from this one" in msg, (rc, msg)
                # the failure is STICKY and nothing is handed out -- a caller that carries on gets the same error on every later
                # call of the plan (round 5 counted the failed run, so the second call skipped the audit, captured the plan and returned 0)
                for o in oi:
                    o.fill_(-7.0)
                for _ in range(3):
                    rc2 = lib.lssvc_engine_iframe(eng, P(x_bl), P(x_el), bits, *[P(o) for o in oi], None)
                    torch.cuda.synchronize()
                    assert rc2 != 0 and "outside what its fp16-split kernel can hold" in lib.lssvc_last_error().decode(), rc2
                    assert all(bool((o == -7.0).all()) for o in oi), "a failed audit must not write the caller's outputs"
        finally:
            lib.lssvc_engine_destroy(eng)
