"""FramePlan mode (hipGraph replay of the per-frame launch sequence, lssvc_amd.intra.FramePlan): bit-identical to the
eager path over whole GOPs -- I-frame plan, first-P plan (feature_adaptor_I path, ref_feature_bl = None) and steady-P
plan -- including the call that captures and the replays after it, at two sizes and in both conv precisions; and the
host cost per frame it exists for."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(seed, gain):
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
    pnet.to(DEV).eval()
    return inet, pnet


def _code(inet, pnet, x_bl, x_el, H, W, gops, frames, lookahead=False, scale=2.0):
    """test.py's loop; returns per frame (bit_bl, bit_el, clones of the four DPB tensors + mv_hat).
    lookahead: the P-frames name the next frame's base-layer input (LSSVC_extend.forward_one_frame's look-ahead protocol)."""
    rows = []
    for _ in range(gops):
        dpb = None
        for t in range(frames):
            ahead = dict(next_x_bl=(x_bl[t + 1:t + 2] if t + 1 < frames else None), frame_id=t) if lookahead else {}
            inet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
            pnet.set_scale_information(scale, (H, W), (0, 0, 0, 0))
            if t == 0:
                r = inet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], None, None)
                dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
                extra = []
            else:
                r = pnet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], dpb, **ahead)
                dpb = r["dpb"]
                extra = [r["mv_hat"].clone(), r["warp_frame"].clone()]
            dpb["ref_frame_bl"].clamp_(0, 1)
            dpb["ref_frame_el"].clamp_(0, 1)
            rows.append((r["bit_bl"], r["bit_el"], [None if v is None else v.clone() for v in dpb.values()] + extra))
    return rows


@pytest.mark.parametrize("H,W,precision", [(128, 128, "f16x3"), (128, 256, "f16x3"), pytest.param(128, 128, "f32", marks=pytest.mark.slow)])
def test_graph_replay_is_bit_identical_to_eager(H, W, precision):
    from lssvc_amd import hip_ops
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    old = hip_ops.CONV_PRECISION
    try:
        hip_ops.set_conv_precision(precision)
        frames, gops = 5, 3
        clip = synth_clip(frames, H, W, seed=9).float() / 255.0
        x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
        inet, pnet = _nets(2, 0.6)
        eager = _code(inet, pnet, x_bl, x_el, H, W, 1, frames)
        inet.set_graph_mode(True)
        pnet.set_graph_mode(True)
        got = _code(inet, pnet, x_bl, x_el, H, W, gops, frames)      # GOP 0: eager warm-up calls + captures, GOP 1-2: replays
        assert any(p.graph is not None for p in pnet._plans.values()) and any(p.graph is not None for p in inet._plans.values())
        assert len(pnet._plans) == 2                                  # first-P and steady-P plans
        for i, (bb, be, tens) in enumerate(got):
            wb, we, wt = eager[i % frames]
            assert (bb, be) == (wb, we), (i, bb, wb, be, we)
            for a, b in zip(tens, wt):
                assert (a is None and b is None) or torch.equal(a, b), i
    finally:
        hip_ops.set_conv_precision(old)


@pytest.mark.parametrize("streams,alias", [(True, False), (True, True), (False, False), pytest.param(False, True, marks=pytest.mark.slow)])
def test_lookahead_base_layer_is_bit_identical(streams, alias):
    """BL(t+1) coded beside EL(t) (forward_one_frame's look-ahead protocol): bits and every DPB tensor of every frame equal the plain
    loop's, eager and replayed from the look-ahead plans, with and without side streams inside the layers."""
    from lssvc_amd import hip_ops
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    H, W, frames, gops = 128, 256, 6, 3
    old = hip_ops.MULTI_STREAM
    try:
        hip_ops.MULTI_STREAM = streams
        clip = synth_clip(frames, H, W, seed=5).float() / 255.0
        x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
        inet, pnet = _nets(3, 0.6)
        want = _code(inet, pnet, x_bl, x_el, H, W, 1, frames)
        got = _code(inet, pnet, x_bl, x_el, H, W, 1, frames, lookahead=True)
        inet.set_graph_mode(True, alias_outputs=alias)      # alias: the DPB the loop hands back IS the look-ahead plans' output
        pnet.set_graph_mode(True, alias_outputs=alias)      # memory, which the next EL then reads in place instead of loading it
        got += _code(inet, pnet, x_bl, x_el, H, W, gops, frames, lookahead=True)     # GOP 0: eager + captures, later: replays
        plans = [k for k in pnet._plans if str(k[0]).startswith("p-ahead")]         # BL(t+1): behind a whole frame + two parities; EL(t): two parities
        # (+ with alias: the first EL plan loads its references, the others are bound.) Single-stream mode captures them too since round 5:
        # what broke it in round 4 was a hipMemsetAsync recorded as a memset node of the frame plans (lssvc_fill_zero is a kernel now).
        assert len(plans) == (6 if alias else 5) and all(pnet._plans[k].graph is not None for k in plans), plans
        for i, (bb, be, tens) in enumerate(got):
            wb, we, wt = want[i % frames]
            assert (bb, be) == (wb, we), (i, bb, wb, be, we)
            for a, b in zip(tens, wt):
                assert (a is None and b is None) or torch.equal(a, b), i
    finally:
        hip_ops.MULTI_STREAM = old


@pytest.mark.parametrize("graph", [False, True])
def test_lookahead_serves_several_sizes_and_ratios_through_one_model(graph):
    """harness._load_nets keeps ONE LSSVC_extend per worker for every dataset and ratio of a run: the look-ahead protocol's
    persistent buffers and the plans that bake their addresses in are per geometry (BL size, EL size, scale), so sizes may alternate --
    same BL size under two EL sizes included -- and every frame equals the plain loop's, eager and replayed."""
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    frames = 4
    jobs = [(128, 256, 2.0, (64, 128)), (192, 192, 1.5, (128, 128)), (256, 256, 2.0, (128, 128)), (128, 256, 2.0, (64, 128)), (192, 192, 1.5, (128, 128))]
    clips = {}
    for H, W, scale, bl in jobs:
        if (H, W, scale) not in clips:
            clip = synth_clip(frames, H, W, seed=H + W).float() / 255.0
            clips[(H, W, scale)] = (imresize_bicubic(clip, bl).clamp_(0, 1).to(DEV), clip.to(DEV))
    inet, pnet = _nets(4, 0.6)
    want = {k: _code(inet, pnet, v[0], v[1], k[0], k[1], 1, frames, scale=k[2]) for k, v in clips.items()}
    inet.set_graph_mode(graph)
    pnet.set_graph_mode(graph)
    assert pnet.MAX_GEOMS >= 3
    # (round 6: the default plan cache holds three geometries' plan sets -- intra._HostModel.MAX_PLANS; rounds 4-5 had to raise it here)
    for rounds in range(3 if graph else 1):            # graph mode: eager first calls, captures, replays
        for H, W, scale, _ in jobs:
            x_bl, x_el = clips[(H, W, scale)]
            got = _code(inet, pnet, x_bl, x_el, H, W, 1, frames, lookahead=True, scale=scale)
            for i, (bb, be, tens) in enumerate(got):
                wb, we, wt = want[(H, W, scale)][i]
                assert (bb, be) == (wb, we), (H, W, scale, rounds, i, bb, wb, be, we)
                for a, b in zip(tens, wt):
                    assert (a is None and b is None) or torch.equal(a, b), (H, W, scale, rounds, i)
    assert len(pnet._lookahead_bufs) == 3


def test_graph_mode_host_time_per_frame():
    """What the plan is for: HOST time to put one frame on the stream (`last_issue_s`: entry of encode_decode to the
    point where every launch has been issued, before the D2H read of the bit counters). Eager: Python + ctypes per launch
    (~400 launches per P-frame); graph: input copies + one hipGraphLaunch. The frame's wall time at 256x256 is GPU-bound
    (~400 dependent small kernels) and does not change; it is printed for DESIGN.md."""
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    H = W = 256
    clip = synth_clip(2, H, W, seed=1).float() / 255.0
    x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
    inet, pnet = _nets(0, 0.55)
    inet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    pnet.set_scale_information(2.0, (H, W), (0, 0, 0, 0))

    def timed(n):
        r = inet.encode_decode(x_bl[0:1], x_el[0:1], None, None)
        dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
        for _ in range(4):
            dpb = pnet.encode_decode(x_bl[1:2], x_el[1:2], dpb)["dpb"]
        i_issue, i_wall = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inet.encode_decode(x_bl[0:1], x_el[0:1], None, None)
            i_wall.append(time.perf_counter() - t0)
            i_issue.append(inet.last_issue_s)
        torch.cuda.synchronize()
        issue, t0 = [], time.perf_counter()
        for _ in range(n):
            dpb = pnet.encode_decode(x_bl[1:2], x_el[1:2], dpb)["dpb"]        # includes the D2H read of the bit counters
            issue.append(pnet.last_issue_s)
        torch.cuda.synchronize()
        return sorted(issue)[n // 2], (time.perf_counter() - t0) / n, sorted(i_issue)[2], sorted(i_wall)[2]

    e_issue, e_wall, ei_issue, ei_wall = timed(20)
    inet.set_graph_mode(True)
    pnet.set_graph_mode(True)
    g_issue, g_wall, gi_issue, gi_wall = timed(20)
    print("256x256 P-frame: host issue eager %.2f ms -> graph %.2f ms; wall per frame eager %.2f ms, graph %.2f ms" %
          (e_issue * 1e3, g_issue * 1e3, e_wall * 1e3, g_wall * 1e3))
    print("256x256 I-frame (configs[0]): host issue eager %.2f ms -> graph %.2f ms; latency eager %.2f ms, graph %.2f ms" %
          (ei_issue * 1e3, gi_issue * 1e3, ei_wall * 1e3, gi_wall * 1e3))
    # (a plan with parallel branches costs the host ~2 ms to launch -- hipGraphLaunch submits every branch to its own
    # stream -- against ~0.2 ms for the single-stream I-frame plan; still a quarter of the eager issue time)
    assert g_issue < 4e-3 and g_issue < 0.5 * e_issue, (e_issue, g_issue)
    assert g_wall <= 1.25 * e_wall, (e_wall, g_wall)          # GPU-bound either way (measured equal); the margin is for a noisy box


@pytest.mark.parametrize("H,W,graph", [(128, 128, False), (128, 256, True), (384, 640, True)])
def test_parallel_branches_are_bit_identical_to_single_stream(H, W, graph):
    """Frame plans issue independent chains of a P-frame (EL SpyNet + reference pyramid || the BL codec; the BL-texture
    pyramid and the layer prior || the EL motion-vector codec; the temporal priors || the encoders) on side streams
    (hip_ops.Fork): same kernels, same per-chain order, so every bit count and every tensor must equal the single-stream
    run -- eager and as captured hipGraph branches, over three GOPs so that buffers recycled across frames and replays of
    the captured branches are covered. 384x640 is the smallest size that dispatches the persistent kernels."""
    from lssvc_amd import hip_ops
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    frames, gops = 4, 3
    clip = synth_clip(frames, H, W, seed=4).float() / 255.0
    x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
    old = hip_ops.MULTI_STREAM
    try:
        hip_ops.MULTI_STREAM = False
        inet, pnet = _nets(3, 0.55)
        want = _code(inet, pnet, x_bl, x_el, H, W, 1, frames)
        hip_ops.MULTI_STREAM = True
        inet, pnet = _nets(3, 0.55)
        inet.set_graph_mode(graph)
        pnet.set_graph_mode(graph)
        got = _code(inet, pnet, x_bl, x_el, H, W, gops, frames)
        torch.cuda.synchronize()
    finally:
        hip_ops.MULTI_STREAM = old
    for i, (bb, be, tens) in enumerate(got):
        wb, we, wt = want[i % frames]
        assert (bb, be) == (wb, we), (i, bb, wb, be, we)
        for a, b in zip(tens, wt):
            assert (a is None and b is None) or torch.equal(a, b), i
