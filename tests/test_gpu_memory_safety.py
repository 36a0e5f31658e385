"""Memory safety of the hand-written kernels as the frame bodies use them (no GPU sanitizer on this pool): every activation buffer of
an I-frame and of P-frames is laid between two sentinel zones that must come back untouched, and the same frames are coded with
every fresh activation buffer pre-filled with NaN, which must not change a single bit count or output value -- an out-of-bounds write
or a read of memory no launch has written would show in one or the other. Both stream modes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup():
    from lssvc_amd import IntraSS, LSSVC_extend
    from lssvc_amd.synth import synth_state_dict, synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    H, W, frames = 128, 256, 3
    clip = synth_clip(frames, H, W, seed=5).float() / 255.0
    x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 3, 0.6)).to(DEV).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 3, 0.6))
    pnet.to(DEV).eval()
    inet.range_audit = pnet.range_audit = False          # (its dry runs would allocate and free the same buffers again)
    for n in (inet, pnet):
        n.set_scale_information(2.0, (H, W), (0, 0, 0, 0))
    return inet, pnet, x_bl, x_el, frames


def _code(inet, pnet, x_bl, x_el, frames, per_frame=None):
    rows, dpb = [], None
    for t in range(frames):
        if t == 0:
            r = inet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], None, None)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
        else:
            r = pnet.encode_decode(x_bl[t:t + 1], x_el[t:t + 1], dpb)
            dpb = r["dpb"]
        if per_frame is not None:
            per_frame(t)
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        rows.append((r["bit_bl"], r["bit_el"], [None if v is None else v.clone() for v in dpb.values()]))
    return rows


@pytest.mark.parametrize("streams", [True, False])
def test_no_launch_writes_outside_its_buffers(streams):
    from lssvc_amd import hip_ops
    inet, pnet, x_bl, x_el, frames = _setup()
    old = hip_ops.MULTI_STREAM
    seen = []

    def check(t):
        torch.cuda.synchronize()
        n = len(hip_ops._GUARDS)
        bad = hip_ops.check_guards()
        seen.append(n)
        assert not bad, (t, bad[:8])
    try:
        hip_ops.MULTI_STREAM = streams
        hip_ops.GUARD_EMPTY = True
        _code(inet, pnet, x_bl, x_el, frames, per_frame=check)
    finally:
        hip_ops.GUARD_EMPTY = False
        hip_ops._GUARDS.clear()
        hip_ops.MULTI_STREAM = old
    assert all(n > 100 for n in seen), seen               # the guards really were around the frames' buffers


@pytest.mark.parametrize("streams", [True, False])
def test_no_launch_reads_memory_that_was_never_written(streams):
    from lssvc_amd import hip_ops
    inet, pnet, x_bl, x_el, frames = _setup()
    old = hip_ops.MULTI_STREAM
    try:
        hip_ops.MULTI_STREAM = streams
        want = _code(inet, pnet, x_bl, x_el, frames)
        hip_ops.POISON_EMPTY = True
        got = _code(inet, pnet, x_bl, x_el, frames)
    finally:
        hip_ops.POISON_EMPTY = False
        hip_ops.MULTI_STREAM = old
    for t, ((wb, we, wt), (gb, ge, gt)) in enumerate(zip(want, got)):
        assert (wb, we) == (gb, ge), (t, wb, gb, we, ge)
        for a, b in zip(wt, gt):
            assert (a is None and b is None) or torch.equal(a, b), t
