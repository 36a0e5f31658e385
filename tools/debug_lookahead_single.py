"""Reproduce / bisect the look-ahead + single-stream graph-replay corruption.
   [LSSVC_FILL_MEMSET=1] python tools/debug_lookahead_single.py [variant ...]     (LSSVC_FILL_MEMSET=1: the round-4 zero fill through hipMemsetAsync, which reproduces it)
variants: sync (device sync around every plan call), eager-bl / eager-el / eager-p (that plan family never captured),
          streams (multi-stream mode, the control)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from lssvc_amd import hip_ops  # noqa: E402
from test_gpu_graph import _nets, _code, DEV  # noqa: E402


def main():
    variants = set(sys.argv[1:])
    from lssvc_amd.synth import synth_clip
    from lssvc_amd.preprocess import imresize_bicubic
    H, W, frames, gops = 128, 256, 6, 3
    hip_ops.MULTI_STREAM = "streams" in variants
    clip = synth_clip(frames, H, W, seed=5).float() / 255.0
    x_bl, x_el = imresize_bicubic(clip, (H // 2, W // 2)).clamp_(0, 1).to(DEV), clip.to(DEV)
    inet, pnet = _nets(3, 0.6)
    want = _code(inet, pnet, x_bl, x_el, H, W, 1, frames)
    got = _code(inet, pnet, x_bl, x_el, H, W, 1, frames, lookahead=True)
    inet.set_graph_mode(True)
    pnet.set_graph_mode(True)
    orig = pnet._run_planned

    def patched(key, tensors, body):
        fam = str(key[0])
        never = ("eager-bl" in variants and fam == "p-ahead-bl") or ("eager-el" in variants and fam == "p-ahead-el") or ("eager-p" in variants and fam == "p")
        if "sync" in variants:
            torch.cuda.synchronize()
        if "watch" in variants:
            before = sums()
        if never:
            from lssvc_amd.hip_ops import T
            ins = {k: (None if v is None else T.from_nchw(v)) for k, v in tensors.items()}
            r = body(ins)
        else:
            r = orig(key, tensors, body)
        if "sync" in variants:
            torch.cuda.synchronize()
        if "watch" in variants:
            after = sums()
            changed = [k for k in after if before.get(k) != after[k]]
            print("call %-11s %-28s changed: %s" % (fam, [x for x in key[1:4] if not isinstance(x, tuple) or len(x) < 4], changed), flush=True)
        return r

    def sums():
        torch.cuda.synchronize()
        out = {}
        for g, b in pnet._lookahead_bufs.items():
            for fam2 in ("stash", "el_out"):
                for par in (0, 1):
                    d = b[fam2][par]
                    if d is not None:
                        for k, t in d.items():
                            out["%s[%d].%s" % (fam2, par, k)] = float(t.buf.double().sum())
        return out

    pnet._run_planned = patched
    if "noempty" in variants:                          # torch.cuda.graph() empties the allocator's cache when a capture begins
        torch.cuda.empty_cache = lambda: None
    if "sharedpool" in variants or "ownstream" in variants:
        real_graph = torch.cuda.graph
        pool = torch.cuda.graph_pool_handle()

        def graph(g, **kw):
            if "sharedpool" in variants:
                kw["pool"] = pool
            if "ownstream" in variants:
                kw["stream"] = torch.cuda.Stream()
            return real_graph(g, **kw)

        torch.cuda.graph = graph
    if "keep" in variants:                             # no buffer of a frame body is recycled before the body ends (what a Fork branch does)
        for name in ("_frame_body", "_ahead_bl_body", "_ahead_el_body"):
            def wrap(fn):
                def inner(*a, **k):
                    hip_ops._TLS.keep = []
                    try:
                        return fn(*a, **k)
                    finally:
                        hip_ops._TLS.keep = None
                return inner
            setattr(pnet, name, wrap(getattr(pnet, name)))
    got += _code(inet, pnet, x_bl, x_el, H, W, gops, frames, lookahead=True)
    if "plain-after" in variants:                      # the plain loop again, after the look-ahead plans exist
        got += _code(inet, pnet, x_bl, x_el, H, W, 2, frames)
    print("variants", sorted(variants), "plans:", [(str(k[0]), p.graph is not None, p.calls) for k, p in pnet._plans.items()])
    names = ["ref_frame_bl", "ref_feature_bl", "ref_frame_el", "ref_feature_el", "mv_hat", "warp_frame"]      # (P-frame dpb order)
    bad = 0
    for i, (bb, be, tens) in enumerate(got):
        wb, we, wt = want[i % frames]
        msg = []
        if (bb, be) != (wb, we):
            msg.append("bits bl %.3f/%.3f el %.3f/%.3f" % (bb, wb, be, we))
        for n, a, b in zip(names, tens, wt):
            if a is None or b is None:
                continue
            if not torch.equal(a, b):
                d = (a - b).abs()
                msg.append("%s max|d| %.3e (%d of %d differ)" % (n, d.max().item(), int((d > 0).sum()), d.numel()))
        if msg:
            bad += 1
            print("pass %d frame %d: %s" % (i // frames, i % frames, "; ".join(msg)))
    print("RESULT: %d of %d frames differ" % (bad, len(got)))


if __name__ == "__main__":
    main()
