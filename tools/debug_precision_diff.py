"""Debug aid: the base-layer analysis / synthesis transform of an I-frame layer by layer in both conv precisions; prints
max |f32 - f16x3| after every block and where it sits:  python tools/debug_precision_diff.py H W seed gain"""
import sys

import torch

sys.path.insert(0, ".")
from lssvc_amd import IntraSS, hip_ops as ops, blocks as B
from lssvc_amd.hip_ops import T
from lssvc_amd.synth import synth_state_dict, synth_clip
from lssvc_amd.preprocess import imresize_bicubic

H, W_, seed, gain = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = "cuda:0"
clip = synth_clip(1, 2 * H, 2 * W_, seed=seed).float() / 255.0
x_bl = imresize_bicubic(clip, (H, W_)).clamp_(0, 1).to(dev)          # (the clip of tests/test_gpu_frames.py at EL = 2H x 2W; for ratio 1.5 pass the BL size)
net = IntraSS.from_state_dict(synth_state_dict("intra_ss", seed, gain)).to(dev).eval()
net.range_audit = False
Wt, p = net.W, "base_layer_model"


def chain(x):
    out = []
    g = p + ".g_a"
    t = B.residual_block_with_stride(Wt, g + ".0", x); out.append(("g_a.0", t))
    t = B.residual_block(Wt, g + ".1", t); out.append(("g_a.1", t))
    t = B.residual_block_with_stride(Wt, g + ".2", t); out.append(("g_a.2", t))
    t = B.residual_block(Wt, g + ".3", t); out.append(("g_a.3", t))
    t = B.residual_block_with_stride(Wt, g + ".4", t); out.append(("g_a.4", t))
    t = B.residual_block(Wt, g + ".5", t); out.append(("g_a.5", t))
    y = ops.conv(Wt, g + ".6", t, stride=2); out.append(("g_a.6 (y)", y))
    return out


res = {}
for prec in ("f32", "f16x3"):
    ops.set_conv_precision(prec)
    res[prec] = [(n, t.to_nchw(copy=True).clone()) for n, t in chain(T.from_nchw(x_bl))]
    torch.cuda.synchronize()
for (n, a), (_, b) in zip(res["f32"], res["f16x3"]):
    d = (a - b).abs()
    i = int(d.argmax())
    idx = []
    for s in reversed(a.shape):
        idx.append(i % s); i //= s
    print("%-12s shape %s  max|d| %.3e at %s  (|a| max %.3g)  n(d > 1e-4 max|a|) %d" % (n, tuple(a.shape), d.max().item(), tuple(reversed(idx)), a.abs().max().item(),
                                                                                   int((d > 1e-4 * a.abs().max()).sum())))
