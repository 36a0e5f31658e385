"""Time the fused DepthConvBlock tail (lssvc_ffn_f16x3) against the unfused three-conv chain at bench resolution."""
import sys, math, torch
sys.path.insert(0, ".")
from lssvc_amd import hip_ops as ops
from lssvc_amd.weights import WeightStore

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
H, W = 1152, 1920
for c, hidden in ((64, 256), (48, 192), (32, 128)):
    g = torch.Generator().manual_seed(c)
    sd = {"f.conv.0.weight": torch.randn(hidden, c, 1, 1, generator=g) / math.sqrt(c), "f.conv.0.bias": torch.randn(hidden, generator=g) * 0.1,
          "f.conv.2.weight": torch.randn(c, hidden, 1, 1, generator=g) / math.sqrt(hidden), "f.conv.2.bias": torch.randn(c, generator=g) * 0.1,
          "p.weight": torch.randn(c, c, 1, 1, generator=g) / math.sqrt(c), "p.bias": torch.randn(c, generator=g) * 0.1}
    Wt = WeightStore(sd, dev)
    t = ops.T.from_nchw(torch.randn(1, c, H, W, device=dev))
    ident = ops.T.from_nchw(torch.randn(1, c, H, W, device=dev))
    out = ident.like()
    ops.set_conv_precision("f16x3")

    def fused():
        ops.ffn_block(Wt, "f", pre_name="p", pre_in=t, ident=ident, out=out)

    def ffn_only():
        ops.ffn_block(Wt, "f", x=ident, out=out)

    def unfused():
        o1 = ops.conv(Wt, "p", t, residual=ident)
        v = ops.conv(Wt, "f.conv.0", o1, act="lrelu", slope=0.1)
        ops.conv(Wt, "f.conv.2", v, act="lrelu", slope=0.1, residual=o1, out=out)

    for name, fn in (("fused conv2+FFN", fused), ("fused FFN only", ffn_only), ("unfused 3 convs", unfused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        macs = H * W * (2 * c * hidden + (c * c if name != "fused FFN only" else 0))
        print("C=%d hidden=%d %-18s %8.1f us  %6.1f TFLOP/s (algorithmic)" % (c, hidden, name, us, 2 * macs / us / 1e6))
