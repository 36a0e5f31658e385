import sys, torch
sys.path.insert(0, "/root/repo")
from lssvc_amd import hip_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H, W = 1152, 1920
a = ops.T(torch.randn(H * W * 64, device=dev), H, W, 64, 64); b = a.like(); c = a.like()
ms = timeit(lambda: ops.add(a, b, out=c)); print("add C=64 1152x1920: %.1f us  %.2f TB/s" % (ms * 1e3, 3 * H * W * 64 * 4 / ms / 1e9))
h2, w2 = H // 2, W // 2
s = ops.T(torch.randn(h2 * w2 * 64, device=dev), h2, w2, 64, 64)
ms = timeit(lambda: ops.resize(s, H, W, out=c)); print("resize x2 C=64 -> 1152x1920: %.1f us  %.2f TB/s" % (ms * 1e3, 1.25 * H * W * 64 * 4 / ms / 1e9))
fl = ops.T(torch.randn(H * W * 2, device=dev) * 3, H, W, 2, 2)
a48 = ops.T(torch.randn(H * W * 48, device=dev), H, W, 48, 48); o48 = a48.like()
ms = timeit(lambda: ops.flow_warp(a48, fl, out=o48)); print("flow_warp C=48 1152x1920: %.1f us  %.2f TB/s" % (ms * 1e3, 2 * H * W * 48 * 4 / ms / 1e9))
ms = timeit(lambda: ops.pool2x2(a, is_max=False)); print("pool2x2 C=64: %.1f us" % (ms * 1e3))
import math
from lssvc_amd.weights import WeightStore
from lssvc_amd._lib import lib, check
w = torch.randn(64, 1, 3, 3) / 3
Wt = WeightStore({"d.weight": w, "d.bias": torch.randn(64)}, dev)
for mode in (1, 0):
    check(lib.lssvc_set_option(b"pointwise_blocks", mode))
    ms = timeit(lambda: ops.dwconv3x3(Wt, "d", a, out=c)); print("dwconv3x3 C=64 1152x1920 (pointwise_blocks=%d): %.1f us  %.2f TB/s" % (mode, ms * 1e3, 2 * H * W * 64 * 4 / ms / 1e9))
    ms = timeit(lambda: ops.resize(s, H, W, out=c)); print("resize x2 C=64 (pointwise_blocks=%d): %.1f us" % (mode, ms * 1e3))
check(lib.lssvc_set_option(b"pointwise_blocks", 1))
