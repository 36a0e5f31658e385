import sys, torch
sys.path.insert(0, "/root/repo")
from lssvc_amd import hip_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H, W = 1152, 1920
for C in (64, 256):
    a = ops.T(torch.randn(H * W * C, device=dev), H, W, C, C); b = a.like()
    ms = timeit(lambda: ops.copy(a, b)); print("lssvc copy   C=%3d: %.3f ms  %.2f TB/s (r+w)" % (C, ms, 2 * H * W * C * 4 / ms / 1e9))
    ms = timeit(lambda: b.buf.copy_(a.buf)); print("torch copy_  C=%3d: %.3f ms  %.2f TB/s (r+w)" % (C, ms, 2 * H * W * C * 4 / ms / 1e9))
    ms = timeit(lambda: b.buf.zero_()); print("torch zero_  C=%3d: %.3f ms  %.2f TB/s (w)" % (C, ms, H * W * C * 4 / ms / 1e9))
    ms = timeit(lambda: ops.lrelu(a, 0.1, out=b)); print("lssvc lrelu  C=%3d: %.3f ms  %.2f TB/s (r+w)" % (C, ms, 2 * H * W * C * 4 / ms / 1e9))
