"""One conv shape launched N times (for rocprofv3 --pmc passes over a single kernel):  python tools/r6_one_conv.py s2|narrow|big|gdn [n]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import hip_ops as ops  # noqa: E402
from lssvc_amd.weights import WeightStore  # noqa: E402

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
ops.set_conv_precision("f16x3")
g = torch.Generator().manual_seed(0)
cin, cout, H, W, stride = {"s2": (48, 64, 1152, 1920, 2), "narrow": (64, 2, 1152, 1920, 1), "big": (64, 64, 1152, 1920, 1), "big48": (48, 48, 1152, 1920, 1)}[what]
w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
Wt = WeightStore({"c.weight": w, "c.bias": torch.randn(cout, generator=g)}, dev)
x = ops.T(torch.randn(H * W * cin, device=dev), H, W, cin, cin)
out = ops.conv(Wt, "c", x, stride=stride)
for _ in range(n):
    ops.conv(Wt, "c", x, stride=stride, out=out)
torch.cuda.synchronize()
print("done")
