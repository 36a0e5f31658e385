"""What mixed read + write traffic can this part sustain? Device-to-device copies and adds of fp32 tensors of the sizes the
HBM-bound kernels move (64-channel 1152x1920 fp32 = 566 MB), timed with HIP events:  python tools/bw_copy_probe.py
Gives the practical ceiling the streaming 1x1 / depthwise / elementwise kernels are priced against."""
import torch

dev = torch.device("cuda:0")
for mb in (141, 566, 1132):
    n = mb * 1000 * 1000 // 4
    a = torch.randn(n, device=dev)
    b = torch.empty_like(a)
    c = torch.randn(n, device=dev)
    for name, fn, traffic in (("copy (1 read + 1 write)", lambda: b.copy_(a), 2), ("add (2 reads + 1 write)", lambda: torch.add(a, c, out=b), 3),
                              ("read-only sum", lambda: a.sum(), 1), ("write-only fill", lambda: b.fill_(1.0), 1)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%5d MB  %-26s %8.1f us  %6.2f TB/s" % (mb, name, ms * 1e3, traffic * 4 * n / ms / 1e9))
