"""Probe: configs[3] (2160p IP12; SIZE=1080: configs[1], 32-frame GOPs) through the frame plans (GRAPH=0: eager), GOP by GOP, with the
allocator's device-malloc counter beside each GOP's time:  [SIZE=1080] [GOPS=n] [RESERVE=<GiB>] [GRAPH=0] python tools/gop_probe.py"""
import sys, json, time, os
sys.path.insert(0, ".")
import torch
import bench
from lssvc_amd import hip_ops, IntraSS, LSSVC_extend
from lssvc_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
gib = float(os.environ.get("RESERVE", "0"))
if gib > 0:
    hip_ops.reserve_device_memory(dev, gib)
bench.HEIGHT, bench.WIDTH = (1080, 1920) if os.environ.get("SIZE") == "1080" else (2160, 3840)
FRAMES = 32 if os.environ.get("SIZE") == "1080" else 12
inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
pnet = LSSVC_extend(); pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN)); pnet.to(dev).eval()
if os.environ.get("GRAPH", "1") == "1":
    inet.set_graph_mode(True, alias_outputs=True); pnet.set_graph_mode(True, alias_outputs=True)
x_bls, x_els, pad, _ = bench.build_inputs(dev, seed=5, frames=FRAMES)
shape_hr = pad["HR_padded_size"]
def stat(k): return torch.cuda.memory_stats(dev).get(k, 0)
with torch.no_grad():
    for g in range(int(os.environ.get("GOPS", "4"))):
        a0, s0 = stat("num_device_alloc"), stat("num_alloc_retries")
        torch.cuda.synchronize(); t0 = time.time()
        bench.encode_gop(inet, pnet, x_bls, x_els, shape_hr)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("GOP %d: %.0f ms (%.2f frames/s), device mallocs during it %d, reserved %.1f GiB, allocated %.1f GiB" % (
            g, 1e3 * dt, FRAMES / dt, stat("num_device_alloc") - a0, torch.cuda.memory_reserved(dev) / 2**30, torch.cuda.memory_allocated(dev) / 2**30), flush=True)
