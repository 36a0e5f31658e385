"""Per-layer conv time of P frames at the bench workload (debug/profiling aid): runs an I + N P frames with the
op log on (bench.encode_gop logs P-frames 1..8) and prints total time per (layer name, shape, kernel)."""
import sys, collections, torch
sys.path.insert(0, ".")
import bench
from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
from lssvc_amd.synth import synth_state_dict

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
pnet = LSSVC_extend(); pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN)); pnet.to(dev).eval()
x_bls, x_els, pad = bench.build_inputs(dev, 0, frames)
with torch.no_grad():
    bench.encode_gop(inet, pnet, x_bls, x_els, pad["HR_padded_size"])
    torch.cuda.synchronize()
    log = []
    bench.encode_gop(inet, pnet, x_bls, x_els, pad["HR_padded_size"], op_log=log)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for e in log:
    ms = e["events"][0].elapsed_time(e["events"][1])
    k = (e["name"], e["kind"], "%dx%d" % (e["hout"], e["wout"]), "%d->%d" % (e["cin"], e["cout"]), e["kernel"])
    a = agg.setdefault(k, [0, 0.0, e["macs"], e["bytes"]])
    a[0] += 1; a[1] += ms
tot = sum(a[1] for a in agg.values())
print("conv total %.1f ms over %d frames" % (tot, frames))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    us = 1e3 * a[1] / a[0]
    print("%-58s %-10s %-10s %-9s %-40s n=%3d avg %7.1f us  %5.1f TF %6.0f GB/s  %4.1f%%" % (
        k[0][-58:], k[1], k[2], k[3], k[4][-40:], a[0], us, 2 * a[2] / us / 1e6, a[3] / us / 1e3, 100 * a[1] / tot))
