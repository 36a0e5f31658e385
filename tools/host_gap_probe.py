"""How much of a GOP's wall time is the host's per-frame round trip (the D2H read of the bit counters that every encode_decode call ends with,
as the reference's .item() does, plus issuing the next frame)? The same GOP timed twice: as shipped, and with BitSlots.fetch() replaced by a
stub that neither copies nor waits (the bits are then meaningless -- timing only).   python tools/host_gap_probe.py [gops]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    gops = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops
    from lssvc_amd.prepost import FramePrep
    from lssvc_amd.synth import synth_clip, synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN))
    pnet.to(dev).eval()
    prep = FramePrep(dev)
    clip = synth_clip(bench.GOP, bench.HEIGHT, bench.WIDTH, seed=0)
    layers = [prep.make_layers_rgb8(clip[t].to(dev), bench.RATIO) for t in range(bench.GOP)]
    x_bls, x_els, shape_hr = [l[0] for l in layers], [l[1] for l in layers], layers[0][2]["HR_padded_size"]
    for net in (inet, pnet):
        net.set_graph_mode(True, alias_outputs=True)
    for _ in range(3):
        bench.encode_gop(inet, pnet, x_bls, x_els, shape_hr)

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(gops):
            bench.encode_gop(inet, pnet, x_bls, x_els, shape_hr)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / gops

    a = timed()
    real = hip_ops.BitSlots.fetch
    hip_ops.BitSlots.fetch = lambda self: [0.0] * 16
    try:
        b = timed()
    finally:
        hip_ops.BitSlots.fetch = real
    c = timed()
    print("GOP of %d frames: %.1f ms as shipped, %.1f ms without the per-frame read of the bit counters (%.2f %%), %.1f ms as shipped again" % (
        bench.GOP, 1e3 * a, 1e3 * b, 100 * (a - b) / a, 1e3 * c))


if __name__ == "__main__":
    main()
