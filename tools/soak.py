"""Soak: the 1080p bench GOP coded over and over through the frame plans (graph replays, side streams, look-ahead, shared graph pools) --
every GOP's 64 bit counts and its last reconstruction must equal the first GOP's, and the reserved memory must not grow.
    python tools/soak.py [gops=40]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lssvc_amd import IntraSS, LSSVC_extend  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402


def main():
    gops = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN))
    pnet.to(dev).eval()
    for net in (inet, pnet):
        net.set_graph_mode(True, alias_outputs=True)
    x_bls, x_els, pad = bench.build_inputs(dev, 0, 32)[:3]
    ref, bad, reserved = None, 0, []
    with torch.no_grad():
        for k in range(gops):
            torch.cuda.synchronize()
            t0 = time.time()
            bits, dpb = bench.encode_gop(inet, pnet, x_bls, x_els, pad["HR_padded_size"])
            torch.cuda.synchronize()
            dt = time.time() - t0
            if ref is None:
                ref = (list(bits), dpb["ref_frame_el"].clone())
            same = list(bits) == ref[0] and torch.equal(dpb["ref_frame_el"], ref[1])
            bad += 0 if same else 1
            reserved.append(torch.cuda.memory_reserved() / 2 ** 30)
            if k < 4 or k % 10 == 9 or not same:
                print("GOP %3d: %.3f s  %.2f frames/s  identical to the first GOP: %s  reserved %.1f GiB" % (k, dt, 32 / dt, same, reserved[-1]), flush=True)
    print("soak: %d GOPs (%d frames), %d differ from the first; reserved memory %.1f GiB after priming -> %.1f GiB at the end" % (
        gops, 32 * gops, bad, reserved[min(4, len(reserved) - 1)], reserved[-1]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
