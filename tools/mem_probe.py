"""Where the reserved HBM of the 1080p bench workload sits (round 5): reserved / allocated bytes after the eager
GOP, after the capture GOP and after a replay GOP, with one graph memory pool per plan and with the shared pools.
    [LSSVC_SHARED_GRAPH_POOL=0] python tools/mem_probe.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def gib(x):
    return "%.1f" % (x / 2 ** 30)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    from lssvc_amd import IntraSS, LSSVC_extend, hip_ops, intra
    from lssvc_amd.prepost import FramePrep
    from lssvc_amd.synth import synth_clip, synth_state_dict
    inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN))
    pnet.to(dev).eval()
    prep = FramePrep(dev)
    clip = synth_clip(frames, bench.HEIGHT, bench.WIDTH, seed=0)
    layers = [prep.make_layers_rgb8(clip[t].to(dev), bench.RATIO) for t in range(frames)]
    x_bls, x_els, shape_hr = [l[0] for l in layers], [l[1] for l in layers], layers[0][2]["HR_padded_size"]
    for net in (inet, pnet):
        net.set_graph_mode(True, alias_outputs=True)

    def report(tag):
        torch.cuda.synchronize()
        print("%-28s reserved %6s GiB  allocated %6s GiB  peak reserved %6s GiB   plans: I %d, P %d" % (
            tag, gib(torch.cuda.memory_reserved(dev)), gib(torch.cuda.memory_allocated(dev)), gib(torch.cuda.max_memory_reserved(dev)),
            len(inet._plans), len(pnet._plans)), flush=True)

    report("nets + inputs")
    for i, la in enumerate((False, False, True, True, True)):
        bench.encode_gop(inet, pnet, x_bls, x_els, shape_hr, lookahead=la)
        report("GOP %d (%s)" % (i, "look-ahead" if la else "frame after frame"))
    torch.cuda.empty_cache()
    report("after empty_cache()")
    print("shared pools:", intra.SHARED_GRAPH_POOL)


if __name__ == "__main__":
    main()
