"""Two GOPs in flight on one GPU from ONE process (the in-process form): two model instances, two host threads, each on a
stream of its own with its own frame plans, against the same two models run one after the other. Round 3 measured this at 0 gain and blamed
the runtime; round 6 found the runtime's 4 hardware queues per process (GPU_MAX_HW_QUEUES) to be a limit of its own, so: again, per setting.
    GPU_MAX_HW_QUEUES=16 python tools/two_gops_threads.py [gops=6]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from lssvc_amd import IntraSS, LSSVC_extend  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402


def main():
    gops = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    x_bls, x_els, pad = bench.build_inputs(dev, 0, 32)[:3]
    shape = pad["HR_padded_size"]
    sets = []
    for i in range(2):
        inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
        pnet = LSSVC_extend()
        pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN))
        pnet.to(dev).eval()
        for net in (inet, pnet):
            net.set_graph_mode(True, alias_outputs=True)
        sets.append((inet, pnet, torch.cuda.Stream(dev)))
    ref = None
    with torch.no_grad():
        for inet, pnet, s in sets:                      # prime one after the other (plan capture uses module-level state), each on its own stream
            with torch.cuda.stream(s):
                for _ in range(3):
                    bits, _ = bench.encode_gop(inet, pnet, x_bls, x_els, shape)
                s.synchronize()
            ref = ref or list(bits)
            assert list(bits) == ref
    bad = [0, 0]

    def run(i, n):
        inet, pnet, s = sets[i]
        with torch.no_grad(), torch.cuda.stream(s):
            for _ in range(n):
                bits, _ = bench.encode_gop(inet, pnet, x_bls, x_els, shape)
                bad[i] += 0 if list(bits) == ref else 1
            s.synchronize()

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        return time.time() - t0

    for rnd in range(2):
        t_one = timed(lambda: run(0, gops))
        t_seq = timed(lambda: (run(0, gops), run(1, gops)))

        def both():
            th = [threading.Thread(target=run, args=(i, gops)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        t_par = timed(both)
        print("GPU_MAX_HW_QUEUES=%s: one model %.2f frames/s; two models one after the other %.2f; two models on two threads / streams %.2f  (x%.3f); GOPs with other bits: %d" % (
            os.environ.get("GPU_MAX_HW_QUEUES"), 32 * gops / t_one, 64 * gops / t_seq, 64 * gops / t_par, t_seq / t_par, sum(bad)), flush=True)


if __name__ == "__main__":
    main()
