"""Why update() -- the CDF-table builder -- is not restated in C++: the tables must equal the reference's entry for
entry (a decoder with other tables reads garbage), and the reference's are outputs of torch's fp32 CPU kernels (Sleef-vectorised
exp / log1p / tanh with <= 1 ulp error, chained: softplus = log1p(exp(x)) rounds twice). This script rebuilds one BitEstimator table
(video_entropy_models.py:168-223) with every transcendental evaluated in double and rounded once to fp32 -- what a libm-based C++
builder would do -- and counts the quantised-CDF entries that differ from lssvc_amd/tables.py's (torch, sha1-pinned to the reference).
    python tools/tables_libm_vs_torch.py     (CPU only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lssvc_amd import tables as Tb  # noqa: E402
from lssvc_amd.entropy_coder import Tables  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402

f32 = np.float32


def main():
    for model, prefix in (("lssvc_extend", "bit_estimator_z"), ("lssvc_extend", "bit_estimator_z_mv"), ("lssvc_extend", "base_layer_model.bit_estimator_z")):
        sd = synth_state_dict(model, 3, 0.6)
        ref = Tb.bit_estimator_tables(sd, prefix)
        P = {k: v.float().numpy().reshape(-1) for k, v in sd.items() if k.startswith(prefix + ".")}
        sp = lambda x: np.where(x.astype(np.float64) > 20, x.astype(np.float64), np.log1p(np.exp(x.astype(np.float64)))).astype(f32)
        th = lambda x: np.tanh(x.astype(np.float64)).astype(f32)
        sg = lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)

        def be(x):
            for i in (1, 2, 3):
                x = (x * sp(P["%s.f%d.h" % (prefix, i)])[:, None]).astype(f32) + P["%s.f%d.b" % (prefix, i)][:, None]
                x = (x + (th(x) * th(P["%s.f%d.a" % (prefix, i)])[:, None]).astype(f32)).astype(f32)
            return sg(((x * sp(P[prefix + ".f4.h"])[:, None]).astype(f32) + P[prefix + ".f4.b"][:, None]).astype(f32))

        C = P[prefix + ".f1.h"].size
        minima, maxima = np.full(C, 50.0, f32), np.full(C, 50.0, f32)
        for i in range(50, 1, -1):
            minima = np.where(be(np.full((C, 1), -i, f32))[:, 0] < f32(0.0001), f32(i), minima)
            maxima = np.where(be(np.full((C, 1), i, f32))[:, 0] > f32(0.9999), f32(i), maxima)
        mi, ma = minima.astype(np.int32), maxima.astype(np.int32)
        L = ma + mi + 1
        samples = np.arange(L.max(), dtype=f32)[None, :] + (0 - mi).astype(f32)[:, None]
        lower, upper = be((samples - f32(0.5)).astype(f32)), be((samples + f32(0.5)).astype(f32))
        mine = Tables.from_pmfs((upper - lower).astype(f32), (lower[:, :1] + (f32(1.0) - upper[:, -1:])).astype(f32), L, -mi)
        same_shape = ref.cdfs.shape == mine.cdfs.shape and np.array_equal(ref.sizes, mine.sizes) and np.array_equal(ref.offsets, mine.offsets)
        d = (ref.cdfs != mine.cdfs) if same_shape else None
        print("%-36s lengths / offsets equal: %s; quantised CDF entries differing: %s" % (
            prefix, same_shape, "%d of %d (%d of %d rows)" % (d.sum(), d.size, d.any(1).sum(), d.shape[0]) if d is not None else "n/a"))


if __name__ == "__main__":
    main()
