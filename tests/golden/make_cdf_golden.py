"""Golden vectors for pmf_to_quantized_cdf from the REFERENCE's own ops.cpp (built by oracle/Makefile into
oracle/_ref from /root/reference/src/cpp/ops/ops.cpp; build container only):

    make -C oracle && python tests/golden/make_cdf_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref"))
import MLCodec_CXX  # noqa: E402  (the reference's pybind11 module)

rng = np.random.default_rng(0)
cases = [[0.1, 0.2, 0.3, 0.4, 1e-9], [1.0], [0.5, 0.5], [1e-9] * 7 + [1.0], [0.0, 0.0, 1.0, 0.0, 0.0, 1e-12]]
for n in (3, 17, 64, 101, 255):
    p = rng.random(n).astype(np.float32) ** 4
    p = p / p.sum()
    cases.append([float(v) for v in p])
    lap = np.exp(-np.abs(np.arange(n) - n // 2) / (0.05 + 3 * rng.random())).astype(np.float32)   # peaky, many ~0 bins
    lap = lap / lap.sum()
    cases.append([float(v) for v in np.append(lap, np.float32(1e-7))])
out = [{"pmf": c, "precision": 16, "cdf": [int(v) for v in MLCodec_CXX.pmf_to_quantized_cdf(c, 16)]} for c in cases]
with open(os.path.join(HERE, "cdf_vectors.json"), "w") as f:
    json.dump(out, f)
print("wrote", len(out), "vectors")
