"""MLCodec_CXX stand-in: pmf_to_quantized_cdf(pmf: list[float], precision: int) -> list[int]
(src/cpp/ops/ops.cpp:24-91; called as `_pmf_to_quantized_cdf(pmf.tolist(), precision)` by
video_entropy_models.py:18-22,72-76 and img_entropy_models.py:30-34). Backed by lssvc_pmf_to_quantized_cdf."""
import numpy as np

from .._lib import lib, check

__name__ = "MLCodec_CXX"


def pmf_to_quantized_cdf(pmf, precision):
    """Return quantized CDF for a given PMF (len(pmf) + 1 entries, cdf[0] = 0, cdf[-1] = 1 << precision)."""
    p = np.ascontiguousarray(pmf, dtype=np.float32).reshape(-1)
    precision = int(precision)
    if not 1 <= precision <= 31:
        raise ValueError("precision must be in [1, 31]")
    out = np.empty(p.size + 1, dtype=np.uint32)
    check(lib.lssvc_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data))
    return out.tolist()
