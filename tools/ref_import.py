"""Import the upstream reference (read-only at /root/reference) in THIS container only.

Used by tools/make_manifest.py and tests/golden/make_golden.py to pin the oracle and to
generate fixtures. The reference never travels to the GPU box; nothing under tests/ (as run by
pytest), bench.py or the product package imports this file.

pytorch_msssim is not installed here; the reference only uses it for training losses / metrics
(LSSVC_net.py:4, dmc_net.py:4), so a metric-only stand-in module is registered in memory.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def import_reference():
    if "pytorch_msssim" not in sys.modules:
        m = types.ModuleType("pytorch_msssim")

        class MS_SSIM(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

        m.MS_SSIM = MS_SSIM
        m.ms_ssim = lambda *a, **k: torch.tensor(0.0)
        sys.modules["pytorch_msssim"] = m
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from src.models.IntraSS import IntraSS
    from src.models.LSSVC_net_extend import LSSVC_extend
    return IntraSS, LSSVC_extend
