"""CDF tables for write_stream=1 -- what the reference's `update(force=True)` builds once per model
(test.py:561-564). Runs on the host in torch CPU fp32, op for op as the reference does, then quantises
each pmf with the C-ABI `lssvc_pmf_to_quantized_cdf`. Returns entropy_coder.Tables objects.

  bit_estimator_tables      BitEstimator.update        video_entropy_models.py:168-223
  laplace_tables            GaussianEncoder.update     video_entropy_models.py:266-307   (256 scales in [0.01, 64])
  gaussian_tables           GaussianConditional.update img_entropy_models.py:623-648     (64 scales in [0.11, 256])
  bottleneck_tables         EntropyBottleneck.update   img_entropy_models.py:436-476
"""
import math

import numpy as np
import scipy.stats
import torch
import torch.nn.functional as F

from .entropy_coder import Tables


def _bit_estimator(x, sd, p):
    """BitEstimator.forward with (1,C,1,1) parameters (video_entropy_models.py:110-129,150-166)."""
    for i in (1, 2, 3):
        x = x * F.softplus(sd["%s.f%d.h" % (p, i)]) + sd["%s.f%d.b" % (p, i)]
        x = x + torch.tanh(x) * torch.tanh(sd["%s.f%d.a" % (p, i)])
    return torch.sigmoid(x * F.softplus(sd[p + ".f4.h"]) + sd[p + ".f4.b"])


def bit_estimator_tables(sd, prefix):
    sd = {k: v.float() for k, v in sd.items() if k.startswith(prefix + ".")}
    channel = sd[prefix + ".f1.h"].numel()
    with torch.no_grad():
        medians = torch.zeros(channel)
        minima = medians + 50
        for i in range(50, 1, -1):
            probs = torch.squeeze(_bit_estimator((torch.zeros_like(medians) - i)[None, :, None, None], sd, prefix))
            minima = torch.where(probs < torch.zeros_like(medians) + 0.0001, torch.zeros_like(medians) + i, minima)
        maxima = medians + 50
        for i in range(50, 1, -1):
            probs = torch.squeeze(_bit_estimator((torch.zeros_like(medians) + i)[None, :, None, None], sd, prefix))
            maxima = torch.where(probs > torch.zeros_like(medians) + 0.9999, torch.zeros_like(medians) + i, maxima)
        minima, maxima = minima.int(), maxima.int()
        offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max()
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = _bit_estimator(samples - 0.5, sd, prefix).squeeze(0)
        upper = _bit_estimator(samples + 0.5, sd, prefix).squeeze(0)
        pmf = (upper - lower)[:, 0, :]
        tail_mass = lower[:, 0, :1] + (1.0 - upper[:, 0, -1:])
    return Tables.from_pmfs(pmf.numpy(), tail_mass.numpy(), pmf_length.numpy(), offset.numpy())


def _scale_table(lo, hi, levels):
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


LAPLACE = {"min": 0.01, "max": 64.0, "levels": 256}
GAUSSIAN = {"min": 0.11, "max": 256.0, "levels": 64}


def index_params(spec, add):
    """(log_min, log_step, add, levels) for lssvc_build_indexes / the reference's build_indexes."""
    lo, hi = math.log(spec["min"]), math.log(spec["max"])
    return lo, (hi - lo) / (spec["levels"] - 1), add, spec["levels"]


def laplace_tables():
    table = _scale_table(LAPLACE["min"], LAPLACE["max"], LAPLACE["levels"])
    with torch.no_grad():
        pmf_center = torch.zeros_like(table) + 50
        lap = torch.distributions.laplace.Laplace(torch.zeros_like(table), torch.zeros_like(table) + table)
        for i in range(50, 1, -1):
            probs = torch.squeeze(lap.cdf(torch.zeros_like(pmf_center) + i))
            pmf_center = torch.where(probs > torch.zeros_like(pmf_center) + 0.9999, torch.zeros_like(pmf_center) + i, pmf_center)
        pmf_center = pmf_center.int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = (torch.arange(max_length) - pmf_center[:, None]).float()
        scales = torch.zeros_like(samples) + table[:, None]
        lap = torch.distributions.laplace.Laplace(torch.zeros_like(scales), scales)
        upper, lower = lap.cdf(samples + 0.5), lap.cdf(samples - 0.5)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
    return Tables.from_pmfs(pmf.numpy(), tail_mass.numpy(), pmf_length.numpy(), (-pmf_center).numpy())


def gaussian_tables(tail_mass=1e-9):
    table = _scale_table(GAUSSIAN["min"], GAUSSIAN["max"], GAUSSIAN["levels"])
    std_cum = lambda x: 0.5 * torch.erfc(float(-(2 ** -0.5)) * x)
    with torch.no_grad():
        multiplier = -scipy.stats.norm.ppf(tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        s = table.unsqueeze(1).float()
        upper, lower = std_cum((0.5 - samples) / s), std_cum((-0.5 - samples) / s)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
    return Tables.from_pmfs(pmf.numpy(), tail.numpy(), pmf_length.numpy(), (-pmf_center).numpy())


def _logits_cumulative(v, sd, p):
    logits = v
    for i in range(5):
        logits = torch.matmul(F.softplus(sd["%s._matrices.%d" % (p, i)]), logits) + sd["%s._biases.%d" % (p, i)]
        if i < 4:
            logits = logits + torch.tanh(sd["%s._factors.%d" % (p, i)]) * torch.tanh(logits)
    return logits


def bottleneck_tables(sd, prefix):
    sd = {k: v.float() for k, v in sd.items() if k.startswith(prefix + ".") and v.is_floating_point()}
    q = sd[prefix + ".quantiles"]
    with torch.no_grad():
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = pmf_length.max()
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = _logits_cumulative(samples - 0.5, sd, prefix)
        upper = _logits_cumulative(samples + 0.5, sd, prefix)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    return Tables.from_pmfs(pmf.numpy(), tail_mass.numpy(), pmf_length.numpy(), offset.numpy()), medians.numpy().copy()
