"""ORACLE (test infrastructure only) -- CPU fp32 restatement of LSSVC's entropy models
(likelihoods, quantisation, bit counts, sigma->index maps).  See blocks.py for the rules.
"""
import math

import torch
import torch.nn.functional as F

LIKELIHOOD_BOUND = 1e-9     # EntropyModel likelihood_lower_bound (img_entropy_models.py:184-192)
SCALE_BOUND = 0.11          # GaussianConditional scale_bound (img_entropy_models.py:586-607)


# ----------------------------------------------------------------------------- factorised prior (I-frames)
def _logits_cumulative(v, p):
    """EntropyBottleneck._logits_cumulative (img_entropy_models.py:483-502); v is (C,1,N)."""
    logits = v
    for i in range(5):
        logits = torch.matmul(F.softplus(p["_matrices.%d" % i]), logits)
        logits = logits + p["_biases.%d" % i]
        if i < 4:
            logits = logits + torch.tanh(p["_factors.%d" % i]) * torch.tanh(logits)
    return logits


def entropy_bottleneck(z, p):
    """EntropyBottleneck.forward in eval mode (img_entropy_models.py:518-554, 505-516):
    z_hat = round(z - median) + median, likelihood = |sig(s*u) - sig(s*l)| >= 1e-9."""
    n, c, h, w = z.shape
    v = z.permute(1, 2, 3, 0).contiguous().reshape(c, 1, -1)
    med = p["quantiles"][:, :, 1:2]
    out = torch.round(v - med) + med
    lower = _logits_cumulative(out - 0.5, p)
    upper = _logits_cumulative(out + 0.5, p)
    sign = -torch.sign(lower + upper)
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = torch.max(lik, torch.tensor([LIKELIHOOD_BOUND]))
    back = lambda t: t.reshape(c, h, w, n).permute(3, 0, 1, 2).contiguous()
    return back(out), back(lik)


# ----------------------------------------------------------------------------- Gaussian conditional (I-frames)
def _std_cumulative(x):
    return 0.5 * torch.erfc(float(-(2 ** -0.5)) * x)


def gaussian_conditional(y, scales, means):
    """GaussianConditional.forward in eval mode (img_entropy_models.py:667-685, 650-665).
    The likelihood sees (round(y-mu)+mu)-mu, not the integer; y_hat = d_quant(y, mu)."""
    out = torch.round(y - means) + means
    values = torch.abs(out - means)
    s = torch.max(scales, torch.tensor([SCALE_BOUND]))
    lik = _std_cumulative((0.5 - values) / s) - _std_cumulative((-0.5 - values) / s)
    lik = torch.max(lik, torch.tensor([LIKELIHOOD_BOUND]))
    r = y - means
    y_hat = r + (torch.round(r) - r) + means
    return y_hat, lik


def bits_from_likelihoods(*liks):
    """(sum log lik_a + sum log lik_b) / -ln 2 (IntraSS.py:163, priors.py:377)."""
    total = None
    for l in liks:
        s = torch.log(l).sum()
        total = s if total is None else total + s
    return total / (-math.log(2))


# ----------------------------------------------------------------------------- Laplace + BitEstimator (P-frames)
def laplace_bits(y_q, sigma):
    """get_y_bits_probs (LSSVC_net.py:154-161 = dmc_net.py:370-377)."""
    mu = torch.zeros_like(sigma)
    sigma = sigma.clamp(1e-5, 1e10)
    lap = torch.distributions.laplace.Laplace(mu, sigma)
    probs = lap.cdf(y_q + 0.5) - lap.cdf(y_q - 0.5)
    return torch.sum(torch.clamp(-1.0 * torch.log(probs + 1e-5) / math.log(2.0), 0, 50))


def bit_estimator(x, p):
    """BitEstimator.forward: three Bitparm + final sigmoid Bitparm, per channel
    (video_entropy_models.py:110-129,150-166)."""
    for i in (1, 2, 3):
        x = x * F.softplus(p["f%d.h" % i]) + p["f%d.b" % i]
        x = x + torch.tanh(x) * torch.tanh(p["f%d.a" % i])
    return torch.sigmoid(x * F.softplus(p["f4.h"]) + p["f4.b"])


def factorized_bits(z_q, p):
    """get_z_bits_probs (LSSVC_net.py:163-167)."""
    prob = bit_estimator(z_q + 0.5, p) - bit_estimator(z_q - 0.5, p)
    return torch.sum(torch.clamp(-1.0 * torch.log(prob + 1e-5) / math.log(2.0), 0, 50))


# ----------------------------------------------------------------------------- sigma -> table index
def laplace_indexes(scales):
    """GaussianEncoder.build_indexes: 256 levels over [0.01, 64] (video_entropy_models.py:247-258,309-313)."""
    lo, hi, levels = math.log(0.01), math.log(64.0), 256
    step = (hi - lo) / (levels - 1)
    s = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
    return ((torch.log(s) - lo) / step).clamp_(0, levels - 1).int()


def gaussian_indexes(scales):
    """GaussianConditional.build_indexes: 64 levels over [0.11, 256], +1 (img_entropy_models.py:589-596,687-691)."""
    lo, hi, levels = math.log(0.11), math.log(256.0), 64
    step = (hi - lo) / (levels - 1)
    s = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
    return ((torch.log(s) - lo) / step + 1).clamp_(0, levels - 1).int()
