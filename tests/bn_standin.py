"""Test infrastructure (NOT product code): torch restatements of the four passes of refign_amd/csrc/bn.hip, same buffer
conventions (sums = (sum x, sum x^2, rows) in fp64, bwd_sums = (sum g', sum g' xhat)), so that the statistics-exchange logic of
refign_amd/bn.py can run on CPU tensors over gloo.  install() puts them in place of the kernel launchers."""
import torch


def stats_fwd(xh, sums):
    x = xh.double().reshape(-1, xh.shape[-1])
    c = x.shape[1]
    sums[:c] = x.sum(0)
    sums[c:2 * c] = (x * x).sum(0)
    sums[2 * c] = x.shape[0]


def _consts(sums, c, eps):
    n = sums[2 * c]
    mean = sums[:c] / n
    var = (sums[c:2 * c] / n - mean * mean).clamp_min(0).float()
    return n.float(), mean.float(), var, torch.rsqrt(var + eps)


def apply_fwd(xh, weight, bias, y, sums, bn, relu):
    c = xh.shape[-1]
    n, mean, var, rstd = _consts(sums, c, bn.eps)
    z = (xh.float() - mean) * rstd * weight + bias
    y.copy_(z.clamp_min(0) if relu else z)
    with torch.no_grad():
        bn.running_mean.mul_(1 - bn.momentum).add_(bn.momentum * mean)
        bn.running_var.mul_(1 - bn.momentum).add_(bn.momentum * var * (n / (n - 1).clamp_min(1)))


def stats_bwd(xh, gy, sums, weight, bias, bsums, eps, relu):
    c = xh.shape[-1]
    _, mean, _, rstd = _consts(sums, c, eps)
    xhat = (xh.float() - mean) * rstd
    g = gy.float()
    if relu:
        g = g * ((xhat * weight + bias) > 0)
    bsums[0] = g.reshape(-1, c).sum(0)
    bsums[1] = (g * xhat).reshape(-1, c).sum(0)


def apply_bwd(xh, gy, sums, bsums, weight, bias, gx, eps, relu):
    c = xh.shape[-1]
    n, mean, _, rstd = _consts(sums, c, eps)
    xhat = (xh.float() - mean) * rstd
    g = gy.float()
    if relu:
        g = g * ((xhat * weight + bias) > 0)
    gx.copy_(weight * rstd * (g - (bsums[0] + xhat * bsums[1]) / n))


def install(bnk):
    bnk._stats_fwd, bnk._apply_fwd, bnk._stats_bwd, bnk._apply_bwd = stats_fwd, apply_fwd, stats_bwd, apply_bwd
