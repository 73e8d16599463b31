"""numpy restatement of the reference's ConvPredictor forward and backward, float64.

TEST INFRASTRUCTURE ONLY (see oracle/rnnt_oracle.c): the checker of rnnt_engine_conv_predictor_*.
Follows /root/reference/rnnt/predictor.py:211-229 line by line —
    embedding -> input_layer_norm -> permute -> conv1 (CausalConv1d k=3) -> gelu -> dropout
    -> conv2 (k=5) -> gelu -> dropout -> permute -> linear -> output_layer_norm
with CausalConv1d = left zero padding of (k-1) then Conv1d, /root/reference/rnnt/causalconv.py:9-32,
torch's LayerNorm (biased variance, eps 1e-5) and exact (erf) GELU.  Pinned by
tests/golden/predictor_*.npz, which were produced by importing the reference module
(tests/golden/make_golden_predictor.py).  Dropout: explicit keep masks (None = eval mode).
"""
from math import sqrt

import numpy as np
from scipy.special import erf

PARAMS = ("embedding.weight", "input_layer_norm.weight", "input_layer_norm.bias", "conv1.conv.weight",
          "conv1.conv.bias", "conv2.conv.weight", "conv2.conv.bias", "linear.weight", "linear.bias",
          "output_layer_norm.weight", "output_layer_norm.bias")


def _ln(x, w, b, eps):
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mean) * rstd
    return xh * w + b, (xh, rstd)


def _ln_bwd(dy, w, cache):
    xh, rstd = cache
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdims=True) - xh * (g * xh).mean(-1, keepdims=True))
    return dx, (dy * xh).reshape(-1, xh.shape[-1]).sum(0), dy.reshape(-1, xh.shape[-1]).sum(0)


def _conv(x, w, b):
    """x [B,U,Cin], w [Cout,Cin,K] (Conv1d layout): y[b,u] = b + sum_j w[:,:,j] x[b,u-(K-1)+j]."""
    B, U, _ = x.shape
    K = w.shape[2]
    xp = np.concatenate([np.zeros((B, K - 1, x.shape[2])), x], axis=1)
    y = np.zeros((B, U, w.shape[0])) + b
    for j in range(K):
        y += xp[:, j:j + U, :] @ w[:, :, j].T
    return y


def _conv_bwd(dy, x, w):
    B, U, _ = x.shape
    K = w.shape[2]
    xp = np.concatenate([np.zeros((B, K - 1, x.shape[2])), x], axis=1)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for j in range(K):
        dw[:, :, j] = np.einsum("buo,bui->oi", dy, xp[:, j:j + U, :])
        dxp[:, j:j + U, :] += dy @ w[:, :, j]
    return dxp[:, K - 1:, :], dw, dy.reshape(-1, dy.shape[-1]).sum(0)


def _gelu(x):
    return 0.5 * x * (1.0 + erf(x / sqrt(2.0)))


def _gelu_grad(x):
    return 0.5 * (1.0 + erf(x / sqrt(2.0))) + x * np.exp(-0.5 * x * x) / sqrt(2.0 * np.pi)


def forward(ids, sd, keep1=None, keep2=None, p=0.0, eps=1e-5):
    """Returns (out [B,U,O], cache)."""
    sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
    scale = 1.0 / (1.0 - p)
    x0 = sd["embedding.weight"][ids]
    x1, c_in = _ln(x0, sd["input_layer_norm.weight"], sd["input_layer_norm.bias"], eps)
    y1 = _conv(x1, sd["conv1.conv.weight"], sd["conv1.conv.bias"])
    g1 = _gelu(y1) * (1.0 if keep1 is None else keep1 * scale)
    y2 = _conv(g1, sd["conv2.conv.weight"], sd["conv2.conv.bias"])
    g2 = _gelu(y2) * (1.0 if keep2 is None else keep2 * scale)
    z = g2 @ sd["linear.weight"].T + sd["linear.bias"]
    out, c_out = _ln(z, sd["output_layer_norm.weight"], sd["output_layer_norm.bias"], eps)
    return out, dict(sd=sd, ids=ids, x1=x1, c_in=c_in, y1=y1, g1=g1, y2=y2, g2=g2, c_out=c_out,
                     m1=1.0 if keep1 is None else keep1 * scale, m2=1.0 if keep2 is None else keep2 * scale)


def backward(grad_out, cache):
    """Gradients of sum(out * grad_out) w.r.t. every parameter, keyed like the state_dict."""
    sd = cache["sd"]
    g = {}
    dz, g["output_layer_norm.weight"], g["output_layer_norm.bias"] = _ln_bwd(
        np.asarray(grad_out, dtype=np.float64), sd["output_layer_norm.weight"], cache["c_out"])
    E = sd["linear.weight"].shape[1]
    g["linear.weight"] = dz.reshape(-1, dz.shape[-1]).T @ cache["g2"].reshape(-1, E)
    g["linear.bias"] = dz.reshape(-1, dz.shape[-1]).sum(0)
    dy2 = (dz @ sd["linear.weight"]) * cache["m2"] * _gelu_grad(cache["y2"])
    dg1, g["conv2.conv.weight"], g["conv2.conv.bias"] = _conv_bwd(dy2, cache["g1"], sd["conv2.conv.weight"])
    dy1 = dg1 * cache["m1"] * _gelu_grad(cache["y1"])
    dx1, g["conv1.conv.weight"], g["conv1.conv.bias"] = _conv_bwd(dy1, cache["x1"], sd["conv1.conv.weight"])
    dx0, g["input_layer_norm.weight"], g["input_layer_norm.bias"] = _ln_bwd(
        dx1, sd["input_layer_norm.weight"], cache["c_in"])
    demb = np.zeros_like(sd["embedding.weight"])
    np.add.at(demb, cache["ids"].reshape(-1), dx0.reshape(-1, dx0.shape[-1]))
    g["embedding.weight"] = demb
    return g
