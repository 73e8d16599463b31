"""ctypes/numpy front-end of oracle/liboracle.so (the C restatement in rnnt_oracle.c).

TEST INFRASTRUCTURE ONLY — see oracle/rnnt_oracle.c for provenance and the
"parity unpinned" note on the loss half.  Each function cites the reference lines
its C counterpart follows (paths relative to /root/reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rnnt_oracle.c", "rnnt_oracle_body.inc")]
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", ctypes.c_double
    if dtype == np.float32:
        return "f32", ctypes.c_float
    raise TypeError(dtype)


def joint_fwd(enc, pred, W, bias, dtype=np.float64):
    """logits[B,T,U1,V] = tanh(enc[:, :, None] + pred[:, None]) @ W.T + bias
    (reference rnnt/joint.py:32-39)."""
    sfx, _ = _sfx(dtype)
    enc, pred, W, bias = (np.ascontiguousarray(x, dtype=dtype) for x in (enc, pred, W, bias))
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    out = np.empty((B, T, U1, V), dtype=dtype)
    getattr(lib(), f"rnnt_oracle_joint_fwd_{sfx}")(
        _ptr(enc), _ptr(pred), _ptr(W), _ptr(bias), B, T, U1, H, V, _ptr(out))
    return out


def rnnt_loss(logits, targets, logit_lens, target_lens, blank=-1, clamp=-1.0,
              dtype=np.float64, want_grad=True, want_work=False):
    """Per-utterance transducer costs and d cost_b / d logits (unscaled), the call made at
    reference rnnt/model.py:35-41.  Returns (costs[B], grad[B,T,U1,V] | None[, work])."""
    sfx, creal = _sfx(dtype)
    logits = np.ascontiguousarray(logits, dtype=dtype)
    B, T, U1, V = logits.shape
    targets = np.ascontiguousarray(targets, dtype=np.int32).reshape(B, U1 - 1)
    logit_lens = np.ascontiguousarray(logit_lens, dtype=np.int32)
    target_lens = np.ascontiguousarray(target_lens, dtype=np.int32)
    costs = np.empty(B, dtype=dtype)
    grad = np.empty_like(logits) if want_grad else None
    al = be = de = None
    if want_work:
        al = np.empty((B, T, U1), dtype=dtype)
        be = np.empty((B, T, U1), dtype=dtype)
        de = np.empty((B, T, U1), dtype=dtype)
    fn = getattr(lib(), f"rnnt_oracle_loss_{sfx}")
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [creal] + [ctypes.c_void_p] * 5
    fn(_ptr(logits), _ptr(targets), _ptr(logit_lens), _ptr(target_lens), B, T, U1, V,
       int(blank), float(clamp), _ptr(costs), _ptr(grad), _ptr(al), _ptr(be), _ptr(de))
    if want_work:
        return costs, grad, dict(alpha=al, beta=be, denom=de)
    return costs, grad


def rnnt_loss_par_f32(logits, targets, logit_lens, target_lens, blank=-1, want_grad=True):
    """bench.py's cpu_baseline only: rnnt_loss in fp32 with the O(T*U1*V) loops spread over all (b,t,u)
    rows by OpenMP (rnnt_oracle_loss_par_f32).  Same call as reference rnnt/model.py:35-41."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    B, T, U1, V = logits.shape
    targets = np.ascontiguousarray(targets, dtype=np.int32).reshape(B, U1 - 1)
    logit_lens = np.ascontiguousarray(logit_lens, dtype=np.int32)
    target_lens = np.ascontiguousarray(target_lens, dtype=np.int32)
    costs = np.empty(B, dtype=np.float32)
    grad = np.empty_like(logits) if want_grad else None
    fn = lib().rnnt_oracle_loss_par_f32
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
    fn.restype = None
    fn(_ptr(logits), _ptr(targets), _ptr(logit_lens), _ptr(target_lens), B, T, U1, V, int(blank),
       _ptr(costs), _ptr(grad))
    return costs, grad


def joint_bwd(enc, pred, W, G, dtype=np.float64):
    """Autograd of reference rnnt/joint.py:32-39 for upstream gradient G[B,T,U1,V]."""
    sfx, _ = _sfx(dtype)
    enc, pred, W, G = (np.ascontiguousarray(x, dtype=dtype) for x in (enc, pred, W, G))
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    ge = np.zeros_like(enc)
    gp = np.zeros_like(pred)
    gW = np.zeros_like(W)
    gb = np.zeros(V, dtype=dtype)
    getattr(lib(), f"rnnt_oracle_joint_bwd_{sfx}")(
        _ptr(enc), _ptr(pred), _ptr(W), _ptr(G), B, T, U1, H, V,
        _ptr(ge), _ptr(gp), _ptr(gW), _ptr(gb))
    return ge, gp, gW, gb


def joint_loss_fwd_bwd(enc, pred, W, bias, targets, logit_lens, target_lens, blank=-1,
                       dtype=np.float64):
    """Mean loss + gradients of the mean loss, i.e. reference rnnt/model.py:32-41 followed
    by loss.backward() (rnnt/train.py:133-134), restricted to the joint+loss path.
    Returns dict(loss, costs, grad_enc, grad_pred, grad_W, grad_bias)."""
    sfx, _ = _sfx(dtype)
    enc, pred, W, bias = (np.ascontiguousarray(x, dtype=dtype) for x in (enc, pred, W, bias))
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    targets = np.ascontiguousarray(targets, dtype=np.int32).reshape(B, U1 - 1)
    logit_lens = np.ascontiguousarray(logit_lens, dtype=np.int32)
    target_lens = np.ascontiguousarray(target_lens, dtype=np.int32)
    costs = np.empty(B, dtype=dtype)
    loss = np.zeros(1, dtype=dtype)
    ge = np.empty_like(enc)
    gp = np.empty_like(pred)
    gW = np.empty_like(W)
    gb = np.empty(V, dtype=dtype)
    fn = getattr(lib(), f"rnnt_oracle_joint_loss_fwd_bwd_{sfx}")
    fn.restype = ctypes.c_int
    rc = fn(_ptr(enc), _ptr(pred), _ptr(W), _ptr(bias), _ptr(targets), _ptr(logit_lens),
            _ptr(target_lens), B, T, U1, H, V, int(blank), _ptr(costs), _ptr(loss),
            _ptr(ge), _ptr(gp), _ptr(gW), _ptr(gb))
    if rc != 0:
        raise MemoryError("oracle: host allocation failed")
    return dict(loss=loss[0], costs=costs, grad_enc=ge, grad_pred=gp, grad_W=gW, grad_bias=gb)
