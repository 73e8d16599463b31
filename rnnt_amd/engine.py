"""ctypes binding of librnnt_engine.so (include/rnnt_engine.h) for torch tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every compute call goes
through the C ABI.  There is NO fallback: if the library is missing or a call fails, a
RuntimeError is raised.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# RNNT_ENGINE_LIB: another BUILD of the same engine (a diagnostic -DRNNT_ABLATE / -DRNNT_STAMPS library of
# tools/exp_*.py, an A/B candidate).  Still the HIP engine and nothing else: a missing file raises.
LIB_PATH = os.environ.get("RNNT_ENGINE_LIB") or os.path.join(_CSRC, "librnnt_engine.so")

DTYPE_F32 = 0
DTYPE_BF16 = 1  # bf16 GEMM operands, fp32 accumulate, fp16 logits (workspace), fp32/fp64 loss; fp32 tensors at the boundary
DTYPE_F32_BF16X3 = 2  # fp32-accurate products as six bf16 MFMA products of 3-way split operands (x3.hip)
DTYPE_F32_F16X2 = 3  # fp32-class products as three fp16 MFMA products of scaled, 2-way split operands (x2.hip): half the bf16x3 route's matrix work
# The arithmetic the product ships (JointNetwork.compute_dtype, the default of rnnt_amd.joint_rnnt_loss and of bench.py).  The low-level
# entry points joint_loss_fwd_bwd / joint_loss_fwd take `dtype` as a REQUIRED keyword.
DEFAULT_DTYPE = "f16x2"
_DTYPES = {"fp32": DTYPE_F32, "f32": DTYPE_F32, "float32": DTYPE_F32, DTYPE_F32: DTYPE_F32,
           "bf16": DTYPE_BF16, "bfloat16": DTYPE_BF16, DTYPE_BF16: DTYPE_BF16,
           "bf16x3": DTYPE_F32_BF16X3, "f32_bf16x3": DTYPE_F32_BF16X3, DTYPE_F32_BF16X3: DTYPE_F32_BF16X3,
           "f16x2": DTYPE_F32_F16X2, "f32_f16x2": DTYPE_F32_F16X2, DTYPE_F32_F16X2: DTYPE_F32_F16X2}


class _Int32:
    """argtypes entry for a C `int`: ctypes' own c_int wraps silently (2**31 -> -2**31) and a dimension
    that does not fit must raise, as must a float where a dimension is expected."""

    @classmethod
    def from_param(cls, v):
        if isinstance(v, ctypes.c_int):
            return v
        if isinstance(v, bool) or not isinstance(v, int):
            raise TypeError(f"rnnt_engine: C int argument needs a Python int, got {type(v).__name__} ({v!r})")
        if not -2 ** 31 <= v < 2 ** 31:
            raise OverflowError(f"rnnt_engine: {v} does not fit a C int")
        return ctypes.c_int(v)


# Argument kinds of every entry point of include/rnnt_engine.h, in declaration order (i = int,
# q = int64_t, z = size_t, f = float, d = double, p = any pointer).  tests/test_abi.py re-derives this
# table from the header, so a prototype change that is not mirrored here fails on the CPU.
_KINDS = {"i": _Int32, "q": ctypes.c_int64, "z": ctypes.c_size_t, "f": ctypes.c_float,
          "d": ctypes.c_double, "p": ctypes.c_void_p}
SIGNATURES = {
    "rnnt_engine_version": "",
    "rnnt_engine_set_flags": "i",
    "rnnt_engine_set_debug": "p",
    "rnnt_engine_debug_query": "i",
    "rnnt_engine_last_error": "",
    "rnnt_engine_workspace_bytes": "iiiiiip",
    "rnnt_engine_loss_workspace_bytes": "iiiiip",
    "rnnt_engine_joint_fwd_workspace_bytes": "iiiiiip",
    "rnnt_engine_joint_fwd": "pppppiiiiiippzp",
    "rnnt_engine_loss_fwd_bwd": "ppppiiiiifipppzp",
    "rnnt_engine_joint_loss_fwd_bwd": "ppppppppiiiiiiffippppppzp",
    "rnnt_engine_joint_loss_fwd": "ppppppppiiiiiiippzp",
    "rnnt_engine_joint_bwd_workspace_bytes": "iiiiiip",
    "rnnt_engine_joint_bwd": "pppppiiiiiipppppzp",
    "rnnt_engine_greedy_scan_workspace_bytes": "iiip",
    "rnnt_engine_greedy_scan": "pqqpppiiiiippzp",
    "rnnt_engine_greedy_decode_workspace_bytes": "iiiiip",
    "rnnt_engine_greedy_decode": "pqipiiiffppppiiiiiiiippppzp",
    "rnnt_engine_greedy_decode_persistent_workspace_bytes": "iiiiiiip",
    "rnnt_engine_greedy_decode_persistent": "pqipiiiffppppiiiiipppppzp",
    "rnnt_engine_greedy_decode_tables_bytes": "iiiiip",
    "rnnt_engine_greedy_decode_build_tables": "piiifppipzp",
    "rnnt_engine_grad_norm_workspace_bytes": "ipp",
    "rnnt_engine_grad_norm": "ippppzp",
    "rnnt_engine_adamw_step": "ipppppdddddqpfip",
    "rnnt_engine_adamw_step_dev": "ippppppddddpppfip",
    "rnnt_engine_conv_predictor_saved_bytes": "iiiiip",
    "rnnt_engine_conv_predictor_fwd": "piiiiipppfffppzp",
    "rnnt_engine_conv_predictor_bwd": "piiiiipppfpppzp",
    "rnnt_engine_linear_fwd": "pqppiiipp",
    "rnnt_engine_linear_bwd_workspace_bytes": "iiip",
    "rnnt_engine_linear_bwd": "pqppiiippppzp",
    "rnnt_engine_linear_x2_workspace_bytes": "iiiip",
    "rnnt_engine_linear_x2_fwd": "pqppiiippzp",
    "rnnt_engine_linear_x2_bwd": "pqppiiippppzp",
    "rnnt_engine_allreduce": "pzpp",
    "rnnt_engine_workspace_layout": "iiiiiip",
    "rnnt_engine_run_stage": "ippppppppiiiiiiffippppppzp",
    "rnnt_engine_run_stages": "iippppppppiiiiiiffippppppzp",
}


def dtype_code(dtype):
    if isinstance(dtype, torch.dtype):
        dtype = {torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
    try:
        return _DTYPES[dtype]
    except (KeyError, TypeError):
        raise ValueError(f"rnnt_amd: unsupported compute dtype {dtype!r} (fp32, bf16, bf16x3 or f16x2)") from None
_lock = threading.Lock()
_lib = None
_workspaces = {}
_captured = {}  # (device, stream) -> workspace a captured HIP graph points into: never freed or replaced

EXPORTS = (
    "rnnt_engine_version", "rnnt_engine_last_error", "rnnt_engine_workspace_bytes",
    "rnnt_engine_loss_workspace_bytes", "rnnt_engine_joint_fwd_workspace_bytes",
    "rnnt_engine_joint_fwd", "rnnt_engine_loss_fwd_bwd", "rnnt_engine_joint_loss_fwd_bwd",
    "rnnt_engine_workspace_layout", "rnnt_engine_run_stage",
    "rnnt_engine_greedy_scan_workspace_bytes", "rnnt_engine_greedy_scan",
    "rnnt_engine_greedy_decode_workspace_bytes", "rnnt_engine_greedy_decode",
    "rnnt_engine_greedy_decode_persistent_workspace_bytes", "rnnt_engine_greedy_decode_persistent",
    "rnnt_engine_greedy_decode_tables_bytes", "rnnt_engine_greedy_decode_build_tables",
    "rnnt_engine_joint_loss_fwd", "rnnt_engine_run_stages",
    "rnnt_engine_joint_bwd_workspace_bytes", "rnnt_engine_joint_bwd",
    "rnnt_engine_grad_norm_workspace_bytes", "rnnt_engine_grad_norm", "rnnt_engine_adamw_step",
    "rnnt_engine_adamw_step_dev",
    "rnnt_engine_conv_predictor_saved_bytes", "rnnt_engine_conv_predictor_fwd",
    "rnnt_engine_conv_predictor_bwd", "rnnt_engine_linear_fwd", "rnnt_engine_linear_bwd_workspace_bytes",
    "rnnt_engine_linear_bwd", "rnnt_engine_allreduce",
    "rnnt_engine_linear_x2_workspace_bytes", "rnnt_engine_linear_x2_fwd", "rnnt_engine_linear_x2_bwd",
)

# per-call kernel variants (include/rnnt_engine.h RNNT_VARIANT_*): bit-identical results
VARIANT_SEPARATE_G = 32
VARIANT_SEPARATE_HIDDEN = 64
VARIANT_FWD_LDS_RING = 128
VARIANT_FWD_ONE_WG_PER_TILE = 256
VARIANT_X3_FP32_FWD = 4096  # bf16x3 route: this stage on the fp32 route's kernel (isolation checks; not bit-identical)
VARIANT_X3_FP32_DH = 8192
# kernels of the diagnostic library only (rnnt_amd/csrc/lab/rnnt_engine_lab.h; tools/build_lab.sh): librnnt_engine.so refuses these bits
VARIANT_LAB_MASK = 0x7FFFC000
VARIANT_X3_FWD_2WG = 16384
VARIANT_X3_FWD_8W = 65536
VARIANT_X3_FWD_Z = 262144
VARIANT_X2_FWD_2WG = 1048576
VARIANT_X2_DW_P16 = 2097152
VARIANT_X2_DW_8W = 524288
VARIANT_X3_DW_P16 = 131072
STAGES_ALL = 255
STAGES_FORWARD = 7  # operand producers + joint-forward GEMM + lattice sweep: costs only


class WsLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in (
        "logits", "hidden", "denom_s", "lpb_s", "lpe_s", "alpha_s", "beta_s", "coef", "wpack",
        "enc_copy", "slab_enc", "slab_pred", "slab_w", "slab_b", "counters", "total", "rows_pad")] + [
        (n, ctypes.c_int) for n in ("n_ublk", "n_ttile", "n_split", "D")] + [
        (n, ctypes.c_size_t) for n in ("g_lo", "aux", "aux_bytes", "ep")]


def build(force: bool = False) -> str:
    """Compile librnnt_engine.so for gfx950 with hipcc (rnnt_amd/csrc/Makefile)."""
    cmd = ["make", "-C", _CSRC, "-j4", "-s", "librnnt_engine.so"]
    if force:
        cmd.insert(1, "-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load the engine; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"rnnt_amd: HIP engine {LIB_PATH} is missing; build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). There is no CPU/torch fallback.")
            L = ctypes.CDLL(LIB_PATH)
            for name in EXPORTS:
                if not hasattr(L, name):
                    raise RuntimeError(f"rnnt_amd: {LIB_PATH} does not export {name}")
            for name, kinds in SIGNATURES.items():  # typed bindings: a wrong kind or an int beyond 2^31 raises
                fn = getattr(L, name)
                fn.argtypes = [_KINDS[k] for k in kinds]
                fn.restype = ctypes.c_int
            L.rnnt_engine_last_error.restype = ctypes.c_char_p
            L.rnnt_engine_set_debug.restype = None
            _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        msg = lib().rnnt_engine_last_error().decode()
        if rc == -1:
            raise ValueError(f"rnnt_engine: {msg}")
        raise RuntimeError(f"rnnt_engine error {rc}: {msg}")


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _nonempty(targets):
    """U = 0 (no labels) gives an empty [B,0] tensor whose data_ptr is NULL; the ABI wants a
    valid pointer, the kernels never dereference it in that case."""
    if targets.numel() == 0:
        return torch.zeros(1, dtype=torch.int32, device=targets.device)
    return targets


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(*tensors):
    dev = tensors[0].device
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "rnnt_amd: tensors must live on a HIP device (got %s); the engine has no CPU path"
                % t.device)
        if t.device != dev:
            raise RuntimeError("rnnt_amd: all tensors must be on the same device")
    return dev


def _require_dtype(dtype, **tensors):
    """The C side reinterprets every pointer: a wrong element type would be read past its
    allocation.  Raise instead."""
    for name, t in tensors.items():
        if t is not None and t.dtype != dtype:
            raise RuntimeError(f"rnnt_amd: {name} must be {dtype} (got {t.dtype})")


def _require_contiguous(**tensors):
    for name, t in tensors.items():
        if t is not None and not t.is_contiguous():
            raise RuntimeError(f"rnnt_amd: {name} must be contiguous")


def workspace(device, nbytes):
    """Grow-only scratch buffer per (device, stream): work enqueued on two streams of one device
    never shares scratch memory (torch owns the memory; 256-byte aligned).

    HIP graphs: a call captured by `torch.cuda.graph` has this buffer's address baked into its
    kernel arguments.  A buffer handed out during a capture is therefore PINNED: it is never
    replaced (growth on that stream raises instead of freeing memory a graph still replays into)
    and `release_workspaces()` keeps it alive — INTEGRATION.md "HIP graphs"."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (device.type, idx, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ws is None or ws.numel() < nbytes:
        if key in _captured:
            raise RuntimeError(
                f"rnnt_amd: this stream's workspace ({ws.numel()} B) is baked into a captured HIP graph and the "
                f"call needs {int(nbytes)} B; warm up with the largest shapes before capturing "
                "(rnnt_amd.engine.workspace)")
        _workspaces.pop(key, None)
        ws = None
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    if capturing:
        _captured[key] = ws
    return ws


def release_workspaces():
    """Drop the cached scratch buffers — except those a captured HIP graph points into — and hand their memory back to
    the driver (torch's caching allocator would otherwise keep a freed 145 GB segment that small tensors then settle in:
    the next, larger workspace — config 4 on the bf16x3 route needs 212 GB of the 288 — could not be allocated beside it)."""
    dropped = False
    for key in list(_workspaces):
        if key not in _captured:
            del _workspaces[key]
            dropped = True
    if dropped and torch.cuda.is_initialized():
        torch.cuda.empty_cache()


def layout(B, T, U1, H, V, dtype="fp32"):
    L = WsLayout()
    _check(lib().rnnt_engine_workspace_layout(B, T, U1, H, V, dtype_code(dtype), ctypes.byref(L)))
    return L


def workspace_bytes(B, T, U1, H, V, dtype="fp32"):
    n = ctypes.c_size_t(0)
    _check(lib().rnnt_engine_workspace_bytes(B, T, U1, H, V, dtype_code(dtype), ctypes.byref(n)))
    return n.value


def _strides3(t):
    return (ctypes.c_int64 * 3)(*t.stride())


def joint_fwd(enc, pred, W, bias):
    """logits[B,T,U1,V] = tanh(enc[:, :, None] + pred[:, None]) @ W.T + bias on the GPU
    (reference rnnt/joint.py:32-39).  `enc` may be a non-contiguous (B,T,H) view."""
    dev = _require_cuda(enc, pred, W, bias)
    _require_dtype(torch.float32, enc=enc, pred=pred, W=W, bias=bias)
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    pred, W, bias = pred.contiguous(), W.contiguous(), bias.contiguous()
    with torch.cuda.device(dev):
        logits = torch.empty((B, T, U1, V), dtype=torch.float32, device=dev)
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_joint_fwd_workspace_bytes(B, T, U1, H, V, DTYPE_F32, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        _check(lib().rnnt_engine_joint_fwd(_p(enc), _strides3(enc), _p(pred), _p(W), _p(bias), B, T, U1,
                                           H, V, DTYPE_F32, _p(logits), _p(ws),
                                           ctypes.c_size_t(ws.numel()), _stream(dev)))
    return logits


def joint_bwd(enc, pred, W, grad_logits):
    """Backward of joint_fwd on the engine (C ABI rnnt_engine_joint_bwd; autograd of reference
    rnnt/joint.py:32-39): (grad_enc, grad_pred, grad_W, grad_bias) for an upstream gradient
    `grad_logits` [B,T,U1,V]."""
    dev = _require_cuda(enc, pred, W, grad_logits)
    _require_dtype(torch.float32, enc=enc, pred=pred, W=W, grad_logits=grad_logits)
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    if tuple(grad_logits.shape) != (B, T, U1, V):
        raise RuntimeError("rnnt_amd.joint_bwd: grad_logits must be [B,T,U1,V]")
    pred, W, grad_logits = pred.contiguous(), W.contiguous(), grad_logits.contiguous()
    with torch.cuda.device(dev):
        ge = torch.empty((B, T, H), dtype=torch.float32, device=dev)
        gp = torch.empty((B, U1, H), dtype=torch.float32, device=dev)
        gW = torch.empty((V, H), dtype=torch.float32, device=dev)
        gb = torch.empty(V, dtype=torch.float32, device=dev)
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_joint_bwd_workspace_bytes(B, T, U1, H, V, DTYPE_F32, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        _check(lib().rnnt_engine_joint_bwd(_p(enc), _strides3(enc), _p(pred), _p(W), _p(grad_logits), B, T,
                                           U1, H, V, DTYPE_F32, _p(ge), _p(gp), _p(gW), _p(gb), _p(ws),
                                           ctypes.c_size_t(ws.numel()), _stream(dev)))
    return ge, gp, gW, gb


def loss_fwd_bwd(logits, targets, logit_lens, target_lens, blank, clamp=-1.0, want_grad=True):
    """Per-utterance costs and d(sum costs)/d logits (reference rnnt/model.py:35-41)."""
    dev = _require_cuda(logits, targets, logit_lens, target_lens)
    _require_dtype(torch.float32, logits=logits)
    _require_dtype(torch.int32, targets=targets, logit_lens=logit_lens, target_lens=target_lens)
    _require_contiguous(logits=logits, targets=targets, logit_lens=logit_lens, target_lens=target_lens)
    B, T, U1, V = logits.shape
    targets = _nonempty(targets)
    with torch.cuda.device(dev):
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(logits) if want_grad else None
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_loss_workspace_bytes(B, T, U1, V, DTYPE_F32, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        _check(lib().rnnt_engine_loss_fwd_bwd(_p(logits), _p(targets), _p(logit_lens), _p(target_lens),
                                              B, T, U1, V, int(blank), ctypes.c_float(clamp), DTYPE_F32,
                                              _p(costs), _p(grad), _p(ws), ctypes.c_size_t(ws.numel()),
                                              _stream(dev)))
    return costs, grad


def _fused_args(enc, pred, W, bias, targets, logit_lens, target_lens, blank, grad_scale, outs, ws,
                dtype=DTYPE_F32):
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    costs, ge, gp, gW, gb = outs
    return (_p(enc), _strides3(enc), _p(pred), _p(W), _p(bias), _p(targets), _p(logit_lens),
            _p(target_lens), B, T, U1, H, V, int(blank), ctypes.c_float(-1.0),
            ctypes.c_float(grad_scale), dtype, _p(costs), _p(ge), _p(gp), _p(gW), _p(gb),
            _p(ws), ctypes.c_size_t(ws.numel()), _stream(enc.device))


def alloc_fused_outputs(enc, pred, W):
    dev = enc.device
    B, T, H = enc.shape
    return (torch.empty(B, dtype=torch.float32, device=dev),
            torch.empty((B, T, H), dtype=torch.float32, device=dev),
            torch.empty(pred.shape, dtype=torch.float32, device=dev),
            torch.empty(W.shape, dtype=torch.float32, device=dev),
            torch.empty(W.shape[0], dtype=torch.float32, device=dev))


def _check_fused_inputs(enc, pred, W, bias, targets, logit_lens, target_lens):
    dev = _require_cuda(enc, pred, W, bias, targets, logit_lens, target_lens)
    _require_dtype(torch.float32, enc=enc, pred=pred, W=W, bias=bias)
    _require_dtype(torch.int32, targets=targets, logit_lens=logit_lens, target_lens=target_lens)
    _require_contiguous(pred=pred, W=W, bias=bias, targets=targets, logit_lens=logit_lens,
                        target_lens=target_lens)
    if enc.dim() != 3 or pred.dim() != 3 or W.dim() != 2 or bias.dim() != 1:
        raise RuntimeError("rnnt_amd: enc [B,T,H], pred [B,U1,H], W [V,H], bias [V] expected")
    B, T, H = enc.shape
    if pred.shape[0] != B or pred.shape[2] != H or W.shape[1] != H or bias.shape[0] != W.shape[0]:
        raise RuntimeError("rnnt_amd: enc / pred / W / bias shape mismatch")
    if logit_lens.shape[0] != B or target_lens.shape[0] != B:
        raise RuntimeError("rnnt_amd: length tensors must have B entries")
    if targets.numel() and tuple(targets.shape) != (B, pred.shape[1] - 1):
        raise RuntimeError("rnnt_amd: targets must be [B, U1-1]")
    return dev


def joint_loss_fwd_bwd(enc, pred, W, bias, targets, logit_lens, target_lens, blank, grad_scale,
                       outs=None, stage=None, *, dtype, stage_mask=None, variant=0):
    """Fused joint + transducer loss forward AND backward (one C-ABI call).  `dtype` is REQUIRED (no default: a caller
    that forgets it must not silently run a route other than the one it means; DEFAULT_DTYPE is what the product ships).
    Returns (costs[B], grad_enc, grad_pred, grad_W, grad_bias); gradients are those of
    grad_scale * sum_b costs[b].  `stage` (0..7) runs a single pipeline stage (bench aid);
    `stage_mask` / `variant` run any subset of stages with per-call kernel variants (VARIANT_*).
    `dtype` "bf16": the three GEMMs take bf16-rounded operands (tensors stay fp32)."""
    dev = _check_fused_inputs(enc, pred, W, bias, targets, logit_lens, target_lens)
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    targets = _nonempty(targets)
    with torch.cuda.device(dev):
        if outs is None:
            outs = alloc_fused_outputs(enc, pred, W)
        else:
            _require_cuda(enc, *outs)
            _require_dtype(torch.float32, **{f"outs[{i}]": o for i, o in enumerate(outs)})
            _require_contiguous(**{f"outs[{i}]": o for i, o in enumerate(outs)})
        code = dtype_code(dtype)
        need = workspace_bytes(B, T, U1, H, V, code)
        if variant & (VARIANT_X3_FP32_FWD | VARIANT_X3_FP32_DH):
            need += layout(B, T, U1, H, V, code).aux_bytes  # fp32 hidden + W pack of the substituted stages
        ws = workspace(dev, need)
        args = _fused_args(enc, pred, W, bias, targets, logit_lens, target_lens, blank, grad_scale,
                           outs, ws, code)
        if stage_mask is not None or variant:
            mask = STAGES_ALL if stage_mask is None else int(stage_mask)
            _check(lib().rnnt_engine_run_stages(mask, int(variant), *args))
        elif stage is None:
            _check(lib().rnnt_engine_joint_loss_fwd_bwd(*args))
        else:
            _check(lib().rnnt_engine_run_stage(int(stage), *args))
    return outs


def joint_loss_fwd(enc, pred, W, bias, targets, logit_lens, target_lens, blank, *, dtype):
    """Costs only (C ABI rnnt_engine_joint_loss_fwd): the fused path's forward — joint GEMM with
    the log-softmax in its epilogue, lattice sweep — and none of the backward kernels or gradient
    buffers.  What RNNTModel.forward costs under torch.no_grad()."""
    dev = _check_fused_inputs(enc, pred, W, bias, targets, logit_lens, target_lens)
    B, T, H = enc.shape
    U1 = pred.shape[1]
    V = W.shape[0]
    targets = _nonempty(targets)
    code = dtype_code(dtype)
    with torch.cuda.device(dev):
        costs = torch.empty(B, dtype=torch.float32, device=dev)
        ws = workspace(dev, workspace_bytes(B, T, U1, H, V, code))
        _check(lib().rnnt_engine_joint_loss_fwd(
            _p(enc), _strides3(enc), _p(pred), _p(W), _p(bias), _p(targets), _p(logit_lens),
            _p(target_lens), B, T, U1, H, V, int(blank), code, _p(costs), _p(ws),
            ctypes.c_size_t(ws.numel()), _stream(dev)))
    return costs


def _rows(x):
    """[.., K] -> a [M, K] view with unit k-stride and a row stride that is a multiple of 4 floats
    (copying when the caller's layout does not allow 16-byte row loads, e.g. the permuted encoder view)."""
    K = x.shape[-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x.contiguous().view(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 4 != 0 or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    return x2


# Work (rows x in-features x out-features) from which "auto" runs a projection on the f16x2 pipes (rnnt_engine_linear_x2_*: three GEMMs at ~3x
# the fp32 matrix rate around ~12 small launches, a ~0.26 ms floor for forward + backward).  Measured, forward + backward
# (tools/bench_linear.py, profiles/r05_f_linear_bench.txt; engine f16x2 / library / engine fp32-MFMA): 32 000 x 1024 x 1024: 0.94 / 1.45 / 2.38 ms;
# 12 864 x 1024 x 1024: 0.53 / 0.66 / 0.98; 6 432 x 1024 x 1024: 0.29 / 0.38 / 0.55; 32 000 x 512 x 1024: 0.69 / 0.85 / 1.33;
# 8 000 x 512 x 1024: 0.26 / 0.22 / 0.36; 808 x 1024 x 1024: 0.26 / 0.15 / 0.22 — the crossover with rocBLAS / hipBLASLt lies near 5e9.
LINEAR_X2_MIN_MKN = 5_500_000_000


def linear_x2_preferred(M, K, N):
    """"auto": the f16x2 kernels take the shape (K, N multiples of 128) and the GEMM is large enough for them to beat the library."""
    return K % 128 == 0 and N % 128 == 0 and M * K * N >= LINEAR_X2_MIN_MKN


def _linear_x2(M, K, N, backend):
    if backend == "x2":
        if K % 128 or N % 128:
            raise RuntimeError("rnnt_amd.linear: the f16x2 kernels need K % 128 == 0 and N % 128 == 0")
        return True
    return backend == "auto" and linear_x2_preferred(M, K, N)


def linear_fwd(x, W, bias, backend="auto"):
    """y = x W^T + b (reference rnnt/joint.py:26-30).  backend "x2": rnnt_engine_linear_x2_fwd (f16x2 matrix pipes), "fp32":
    rnnt_engine_linear_fwd (fp32-MFMA small-GEMM kernels), "auto": x2 from LINEAR_X2_MIN_MKN (rows x K x N) when the shape allows it."""
    dev = _require_cuda(x, W, bias)
    _require_dtype(torch.float32, x=x, W=W, bias=bias)
    N, K = W.shape
    if x.shape[-1] != K or bias.shape != (N,):
        raise RuntimeError("rnnt_amd.linear: x [..,K], W [N,K], bias [N] expected")
    x2 = _rows(x)
    M = x2.shape[0]
    W, bias = W.contiguous(), bias.contiguous()
    if _linear_x2(M, K, N, backend):
        with torch.cuda.device(dev):
            y = torch.empty((M, N), dtype=torch.float32, device=dev)
            n = ctypes.c_size_t(0)
            _check(lib().rnnt_engine_linear_x2_workspace_bytes(M, K, N, 0, ctypes.byref(n)))
            ws = workspace(dev, n.value)
            _check(lib().rnnt_engine_linear_x2_fwd(_p(x2), ctypes.c_int64(x2.stride(0)), _p(W), _p(bias), M, K, N, _p(y), _p(ws),
                                                   ctypes.c_size_t(ws.numel()), _stream(dev)))
        return y.view(*x.shape[:-1], N)
    with torch.cuda.device(dev):
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        _check(lib().rnnt_engine_linear_fwd(_p(x2), ctypes.c_int64(x2.stride(0)), _p(W), _p(bias), M, K, N,
                                            _p(y), _stream(dev)))
    return y.view(*x.shape[:-1], N)


def linear_bwd(x, W, dy, need_dx=True, backend="auto"):
    """(dx, dW, db) of linear_fwd (C ABI rnnt_engine_linear_x2_bwd / rnnt_engine_linear_bwd, chosen as in linear_fwd)."""
    dev = _require_cuda(x, W, dy)
    _require_dtype(torch.float32, x=x, W=W, dy=dy)
    N, K = W.shape
    x2 = _rows(x)
    M = x2.shape[0]
    dy2 = dy.reshape(M, N).contiguous()
    W = W.contiguous()
    with torch.cuda.device(dev):
        dW = torch.empty((N, K), dtype=torch.float32, device=dev)
        db = torch.empty(N, dtype=torch.float32, device=dev)
        dx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_dx else None
        n = ctypes.c_size_t(0)
        if _linear_x2(M, K, N, backend):
            _check(lib().rnnt_engine_linear_x2_workspace_bytes(M, K, N, 1, ctypes.byref(n)))
            ws = workspace(dev, n.value)
            _check(lib().rnnt_engine_linear_x2_bwd(_p(x2), ctypes.c_int64(x2.stride(0)), _p(W), _p(dy2), M, K, N,
                                                   _p(dx), _p(dW), _p(db), _p(ws), ctypes.c_size_t(ws.numel()), _stream(dev)))
            return (dx.view(x.shape) if need_dx else None), dW, db
        _check(lib().rnnt_engine_linear_bwd_workspace_bytes(M, K, N, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        _check(lib().rnnt_engine_linear_bwd(_p(x2), ctypes.c_int64(x2.stride(0)), _p(W), _p(dy2), M, K, N,
                                            _p(dx), _p(dW), _p(db), _p(ws), ctypes.c_size_t(ws.numel()),
                                            _stream(dev)))
    return (dx.view(x.shape) if need_dx else None), dW, db


GREEDY_SCAN_MAX_FRAMES = 128


def greedy_scan(enc, pred, W, bias, t0, nframes, blank):
    """Greedy-decode scan (C ABI rnnt_engine_greedy_scan; reference rnnt/model.py:108-125 inner
    loop): frames t0 .. t0+nframes-1 of `enc` [T,H] (any strides) against ONE predictor state
    `pred` [H].  Returns an int32 device tensor [2+nframes]: first non-blank frame (t0+nframes if
    none), its token, then the argmax of every frame.  No host synchronisation."""
    dev = _require_cuda(enc, pred, W, bias)
    if enc.dim() != 2 or pred.dim() != 1 or enc.shape[1] != pred.shape[0] or W.shape[1] != pred.shape[0]:
        raise RuntimeError("greedy_scan: enc [T,H], pred [H], W [V,H] expected")
    _require_dtype(torch.float32, enc=enc, pred=pred, W=W, bias=bias)
    T, H = enc.shape
    V = W.shape[0]
    nframes = int(nframes)
    if t0 < 0 or nframes < 1 or t0 + nframes > T:
        raise ValueError(f"greedy_scan: frames [{t0}, {t0 + nframes}) outside [0, {T})")
    with torch.cuda.device(dev):
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_greedy_scan_workspace_bytes(nframes, H, V, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        out = torch.empty(2 + nframes, dtype=torch.int32, device=dev)
        _check(lib().rnnt_engine_greedy_scan(_p(enc), ctypes.c_int64(enc.stride(0)), ctypes.c_int64(enc.stride(1)),
                                             _p(pred.contiguous()), _p(W.contiguous()), _p(bias.contiguous()),
                                             int(t0), nframes, H, V, int(blank), _p(out), _p(ws),
                                             ctypes.c_size_t(ws.numel()), _stream(dev)))
    return out


def _eps_pair(ln_eps):
    """`ln_eps` of the decode entry points: one float (both LayerNorms of the ConvPredictor) or (input_layer_norm.eps, output_layer_norm.eps)."""
    if isinstance(ln_eps, (tuple, list)):
        return float(ln_eps[0]), float(ln_eps[1])
    return float(ln_eps), float(ln_eps)


class _PredParams(ctypes.Structure):  # include/rnnt_engine.h: rnnt_conv_predictor_params
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "embedding", "ln_in_w", "ln_in_b", "conv1_w", "conv1_b", "conv2_w", "conv2_b", "linear_w",
        "linear_b", "ln_out_w", "ln_out_b")]


def _decode_inputs(frames, pred_params, text_W, text_b, W, bias):
    params = [t.contiguous() for t in pred_params]
    dev = _require_cuda(frames, W, bias, *params)
    _require_dtype(torch.float32, frames=frames, W=W, bias=bias, **{f"param{i}": t for i, t in enumerate(params)})
    if frames.dim() != 2 or frames.stride(1) != 1:
        frames = frames.contiguous()
    W, bias = W.contiguous(), bias.contiguous()
    if text_W is not None:
        _require_cuda(text_W, text_b)
        _require_dtype(torch.float32, text_W=text_W, text_b=text_b)
        text_W, text_b = text_W.contiguous(), text_b.contiguous()
    return dev, frames, params, text_W, text_b, W, bias


def greedy_decode_persistent_supported(T, S, E, O, H, V, has_text):
    """Whether rnnt_engine_greedy_decode_persistent takes these sizes (else: greedy_decode_loop's kernel-per-layer chain)."""
    n = ctypes.c_size_t(0)
    return lib().rnnt_engine_greedy_decode_persistent_workspace_bytes(int(T), int(S), int(E), int(O), int(H), int(V), int(bool(has_text)),
                                                                      ctypes.byref(n)) == 0


def greedy_decode_tables(pred_params, ln_eps, text_W, text_b, H):
    """The model's part of the persistent decode's memory (C ABI rnnt_engine_greedy_decode_build_tables): conv2's weight pack, conv1 as
    three tap tables over the symbols, joint.text_ln folded into the predictor's linear layer — a function of the parameters only.  Returns a
    device buffer to pass as `tables=` to greedy_decode_persistent for every utterance decoded with THESE parameter values (build again after
    the weights change); enqueued on the current stream, no synchronisation."""
    params = [t.contiguous() for t in pred_params]
    dev = _require_cuda(*params)
    _require_dtype(torch.float32, **{f"param{i}": t for i, t in enumerate(params)})
    if text_W is not None:
        _require_cuda(text_W, text_b)
        _require_dtype(torch.float32, text_W=text_W, text_b=text_b)
        text_W, text_b = text_W.contiguous(), text_b.contiguous()
    S, E = params[0].shape
    O = params[7].shape[0]
    with torch.cuda.device(dev):
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_greedy_decode_tables_bytes(S, E, O, int(H), 1 if text_W is not None else 0, ctypes.byref(n)))
        tables = torch.empty(n.value, dtype=torch.uint8, device=dev)
        st = _PredParams(*[t.data_ptr() for t in params])
        _check(lib().rnnt_engine_greedy_decode_build_tables(ctypes.byref(st), S, E, O, ctypes.c_float(_eps_pair(ln_eps)[0]), _p(text_W), _p(text_b), int(H),
                                                            _p(tables), ctypes.c_size_t(n.value), _stream(dev)))
        tables._keepalive = (params, text_W, text_b)
    return tables


def greedy_decode_persistent(frames, pred_params, ln_eps, text_W, text_b, W, bias, blank, max_length, max_per_frame=10, tables=None):
    """Device-resident greedy decode of one utterance as ONE persistent launch (C ABI rnnt_engine_greedy_decode_persistent;
    reference rnnt/model.py:108-125 with the stateless ConvPredictor).  Arguments as greedy_decode_loop.  Returns (state int32[8],
    tokens int32[max_length]) device tensors WITHOUT synchronising: after one synchronisation tokens[1 : 1 + state[2]] are the
    decoded ids; state[7] != 0 means the loop gave up on a hand-off (check_decode_state raises).  `tables`: greedy_decode_tables(...) of
    the same parameters (else they are rebuilt inside this call, ~0.13 ms)."""
    dev, frames, params, text_W, text_b, W, bias = _decode_inputs(frames, pred_params, text_W, text_b, W, bias)
    T, H = frames.shape
    V = W.shape[0]
    S, E = params[0].shape
    O = params[7].shape[0]
    with torch.cuda.device(dev):
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_greedy_decode_persistent_workspace_bytes(T, S, E, O, H, V, 1 if text_W is not None else 0, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        both = torch.empty(8 + int(max_length), dtype=torch.int32, device=dev)  # state | tokens: ONE device-to-host copy reads the whole result
        state, tokens = both[:8], both[8:]
        st = _PredParams(*[t.data_ptr() for t in params])
        _check(lib().rnnt_engine_greedy_decode_persistent(
            _p(frames), ctypes.c_int64(frames.stride(0)), T, ctypes.byref(st), S, E, O, ctypes.c_float(_eps_pair(ln_eps)[0]), ctypes.c_float(_eps_pair(ln_eps)[1]),
            _p(text_W), _p(text_b), _p(W), _p(bias), H, V, int(blank), int(max_length), int(max_per_frame),
            _p(tables), None, _p(state), _p(tokens), _p(ws), ctypes.c_size_t(ws.numel()), _stream(dev)))
        state._keepalive = (frames, params, W, bias, text_W, text_b, tables)  # until the caller has synchronised
        state._with_tokens = both
    return state, tokens


DECODE_RANGE_CODES = (10, 11)  # decode.hip DP_CODE_FRAME_RANGE / DP_CODE_TEXT_RANGE


def check_decode_state(state_list):
    """After the synchronisation: raise if the persistent loop did not decode — state[7]: the hand-off that never arrived, or 10 / 11: an
    audio frame / a text vector with an entry beyond +-30 (or non-finite), where tanh(e + p) = 1 - 2 / (1 + exp 2e exp 2p) is not exact
    (greedy_decode_loop takes tanh of the sum and has no such limit)."""
    if state_list[7] in DECODE_RANGE_CODES:
        raise RuntimeError(f"rnnt_engine: the persistent greedy decode met an {'audio frame' if state_list[7] == 10 else 'text vector'} "
                           "entry beyond +-30 (or non-finite): use greedy_decode_loop for this utterance")
    if state_list[7] != 0:
        raise RuntimeError(f"rnnt_engine: the persistent greedy decode gave up waiting for hand-off {state_list[7]} "
                           f"(iteration {state_list[5]}, {state_list[6]} workgroups): are all of its workgroups resident?")


def greedy_decode_loop(frames, pred_params, ln_eps, text_W, text_b, W, bias, blank, max_length,
                       max_per_frame=10, scan_frames=64, chunk=8):
    """Device-resident greedy decode of one utterance (C ABI rnnt_engine_greedy_decode; reference
    rnnt/model.py:108-125 with the stateless ConvPredictor).  `frames` [T,H] fp32 audio frames (after audio_ln),
    `pred_params` the 11 ConvPredictor parameter tensors in rnnt_conv_predictor_params order, `text_W` / `text_b`
    joint.text_ln's parameters or None.  Iterations are enqueued `chunk` at a time until the device has raised the
    end-of-loop flag in a pinned host word (polled, never waited for) or the loop's upper bound is reached.
    Returns (state int32[8], tokens int32[max_length]) device tensors WITHOUT synchronising: after one
    synchronisation tokens[1 : 1 + state[2]] are the decoded ids."""
    dev, frames, params, text_W, text_b, W, bias = _decode_inputs(frames, pred_params, text_W, text_b, W, bias)
    T, H = frames.shape
    V = W.shape[0]
    S, E = params[0].shape
    O = params[7].shape[0]
    scan_frames, chunk = int(scan_frames), max(1, int(chunk))
    bound = int(max_length) + (T + scan_frames - 1) // scan_frames + 1
    with torch.cuda.device(dev):
        n = ctypes.c_size_t(0)
        _check(lib().rnnt_engine_greedy_decode_workspace_bytes(H, V, E, O, scan_frames, ctypes.byref(n)))
        ws = workspace(dev, n.value)
        state = torch.empty(8, dtype=torch.int32, device=dev)
        tokens = torch.empty(int(max_length), dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        st = _PredParams(*[t.data_ptr() for t in params])
        stream = _stream(dev)
        done = 0
        while done < bound and (done == 0 or int(flag[0]) == 0):
            it = min(chunk, bound - done)
            _check(lib().rnnt_engine_greedy_decode(
                _p(frames), ctypes.c_int64(frames.stride(0)), T, ctypes.byref(st), S, E, O, ctypes.c_float(_eps_pair(ln_eps)[0]), ctypes.c_float(_eps_pair(ln_eps)[1]),
                _p(text_W), _p(text_b), _p(W), _p(bias), H, V, int(blank), int(max_length), int(max_per_frame),
                scan_frames, it, 1 if done == 0 else 0, ctypes.c_void_p(flag.data_ptr()), _p(state), _p(tokens), _p(ws),
                ctypes.c_size_t(ws.numel()), stream))
            done += it
        state._keepalive = (flag, frames, params, W, bias, text_W, text_b)  # until the caller has synchronised
    return state, tokens
