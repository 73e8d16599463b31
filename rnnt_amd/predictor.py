"""ConvPredictor with the constructor, attribute names and state-dict keys of the reference
(rnnt/predictor.py:189-229) whose forward AND backward run on the HIP engine
(C ABI rnnt_engine_conv_predictor_fwd / _bwd).

    predictor:
      _target_: rnnt_amd.predictor.ConvPredictor      # was rnnt.predictor.ConvPredictor
      num_symbols: ..., output_dim: 1024, symbol_embedding_dim: 512, dropout: 0.3

`CausalConv1d` only holds the parameters under the reference's names (`conv1.conv.weight`, ...); its
own forward is plain torch and not on the engine's path.  Dropout masks are drawn with torch's
generator (so `torch.manual_seed` governs them) and handed to the kernels as keep bytes.
CPU tensors are rejected: there is no fallback.
"""
import ctypes

import torch

from . import engine


class CausalConv1d(torch.nn.Module):
    """Parameter holder with the reference's layout (rnnt/causalconv.py:9-32): `self.conv` is a
    Conv1d(in, out, kernel_size, stride, dilation) applied after (kernel_size-1)*dilation left zeros."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation, additional_context: int = 0):
        super().__init__()
        if additional_context != 0 or stride != 1 or dilation != 1:
            raise NotImplementedError("rnnt_amd CausalConv1d: stride 1, dilation 1, no look-ahead (ConvPredictor's use)")
        self.conv = torch.nn.Conv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation)
        self.left_padding = (kernel_size - 1) * dilation - stride + 1

    def forward(self, x):
        return self.conv(torch.nn.functional.pad(x, (self.left_padding, 0)))


class _Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "embedding", "ln_in_w", "ln_in_b", "conv1_w", "conv1_b", "conv2_w", "conv2_b", "linear_w",
        "linear_b", "ln_out_w", "ln_out_b")]


def _struct(tensors):
    return _Params(*[t.data_ptr() for t in tensors])


class _ConvPredictorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, keep1, keep2, p, eps_in, eps_out, *params):
        dev = engine._require_cuda(ids, *params)
        engine._require_dtype(torch.float32, **{f"param{i}": t for i, t in enumerate(params)})
        engine._require_dtype(torch.int64, ids=ids)
        ids = ids.contiguous()
        params = tuple(t.contiguous() for t in params)
        B, U1 = ids.shape
        S, E = params[0].shape
        O = params[7].shape[0]
        lib = engine.lib()
        with torch.cuda.device(dev):
            n = ctypes.c_size_t(0)
            engine._check(lib.rnnt_engine_conv_predictor_saved_bytes(B, U1, S, E, O, ctypes.byref(n)))
            saved = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
            out = torch.empty((B, U1, O), dtype=torch.float32, device=dev)
            st = _struct(params)
            engine._check(lib.rnnt_engine_conv_predictor_fwd(
                engine._p(ids), B, U1, S, E, O, ctypes.byref(st), engine._p(keep1), engine._p(keep2),
                ctypes.c_float(p), ctypes.c_float(eps_in), ctypes.c_float(eps_out), engine._p(out), engine._p(saved),
                ctypes.c_size_t(saved.numel()), engine._stream(dev)))
        ctx.save_for_backward(ids, keep1, keep2, saved, *params)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ids, keep1, keep2, saved, *params = ctx.saved_tensors
        dev = ids.device
        B, U1 = ids.shape
        S, E = params[0].shape
        O = params[7].shape[0]
        grad_out = grad_out.contiguous().float()
        lib = engine.lib()
        with torch.cuda.device(dev):
            grads = [torch.empty_like(t) for t in params]
            sp, sg = _struct(params), _struct(grads)
            engine._check(lib.rnnt_engine_conv_predictor_bwd(
                engine._p(ids), B, U1, S, E, O, ctypes.byref(sp), engine._p(keep1), engine._p(keep2),
                ctypes.c_float(ctx.p), engine._p(grad_out), ctypes.byref(sg), engine._p(saved),
                ctypes.c_size_t(saved.numel()), engine._stream(dev)))
        return (None, None, None, None, None, None, *grads)


class ConvPredictor(torch.nn.Module):
    def __init__(self, num_symbols: int, output_dim: int, symbol_embedding_dim: int, dropout: float) -> None:
        super().__init__()
        self.embedding = torch.nn.Embedding(num_symbols, symbol_embedding_dim)
        self.input_layer_norm = torch.nn.LayerNorm(symbol_embedding_dim)
        self.conv1 = CausalConv1d(symbol_embedding_dim, symbol_embedding_dim, kernel_size=3, stride=1, dilation=1)
        self.conv2 = CausalConv1d(symbol_embedding_dim, symbol_embedding_dim, kernel_size=5, stride=1, dilation=1)
        self.linear = torch.nn.Linear(symbol_embedding_dim, output_dim)
        self.output_layer_norm = torch.nn.LayerNorm(output_dim)
        self.dropout = torch.nn.Dropout(p=dropout)

    def _params(self):
        return (self.embedding.weight, self.input_layer_norm.weight, self.input_layer_norm.bias,
                self.conv1.conv.weight, self.conv1.conv.bias, self.conv2.conv.weight, self.conv2.conv.bias,
                self.linear.weight, self.linear.bias, self.output_layer_norm.weight, self.output_layer_norm.bias)

    def forward(self, input, keep_masks=None):
        """input [B,U] int64 symbol ids -> [B,U,output_dim] (reference predictor.py:211-229).
        `keep_masks` (test aid): explicit (keep1, keep2) uint8 tensors [B,U,E] instead of drawing them."""
        p = float(self.dropout.p) if self.training else 0.0
        keep1 = keep2 = None
        if keep_masks is not None:
            keep1, keep2 = (k.to(torch.uint8).contiguous() for k in keep_masks)
        elif p > 0.0:
            shape = (*input.shape, self.embedding.embedding_dim)
            keep1 = (torch.rand(shape, device=input.device) >= p).to(torch.uint8)
            keep2 = (torch.rand(shape, device=input.device) >= p).to(torch.uint8)
        return _ConvPredictorFn.apply(input, keep1, keep2, p, float(self.input_layer_norm.eps), float(self.output_layer_norm.eps), *self._params())
