"""Diagnostic: every INPUT of the fused call (enc, pred, W, bias, targets, lengths) placed so that it ends at
an unmapped page (tools/guard_alloc.hip, HIP virtual-memory API): a kernel that reads past the end of an
input faults instead of silently reading a neighbouring allocation.  Prints one line per case; a memory
access fault aborts the process (the case printed last + the ranges printed for it name the culprit).
   hipcc -shared -fPIC -o tools/libguard.so tools/guard_alloc.hip && python tools/guard_sweep.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rnnt_amd as amd
from helpers import make_inputs, oracle_fused, oracle_fused_bf16, assert_close_loss, BF16_LOSS_RTOL

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libguard.so"))
torch.cuda.init(); torch.zeros(1, device="cuda")


class _Arr:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def guarded(a: np.ndarray):
    base, mapped, end = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_void_p()
    nbytes = max(a.nbytes, 16)
    rc = lib.guard_alloc(ctypes.c_size_t(nbytes), 0, ctypes.byref(base), ctypes.byref(mapped), ctypes.byref(end))
    assert rc == 0, f"guard_alloc failed at step {rc}"
    t = torch.as_tensor(_Arr(end.value, a.shape if a.size else (max(a.size, 1),), a.dtype.str), device="cuda")
    if a.size:
        t.copy_(torch.from_numpy(a))
    else:
        t = t[:0].view(a.shape)
    return t, (end.value, end.value + a.nbytes, base.value + mapped.value)


CASES = [("fp32", s) for s in [(1, 1, 0, 8, 4), (2, 5, 2, 16, 8), (3, 23, 19, 36, 132), (2, 40, 33, 72, 520), (2, 7, 3, 12, 8),
                               (4, 30, 12, 520, 260), (1, 64, 40, 32, 1300), (3, 21, 18, 1024, 64), (2, 13, 6, 640, 1024),
                               (2, 9, 4, 516, 96), (2, 19, 7, 1536, 32), (2, 11, 20, 1152, 96), (3, 33, 5, 1028, 64),
                               (2, 30, 9, 1024, 260), (3, 37, 11, 128, 132), (2, 50, 101, 512, 1024), (2, 130, 50, 512, 256)]] + \
        [("bf16", s) for s in [(1, 1, 0, 128, 128), (2, 9, 4, 128, 128), (3, 23, 19, 256, 384), (2, 40, 33, 512, 1024),
                               (2, 13, 20, 1024, 256), (3, 21, 9, 640, 128), (1, 43, 27, 256, 128), (2, 50, 101, 512, 1024)]]
for dtype, (B, T, U, H, V) in CASES:
    d = make_inputs(B, T, U, H, V, seed=B + T + U + H + V)
    g, ranges = {}, {}
    for k, v in d.items():
        g[k], ranges[k] = guarded(np.ascontiguousarray(v))
    print(f"{dtype} B={B} T={T} U={U} H={H} V={V} " + " ".join(f"{k}=[{r[0]:#x},{r[1]:#x}) unmapped from {r[2]:#x}" for k, r in ranges.items()), flush=True)
    outs = amd.engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
                                         V - 1, 1.0 / B, dtype=dtype)
    torch.cuda.synchronize()
    ref = oracle_fused_bf16(d) if dtype == "bf16" else oracle_fused(d)
    assert_close_loss("costs", outs[0].cpu().numpy(), ref["costs"], rtol=BF16_LOSS_RTOL if dtype == "bf16" else 1e-4)
    print("   ok", flush=True)
print("guard sweep clean:", len(CASES), "cases")
