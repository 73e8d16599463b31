import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd())
import numpy as np, torch
import rnnt_amd
from tests.helpers import make_inputs, oracle_fused, assert_close_grad, assert_close_loss
from tests.test_gpu_parity import _run_fused, _compare, FUSED_SHAPES
shapes = [(2, 9, 4, 128, 128), (3, 23, 19, 256, 384), (2, 40, 33, 512, 1024), (2, 13, 20, 1024, 256), (3, 21, 9, 640, 128),
          (4, 30, 12, 520, 260), (2, 130, 50, 512, 256)]
for sh in shapes:
    d = make_inputs(*sh, seed=sum(sh))
    ref = oracle_fused(d)
    r32 = _run_fused(rnnt_amd, d)
    r = _run_fused(rnnt_amd, d, dtype="bf16x3")
    def err(r, k): return np.abs(r[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
    print(sh, "x3 err", {k: "%.1e" % err(r, k) for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias")},
          "f32 err", {k: "%.1e" % err(r32, k) for k in ("grad_W", "grad_bias")}, flush=True)
    _compare(r, ref)
print("OK")
