"""f16x2 route against the fp64 oracle and beside the fp32 / bf16x3 routes, per isolation variant.
    python3 tools/dbg_x2.py [variant ...]      variant: dw_only | dw_dhidden | all"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import make_inputs, oracle_fused
from tests.test_gpu_parity import _dev
import rnnt_amd
e = rnnt_amd.engine
VAR = {"dw_only": e.VARIANT_X3_FP32_FWD | e.VARIANT_X3_FP32_DH, "dw_dhidden": e.VARIANT_X3_FP32_FWD, "all": 0,
       "fwd_2wg": e.VARIANT_X2_FWD_2WG, "dwp_only": e.VARIANT_X3_FP32_FWD | e.VARIANT_X3_FP32_DH | e.VARIANT_X2_DW_P16, "dw8_only": e.VARIANT_X3_FP32_FWD | e.VARIANT_X3_FP32_DH | e.VARIANT_X2_DW_8W}
shapes = [(2, 9, 4, 128, 128), (3, 23, 19, 256, 384), (2, 40, 33, 512, 1024), (2, 13, 20, 1024, 256), (3, 21, 9, 640, 128), (2, 130, 50, 512, 256), (4, 100, 24, 512, 1024)]
for variant in sys.argv[1:] or ["dw_only"]:
    for shape in shapes:
        B, T, U, H, V = shape
        d = make_inputs(B, T, U, H, V, seed=sum(shape))
        g = _dev(d)
        ref = oracle_fused(d)
        line = []
        for dt, var in (("fp32", 0), ("bf16x3", 0), ("f16x2", VAR[variant])):
            outs = e.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"], V - 1, 1.0 / B, dtype=dt, variant=var)
            torch.cuda.synchronize()
            errs = [float(np.abs(outs[0].cpu().numpy().astype(np.float64) - ref["costs"]).max() / np.abs(ref["costs"]).max())]
            for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias")):
                x = o.cpu().numpy()
                errs.append(float(np.abs(x - ref[k]).max() / np.abs(ref[k]).max()) if np.isfinite(x).all() else float("nan"))
            line.append(dt + " " + " ".join("%.1e" % v for v in errs))
        print(variant, shape, " | ".join(line), flush=True)
