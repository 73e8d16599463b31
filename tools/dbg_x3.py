"""debug: which bf16x3 kernel form breaks a shape (fp64 oracle check per variant)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from rnnt_amd import engine
from tests.helpers import make_inputs, oracle_fused
shape = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (4, 100, 24, 512, 1024)
B, T, U, H, V = shape
d = make_inputs(B, T, U, H, V, seed=1234)
ref = oracle_fused(d)
g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
print("lens", d["logit_lens"], d["target_lens"])
for name, var in (("default", 0), ("fwd 2wg", engine.VARIANT_X3_FWD_2WG), ("fwd 8w", engine.VARIANT_X3_FWD_8W)):
    outs = engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
                                     V - 1, 1.0 / B, dtype="bf16x3", variant=var)
    torch.cuda.synchronize()
    err = {k: float(np.abs(o.cpu().numpy() - ref[k]).max() / np.abs(ref[k]).max()) for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias"))}
    gw = outs[3].cpu().numpy()
    bad = np.argwhere(~np.isfinite(gw))
    print(name, {k: "%.1e" % v for k, v in err.items()}, "non-finite grad_W entries:", len(bad), bad[:5].tolist() if len(bad) else "")
