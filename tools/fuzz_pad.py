"""Randomised sweep with H and V that are NOT multiples of 4 (the Python wrapper pads on the host
side: zero W columns, bias -1e30 for padded vocabulary rows): fused path and plain joint.
   python tools/fuzz_pad.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rnnt_amd as amd
from oracle import cpu_oracle
from helpers import make_inputs, oracle_fused, assert_close_grad, assert_close_loss

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
    bad = 0
    for it in range(n):
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 40)); U = int(rng.integers(0, 25))
        H = int(rng.integers(1, 300)); V = int(rng.integers(2, 200))
        d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
        ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
        ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
        d["logit_lens"] = ll.astype(np.int32); d["target_lens"] = tl.astype(np.int32)
        tag = f"pad B={B} T={T} U={U} H={H} V={V} ll={ll.tolist()} tl={tl.tolist()}"
        try:
            g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
            enc = g["enc"].requires_grad_(True); pred = g["pred"].requires_grad_(True)
            W = g["W"].requires_grad_(True); bias = g["bias"].requires_grad_(True)
            loss, costs = amd.joint_rnnt_loss(enc, pred, W, bias, g["targets"], g["logit_lens"], g["target_lens"],
                                              blank=-1, reduction="mean", return_costs=True)
            loss.backward()
            ref = oracle_fused(d)
            assert_close_loss("costs", costs.detach().cpu().numpy(), ref["costs"])
            for k, t in (("grad_enc", enc), ("grad_pred", pred), ("grad_W", W), ("grad_bias", bias)):
                assert_close_grad(k, t.grad.cpu().numpy(), ref[k])
            lg = amd.joint_logits(g["enc"], g["pred"], g["W"], g["bias"])
            assert tuple(lg.shape) == (B, T, U + 1, V)
            assert_close_grad("logits", lg.detach().cpu().numpy(), cpu_oracle.joint_fwd(d["enc"], d["pred"], d["W"], d["bias"], dtype=np.float64))
            print("ok  ", tag, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, "::", str(e)[:300], flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
