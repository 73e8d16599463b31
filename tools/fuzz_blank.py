"""Randomised sweep with an arbitrary blank index (the reference always uses V-1): fused path, both
routes, and the standalone loss, targets drawn from the non-blank symbols.
   python tools/fuzz_blank.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rnnt_amd as amd
from oracle import cpu_oracle
from helpers import make_inputs, bf16_round, assert_close_grad, assert_close_loss, BF16_LOSS_RTOL, BF16_GRAD_RTOL

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
    bad = 0
    for it in range(n):
        bf = it % 3 == 2
        B = int(rng.integers(1, 5)); T = int(rng.integers(1, 50)); U = int(rng.integers(0, 30))
        if bf:
            H = int(rng.choice([128, 256])); V = int(rng.choice([128, 256]))
        else:
            H = 4 * int(rng.integers(1, 150)); V = 4 * int(rng.integers(1, 70))
        blank = int(rng.integers(0, V))
        d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
        t = rng.integers(0, V - 1, (B, max(U, 0))); t = t + (t >= blank)  # skip the blank symbol
        d["targets"] = t.astype(np.int32)
        ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
        ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
        d["logit_lens"] = ll.astype(np.int32); d["target_lens"] = tl.astype(np.int32)
        tag = f"{'bf16' if bf else 'fp32'} blank={blank} B={B} T={T} U={U} H={H} V={V} ll={ll.tolist()} tl={tl.tolist()}"
        try:
            g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
            enc = g["enc"].requires_grad_(True); pred = g["pred"].requires_grad_(True)
            W = g["W"].requires_grad_(True); bias = g["bias"].requires_grad_(True)
            loss, costs = amd.joint_rnnt_loss(enc, pred, W, bias, g["targets"], g["logit_lens"], g["target_lens"],
                                              blank=blank, reduction="mean", return_costs=True,
                                              dtype="bf16" if bf else "fp32")
            loss.backward()
            if not bf:
                ref = cpu_oracle.joint_loss_fwd_bwd(d["enc"], d["pred"], d["W"], d["bias"], d["targets"],
                                                    d["logit_lens"], d["target_lens"], blank=blank, dtype=np.float64)
                assert_close_loss("costs", costs.detach().cpu().numpy(), ref["costs"])
                for k, tt in (("grad_enc", enc), ("grad_pred", pred), ("grad_W", W), ("grad_bias", bias)):
                    assert_close_grad(k, tt.grad.cpu().numpy(), ref[k])
            else:  # the rounding-point oracle of tests/helpers.py with this blank
                hidden = bf16_round(np.tanh(d["enc"][:, :, None, :].astype(np.float64) + d["pred"][:, None, :, :].astype(np.float64)).astype(np.float32)).astype(np.float64)
                Wb = bf16_round(d["W"]).astype(np.float64)
                logits = (hidden.reshape(-1, H) @ Wb.T + d["bias"].astype(np.float64)).astype(np.float32)
                logits = logits.astype(np.float16).astype(np.float32).reshape(B, T, U + 1, V)
                c, G = cpu_oracle.rnnt_loss(logits, d["targets"], d["logit_lens"], d["target_lens"], blank=blank, dtype=np.float64)
                Gb = bf16_round((G / B).astype(np.float32)).astype(np.float64).reshape(-1, V)
                dpre = (Gb @ Wb).reshape(B, T, U + 1, H) * (1.0 - hidden * hidden)
                assert_close_loss("costs", costs.detach().cpu().numpy(), c, rtol=BF16_LOSS_RTOL)
                for k, tt, r in (("grad_enc", enc, dpre.sum(2)), ("grad_pred", pred, dpre.sum(1)),
                                 ("grad_W", W, Gb.T @ hidden.reshape(-1, H)), ("grad_bias", bias, Gb.sum(0))):
                    assert_close_grad(k, tt.grad.cpu().numpy(), r, rtol=BF16_GRAD_RTOL)
            # standalone loss on the same lattice shape
            lg = (rng.standard_normal((B, T, U + 1, V)) * 2).astype(np.float32)
            lt = torch.from_numpy(lg).cuda().requires_grad_(True)
            cs = amd.rnnt_loss(lt, g["targets"], g["logit_lens"], g["target_lens"], blank=blank, reduction="none")
            cs.sum().backward()
            rc, rg = cpu_oracle.rnnt_loss(lg, d["targets"], d["logit_lens"], d["target_lens"], blank=blank)
            assert_close_loss("loss costs", cs.detach().cpu().numpy(), rc)
            assert_close_grad("grad_logits", lt.grad.cpu().numpy(), rg)
            print("ok  ", tag, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, "::", str(e)[:300], flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
