"""Randomised sweep of the remaining entry points: the plain joint (logits + autograd), the decode
scan, and the fused path on long / wide lattices with a permuted (B,C,T) encoder view.
   python tools/fuzz_misc.py [n_each] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rnnt_amd as amd
from oracle import cpu_oracle
from helpers import make_inputs, oracle_fused, assert_close_grad, assert_close_loss

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    bad = 0

    def report(ok, tag, e=None):
        global bad
        if ok:
            print("ok  ", tag, flush=True)
        else:
            bad += 1
            print("FAIL", tag, "::", str(e)[:300], flush=True)

    for it in range(n):  # plain joint: logits and gradients for a random upstream gradient
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 50)); U = int(rng.integers(0, 30))
        H = 4 * int(rng.integers(1, 200)); V = 4 * int(rng.integers(1, 90))
        d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
        tag = f"joint B={B} T={T} U={U} H={H} V={V}"
        try:
            g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
            enc = g["enc"].requires_grad_(True); pred = g["pred"].requires_grad_(True)
            W = g["W"].requires_grad_(True); bias = g["bias"].requires_grad_(True)
            logits = amd.joint_logits(enc, pred, W, bias)
            up = rng.standard_normal(logits.shape).astype(np.float32)
            logits.backward(torch.from_numpy(up).cuda())
            ref = cpu_oracle.joint_fwd(d["enc"], d["pred"], d["W"], d["bias"], dtype=np.float64)
            assert_close_grad("logits", logits.detach().cpu().numpy(), ref)
            ge, gp, gW, gb = cpu_oracle.joint_bwd(d["enc"], d["pred"], d["W"], up.astype(np.float64), dtype=np.float64)
            for k, t, r in (("grad_enc", enc, ge), ("grad_pred", pred, gp), ("grad_W", W, gW), ("grad_bias", bias, gb)):
                assert_close_grad(k, t.grad.cpu().numpy(), r)
            report(True, tag)
        except Exception as e:  # noqa: BLE001
            report(False, tag, e)
    for it in range(n):  # decode scan
        T = int(rng.integers(1, 300)); H = 8 * int(rng.integers(1, 130)); V = 4 * int(rng.integers(1, 275))  # the C entry wants V % 4 == 0, H % 8 == 0
        nfr = int(rng.integers(1, min(T, 128) + 1)); t0 = int(rng.integers(0, T - nfr + 1))
        tag = f"scan T={T} H={H} V={V} n={nfr} t0={t0}"
        try:
            torch.manual_seed(int(rng.integers(1 << 30)))
            enc = torch.randn(H, T, device="cuda").permute(1, 0) if it % 2 else torch.randn(T, H, device="cuda")
            pred = torch.randn(H, device="cuda"); W = torch.randn(V, H, device="cuda") / H ** 0.5
            bias = torch.randn(V, device="cuda") * 0.1; blank = V - 1; bias[blank] += 1.0
            out = amd.engine.greedy_scan(enc, pred, W, bias, t0, nfr, blank).cpu().numpy()
            logits = torch.tanh(enc[t0:t0 + nfr].double() + pred.double()) @ W.double().T + bias.double()
            ref = logits.argmax(dim=-1).cpu().numpy()
            top2 = logits.topk(2, dim=-1).values if V > 1 else None
            clear = ((top2[:, 0] - top2[:, 1]) > 1e-4).cpu().numpy()
            assert (out[2:][clear] == ref[clear]).all(), "per-frame argmax"
            if clear.all():
                nb = np.nonzero(ref != blank)[0]
                exp = (t0 + nb[0], ref[nb[0]]) if len(nb) else (t0 + nfr, blank)
                assert (out[0], out[1]) == exp, (out[:2], exp)
            report(True, tag)
        except Exception as e:  # noqa: BLE001
            report(False, tag, e)
    for it in range(n):  # fused, long / wide lattice, permuted encoder view, small H and V
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 260)); U = int(rng.integers(0, 280))
        H = 4 * int(rng.integers(1, 12)); V = 4 * int(rng.integers(1, 8))
        d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
        ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
        ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
        d["logit_lens"] = ll.astype(np.int32); d["target_lens"] = tl.astype(np.int32)
        tag = f"fused-long B={B} T={T} U={U} H={H} V={V} ll={ll.tolist()} tl={tl.tolist()}"
        try:
            g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
            enc_ct = g["enc"].permute(0, 2, 1).contiguous().requires_grad_(True)  # (B,C,T) as the encoder emits
            pred = g["pred"].requires_grad_(True); W = g["W"].requires_grad_(True); bias = g["bias"].requires_grad_(True)
            loss, costs = amd.joint_rnnt_loss(enc_ct.permute(0, 2, 1), pred, W, bias, g["targets"], g["logit_lens"],
                                              g["target_lens"], blank=-1, reduction="mean", return_costs=True)
            loss.backward()
            ref = oracle_fused(d)
            assert_close_loss("costs", costs.detach().cpu().numpy(), ref["costs"])
            assert_close_grad("grad_enc", enc_ct.grad.permute(0, 2, 1).cpu().numpy(), ref["grad_enc"])
            for k, t in (("grad_pred", pred), ("grad_W", W), ("grad_bias", bias)):
                assert_close_grad(k, t.grad.cpu().numpy(), ref[k])
            report(True, tag)
        except Exception as e:  # noqa: BLE001
            report(False, tag, e)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
