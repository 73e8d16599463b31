"""Randomised sweep of the loss kernels on long / wide lattices (several chained waves, the barrier
fallback, ragged lengths) — the standalone rnnt_loss entry against the fp64 oracle — and of the
fused path with H > 512 (two-kernel backward) and small V.
   python tools/fuzz_lattice.py [n_loss] [n_fused] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import rnnt_amd as amd
from oracle import cpu_oracle
from helpers import make_inputs, oracle_fused, assert_close_grad, assert_close_loss

if __name__ == "__main__":
    n_loss = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n_fused = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 11)
    bad = 0
    for it in range(n_loss):
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 400)); U = int(rng.integers(0, 330)); V = 4 * int(rng.integers(1, 6))
        if it % 7 == 6:
            T, U = int(rng.integers(1300, 1600)), int(rng.integers(280, 330))  # mailboxes > 64 KB: barrier kernel
        logits = (rng.standard_normal((B, T, U + 1, V)) * 2).astype(np.float32)
        targets = rng.integers(0, V - 1, (B, max(U, 0))).astype(np.int32) if V > 1 else np.zeros((B, U), np.int32)
        ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
        ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
        ll = ll.astype(np.int32); tl = tl.astype(np.int32)
        tag = f"loss B={B} T={T} U={U} V={V} ll={ll.tolist()} tl={tl.tolist()}"
        try:
            lt = torch.from_numpy(logits).cuda().requires_grad_(True)
            costs = amd.rnnt_loss(lt, torch.from_numpy(targets).cuda(), torch.from_numpy(ll).cuda(),
                                  torch.from_numpy(tl).cuda(), blank=-1, reduction="none")
            costs.sum().backward()
            ref_c, ref_g = cpu_oracle.rnnt_loss(logits, targets, ll, tl)
            assert_close_loss("costs", costs.detach().cpu().numpy(), ref_c)
            assert_close_grad("grad_logits", lt.grad.cpu().numpy(), ref_g)
            print("ok  ", tag, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, "::", str(e)[:300], flush=True)
            g = lt.grad.cpu().numpy()
            nf = ~np.isfinite(g)
            print("     costs", costs.detach().cpu().numpy(), "ref", ref_c, "non-finite", int(nf.sum()))
            for b in range(B):
                bb = np.argwhere(nf[b].any(-1))
                if len(bb):
                    print("     b", b, "t", bb[:, 0].min(), bb[:, 0].max(), "u", bb[:, 1].min(), bb[:, 1].max(), "cells", len(bb))
            # same inputs again: deterministic?
            lt2 = torch.from_numpy(logits).cuda().requires_grad_(True)
            c2 = amd.rnnt_loss(lt2, torch.from_numpy(targets).cuda(), torch.from_numpy(ll).cuda(),
                               torch.from_numpy(tl).cuda(), blank=-1, reduction="none")
            c2.sum().backward()
            print("     rerun non-finite", int((~torch.isfinite(lt2.grad)).sum()), flush=True)
            # dump the engine's alpha/beta (standalone workspace layout) against a numpy recursion
            from rnnt_amd import engine as eng
            ws = list(eng._workspaces.values())[0]
            U1 = U + 1; D = T + U1 - 1; skew = B * D * U1
            al = lambda n: (n + 255) // 256 * 256
            o_lpb, o_lpe, o_a = al(skew * 4), 2 * al(skew * 4), 3 * al(skew * 4)
            o_b = o_a + al(skew * 8)
            f32 = lambda o: ws[o:o + skew * 4].view(torch.float32).view(B, D, U1).cpu().numpy()
            f64 = lambda o: ws[o:o + skew * 8].view(torch.float64).view(B, D, U1).cpu().numpy()
            lpb_s, lpe_s, a_s, b_s = f32(o_lpb), f32(o_lpe), f64(o_a), f64(o_b)
            lp = logits.astype(np.float64); lp = lp - np.log(np.exp(lp - lp.max(-1, keepdims=True)).sum(-1, keepdims=True)) - lp.max(-1, keepdims=True)
            for b in range(B):
                Tb, Ub = int(ll[b]), int(tl[b])
                alpha = np.full((Tb, Ub + 1), -np.inf); alpha[0, 0] = 0
                for t in range(Tb):
                    for u in range(Ub + 1):
                        if t == 0 and u == 0: continue
                        x = alpha[t - 1, u] + lp[b, t - 1, u, V - 1] if t > 0 else -np.inf
                        y = alpha[t, u - 1] + lp[b, t, u - 1, targets[b, u - 1]] if u > 0 else -np.inf
                        alpha[t, u] = np.logaddexp(x, y)
                got = np.array([[a_s[b, t + u, u] for u in range(Ub + 1)] for t in range(Tb)])
                err = np.abs(got - alpha); err[~np.isfinite(err)] = 1e30
                w = np.argwhere(err > 1e-3)
                print("     b", b, "alpha max err", err.max(), "first bad", w[:3].tolist() if len(w) else None)
                if len(w):
                    t0_, u0_ = w[0]
                    print("       got", got[t0_, max(0, u0_ - 2):u0_ + 2], "ref", alpha[t0_, max(0, u0_ - 2):u0_ + 2],
                          "lpe_s", lpe_s[b, t0_ + u0_ - 1, u0_ - 1], "lp", lp[b, t0_, u0_ - 1, targets[b, u0_ - 1]], flush=True)
    for it in range(n_fused):
        B = int(rng.integers(1, 4)); T = int(rng.integers(1, 60)); U = int(rng.integers(0, 30))
        H = 4 * int(rng.integers(129, 300)); V = 4 * int(rng.integers(1, 40))
        d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
        ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
        ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
        d["logit_lens"] = ll.astype(np.int32); d["target_lens"] = tl.astype(np.int32)
        tag = f"fused B={B} T={T} U={U} H={H} V={V} ll={ll.tolist()} tl={tl.tolist()}"
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
            print("ok  ", tag, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, "::", str(e)[:300], flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)
