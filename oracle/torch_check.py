"""Independent torch restatements used to cross-check the C oracle.

TEST INFRASTRUCTURE ONLY.
 * joint_torch: the same torch op sequence as reference rnnt/joint.py:32-39
   (unsqueeze-broadcast add -> tanh -> linear).  Also the torch-CPU half of bench.py's
   cpu_baseline (the reference's own CPU arithmetic for the joint is exactly these ops).
 * rnnt_loss_torch: log-space alpha recursion written with differentiable torch ops, so
   torch.autograd yields a gradient that never touches the closed-form beta/gradient
   formulas of rnnt_oracle.c (argument meaning as at reference rnnt/model.py:35-41).
"""
import torch


def joint_torch(enc, pred, W, bias):
    hidden = torch.tanh(enc.unsqueeze(2) + pred.unsqueeze(1))
    return torch.nn.functional.linear(hidden, W, bias)


def rnnt_loss_torch(logits, targets, logit_lens, target_lens, blank=-1, reduction="mean"):
    B, T, U1, V = logits.shape
    if blank < 0:
        blank += V
    lp = torch.log_softmax(logits, dim=-1)
    costs = []
    for b in range(B):
        Tb, Ub = int(logit_lens[b]), int(target_lens[b])
        y = targets[b]
        alpha = [[None] * (Ub + 1) for _ in range(Tb)]
        for t in range(Tb):
            for u in range(Ub + 1):
                if t == 0 and u == 0:
                    alpha[0][0] = lp.new_zeros(())
                    continue
                terms = []
                if t > 0:
                    terms.append(alpha[t - 1][u] + lp[b, t - 1, u, blank])
                if u > 0:
                    terms.append(alpha[t][u - 1] + lp[b, t, u - 1, int(y[u - 1])])
                alpha[t][u] = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
        costs.append(-(alpha[Tb - 1][Ub] + lp[b, Tb - 1, Ub, blank]))
    costs = torch.stack(costs)
    if reduction == "mean":
        return costs.mean(), costs
    if reduction == "sum":
        return costs.sum(), costs
    return costs, costs
