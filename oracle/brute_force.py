"""Brute-force transducer likelihood by explicit alignment enumeration.

TEST INFRASTRUCTURE ONLY.  Independent of the alpha/beta recurrences: enumerates every
monotone path through the (T x U+1) lattice (pure-Python loops, tiny cases only) and sums
path probabilities in double precision.  Used to pin oracle/rnnt_oracle.c's loss, which
restates the call at reference rnnt/model.py:35-41.
"""
import itertools
import math

import numpy as np


def log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def nll_bruteforce(logits, targets, T, U, blank):
    """-log sum_{alignments} prod p(step).  logits [Tmax,U1max,V]; uses t<T, u<=U.
    A path is a sequence of T blanks and U emits whose last symbol is the blank leaving
    (T-1, U)."""
    lp = log_softmax(np.asarray(logits, dtype=np.float64))
    total = -math.inf
    # choose positions of the U emits among the first T+U-1 steps (last step is a blank)
    n = T + U - 1
    for emit_pos in itertools.combinations(range(n), U):
        emit_pos = set(emit_pos)
        t = u = 0
        s = 0.0
        ok = True
        for i in range(n):
            if i in emit_pos:
                s += lp[t, u, targets[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
                if t >= T:
                    ok = False
                    break
        if not ok or t != T - 1 or u != U:
            continue
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return -total


def grad_bruteforce(logits, targets, T, U, blank, eps=1e-6):
    """Central finite differences of nll_bruteforce w.r.t. logits (tiny cases only)."""
    logits = np.array(logits, dtype=np.float64)
    g = np.zeros_like(logits)
    it = np.nditer(logits, flags=["multi_index"])
    for _ in it:
        idx = it.multi_index
        if idx[0] >= T or idx[1] > U:
            continue
        old = logits[idx]
        logits[idx] = old + eps
        fp = nll_bruteforce(logits, targets, T, U, blank)
        logits[idx] = old - eps
        fm = nll_bruteforce(logits, targets, T, U, blank)
        logits[idx] = old
        g[idx] = (fp - fm) / (2 * eps)
    return g
