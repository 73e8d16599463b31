"""Experiment: cfg2 shapes with ragged lengths (uniform in [50 %, 100 %] of T and U, first utterance
full) against full lengths: a ragged batch should cost what its live cells cost."""
import sys
sys.path.insert(0, ".")
import torch
from rnnt_amd import engine
import bench

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    B, T, U, H, V = bench.CONFIGS["cfg2"]
    enc, pred, W, bias, targets, ll, tl = bench.synth(B, T, U, H, V, 1234, dev)
    g = torch.Generator().manual_seed(7)
    ll_r = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32); ll_r[0] = T
    tl_r = torch.randint(U // 2, U + 1, (B,), generator=g, dtype=torch.int32); tl_r[0] = U
    names = ["prod", "fwd", "lattice", "coef", "dhidden", "dh_red", "dw", "dw_red"]
    for tag, l1, l2 in (("full", ll, tl), ("ragged", ll_r.to(dev), tl_r.to(dev))):
        def run(stage=None):
            kw = {} if stage is None else {"stage": stage}
            engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, l1, l2, V - 1, 1.0 / 32, dtype=dtype, **kw)
        def timed(stage=None, reps=5):
            run(stage); run(stage)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(stage)
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps
        live = float((l1.cpu().double() * (l2.cpu().double() + 1)).sum() / (B * T * (U + 1)))
        live_t = float(l1.cpu().double().sum() / (B * T))
        st = {n: round(timed(s), 2) for s, n in enumerate(names)}
        print(f"{tag:7s} live cells {live:.3f} (time steps {live_t:.3f})  step {timed():7.2f} ms  {st}", flush=True)
