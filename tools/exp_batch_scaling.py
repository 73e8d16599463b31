"""Experiment: step time of the per-rank shard the strong-scaling bench hands one GPU
(cfg2 with B = 32/N for N = 1,2,4,8), on one GPU.  Ideal: ms(B) = ms(32) * B/32."""
import sys
sys.path.insert(0, ".")
import torch
from rnnt_amd import engine
import bench

if __name__ == "__main__":
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    dev = torch.device("cuda", 0)
    _, T, U, H, V = bench.CONFIGS["cfg2"]
    names = ["prod", "fwd", "lattice", "coef", "dhidden", "dh_red", "dw", "dw_red"]
    base = None
    for B in (32, 16, 8, 4):
        enc, pred, W, bias, targets, ll, tl = bench.synth(B, T, U, H, V, 1234, dev)
        def run(stage=None):
            kw = {} if stage is None else {"stage": stage}
            engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1.0 / 32, dtype=dtype, **kw)
        def timed(stage=None, reps=5):
            run(stage); run(stage)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(stage)
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps
        ms = timed()
        if base is None:
            base = ms
        st = {n: round(timed(s), 2) for s, n in enumerate(names)}
        print(f"B={B:2d} {ms:8.2f} ms  ideal {base * B / 32:7.2f}  eff {base * B / 32 / ms:.3f}  {st}", flush=True)
