"""Soak: the fused call repeated for a while on one input; every output must stay bit-identical to the
first call's (deterministic reductions, no spin time-outs, no stale scratch).
   python tools/soak.py [config] [seconds] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rnnt_amd import engine

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp32"
B, T, U, H, V = bench.CONFIGS[cfg]
dev = torch.device("cuda:0")
enc, pred, W, bias, targets, ll, tl = bench.synth(B, T, U, H, V, 1234, dev)
# ragged lengths: the data-dependent walks (live-granule list, dead tiles) are part of the soak
g = torch.Generator().manual_seed(1)
ll = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32); ll[0] = T
tl = torch.randint(U // 2, U + 1, (B,), generator=g, dtype=torch.int32); tl[0] = U
ll, tl = ll.to(dev), tl.to(dev)
ref = [o.clone() for o in engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1.0 / B, dtype=dtype)]
torch.cuda.synchronize()
assert all(torch.isfinite(r).all() for r in ref)
outs = engine.alloc_fused_outputs(enc, pred, W)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < secs:
    for _ in range(20):
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1.0 / B, outs=outs, dtype=dtype)
        n += 1
    torch.cuda.synchronize()
    if not all(torch.equal(a, b) for a, b in zip(outs, ref)):
        bad += 1
        print("MISMATCH after", n, "calls", flush=True)
    print(f"{n} calls, {time.time() - t0:.0f} s, mismatching checks: {bad}", flush=True)
print("soak", cfg, dtype, "calls", n, "mismatches", bad)
sys.exit(1 if bad else 0)
