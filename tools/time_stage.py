"""Median time of pipeline stages of one route (the library comes from RNNT_ENGINE_LIB or the shipped build).
   python tools/time_stage.py bf16x3 cfg2 1,4,6"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, CONFIGS
from rnnt_amd import engine
dtype, cfg = sys.argv[1], sys.argv[2]
stages = [int(x) for x in sys.argv[3].split(",")]
B, T, U, H, V = CONFIGS[cfg]
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
def run(stage): engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=stage, dtype=dtype)
for s in range(8): run(s)
res = []
for st in stages:
    for s in range(4): run(s)  # a consistent workspace in front of the timed stage
    run(st); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        if st == 4:
            for s in (1, 2, 3): run(s)  # dHidden consumes the logits: regenerate them
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    res.append("stage %d: %.3f ms" % (st, sorted(ts)[2]))
print(os.path.basename(os.environ.get("RNNT_ENGINE_LIB", "shipped")), " ".join(res), flush=True)
