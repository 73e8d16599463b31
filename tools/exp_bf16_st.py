"""[needs a diagnostic build: make -C rnnt_amd/csrc clean && make -C rnnt_amd/csrc EXTRA=-DRNNT_ABLATE] Experiment: do the G / logits stores cost time by themselves or through the in-order vmcnt?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from rnnt_amd import engine
B, T, U, H, V = 32, 1000, 200, 512, 1024
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
def run(stage): engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=stage, dtype="bf16")
for s in (0, 1, 2, 3): run(s)
def timeit(stage, n=3):
    run(stage); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(stage)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for stage, nm in ((4, "dhidden"), (1, "fwd")):
    for flags, name in ((8192 + 2048 + 4096, "no loads/epilogue, WITH stores"), (8192 + 256 + 2048 + 4096, "no loads/epilogue/stores"),
                        (8192 + 2048, "no W loads/epilogue, with stores"), (8192 + 2048 + 256, "no W loads/epilogue/stores"),
                        (512 + 2048 + 4096, "fwd: no loads/stats, WITH stores"), (512 + 256 + 2048 + 4096, "fwd: no loads/stats/stores")):
        if (stage == 1) != name.startswith("fwd"): continue
        engine.lib().rnnt_engine_set_flags(flags)
        print(f"{nm:8s} {name:36s}: {timeit(stage):.2f} ms", flush=True)
engine.lib().rnnt_engine_set_flags(0)
