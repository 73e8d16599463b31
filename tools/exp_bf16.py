"""[needs a diagnostic build: make -C rnnt_amd/csrc clean && make -C rnnt_amd/csrc EXTRA=-DRNNT_ABLATE] Experiment: bf16 stage times with parts switched off (rnnt_engine_set_flags bits 256 no stores,
512 no statistics, 1024 no MFMA, 8192 no dHidden epilogue)."""
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
names = {1: "fwd", 4: "dhidden", 6: "dw"}
cases = [(0, "normal"), (4096, "no A/logits loads only"), (2048, "no W loads only"), (512, "no stats only"), (1024, "no MFMA only"), (256, "no stores"), (768, "no stores/stats"),
         (1024 + 768, "no MFMA/stores/stats"), (1024 + 768 + 2048, "..and no W loads"),
         (1024 + 768 + 4096, "..and no A/logits loads"), (1024 + 768 + 2048 + 4096, "..and neither"),
         (8192 + 256, "no dh epilogue/stores"), (8192 + 256 + 2048, "..and no W loads"),
         (8192 + 256 + 4096, "..and no logits loads"), (8192 + 256 + 2048 + 4096, "..and neither"),
         (8192 + 256 + 2048 + 4096 + 1024, "..and no MFMA")]
for stage in (1, 4, 6):
    for flags, name in cases:
        if stage == 6: continue
        if stage == 1 and flags >= 8192: continue
        if stage == 4 and (flags & 512): continue
        engine.lib().rnnt_engine_set_flags(flags)
        print(f"{names[stage]:8s} {name:24s}: {timeit(stage):.2f} ms", flush=True)
engine.lib().rnnt_engine_set_flags(0)
for s in (0, 1, 2, 3): run(s)
