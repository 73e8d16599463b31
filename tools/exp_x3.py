"""[diagnostic build: RNNT_ENGINE_LIB=build_variants/ablate/librnnt_engine_ablate.so] bf16x3 stage times with parts
switched off (flags 1024 no MFMA, 4096 no fragment reads, 8192 no DMA)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, CONFIGS
from rnnt_amd import engine
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
stages = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6]
B, T, U, H, V = CONFIGS[cfg]
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
def run(stage): engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=stage, dtype="bf16x3")
for s in range(8): run(s)
def timeit(stage, n=5):
    run(stage); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(stage); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]
names = {1: "fwd", 4: "dhidden", 6: "dw"}
for stage in stages:
    extra = ((256, "no G stores"), (16384, "no epilogue"), (256 + 16384, "no G stores, no epilogue"), (256 + 8192, "no G stores, no DMA"),
             (1024 + 256 + 4096 + 8192 + 16384, "production + barriers only")) if stage == 4 else ()
    for flags, label in ((0, "shipped"),) + extra + ( (1024, "no MFMA"), (8192, "no DMA"), (4096, "no fragment reads"), (1024 + 4096, "DMA only"),
                         (8192 + 4096, "MFMA only"), (1024 + 8192, "reads only"), (1024 + 4096 + 8192, "skeleton (barriers)")):
        if flags == 0 and label != "shipped": continue
        engine.lib().rnnt_engine_set_flags(flags)
        print(f"{names[stage]:8s} {label:24s} {timeit(stage):7.3f} ms", flush=True)
    engine.lib().rnnt_engine_set_flags(0)
    for s in range(8): run(s)
