"""[diagnostic build: RNNT_ENGINE_LIB=build_variants/ablate/librnnt_engine_ablate.so, made with EXTRA=-DRNNT_ABLATE]
VERDICT r2 item 1: is k_joint_fwd_bf16 / k_dw_bf16 bound by the traffic it generates?  Same kernel, all loads and
stores kept, MFMAs (and the softmax statistics / bias dot products) switched off: if the time barely moves the
kernel is traffic-bound."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from rnnt_amd import engine
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
from bench import CONFIGS
B, T, U, H, V = CONFIGS[cfg]
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
def run(stage): engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=stage, dtype="bf16")
for s in range(8): run(s)
def timeit(stage, n=5):
    run(stage); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(stage); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]
res = {"config": cfg}
for stage, name, cases in ((1, "fwd", ((0, "shipped"), (1024, "no MFMA"), (512, "no statistics"), (1024 + 512, "loads+stores only (no MFMA, no statistics)"),
                                       (1024 + 512 + 256, "loads only"), (1024 + 512 + 4096, "W loads + stores only (no A loads)"))),
                           (6, "dw", ((0, "shipped"), (1024, "no MFMA"), (1024 + 2048, "DMA + transposed reads only (no MFMA, no dot2)"),
                                      (1024 + 2048 + 4096, "DMA only")))):
    for flags, label in cases:
        old = engine.lib().rnnt_engine_set_flags(flags)
        t = timeit(stage)
        res[f"{name}: {label}"] = round(t, 3)
        print(f"{name:4s} {label:50s} {t:7.3f} ms", flush=True)
    engine.lib().rnnt_engine_set_flags(0)
    for s in range(8): run(s)  # restore a consistent workspace for the next kernel
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/exp_bf16_traffic.json", "w"), indent=1)
