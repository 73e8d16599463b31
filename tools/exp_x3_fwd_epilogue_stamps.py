"""[diagnostic build: RNNT_ENGINE_LIB=build_variants/x3/lib_estamps.so = x3.hip + engine.hip with -DRNNT_STAMPS]
k_joint_fwd_x3, workgroup 0, wave XS_WAVE, its third tile: cycles of the passes' k loops and epilogues."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth
from rnnt_amd import engine
B, T, U, H, V = 32, 1000, 200, 512, 1024
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
run = lambda st: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=st, dtype="bf16x3")
for s in range(8): run(s)
torch.cuda.synchronize()
dbg = torch.zeros(256, dtype=torch.int64, device="cuda")
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(dbg.data_ptr()))
run(1); torch.cuda.synchronize()
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(0))
x = dbg.cpu().numpy()[128:]
print(os.path.basename(os.environ.get("RNNT_ENGINE_LIB", "")))
for p in range(V // 512):
    b = 32 * p
    print(f"pass {p}: k loop {x[b+1]-x[b]}, epilogue quarters {x[b+2]-x[b+1]} {x[b+3]-x[b+2]} {x[b+4]-x[b+3]} {x[b+5]-x[b+4]}, total epilogue {x[b+6]-x[b+1]}")
print(f"store drain {x[97]-x[96]}, finalisation {x[98]-x[97]}, tile {x[98]-x[0]}")
dt, dr = x[102] - x[100], x[103] - x[101]
print(f"workgroup 0 lifetime: {dt} s_memtime ticks, {dr} s_memrealtime ticks (100 MHz): {dr / 100e6 * 1e3:.2f} ms, s_memtime rate {dt / dr * 100:.0f} MHz")
