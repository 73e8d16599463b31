"""[diagnostic build: RNNT_ENGINE_LIB=build_variants/bf16/lib_stamps.so = bf16.hip + engine.hip with -DRNNT_STAMPS]
k_joint_fwd_bf16_ra, workgroup FRS_BLOCK, all four waves: cycles between the stamps of one tile."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth
from rnnt_amd import engine
B, T, U, H, V = 32, 1000, 200, 512, 1024
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
run = lambda st: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=st, dtype="bf16")
for s in range(8): run(s)
torch.cuda.synchronize()
dbg = torch.zeros(4 * 128, dtype=torch.int64, device="cuda")
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(dbg.data_ptr()))
run(1); torch.cuda.synchronize()
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(0))
d = dbg.cpu().numpy().reshape(4, 128)
print(os.path.basename(os.environ.get("RNNT_ENGINE_LIB", "")))
for w in range(4):
    x = d[w]
    print(f"wave {w}: DMA issue {x[1]-x[0]}, production {x[2]-x[1]}, first barrier {x[3]-x[2]}, total {x[60]-x[0]}")
    print("   pass: chunks / epilogue:", " ".join(f"{x[5+3*p]-x[4+3*p]}/{x[6+3*p]-x[5+3*p]}" for p in range(V // 128)))
    print("   pass 3 chunks (vmcnt wait | barrier | body):", " ".join(
        f"{x[65+3*c]-x[64+3*c]}|{x[66+3*c]-x[65+3*c]}" for c in range(H // 64)),
        " gaps:", " ".join(f"{x[64+3*(c+1)]-x[66+3*c]}" for c in range(H // 64 - 1)))
