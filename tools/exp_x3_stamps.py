"""[diagnostic build: RNNT_ENGINE_LIB=build_variants/x3/lib_stamps<w>.so, x3.hip + engine.hip with -DRNNT_STAMPS -DXS_WAVE=w]
k_joint_fwd_x3, workgroup 0, k-steps 8..23 of its first tile: cycles between the stamps of a k-step."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth
from rnnt_amd import engine
B, T, U, H, V = 32, 1000, 200, 512, 1024
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
run = lambda st: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V-1, 1/B, outs=outs, stage=st, dtype=os.environ.get("STAMP_DTYPE", "bf16x3"))
for s in range(8): run(s)
torch.cuda.synchronize()
dbg = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(dbg.data_ptr()))
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # 1: k_joint_fwd_x3, 4: k_dhidden_x3
run(STAGE); torch.cuda.synchronize()
engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(0))
d = dbg.cpu().numpy().reshape(16, 8)
names = ["vmcnt wait", "barrier", "reads issue+land", "blocks 0-1", "block 2 (+loads)", "block 3", "blocks 4-5 (+stores)"]
seg = np.diff(d[:, :7], axis=1)
print(os.path.basename(os.environ.get("RNNT_ENGINE_LIB", "")))
labels = ["0->1 vmcnt wait", "1->2 barrier", "2->3 first reads land", "3->4 block 0", "4->5 block 1", "5->6 block 2 (+loads, stores)"] if os.environ.get("STAMP_DTYPE") == "f16x2" else ["0->1 vmcnt wait", "1->2 barrier", "2->3 first reads land", "3->4 blocks 0,1", "4->5 blocks 2,3", "5->6 blocks 4,5"] if STAGE == 1 else \
         ["0->1 vmcnt wait", "1->2 barrier", "2->3 first reads land", "3->4 blocks 0,1", "4->5 block 2, stores, loads, block 3", "5->6 blocks 4,5"]
for i, n in enumerate(labels):
    print(f"  {n:24s} median {np.median(seg[:, i]):7.0f}  min {seg[:, i].min():6d} max {seg[:, i].max():6d}")
step = np.diff(d[:, 0])
print("  k-step period (stamp 0 to next stamp 0): median", np.median(step), " -> ideal 96 (f16x2: 48) MFMAs x 32 = 3072 (1536)")
