"""cfg2 (full size) on the f16x2 route, repeated: loss and gradient checksums of every call against the fp32-MFMA route's —
a race shows up as a run-to-run difference.   python3 tools/dbg_x2_loss.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, CONFIGS
from rnnt_amd import engine
B, T, U, H, V = CONFIGS[os.environ.get("CFG", "cfg2")]
enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
outs = engine.alloc_fused_outputs(enc, pred, W)
def run(dt, var=0):
    engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype=dt, variant=var)
    torch.cuda.synchronize()
    return [float(outs[0].double().sum())] + [float(o.double().abs().sum()) for o in outs[1:]]
ref = run("fp32")
print("fp32 ", ["%.9g" % v for v in ref])
bad = 0
for name, var in (("f16x2", 0), ("f16x2 fwd=2wg", engine.VARIANT_X2_FWD_2WG), ("f16x2 dw=p16", engine.VARIANT_X2_DW_P16), ("f16x2 fwd=fp32", engine.VARIANT_X3_FP32_FWD), ("f16x2 fwd,dh=fp32", engine.VARIANT_X3_FP32_FWD | engine.VARIANT_X3_FP32_DH)):
    first = None
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        r = run("f16x2", var)
        first = first or r
        rel = [abs(a - b) / abs(b) for a, b in zip(r, ref)]
        # (checksums are sums of |entries|: the bias gradient's carries the cancellation noise of 6.4 M cells, 1e-5 on every route)
        flag = "" if r == first and rel[0] < 1e-6 and max(rel[:4]) < 1e-4 and rel[4] < 1e-3 else "   <-- DIFFERS"
        bad += bool(flag)
        print(f"{name:18s}", ["%.9g" % v for v in r], "rel", ["%.1e" % v for v in rel], flag, flush=True)
print("bad runs:", bad)
for name, var in (("k_joint_fwd_x2", 0), ("k_joint_fwd_x2d (2 WG / CU)", engine.VARIANT_X2_FWD_2WG), ("k_joint_fwd_x2", 0), ("k_joint_fwd_x2d (2 WG / CU)", engine.VARIANT_X2_FWD_2WG)):
    ts = []
    for _ in range(7):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype="f16x2", stage_mask=2, variant=var)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"forward stage {name:30s} {sorted(ts)[3]:7.3f} ms (min {min(ts):.3f})", flush=True)
engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype="f16x2")  # G and hidden planes in place for the dW stage
for name, var in (("k_dw_x2<4>", 0), ("k_dw_x2p (16x16x32)", engine.VARIANT_X2_DW_P16), ("k_dw_x2<8>", engine.VARIANT_X2_DW_8W)) * 2:
    ts = []
    for _ in range(7):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype="f16x2", stage_mask=64, variant=var)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"dW stage {name:30s} {sorted(ts)[3]:7.3f} ms (min {min(ts):.3f})", flush=True)
