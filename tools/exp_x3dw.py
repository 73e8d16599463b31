"""dW stage time of the bf16x3 route: shipped library against X3_EXP variant libraries.   python3 tools/exp_x3dw.py [config] [exp ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(cfg):
    import torch
    from bench import synth, CONFIGS
    from rnnt_amd import engine
    B, T, U, H, V = CONFIGS[cfg]
    enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
    outs = engine.alloc_fused_outputs(enc, pred, W)

    def run(mask, var=0):
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype="bf16x3", stage_mask=mask, variant=var)

    run(255)
    res = {}
    for _ in range(5):  # interleaved rounds: the 16x16x32 pair form (default) against the 32x32x16 kernel
        for name, var in (("pair16", engine.VARIANT_X3_DW_P16), ("v1_32", 0)):
            run(64, var)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(64, var); run(64, var); e1.record(); e1.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 2)
    return "  ".join(f"dw[{n}] {sorted(v)[2]:7.3f} ms (min {min(v):7.3f})" for n, v in res.items())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(measure(sys.argv[2]), flush=True)
        sys.exit(0)
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    for e in [int(x) for x in sys.argv[2:]] or [0]:
        env = dict(os.environ)
        if e:
            env["RNNT_ENGINE_LIB"] = os.path.join(ROOT, "build_variants", "x3", f"lib_{e}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg], env=env, capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]
        print(f"{cfg} X3_EXP={e:8d} {line}", flush=True)
