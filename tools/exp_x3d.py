"""Stage time of the bf16x3 forward's forms (round 4).  With no arguments: the shipped library, the three forms of the
forward (8-wave workgroup / two 4-wave workgroups per CU / round-3 kernel).  With X3_EXP variant numbers as arguments
(libraries built by tools/build_x3_variants.sh): one child process per library (RNNT_ENGINE_LIB), same measurement.
    python3 tools/exp_x3d.py [config] [exp ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LABEL = {0: "shipped", 1: "no MFMA", 8: "no W DMA", 128: "no hidden stores", 16: "no pass epilogue", 2: "no logits stores", 32: "no statistics",
         9: "no MFMA, no DMA", 136: "no DMA, no hidden stores", 129: "no MFMA, no hidden stores", 137: "no MFMA/DMA/hidden stores",
         144: "no hidden stores, no epilogue", 16384: "static prio waves 4-7 (nw8)", 8192: "no operand loads", 8322: "no operand loads, no hidden / logits stores", 8330: "no loads / DMA / stores", 4096: "nontemporal hidden stores", 1024: "plain logits stores", 2048: "paced logits stores", 130: "no hidden / logits stores", 34: "no logits stores, no stats",
         1152: "plain logits stores, no hidden stores", 2176: "paced logits stores, no hidden stores", 138: "no DMA / hidden / logits stores", 152: "MFMA + production only", 153: "production only"}


def measure(cfg):
    import torch
    from bench import synth, CONFIGS
    from rnnt_amd import engine
    B, T, U, H, V = CONFIGS[cfg]
    enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
    outs = engine.alloc_fused_outputs(enc, pred, W)

    def run(mask, var):
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, dtype="bf16x3", stage_mask=mask, variant=var)

    run(255, 0)
    out = []
    for name, var in (("z", engine.VARIANT_X3_FWD_Z), ("nw4", engine.VARIANT_X3_FWD_2WG), ("v1", 0)):
        ts = []
        for _ in range(5):
            run(2, var)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(2, var); run(2, var); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 2)
        out.append(f"{name} {sorted(ts)[2]:7.3f}")
    return "  ".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(measure(sys.argv[2]), flush=True)
        sys.exit(0)
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    exps = [int(x) for x in sys.argv[2:]] or [0]
    for e in exps:
        env = dict(os.environ)
        if e:
            env["RNNT_ENGINE_LIB"] = os.path.join(ROOT, "build_variants", "x3", f"lib_{e}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg], env=env, capture_output=True, text=True, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-300:]
        print(f"fwd {cfg} X3_EXP={e:4d} {LABEL.get(e, ''):32s} {line}", flush=True)
