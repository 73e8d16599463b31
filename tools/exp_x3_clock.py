"""[diagnostic libraries of tools/build_x3_stamp_variants.sh] the core clock k_joint_fwd_x3 runs at, full kernel against
stripped variants: workgroup 0 stamps s_memtime and s_memrealtime (100 MHz) at its start and end.
    python3 tools/exp_x3_clock.py <exp> [<exp> ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import ctypes, numpy as np, torch
    from bench import synth
    from rnnt_amd import engine
    B, T, U, H, V = 32, 1000, 200, 512, 1024
    enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
    outs = engine.alloc_fused_outputs(enc, pred, W)
    run = lambda st: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, stage=st, dtype=os.environ.get("STAMP_DTYPE", "bf16x3"))
    for s in range(8): run(s)
    dbg = torch.zeros(512, dtype=torch.int64, device="cuda")
    out = []
    for name, st, slot in (("forward", 1, 228), ("dhidden", 4, 236), ("dw", 6, 232)):
        for _ in range(20): run(st)  # sustained load first: the clock settles
        torch.cuda.synchronize()
        engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(dbg.data_ptr()))
        res = []
        for _ in range(5):
            if st == 4: run(1); run(2); run(3)  # fresh logits for the in-place G
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(st); e1.record(); e1.synchronize()
            d = dbg.cpu().numpy()
            t0, r0, t1, r1 = d[slot:slot + 4]
            res.append((e0.elapsed_time(e1), (t1 - t0) / max(1, (r1 - r0)) * 0.1))
        engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(0))
        res.sort()
        out.append(f"{name} {res[2][0]:7.3f} ms at {res[2][1]:.3f} GHz")
    return "  ".join(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(measure(), flush=True)
        sys.exit(0)
    for e in sys.argv[1:]:
        # <exp>: an X3_EXP variant of tools/build_x3_stamp_variants.sh; "x2": tools/build_x2_stamps.sh's library on the f16x2 route
        env = dict(os.environ, RNNT_ENGINE_LIB=os.path.join(ROOT, "build_variants", "x3", "lib_x2_stamps.so" if e == "x2" else f"lib_stamps_{e}.so"))
        if e == "x2":
            env["STAMP_DTYPE"] = "f16x2"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
        print(f"X3_EXP={e:>5s}  {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr[-300:]}", flush=True)
