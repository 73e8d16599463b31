"""[diagnostic libraries of tools/x2_whatif_clock.sh] wall clock of the f16x2 route's three GEMM kernels at cfg2 and the core clock workgroup 0
of each ran at (s_memtime / s_memrealtime stamps at its start and end), for one variant library:
    python3 tools/exp_x2_clock.py F128:fwd      (library build_variants/x2c/lib_F128.so = -DX2_EXP=128; G<bits> = -DXG2_EXP; kernels: fwd, dh, dw or all)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LEGEND = {"F": {128: "MFMAs", 2: "W's bytes", 2048: "fragment reads", 256: "logits stores", 32: "production arithmetic"},
          "G": {1: "MFMAs", 2: "W's bytes", 8: "G line stores", 16: "logits loads", 32: "fragment reads"}}


def measure(which):
    import ctypes, torch
    from bench import synth
    from rnnt_amd import engine
    B, T, U, H, V = 32, 1000, 200, 512, 1024
    enc, pred, W, bias, targets, ll, tl = synth(B, T, U, H, V, 1, "cuda")
    outs = engine.alloc_fused_outputs(enc, pred, W)
    run = lambda st: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1 / B, outs=outs, stage=st, dtype="f16x2")
    for s in range(8): run(s)
    dbg = torch.zeros(512, dtype=torch.int64, device="cuda")
    out = []
    for name, st, slot in (("fwd", 1, 228), ("dh", 4, 236), ("dw", 6, 232)):
        if which not in ("all", name):
            continue
        for _ in range(12): run(st)  # sustained load first: the clock settles
        torch.cuda.synchronize()
        engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(dbg.data_ptr()))
        res = []
        for _ in range(5):
            if st == 4: run(1); run(2); run(3)  # fresh logits for the in-place G
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); run(st); e1.record(); e1.synchronize()
            d = dbg.cpu().numpy()
            t0, r0, t1, r1 = d[slot:slot + 4]
            res.append((e0.elapsed_time(e1), (t1 - t0) / max(1, (r1 - r0)) * 0.1, int(t1 - t0)))
        engine.lib().rnnt_engine_set_debug(ctypes.c_void_p(0))
        res.sort()
        out.append(f"{name} {res[2][0]:7.3f} ms at {res[2][1]:.2f} GHz (workgroup 0: {res[2][2]} cycles)")
    return "  |  ".join(out)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        print(measure(sys.argv[2]), flush=True)
        sys.exit(0)
    for spec in sys.argv[1:]:
        lib, which = spec.split(":")
        var, bits = lib[0], lib[1:]
        off = " + ".join(n for b, n in LEGEND[var].items() if int(bits) & b) or "nothing"
        env = dict(os.environ, RNNT_ENGINE_LIB=os.path.join(ROOT, "build_variants", "x2c", f"lib_{lib}.so"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=env, capture_output=True, text=True, timeout=600)
        print(f"{lib:>10s}  without {off:45s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED ' + r.stderr[-400:]}", flush=True)
