"""Eager enqueue vs HIP-graph replay of the fused joint+loss call at launch-bound sizes.
   python tools/bench_graph.py [config ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rnnt_amd import engine

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["cfg1", "small", "ref1024"]):
    for dtype in ("fp32", "bf16"):
        B, T, U, H, V = bench.CONFIGS[cfg]
        enc, pred, W, bias, targets, ll, tl = bench.synth(B, T, U, H, V, 1234, dev)
        outs = engine.alloc_fused_outputs(enc, pred, W)
        call = lambda: engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1.0 / B, outs=outs, dtype=dtype)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                call()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                call()
            n = 200
            def timed(f):
                s.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    f()
                s.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            e1, g1, e2, g2 = timed(call), timed(g.replay), timed(call), timed(g.replay)
        print(f"{cfg} {dtype}: B={B},T={T},U={U},H={H},V={V}  eager {min(e1, e2):.3f} ms/step   graph replay {min(g1, g2):.3f} ms/step   ({(min(g1, g2) / min(e1, e2) - 1) * 100:+.1f} %)", flush=True)
