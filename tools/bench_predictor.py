"""ConvPredictor forward+backward: the engine's kernels (rnnt_amd.ConvPredictor) vs the same op
sequence in eager torch (embedding, LayerNorm, pad + Conv1d, gelu, Linear, LayerNorm on rocBLAS /
MIOpen), at the reference's sizes (S=1024, E=512, O=1024).  python tools/bench_predictor.py [BxU1 ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd  # noqa: E402


class TorchPredictor(torch.nn.Module):
    def __init__(self, S, O, E):
        super().__init__()
        self.embedding = torch.nn.Embedding(S, E)
        self.input_layer_norm = torch.nn.LayerNorm(E)
        self.c1 = torch.nn.Conv1d(E, E, 3)
        self.c2 = torch.nn.Conv1d(E, E, 5)
        self.linear = torch.nn.Linear(E, O)
        self.output_layer_norm = torch.nn.LayerNorm(O)

    def forward(self, ids):
        x = self.input_layer_norm(self.embedding(ids)).permute(0, 2, 1)
        x = torch.nn.functional.gelu(self.c1(torch.nn.functional.pad(x, (2, 0))))
        x = torch.nn.functional.gelu(self.c2(torch.nn.functional.pad(x, (4, 0))))
        return self.output_layer_norm(self.linear(x.permute(0, 2, 1)))


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2, 51), (8, 51), (8, 101), (32, 101), (32, 201)]
for B, U1 in SIZES:
    S, E, O = 1024, 512, 1024
    ids = torch.randint(0, S, (B, U1), device="cuda")
    G = torch.randn(B, U1, O, device="cuda")
    res = {}
    for name, m in (("engine", rnnt_amd.ConvPredictor(S, O, E, 0.0).cuda()), ("torch", TorchPredictor(S, O, E).cuda())):
        def step():
            m.zero_grad(set_to_none=True)
            (m(ids) * G).sum().backward()
        res[name] = timeit(step)
        with torch.no_grad():
            res[name + "_fwd"] = timeit(lambda: m(ids))
    M = B * U1
    gflop = 2 * M * (3 + 5) * E * E * 3 / 1e9 + 2 * M * E * O * 3 / 1e9
    print(f"B={B} U1={U1} rows={M}: engine fwd+bwd {res['engine']:.3f} ms (fwd {res['engine_fwd']:.3f}), "
          f"torch eager {res['torch']:.3f} ms (fwd {res['torch_fwd']:.3f}); {gflop:.1f} GFLOP")
