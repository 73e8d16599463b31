"""Generate tests/golden/predictor_*.npz by importing the REFERENCE's own
rnnt.predictor.ConvPredictor (reference rnnt/predictor.py:189-229, rnnt/causalconv.py).

Run in the build container only (the reference never travels to the GPU box):
    PYTHONPATH=/root/reference python tests/golden/make_golden_predictor.py

Stored per case: input ids (with the leading blank the model prepends, rnnt/model.py:20-21),
the module's state_dict, its output in eval mode (dropout off) from an fp32 and an fp64 copy of the
module, and torch-autograd gradients (fp64 module) of sum(out * G) for a fixed random upstream G
w.r.t. every parameter.  Fixtures are data (arrays); no reference source is copied.
"""
import os

import numpy as np
import torch

from rnnt.predictor import ConvPredictor  # reference, via PYTHONPATH=/root/reference

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (num_symbols, output_dim, symbol_embedding_dim, B, U1, seed)
    "predictor_small": (40, 48, 32, 3, 9, 41),
    "predictor_mid": (1024, 256, 128, 2, 21, 42),   # the reference's vocabulary size
    "predictor_one": (16, 8, 8, 1, 1, 43),          # U1 = 1: only the prepended blank (empty target)
}


def make(name, S, O, E, B, U1, seed):
    torch.manual_seed(seed)
    m = ConvPredictor(S, O, E, dropout=0.3).eval()
    with torch.no_grad():  # non-trivial affine LayerNorm parameters and biases
        for n_, p in m.named_parameters():
            if "layer_norm" in n_:
                p.add_(torch.randn_like(p) * 0.2)
    ids = torch.randint(0, S, (B, U1))
    ids[:, 0] = S - 1
    G = torch.randn(B, U1, O, dtype=torch.float64)
    out32 = m(ids).detach().numpy()
    m64 = ConvPredictor(S, O, E, dropout=0.3).double().eval()
    m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    o64 = m64(ids)
    (o64 * G).sum().backward()
    out = {"ids": ids.numpy(), "G": G.numpy(), "out_f32": out32, "out_f64": o64.detach().numpy(),
           "dims": np.array([S, O, E, B, U1])}
    for k, v in m.state_dict().items():
        out["sd__" + k.replace(".", "__")] = v.numpy()
    for k, p in m64.named_parameters():
        out["grad__" + k.replace(".", "__")] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


if __name__ == "__main__":
    for k, v in CASES.items():
        make(k, *v)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("predictor_")))
