"""CPU: the numpy ConvPredictor oracle (oracle/predictor_oracle.py) against the golden vectors
produced by importing the reference's rnnt.predictor.ConvPredictor (tests/golden/predictor_*.npz)."""
import os

import numpy as np
import pytest

from oracle import predictor_oracle as po


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = {k[4:].replace("__", "."): z[k] for k in z.files if k.startswith("sd__")}
    grads = {k[6:].replace("__", "."): z[k] for k in z.files if k.startswith("grad__")}
    return z, sd, grads


@pytest.mark.parametrize("name", ["predictor_small", "predictor_mid", "predictor_one"])
def test_predictor_oracle_matches_reference_golden(golden_dir, name):
    z, sd, grads = load_case(golden_dir, name)
    assert set(sd) == set(po.PARAMS)  # the reference's state-dict keys
    out, cache = po.forward(z["ids"], sd)
    np.testing.assert_allclose(out, z["out_f64"], rtol=0, atol=5e-6)  # fp32-stored weights, fp64 run
    np.testing.assert_allclose(out, z["out_f32"], rtol=0, atol=2e-4)
    g = po.backward(z["G"], cache)
    for k in po.PARAMS:
        scale = np.abs(grads[k]).max() + 1e-12
        assert np.abs(g[k] - grads[k]).max() <= 1e-5 * scale, k


def test_predictor_oracle_dropout_masks_and_causality():
    rng = np.random.default_rng(0)
    S, O, E, B, U = 12, 8, 8, 2, 7
    sd = {"embedding.weight": rng.normal(size=(S, E)), "input_layer_norm.weight": rng.normal(size=E) + 1,
          "input_layer_norm.bias": rng.normal(size=E), "conv1.conv.weight": rng.normal(size=(E, E, 3)) * .3,
          "conv1.conv.bias": rng.normal(size=E), "conv2.conv.weight": rng.normal(size=(E, E, 5)) * .3,
          "conv2.conv.bias": rng.normal(size=E), "linear.weight": rng.normal(size=(O, E)),
          "linear.bias": rng.normal(size=O), "output_layer_norm.weight": rng.normal(size=O) + 1,
          "output_layer_norm.bias": rng.normal(size=O)}
    ids = rng.integers(0, S, (B, U))
    out, _ = po.forward(ids, sd)
    ids2 = ids.copy()
    ids2[:, 4:] = (ids2[:, 4:] + 3) % S  # causal: the first 4 outputs cannot see later symbols
    out2, _ = po.forward(ids2, sd)
    np.testing.assert_allclose(out[:, :4], out2[:, :4], atol=1e-12)
    assert np.abs(out[:, 4:] - out2[:, 4:]).max() > 1e-3
    # finite differences through the dropout masks
    k1 = (rng.random((B, U, E)) > 0.3).astype(np.float64)
    k2 = (rng.random((B, U, E)) > 0.3).astype(np.float64)
    G = rng.normal(size=(B, U, O))
    o, cache = po.forward(ids, sd, k1, k2, p=0.3)
    g = po.backward(G, cache)
    for key, idx in (("conv1.conv.weight", (1, 2, 0)), ("linear.weight", (3, 4)), ("embedding.weight", (int(ids[0, 0]), 1)),
                     ("input_layer_norm.weight", (2,)), ("conv2.conv.bias", (5,))):
        h = 1e-6
        sp = {k: v.copy() for k, v in sd.items()}
        sp[key][idx] += h
        sm = {k: v.copy() for k, v in sd.items()}
        sm[key][idx] -= h
        fd = ((po.forward(ids, sp, k1, k2, 0.3)[0] - po.forward(ids, sm, k1, k2, 0.3)[0]) * G).sum() / (2 * h)
        assert abs(fd - g[key][idx]) <= 1e-5 * max(1.0, abs(fd)), key
