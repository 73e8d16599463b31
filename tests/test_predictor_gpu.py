"""-m gpu: rnnt_amd.ConvPredictor (C ABI rnnt_engine_conv_predictor_fwd/_bwd) and the engine's Linear
(rnnt_engine_linear_fwd/_bwd) against the golden vectors produced by the reference's own
rnnt.predictor.ConvPredictor (tests/golden/predictor_*.npz) and the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import predictor_oracle as po
from tests.helpers import assert_close_grad
from tests.test_predictor_oracle import load_case

pytestmark = pytest.mark.gpu


def _module(amd, sd, S, O, E, p=0.3):
    m = amd.ConvPredictor(S, O, E, dropout=p).cuda()
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)).float() for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys  # the reference's state-dict keys
    return m


@pytest.mark.parametrize("name", ["predictor_small", "predictor_mid", "predictor_one"])
def test_conv_predictor_matches_reference_golden(golden_dir, name):
    import rnnt_amd
    z, sd, grads = load_case(golden_dir, name)
    S, O, E, B, U1 = (int(v) for v in z["dims"])
    m = _module(rnnt_amd, sd, S, O, E).eval()
    out = m(torch.from_numpy(z["ids"]).cuda())
    (out * torch.from_numpy(z["G"]).float().cuda()).sum().backward()
    assert_close_grad("out", out.detach().cpu().numpy(), z["out_f64"], rtol=1e-5, atol=1e-5)
    got = dict(m.named_parameters())
    for k in po.PARAMS:
        assert_close_grad(k, got[k].grad.cpu().numpy(), grads[k])


@pytest.mark.parametrize("B,U1,S,E,O,p", [(8, 51, 1024, 512, 1024, 0.3), (3, 70, 33, 36, 20, 0.5), (32, 5, 64, 128, 64, 0.0)])
def test_conv_predictor_training_mode_vs_oracle(B, U1, S, E, O, p):
    """Training mode with explicit dropout keep masks (the reference draws them inside nn.Dropout),
    at the reference's real sizes (E=512, O=1024, S=1024; basic_sp_convjs_fullcausal.yaml:20-25)."""
    import rnnt_amd
    torch.manual_seed(B + U1)
    m = rnnt_amd.ConvPredictor(S, O, E, dropout=p).cuda().train()
    ids = torch.randint(0, S, (B, U1), device="cuda")
    k1 = (torch.rand(B, U1, E, device="cuda") >= p).to(torch.uint8)
    k2 = (torch.rand(B, U1, E, device="cuda") >= p).to(torch.uint8)
    G = torch.randn(B, U1, O, device="cuda")
    out = m(ids, keep_masks=(k1, k2))
    (out * G).sum().backward()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref, cache = po.forward(ids.cpu().numpy(), sd, k1.cpu().numpy().astype(np.float64),
                            k2.cpu().numpy().astype(np.float64), p=p)
    assert_close_grad("out", out.detach().cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    g = po.backward(G.cpu().numpy(), cache)
    got = dict(m.named_parameters())
    for k in po.PARAMS:
        assert_close_grad(k, got[k].grad.cpu().numpy(), g[k], rtol=2e-4)
    # masks drawn by the module itself: governed by torch's generator, ~p of the units dropped
    if p > 0:
        torch.manual_seed(5); a = m(ids)
        torch.manual_seed(5); b = m(ids)
        torch.manual_seed(6); c = m(ids)
        assert torch.equal(a, b) and not torch.equal(a, c)
        assert torch.equal(m.eval()(ids), m(ids))  # eval: deterministic


def test_conv_predictor_is_deterministic_with_repeated_symbols():
    """The embedding gradient sums rows in a fixed order (no atomics): bitwise reproducible."""
    import rnnt_amd
    torch.manual_seed(0)
    m = rnnt_amd.ConvPredictor(8, 32, 16, dropout=0.0).cuda()
    ids = torch.randint(0, 8, (16, 40), device="cuda")  # every symbol ~80 times
    G = torch.randn(16, 40, 32, device="cuda")
    gs = []
    for _ in range(2):
        m.zero_grad()
        (m(ids) * G).sum().backward()
        gs.append([p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(x, y) for x, y in zip(*gs))


@pytest.mark.parametrize("M,K,N", [(1, 8, 4), (37, 24, 32), (408, 512, 1024), (2100, 1024, 640), (130, 20, 36)])
def test_engine_linear_fwd_bwd_vs_torch(M, K, N):
    import rnnt_amd
    torch.manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", requires_grad=True)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    G = torch.randn(M, N, device="cuda")
    y = rnnt_amd.linear(x, W, b)
    (y * G).sum().backward()
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    ref = torch.nn.functional.linear(x64, W64, b64)
    (ref * G.double()).sum().backward()
    assert_close_grad("y", y.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5)
    for name, a_, r_ in (("dx", x, x64), ("dW", W, W64), ("db", b, b64)):
        assert_close_grad(name, a_.grad.cpu().numpy(), r_.grad.cpu().numpy())


@pytest.mark.parametrize("M,K,N", [(1, 128, 128), (37, 128, 256), (130, 384, 128), (408, 512, 1024), (2100, 1024, 640), (6432, 1024, 1024),
                                   (513, 640, 1152), (700, 640, 512), (300, 896, 1024)])  # (the last two: dW on whole-block + tall tiles, k_dw_x2m)
@pytest.mark.parametrize("xscale,gscale", [(1.0, 1.0), (1e-3, 40.0), (300.0, 1e-5)])
def test_engine_linear_x2_fwd_bwd_vs_float64(M, K, N, xscale, gscale):
    """The projections on the f16x2 matrix pipes (rnnt_engine_linear_x2_fwd / _bwd, round 5: reference rnnt/joint.py:8-12,26-30 —
    y through the joint forward's pipeline as a plain GEMM, dx through the same kernel on W^T, dW / db through the joint's dW kernel)
    against float64 torch at the fp32 tolerances (1e-4 of each result's largest entry), over operand magnitudes that need the
    device-found power-of-two scales, row counts that are not multiples of any tile, a permuted row stride."""
    import rnnt_amd
    torch.manual_seed(M + K + N)
    xbuf = torch.randn(M, K + 8, device="cuda") * xscale
    x = xbuf[:, :K].requires_grad_(True)  # rows K + 8 floats apart
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device="cuda", requires_grad=True)
    G = torch.randn(M, N, device="cuda") * gscale
    y = rnnt_amd.linear(x, W, b, backend="x2")
    (y * G).sum().backward()
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    ref = torch.nn.functional.linear(x64, W64, b64)
    (ref * G.double()).sum().backward()
    assert_close_grad("y", y.detach().cpu().numpy(), ref.detach().cpu().numpy())
    for name, a_, r_ in (("dx", x, x64), ("dW", W, W64), ("db", b, b64)):
        assert_close_grad(name, a_.grad.cpu().numpy(), r_.grad.cpu().numpy())
    # bit-reproducible (fixed summation orders) and no write outside the outputs' rows (M is not a multiple of the 128-row tile)
    y2 = rnnt_amd.linear(x.detach(), W.detach(), b.detach(), backend="x2")
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("where", ["x", "W", "dy"])
@pytest.mark.parametrize("what", [float("nan"), float("inf"), float("-inf")])
def test_engine_linear_x2_propagates_non_finite_operands(where, what):
    """A NaN or an infinity in x, W or dy must not come out finite (round-5 advice: the operand split clamps into fp16's range, the
    magnitude search drops NaNs — a diverged run would keep training on garbage).  Wherever torch.nn.functional.linear and its
    autograd give a non-finite entry, rnnt_engine_linear_x2_fwd / _bwd do too (they may poison more: the whole result the operand feeds);
    results the operand does not feed stay exactly what they are without it."""
    import rnnt_amd
    torch.manual_seed(3)
    M, K, N = 200, 256, 128
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    G = torch.randn(M, N, device="cuda")

    def run(x_, W_, G_, fn):
        xr, Wr, br = (t.clone().requires_grad_(True) for t in (x_, W_, b))
        y = fn(xr, Wr, br)
        y.backward(G_)
        return y.detach(), xr.grad, Wr.grad, br.grad

    clean = run(x, W, G, lambda a_, w_, b_: rnnt_amd.linear(a_, w_, b_, backend="x2"))
    xs, Ws, Gs = x.clone(), W.clone(), G.clone()
    {"x": xs, "W": Ws, "dy": Gs}[where][7, 33] = what
    got = run(xs, Ws, Gs, lambda a_, w_, b_: rnnt_amd.linear(a_, w_, b_, backend="x2"))
    ref = run(xs, Ws, Gs, torch.nn.functional.linear)
    for name, g, r, c in zip(("y", "dx", "dW", "db"), got, ref, clean):
        bad_ref = ~torch.isfinite(r)
        assert not torch.isfinite(g[bad_ref]).any(), (name, where, what)
        if not bad_ref.any():  # the operand does not reach this result: untouched
            assert torch.equal(g, c), (name, where, what)
    fed = {"x": ("y", "dW"), "W": ("y", "dx"), "dy": ("dx", "dW", "db")}[where]
    for name, r in zip(("y", "dx", "dW", "db"), ref):
        assert (name in fed) == bool((~torch.isfinite(r)).any()), (name, where)  # (the test's own premise about what feeds what)


def test_joint_projections_default_to_the_engine_from_a_work_threshold():
    """JointNetwork.projection_backend = "auto": audio_ln (B*T rows) on the engine's f16x2 kernels from engine.LINEAR_X2_MIN_MKN of work
    (rows x in x out), text_ln (B*U1 rows, below it here) through the library; both within the fp32 bar of float64 torch, gradients included."""
    import rnnt_amd
    torch.manual_seed(5)
    j = rnnt_amd.JointNetwork(256, 128, 128, 128).cuda()
    assert j.projection_backend == "auto"
    rnnt_amd.engine.LINEAR_X2_MIN_MKN, keep = 2048 * 256 * 128, rnnt_amd.engine.LINEAR_X2_MIN_MKN  # (a small batch stands in for the real threshold)
    a = torch.randn(4, 600, 256, device="cuda", requires_grad=True)   # 2 400 rows: engine
    t = torch.randn(4, 9, 128, device="cuda", requires_grad=True)     # 36 rows: library
    calls = []
    orig = rnnt_amd.engine.linear_fwd
    rnnt_amd.engine.linear_fwd = lambda *args, **kw: (calls.append(kw.get("backend")), orig(*args, **kw))[1]
    try:
        af, tf = j._project(a, t)
    finally:
        rnnt_amd.engine.linear_fwd = orig
        rnnt_amd.engine.LINEAR_X2_MIN_MKN = keep
    assert calls == ["x2"]
    (af.sum() + tf.sum()).backward()
    a64 = a.detach().double().requires_grad_(True)
    ref = torch.nn.functional.linear(a64, j.audio_ln.weight.detach().double(), j.audio_ln.bias.detach().double())
    ref.sum().backward()
    assert_close_grad("audio_ln", af.detach().cpu().numpy(), ref.detach().cpu().numpy())
    assert_close_grad("d audio", a.grad.cpu().numpy(), a64.grad.cpu().numpy())


def test_engine_linear_on_permuted_encoder_view():
    """audio_ln applied to the (N,C,L)->(N,L,C) permuted view of reference rnnt/model.py:28."""
    import rnnt_amd
    torch.manual_seed(1)
    enc = torch.randn(3, 24, 50, device="cuda", requires_grad=True)  # (N,C,L)
    lin = torch.nn.Linear(24, 64).cuda()
    y = rnnt_amd.linear(enc.permute(0, 2, 1), lin.weight, lin.bias)
    G = torch.randn_like(y)
    (y * G).sum().backward()
    g_eng = enc.grad.clone(); gw = lin.weight.grad.clone()
    enc.grad = None; lin.zero_grad()
    (lin(enc.permute(0, 2, 1)) * G).sum().backward()
    assert torch.allclose(y, lin(enc.permute(0, 2, 1)), atol=1e-5)
    assert torch.allclose(enc.grad, g_eng, atol=1e-4) and torch.allclose(lin.weight.grad, gw, atol=1e-4)
