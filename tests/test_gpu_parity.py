"""GPU parity tests (-m gpu): the HIP engine, called through the C ABI (ctypes), against the
CPU oracle on the same seeded inputs, against the committed golden fixtures, and — at
BASELINE.json's full sizes — through size-independent properties."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import (BF16_GRAD_RTOL, BF16_GRAD_RTOL_EXACT, BF16_LOSS_RTOL,
                           BF16_LOSS_RTOL_EXACT, GRAD_RTOL, LOSS_RTOL, assert_close_digest, assert_close_grad, assert_close_loss,
                           lgamma_paths_cost, make_inputs, oracle_fused, oracle_fused_bf16, published_kat_cases)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import rnnt_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    rnnt_amd.engine.lib()  # fail loudly if the HIP extension is missing
    return rnnt_amd


def _dev(d):
    return {k: torch.from_numpy(v).cuda() for k, v in d.items()}


# Every route that claims the fp32 bar (LOSS_RTOL / GRAD_RTOL against the plain fp64 oracle).  "f16x2" is what
# RNNTModel.forward ships (rnnt_amd/joint.py: compute_dtype; round 4: three fp16 products of scaled, 2-way split operands),
# "bf16x3" its predecessor (six bf16 products of 3-way split operands), "fp32" the exact-fp32 MFMA route.  `_run_fused` has NO
# default dtype and every fp32-bar test takes the `route` fixture, so a new test cannot silently skip the shipped
# default (round-3 verdict item 8).
FP32_BAR_ROUTES = ("fp32", "bf16x3", "f16x2")


@pytest.fixture(params=FP32_BAR_ROUTES)
def route(request):
    return request.param


def _run_fused(amd, d, dtype, enc_override=None):
    g = _dev(d)
    enc = enc_override if enc_override is not None else g["enc"]
    enc = enc.detach().requires_grad_(True)
    pred = g["pred"].requires_grad_(True)
    W = g["W"].requires_grad_(True)
    bias = g["bias"].requires_grad_(True)
    loss, costs = amd.joint_rnnt_loss(enc, pred, W, bias, g["targets"], g["logit_lens"],
                                      g["target_lens"], blank=-1, reduction="mean",
                                      return_costs=True, dtype=dtype)
    loss.backward()
    torch.cuda.synchronize()
    return dict(loss=loss.item(), costs=costs.cpu().numpy(), grad_enc=enc.grad.cpu().numpy(),
                grad_pred=pred.grad.cpu().numpy(), grad_W=W.grad.cpu().numpy(),
                grad_bias=bias.grad.cpu().numpy())


def _compare(r, ref):
    assert_close_loss("loss", r["loss"], ref["loss"])
    assert_close_loss("costs", r["costs"], ref["costs"])
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k])


# (B, T, U, H, V): edges — single cell, U=0, T=1, ragged, H%8!=0, V%128!=0, V%4!=0, H%4!=0,
# several u-blocks / t-tiles / forward passes / dW tiles
FUSED_SHAPES = [
    (1, 1, 0, 8, 4), (1, 1, 3, 8, 8), (2, 5, 0, 16, 8), (2, 5, 2, 16, 8), (3, 17, 8, 64, 32),
    (2, 9, 4, 20, 12), (3, 23, 19, 36, 132), (2, 12, 5, 128, 1024), (2, 40, 33, 72, 520),
    (2, 7, 3, 10, 7), (4, 30, 12, 520, 260), (1, 64, 40, 32, 1300),
    # H > 512 (the reference's joint is 1024 wide): k_dhidden_gen covers columns 0-511 and every further
    # whole group of 512 (reading G), the persistent k_dhidden the H % 512 rest (4-row t tiles: two
    # dPred slab heights in one reduction)
    (3, 21, 18, 1024, 64), (2, 10, 35, 768, 160), (2, 13, 6, 640, 1024), (2, 9, 4, 516, 96),
    (2, 19, 7, 1536, 32), (2, 11, 20, 1152, 96), (3, 33, 5, 1028, 64),
    # H % 256 == 128 with whole h blocks beside the odd one and V % 512 == 0: the f16x2 route's dW runs whole-block AND tall tiles
    # (k_dw_x2m, round 6) — three h blocks + the odd one, one v pair / two; ragged lengths split the live rows over 25 / 28 splits
    (3, 40, 17, 896, 512), (2, 300, 20, 640, 1024),
    # BASELINE config 2's / config 4's lattice lengths through the whole fused pipeline at a joint small
    # enough for the oracle: 1 200 / 2 100 dependent sweep steps (chained waves / the barrier kernel of
    # lattices wider than the mailbox), alpha and beta around 1e3-1e4
    (2, 1000, 200, 32, 64), (1, 1500, 600, 16, 32),
]


@pytest.mark.parametrize("shape", FUSED_SHAPES)
def test_fused_joint_loss_vs_oracle(amd, shape, route):
    """(bf16x3: the host side zero-pads H and V to multiples of 128.)"""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=sum(shape))
    _compare(_run_fused(amd, d, route), oracle_fused(d))


@pytest.mark.parametrize("dtype,H", [("fp32", 128), ("bf16x3", 128), ("f16x2", 128), ("bf16", 128), ("fp32", 640), ("bf16x3", 640), ("f16x2", 640)])
def test_fused_very_ragged_batch_vs_oracle(amd, dtype, H):
    """Utterances of 1, 2 and a few time steps next to a full one, empty and full targets: the dW
    GEMM walks only the live rows (k_dw_table: ranges rounded out to 16/32-cell granules, merged
    where they touch), dead forward / dHidden tiles are skipped or zero-filled.  T*U1 = 37*11 is
    odd, so every utterance starts in the middle of a granule."""
    B, T, U, V = 6, 37, 10, 128  # H = 640: the two-kernel backward (k_make_g + k_dhidden)
    d = make_inputs(B, T, U, H, V, seed=77)
    d["logit_lens"] = np.array([37, 1, 2, 9, 36, 17], dtype=np.int32)
    d["target_lens"] = np.array([10, 0, 1, 10, 0, 5], dtype=np.int32)
    r = _run_fused(amd, d, dtype)
    if dtype in FP32_BAR_ROUTES:
        _compare(r, oracle_fused(d))
    else:
        ref = oracle_fused_bf16(d)
        assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL)
        for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
            assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL)
    # twice on the same workspace: rows the first call left behind must not leak into the second
    d2 = dict(d)
    d2["logit_lens"] = np.array([3, 37, 30, 1, 1, 37], dtype=np.int32)
    d2["target_lens"] = np.array([2, 10, 0, 0, 10, 3], dtype=np.int32)
    r2 = _run_fused(amd, d2, dtype)
    if dtype in FP32_BAR_ROUTES:
        _compare(r2, oracle_fused(d2))
    else:
        ref2 = oracle_fused_bf16(d2)
        for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
            assert_close_grad(k, r2[k], ref2[k], rtol=BF16_GRAD_RTOL)


def _random_case(rng, bf16):
    B = int(rng.integers(1, 6)); T = int(rng.integers(1, 70)); U = int(rng.integers(0, 40))
    if bf16:
        H = int(rng.choice([128, 256, 384, 512])); V = int(rng.choice([128, 256, 384]))
    else:
        H = 4 * int(rng.integers(1, 161)); V = 4 * int(rng.integers(1, 80))
    d = make_inputs(B, T, U, H, V, seed=int(rng.integers(1 << 30)))
    ll = rng.integers(1, T + 1, B); tl = rng.integers(0, U + 1, B)
    ll[rng.integers(B)] = T; tl[rng.integers(B)] = U
    d["logit_lens"] = ll.astype(np.int32); d["target_lens"] = tl.astype(np.int32)
    return d


def test_fused_random_shapes_and_lengths_vs_oracle(amd, route):
    """Seeded sweep (tools/fuzz_parity.py runs the long version): random shapes with arbitrary
    lengths per utterance.  First the case that sweep found: T*U1 = 205 cells per utterance, short
    utterances — the dW granule of the NEXT utterance's first cell reaches back into dead time
    steps of this one that no dHidden tile had zero-filled."""
    d = make_inputs(5, 41, 4, 480, 192, seed=5)
    d["logit_lens"] = np.array([20, 8, 41, 19, 28], dtype=np.int32)
    d["target_lens"] = np.array([1, 4, 0, 4, 3], dtype=np.int32)
    _compare(_run_fused(amd, d, route), oracle_fused(d))
    rng = np.random.default_rng(31337)
    for it in range(24):
        bf = it % 4 == 3  # (H, V multiples of 128: the bf16 route's arithmetic when the fp32 route is under test, bf16x3 without padding otherwise)
        d = _random_case(rng, bf)
        bf = bf and route == "fp32"
        r = _run_fused(amd, d, "bf16" if bf else route)
        if bf:
            ref = oracle_fused_bf16(d)
            assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL)
            for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
                assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL)
        else:
            _compare(r, oracle_fused(d))


@pytest.mark.parametrize("pattern", [0x7FA00000, 0xFFFFFFFF, 0x7F800000])
def test_poisoned_workspace_does_not_leak(amd, pattern):
    """The workspace is caller-owned scratch: whatever it holds (signalling NaNs, quiet NaNs, +inf)
    must not reach a result.  Slots no kernel writes (lattice arrays outside the lattice, logits /
    G / slabs of skipped tiles) are only ever dropped by a select or a range check — a bare
    v_min_f32 let a signalling NaN through once (tools/fuzz_lattice.py)."""
    from oracle import cpu_oracle
    dev = torch.device("cuda", 0)

    def poison():
        ws = amd.engine.workspace(dev, 1)
        ws.view(torch.int32)[: ws.numel() // 4].fill_(pattern - (1 << 32) if pattern >= (1 << 31) else pattern)

    rng = np.random.default_rng(3)
    # (the wide lattices: dead stretches longer than a dW granule / a dHidden tile, so only the dead
    # cells NEXT to lattice cells are zeroed by k_zero_dead_hidden and the rest keeps the poison)
    for dtype, (B, T, U, H, V) in (("fp32", (3, 41, 13, 136, 68)), ("fp32", (2, 30, 9, 640, 64)),
                                   ("bf16", (3, 41, 13, 128, 128)), ("fp32", (3, 61, 70, 64, 96)),
                                   ("fp32", (3, 50, 100, 32, 36)), ("bf16x3", (3, 41, 13, 128, 128)),
                                   ("bf16x3", (2, 30, 9, 640, 256)), ("bf16x3", (3, 61, 70, 128, 128)),
                                   ("f16x2", (3, 41, 13, 128, 128)), ("f16x2", (2, 30, 9, 640, 256)), ("f16x2", (3, 61, 70, 128, 128))):
        d = make_inputs(B, T, U, H, V, seed=pattern & 0xffff)
        d["logit_lens"] = np.array(([T, 7, 23] if B == 3 else [11, T]), dtype=np.int32)
        d["target_lens"] = np.array(([4, U, 0] if B == 3 else [U, 2]), dtype=np.int32)
        if U >= 70:
            d["target_lens"] = np.array([U, 9, 33], dtype=np.int32)
        _run_fused(amd, d, dtype)  # sizes the workspace
        poison()
        r = _run_fused(amd, d, dtype)
        if dtype in FP32_BAR_ROUTES:
            _compare(r, oracle_fused(d))
        else:
            ref = oracle_fused_bf16(d)
            assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL)
            for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
                assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL)
    # standalone loss, wide lattice (five chained waves), utterances shorter than the batch
    B, T, U, V = 3, 60, 300, 4
    logits = (rng.standard_normal((B, T, U + 1, V)) * 2).astype(np.float32)
    targets = rng.integers(0, V - 1, (B, U)).astype(np.int32)
    ll = np.array([31, 60, 44], dtype=np.int32); tl = np.array([300, 280, 190], dtype=np.int32)
    args = (torch.from_numpy(targets).cuda(), torch.from_numpy(ll).cuda(), torch.from_numpy(tl).cuda())
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    amd.rnnt_loss(lt, *args, blank=-1, reduction="none")
    poison()
    costs = amd.rnnt_loss(lt, *args, blank=-1, reduction="none")
    costs.sum().backward()
    ref_c, ref_g = cpu_oracle.rnnt_loss(logits, targets, ll, tl)
    assert_close_loss("costs", costs.detach().cpu().numpy(), ref_c)
    assert_close_grad("grad_logits", lt.grad.cpu().numpy(), ref_g)


def test_fused_config1_plumbing_shape_vs_oracle(amd, route):
    """BASELINE.json configs[0]'s shape (B=2, T~200, U~50, H=1024, V=1024): H > 512 — a second dHidden launch
    for columns 512-1023 on every route."""
    d = make_inputs(2, 208, 50, 1024, 1024, seed=208)
    _compare(_run_fused(amd, d, route), oracle_fused(d))


def test_fused_noncontiguous_encoder_view(amd, route):
    """The reference hands the joint a permuted (B,C,T)->(B,T,C) view (rnnt/model.py:27-28)."""
    d = make_inputs(2, 21, 6, 48, 64, seed=5)
    enc_bct = torch.from_numpy(np.ascontiguousarray(d["enc"].transpose(0, 2, 1))).cuda()
    view = enc_bct.permute(0, 2, 1)
    assert not view.is_contiguous()
    _compare(_run_fused(amd, d, route, enc_override=view), oracle_fused(d))


@pytest.mark.parametrize("shape", [(1, 1, 0, 4), (2, 6, 3, 8), (3, 31, 17, 40), (2, 20, 9, 1024),
                                   (2, 8, 4, 7), (1, 130, 70, 16), (2, 70, 200, 8), (2, 1400, 300, 8)])
def test_loss_only_vs_oracle(amd, shape):
    """rnnt_loss on given logits == the call at reference rnnt/model.py:35-41.  The last two shapes
    put the lattice sweep on four chained waves and (mailboxes > 64 KB) on the barrier kernel."""
    from oracle import cpu_oracle
    B, T, U, V = shape
    rng = np.random.default_rng(sum(shape))
    logits = (rng.standard_normal((B, T, U + 1, V)) * 2).astype(np.float32)
    d = make_inputs(B, T, U, 4, V, seed=sum(shape))
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    g = _dev(d)
    costs = amd.rnnt_loss(lt, g["targets"], g["logit_lens"], g["target_lens"], blank=-1,
                          reduction="none")
    w = torch.arange(1, B + 1, dtype=torch.float32, device="cuda")
    (costs * w).sum().backward()
    ref_c, ref_g = cpu_oracle.rnnt_loss(logits, d["targets"], d["logit_lens"], d["target_lens"])
    assert_close_loss("costs", costs.detach().cpu().numpy(), ref_c)
    assert_close_grad("grad_logits", lt.grad.cpu().numpy(),
                      ref_g * np.arange(1, B + 1).reshape(B, 1, 1, 1))
    mean = amd.rnnt_loss(lt.detach(), g["targets"], g["logit_lens"], g["target_lens"])
    assert_close_loss("mean", mean.item(), ref_c.mean())


@pytest.mark.parametrize("case", published_kat_cases(), ids=lambda c: c["name"])
def test_loss_matches_published_known_answers(amd, case):
    """rnnt_amd.rnnt_loss — the drop-in for the torchaudio call at reference rnnt/model.py:35-41 —
    against the published warp-transducer / torchaudio known-answer vectors (third-party published
    unit-test data, tests/golden/published_transducer_kat.json): costs to 1e-6, gradients to 1e-6."""
    lt = torch.from_numpy(case["logits"]).cuda().requires_grad_(True)
    tg = torch.from_numpy(case["targets"]).cuda()
    ll = torch.from_numpy(case["logit_lens"]).cuda()
    tl = torch.from_numpy(case["target_lens"]).cuda()
    costs = amd.rnnt_loss(lt, tg, ll, tl, blank=case["blank"], clamp=-1, reduction="none")
    costs.sum().backward()
    np.testing.assert_allclose(costs.detach().cpu().numpy(), case["costs"], rtol=1e-6)
    g = lt.grad.cpu().numpy().astype(np.float64)
    if case["grads"] is not None:
        assert np.abs(g - case["grads"]).max() <= max(case["grad_atol"], 1e-6)
    assert np.abs(g.sum(-1)).max() < 1e-6
    mean = amd.rnnt_loss(lt.detach(), tg, ll, tl, blank=case["blank"], reduction="mean")
    np.testing.assert_allclose(mean.item(), case["costs"].mean(), rtol=1e-6)


def test_loss_clamp(amd):
    from oracle import cpu_oracle
    rng = np.random.default_rng(3)
    logits = (rng.standard_normal((2, 7, 4, 12)) * 3).astype(np.float32)
    d = make_inputs(2, 7, 3, 4, 12, seed=3)
    lt = torch.from_numpy(logits).cuda().requires_grad_(True)
    g = _dev(d)
    amd.rnnt_loss(lt, g["targets"], g["logit_lens"], g["target_lens"], clamp=0.05,
                  reduction="sum").backward()
    _, ref_g = cpu_oracle.rnnt_loss(logits, d["targets"], d["logit_lens"], d["target_lens"], clamp=0.05)
    assert_close_grad("grad_logits", lt.grad.cpu().numpy(), ref_g)


def _sd(z):
    return {k[4:].replace("__", "."): torch.from_numpy(z[k]) for k in z.files if k.startswith("sd__")}


@pytest.mark.parametrize("name,backend", [("joint_tiny", "library"), ("joint_mid", "library"), ("joint_proj", "library"),
                                          ("joint_proj", "engine"), ("joint_v1024", "library")])
def test_joint_forward_golden(amd, golden_dir, name, backend):
    """JointNetwork.forward on the engine vs logits/grads produced by the reference's own
    rnnt.joint.JointNetwork (tests/golden/make_golden.py); the input projections through the library GEMM (the
    default) and through the engine's own linear kernels."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = amd.JointNetwork(*[int(x) for x in z["ctor"]]).cuda()
    m.projection_backend = backend
    m.load_state_dict(_sd(z))
    a = torch.from_numpy(z["audio"]).cuda().requires_grad_(True)
    t = torch.from_numpy(z["text"]).cuda().requires_grad_(True)
    logits = m(a, t)
    assert_close_grad("logits", logits.detach().cpu().numpy(), z["logits_f64"], rtol=1e-5, atol=1e-5)
    (logits * torch.from_numpy(z["G"]).cuda()).sum().backward()
    assert_close_grad("grad_audio", a.grad.cpu().numpy(), z["grad_audio"])
    assert_close_grad("grad_text", t.grad.cpu().numpy(), z["grad_text"])
    for k, p in m.named_parameters():
        assert_close_grad(k, p.grad.cpu().numpy(), z["grad__" + k.replace(".", "__")])


@pytest.mark.parametrize("name,backend", [("e2e_tiny", "library"), ("e2e_mid", "library"), ("e2e_proj", "library"),
                                          ("e2e_proj", "engine"), ("e2e_v1024", "library")])
def test_fused_golden(amd, golden_dir, name, backend, route):
    """fused_loss (joint + loss + backward in one engine call) vs the golden end-to-end
    fixtures (reference JointNetwork in fp64 + independent autograd loss), on both fp32-bar routes."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = amd.JointNetwork(*[int(x) for x in z["ctor"]]).cuda()
    m.projection_backend = backend
    m.load_state_dict(_sd(z))
    a = torch.from_numpy(z["audio"]).cuda().requires_grad_(True)
    t = torch.from_numpy(z["text"]).cuda().requires_grad_(True)
    loss = m.fused_loss(a, t, torch.from_numpy(z["targets"]).cuda(),
                        torch.from_numpy(z["logit_lens"]).cuda(),
                        torch.from_numpy(z["target_lens"]).cuda(), dtype=route)
    loss.backward()
    assert_close_loss("loss", loss.item(), float(z["loss"]))
    assert_close_grad("grad_audio", a.grad.cpu().numpy(), z["grad_audio"])
    assert_close_grad("grad_text", t.grad.cpu().numpy(), z["grad_text"])
    for k, p in m.named_parameters():
        assert_close_grad(k, p.grad.cpu().numpy(), z["grad__" + k.replace(".", "__")])


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3", "f16x2", "bf16"])
def test_fused_golden_inputs_from_reference_modules(amd, golden_dir, dtype):
    """The joint's inputs come from the reference's own AudioEncoder / ConvPredictor (fixture
    e2e_refmodules: call sequence of rnnt/model.py:20-29); the encoder output is handed over as
    the permuted, non-contiguous (N,C,L)->(N,L,C) view exactly as model.py:28 does."""
    z = np.load(os.path.join(golden_dir, "e2e_refmodules.npz"))
    m = amd.JointNetwork(*[int(x) for x in z["ctor"]]).cuda()
    m.load_state_dict(_sd(z))
    enc_ncl = torch.from_numpy(z["enc_ncl"]).cuda().requires_grad_(True)
    t = torch.from_numpy(z["text"]).cuda().requires_grad_(True)
    a = enc_ncl.permute(0, 2, 1)
    assert not a.is_contiguous()
    loss = m.fused_loss(a, t, torch.from_numpy(z["targets"]).cuda(),
                        torch.from_numpy(z["logit_lens"]).cuda(),
                        torch.from_numpy(z["target_lens"]).cuda(), dtype=dtype)
    loss.backward()
    lt, gt = (LOSS_RTOL, GRAD_RTOL) if dtype in FP32_BAR_ROUTES else (BF16_LOSS_RTOL_EXACT, BF16_GRAD_RTOL_EXACT)
    assert_close_loss("loss", loss.item(), float(z["loss"]), rtol=lt)
    assert_close_grad("grad_audio", enc_ncl.grad.permute(0, 2, 1).cpu().numpy(), z["grad_audio"], rtol=gt)
    assert_close_grad("grad_text", t.grad.cpu().numpy(), z["grad_text"], rtol=gt)
    for k, p in m.named_parameters():
        assert_close_grad(k, p.grad.cpu().numpy(), z["grad__" + k.replace(".", "__")], rtol=gt)


def test_model_forward_matches_unfused(amd):
    """RNNTModel.forward (reference rnnt/model.py:17-43 call sequence) with stand-in encoder /
    predictor modules: fused loss == joint logits -> rnnt_loss, and grads reach every module."""
    torch.manual_seed(0)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv1d(10, 32, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    class Pred(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.e = torch.nn.Embedding(16, 32)

        def forward(self, ids):
            return self.e(ids)

    model = amd.RNNTModel(Pred(), Enc(), amd.JointNetwork(-1, -1, 32, 16)).cuda()
    mel = torch.randn(3, 10, 40, device="cuda")
    mel_lens = torch.tensor([40, 33, 21], device="cuda")
    ids = torch.randint(0, 15, (3, 6), device="cuda")
    id_lens = torch.tensor([6, 4, 2], device="cuda")
    loss = model(mel, mel_lens, ids, id_lens, 15)
    loss.backward()
    g_fused = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    # unfused: same modules, logits materialised, separate loss op
    start = torch.full((3, 1), 15, dtype=ids.dtype, device="cuda")
    dec = model.predictor(torch.cat([start, ids], 1))
    aud = model.encoder(mel).permute(0, 2, 1)
    logits = model.joint(aud, dec)
    loss2 = amd.rnnt_loss(logits, ids.int(), model.encoder.calc_output_lens(mel_lens).int(),
                          id_lens.int(), blank=-1)
    loss2.backward()
    assert_close_loss("loss", loss.item(), loss2.item(), rtol=1e-5)
    for k, p in model.named_parameters():
        assert_close_grad(k, g_fused[k].cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-4)


def test_midsize_long_lattice_loss_vs_oracle(amd, route):
    """Long lattice at the BASELINE H and V (B=2,T=500,U=100,H=512,V=1024, ragged): per-utterance
    costs of the fused path vs the fp64 oracle (loss only: the oracle's forward is OpenMP
    parallel, its backward is not).  Catches errors that only show with many k-chunks, many
    tiles and multi-GiB buffers (a register-aliasing bug in an inline-asm load did)."""
    from oracle import cpu_oracle
    d = make_inputs(2, 500, 100, 512, 1024, seed=5)
    logits = cpu_oracle.joint_fwd(d["enc"], d["pred"], d["W"], d["bias"], dtype=np.float64)
    ref, _ = cpu_oracle.rnnt_loss(logits, d["targets"], d["logit_lens"], d["target_lens"],
                                  want_grad=False)
    g = _dev(d)
    outs = amd.engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"],
                                         g["logit_lens"], g["target_lens"], 1023, 0.5, dtype=route)
    assert_close_loss("costs", outs[0].cpu().numpy(), ref, rtol=1e-6)
    for o in outs[1:]:
        assert torch.isfinite(o).all()


def test_fused_path_is_bitwise_reproducible(amd):
    """Work is distributed with atomics (persistent waves) and split-K slabs, but every output
    element is summed in a fixed order: two runs must agree bit for bit."""
    d = make_inputs(3, 61, 23, 136, 260, seed=9)
    g = _dev(d)
    run = lambda: [o.clone() for o in amd.engine.joint_loss_fwd_bwd(
        g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
        259, 1.0 / 3, dtype="fp32")]  # (H, V not multiples of 128: the split routes' twins are tests/test_x2_gpu.py / test_x3_gpu.py)
    a, b = run(), run()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shape", [(3, 61, 23, 128, 260), (2, 300, 40, 512, 1024)])
def test_forward_kernel_variants_agree_bitwise(amd, shape):
    """The pipeline has per-call kernel variants (rnnt_engine_run_stages, RNNT_VARIANT_*): forward
    with persistent workgroups and register-streamed W fragments (default) / hidden from the
    separate k_make_hidden pass / the LDS-DMA ring main loop / one workgroup per tile; G from the
    separate k_make_g pass + the persistent dHidden kernel.  They multiply the same numbers in the
    same order (the forward ones exactly; the dHidden variant sums its k chunks in a rotated order,
    so it is held to the parity tolerance instead)."""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=21)
    g = _dev(d)
    E = amd.engine
    run = lambda variant: [o.clone() for o in E.joint_loss_fwd_bwd(
        g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
        V - 1, 1.0 / B, variant=variant, dtype="fp32")]  # (the exact-fp32 route's own kernel variants)
    ref = run(0)
    for variant in (E.VARIANT_SEPARATE_HIDDEN, E.VARIANT_FWD_LDS_RING, E.VARIANT_FWD_ONE_WG_PER_TILE,
                    E.VARIANT_SEPARATE_HIDDEN | E.VARIANT_FWD_ONE_WG_PER_TILE):
        for x, y in zip(run(variant), ref):
            assert torch.equal(x, y), variant
    sep = run(E.VARIANT_SEPARATE_G)
    assert torch.equal(sep[0], ref[0])  # costs: same forward
    for x, y in zip(sep[1:], ref[1:]):
        assert_close_grad("separate-G variant", x.cpu().numpy(), y.cpu().numpy())
    # the shipped library has no process-wide switches: set_flags is a no-op
    assert E.lib().rnnt_engine_set_flags(64 | 128 | 256) == 0
    for x, y in zip(run(0), ref):
        assert torch.equal(x, y)


@pytest.mark.parametrize("H,V", [(64, 96), (640, 64), (128, 260)])
def test_short_targets_skip_dead_rows_vs_oracle(amd, H, V, route):
    """Utterances whose targets are much shorter than U: most cells of every live time step lie past
    U_b.  The forward skips wave tiles without a lattice cell, dW walks the device-built list of live
    16-cell granules (the list walk, not the contiguous-range one), dHidden skips u blocks past U_b."""
    d = make_inputs(5, 37, 100, H, V, seed=H + V)
    d["logit_lens"] = np.array([37, 20, 37, 1, 30], dtype=np.int32)
    d["target_lens"] = np.array([100, 7, 0, 55, 16], dtype=np.int32)
    _compare(_run_fused(amd, d, route), oracle_fused(d))
    # all targets full but ragged time steps: the contiguous-range walk
    d["target_lens"] = np.full(5, 100, dtype=np.int32)
    _compare(_run_fused(amd, d, route), oracle_fused(d))


def test_forward_only_costs_match_and_skip_backward(amd, route):
    """torch.no_grad() / no input requires grad: RNNTModel.forward's loss comes from the forward
    kernels alone (rnnt_engine_joint_loss_fwd) and equals the training-mode loss bit for bit."""
    d = make_inputs(3, 37, 11, 256, 256, seed=5)
    g = _dev(d)
    r = _run_fused(amd, d, route)  # (bit-for-bit comparisons below: the same route's forward kernels)
    with torch.no_grad():
        loss, costs = amd.joint_rnnt_loss(g["enc"].requires_grad_(True), g["pred"], g["W"], g["bias"],
                                          g["targets"], g["logit_lens"], g["target_lens"], return_costs=True, dtype=route)
    assert not loss.requires_grad
    assert loss.item() == r["loss"] and np.array_equal(costs.cpu().numpy(), r["costs"])
    loss2 = amd.joint_rnnt_loss(g["enc"].detach(), g["pred"], g["W"], g["bias"], g["targets"],
                                g["logit_lens"], g["target_lens"], dtype=route)
    assert loss2.item() == r["loss"] and not loss2.requires_grad
    c3 = amd.engine.joint_loss_fwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"],
                                   g["target_lens"], 255, dtype="bf16")
    assert np.allclose(c3.cpu().numpy(), r["costs"], rtol=2e-2)


def test_unfused_backward_runs_on_the_engine(amd):
    """joint_logits(...).backward(): rnnt_engine_joint_bwd (autograd of rnnt/joint.py:32-39) against
    plain torch ops in float64, with a random upstream gradient, a permuted encoder view, H > 512
    and H / V that need host-side padding."""
    for (B, T, U1, H, V) in ((2, 19, 7, 64, 40), (1, 33, 18, 640, 96), (2, 9, 5, 30, 10)):
        torch.manual_seed(B + T + H)
        enc_ct = torch.randn(B, H, T, device="cuda")
        enc = enc_ct.permute(0, 2, 1).requires_grad_(True)
        pred = torch.randn(B, U1, H, device="cuda", requires_grad=True)
        W = (torch.randn(V, H, device="cuda") / H ** 0.5).requires_grad_(True)
        bias = torch.randn(V, device="cuda", requires_grad=True)
        up = torch.randn(B, T, U1, V, device="cuda")
        out = amd.joint_logits(enc, pred, W, bias)
        (out * up).sum().backward()
        e64, p64, W64, b64 = (x.detach().double().requires_grad_(True) for x in (enc, pred, W, bias))
        ref = torch.tanh(e64.unsqueeze(2) + p64.unsqueeze(1)) @ W64.T + b64
        (ref * up.double()).sum().backward()
        assert_close_grad("logits", out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5)
        for name, a_, b_ in (("enc", enc, e64), ("pred", pred, p64), ("W", W, W64), ("bias", bias, b64)):
            assert_close_grad(name, a_.grad.cpu().numpy(), b_.grad.cpu().numpy())


def test_engine_rejects_wrong_dtypes(amd):
    """The C side reinterprets pointers: a bf16 / fp64 / int64 tensor must raise, not be read past
    its allocation."""
    d = make_inputs(2, 6, 3, 16, 8, seed=1)
    g = _dev(d)
    args = [g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"], 7, 0.5]
    for i, bad in ((2, g["W"].bfloat16()), (3, g["bias"].double()), (4, g["targets"].long()),
                   (5, g["logit_lens"].long()), (0, g["enc"].half())):
        a = list(args)
        a[i] = bad
        with pytest.raises(RuntimeError, match="must be torch"):
            amd.engine.joint_loss_fwd_bwd(*a, dtype=amd.engine.DEFAULT_DTYPE)
    with pytest.raises(TypeError, match="dtype"):  # no default route at the engine level
        amd.engine.joint_loss_fwd_bwd(*args)
    with pytest.raises(RuntimeError, match="must be torch"):
        amd.engine.joint_fwd(g["enc"], g["pred"], g["W"].bfloat16(), g["bias"])
    with pytest.raises(RuntimeError, match="float32"):
        amd.joint_rnnt_loss(g["enc"], g["pred"], g["W"].bfloat16(), g["bias"], g["targets"],
                            g["logit_lens"], g["target_lens"])


def test_bad_lengths_are_clamped_on_the_device(amd, route):
    """check_lengths=False (no host sync) with lengths outside the lattice: the kernels clamp what
    they read (include/rnnt_engine.h), so the result is the loss of the clamped lattice, finite and
    equal to the oracle on the clamped lengths — never an out-of-bounds access."""
    d = make_inputs(3, 20, 70, 64, 32, seed=9)  # U1 = 71: two chained waves in the sweep
    d["logit_lens"] = np.array([20, 99, 0], dtype=np.int32)
    d["target_lens"] = np.array([70, -3, 500], dtype=np.int32)
    g = _dev(d)
    loss, costs = amd.joint_rnnt_loss(g["enc"], g["pred"], g["W"], g["bias"], g["targets"],
                                      g["logit_lens"], g["target_lens"], check_lengths=False,
                                      return_costs=True, dtype=route)
    dc = dict(d)
    dc["logit_lens"] = np.array([20, 20, 1], dtype=np.int32)
    dc["target_lens"] = np.array([70, 0, 70], dtype=np.int32)
    ref = oracle_fused(dc)
    assert_close_loss("costs", costs.cpu().numpy(), ref["costs"])
    # ... and the GRADIENTS are those of the clamped lattice (every backward kernel reads the same clamped
    # lengths: dead tiles skipped, dead rows zero), on both arithmetic routes that share the lattice code
    leaves = [g[k].clone().requires_grad_(True) for k in ("enc", "pred", "W", "bias")]
    loss, costs = amd.joint_rnnt_loss(*leaves, g["targets"], g["logit_lens"], g["target_lens"], check_lengths=False,
                                      return_costs=True, dtype=route)  # (pads H, V for the split routes)
    loss.backward()
    torch.cuda.synchronize()
    outs = [costs] + [x.grad for x in leaves]
    assert_close_loss("costs", outs[0].cpu().numpy(), ref["costs"])
    for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias")):
        assert_close_grad(k, o.cpu().numpy(), ref[k])
    # utterance 1 was clamped to U_b = 0, utterance 2 to T_b = 1: nothing outside those lattices has a gradient
    assert float(outs[2][1, 1:].abs().max()) == 0.0 and float(outs[1][2, 1:].abs().max()) == 0.0
    db = make_inputs(3, 20, 70, 128, 128, seed=10)
    db["logit_lens"], db["target_lens"] = d["logit_lens"], d["target_lens"]
    gb = _dev(db)
    outs = amd.engine.joint_loss_fwd_bwd(gb["enc"], gb["pred"], gb["W"], gb["bias"], gb["targets"], gb["logit_lens"],
                                         gb["target_lens"], 127, 1.0 / 3, dtype="bf16")
    torch.cuda.synchronize()
    dbc = dict(db)
    dbc["logit_lens"], dbc["target_lens"] = dc["logit_lens"], dc["target_lens"]
    refb = oracle_fused_bf16(dbc)
    assert_close_loss("costs", outs[0].cpu().numpy(), refb["costs"], rtol=BF16_LOSS_RTOL)
    for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias")):
        assert_close_grad(k, o.cpu().numpy(), refb[k], rtol=BF16_GRAD_RTOL)


def test_inputs_ending_at_unmapped_pages():
    """Input OVER-READS (the class of the round-2 bf16 fault: a bias read 512 B past the vector at V = 128,
    commit 1f6212f — invisible to every parity test because the neighbouring allocation was mapped): every
    input of the fused call placed so that it ends at an unmapped page (HIP virtual-memory API,
    tools/guard_alloc.hip) for 12 fused shapes on the three routes — V = 128 bf16 and H = 1024 among them — plus one
    call of every other entry point.  Runs tools/guard_sweep.py --slice in a CHILD process: a memory fault
    kills the child and fails this test, the rest of the run goes on.  Run once; never looped."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tools", "libguard.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-o", so,
                               os.path.join(root, "tools", "guard_alloc.hip")])
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "guard_sweep.py"), "--slice"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    tail = (out.stdout[-1500:] + "\n" + out.stderr[-1500:])
    assert out.returncode == 0, "guard sweep died (memory fault = an input over-read):\n" + tail
    assert "guard sweep clean: 15 cases" in out.stdout and "guard sweep of the other entry points clean" in out.stdout, tail


# ---------------------------------------------------------------- full-size properties
def _full(amd, B, T, U, H, V, seed, dtype):
    d = make_inputs(B, T, U, H, V, seed, ragged=False)
    d["W"] = np.zeros_like(d["W"])  # logits == bias in every cell: closed-form loss
    r = _run_fused(amd, d, dtype)
    bias = d["bias"].astype(np.float64)
    lp = bias - np.log(np.exp(bias).sum())
    for b in range(B):
        ref = lgamma_paths_cost(T, U, lp[V - 1], lp[d["targets"][b]].sum())
        assert abs(r["costs"][b] - ref) / ref < 1e-5, (b, r["costs"][b], ref)
    assert np.abs(r["grad_enc"]).max() == 0 and np.abs(r["grad_pred"]).max() == 0
    # every gradient row sums to zero => so does grad_bias; blank column carries -T/B... total
    assert abs(r["grad_bias"].sum()) < 1e-3 * np.abs(r["grad_bias"]).sum()
    return r


def test_fullsize_config2_closed_form(amd, route):
    """BASELINE config 2 (B=32,T=1000,U=200,H=512,V=1024): known-answer loss at full size."""
    amd.engine.release_workspaces()
    _full(amd, 32, 1000, 200, 512, 1024, seed=2, dtype=route)
    amd.engine.release_workspaces()


def test_fullsize_config4_long_utterance_closed_form(amd, route):
    """BASELINE config 4 (B=8,T=4000,U=600,H=640,V=1024): ~145 GB workspace (212 GB on the bf16x3 route: 73 % of
    HBM), 4600-step lattice sweep, H not a multiple of 256/512 tiles (bf16x3: a second dHidden launch for
    columns 512-639 that re-reads G's planes): known-answer loss at full size."""
    amd.engine.release_workspaces()
    _full(amd, 8, 4000, 600, 640, 1024, seed=4, dtype=route)
    amd.engine.release_workspaces()


def test_fullsize_config5_large_vocab_closed_form(amd, route):
    """BASELINE config 5 (B=16,T=800,U=150,H=512,V=16384): 32 forward passes over V, 127 GB of
    logits, W (32 MiB) larger than an XCD's L2: known-answer loss at full size."""
    amd.engine.release_workspaces()
    _full(amd, 16, 800, 150, 512, 16384, seed=5, dtype=route)
    amd.engine.release_workspaces()


def _fused_vs_unfused(amd, d, route):
    """Fused engine path (on `route`) vs the unfused GPU path (joint GEMM on the exact-fp32 kernels -> rnnt_loss
    kernels -> plain torch-op backward of the joint in FLOAT64, written out here): independent backward arithmetic on dense
    data.  Round 5: the reference products run in float64 (the fp32 library GEMM's own rounding over 200 k - 2.4 M rows was
    why this comparison used to allow 5e-4), so the gradients are held to the same 1e-4 as every oracle comparison."""
    amd.engine.release_workspaces()
    r = _run_fused(amd, d, route)
    amd.engine.release_workspaces()
    g = _dev(d)
    logits = amd.joint_logits(g["enc"], g["pred"], g["W"], g["bias"]).requires_grad_(True)
    loss = amd.rnnt_loss(logits, g["targets"], g["logit_lens"], g["target_lens"], blank=-1)
    loss.backward()
    G = logits.grad
    del logits
    f64 = torch.float64
    ge = torch.zeros_like(g["enc"], dtype=f64); gp = torch.zeros_like(g["pred"], dtype=f64); gW = torch.zeros_like(g["W"], dtype=f64)
    gb = G.sum((0, 1, 2), dtype=f64)
    W64 = g["W"].double()
    TC = 250  # time steps per chunk: [TC, U1, V] and [TC, U1, H] float64 temporaries
    for b in range(G.shape[0]):  # joint.py:32-39 backwards, one utterance and one block of time steps at a time (memory)
        p64 = g["pred"][b].double().unsqueeze(0)
        for t0 in range(0, G.shape[1], TC):
            Gc = G[b, t0:t0 + TC].double()
            hid = torch.tanh(g["enc"][b, t0:t0 + TC].double().unsqueeze(1) + p64)
            dh = torch.matmul(Gc, W64) * (1 - hid * hid)
            ge[b, t0:t0 + TC] = dh.sum(1); gp[b] += dh.sum(0)
            gW += torch.matmul(Gc.reshape(-1, Gc.shape[-1]).t(), hid.reshape(-1, hid.shape[-1]))
            del hid, dh, Gc
    assert_close_loss("loss", r["loss"], loss.item(), rtol=1e-5)
    assert_close_grad("grad_enc", r["grad_enc"], ge.cpu().numpy(), rtol=GRAD_RTOL)
    assert_close_grad("grad_pred", r["grad_pred"], gp.cpu().numpy(), rtol=GRAD_RTOL)
    assert_close_grad("grad_W", r["grad_W"], gW.cpu().numpy(), rtol=GRAD_RTOL)
    assert_close_grad("grad_bias", r["grad_bias"], gb.cpu().numpy(), rtol=GRAD_RTOL)
    assert np.abs(r["grad_enc"]).max() > 0 and np.abs(r["grad_pred"]).max() > 0
    del G, ge, gp, gW
    torch.cuda.empty_cache()
    amd.engine.release_workspaces()


def test_fullsize_config2_one_utterance_vs_oracle_fixture(amd, route, golden_dir):
    """ORACLE gradients at BASELINE config 2's full T, U, H, V (round-4 verdict item 1c): one utterance (T=1000, U=200, H=512, V=1024:
    201 000 cells, 1 200 sweep steps, 64 k-steps x 2 passes per forward tile, every dW split) against the fp64 CPU oracle's cost and
    four gradients, generated once in the build container (tests/golden/make_fullsize_fixture.py; the reference's call sequence
    rnnt/model.py:32-41 + train.py:134) — the inputs are regenerated from the seed and checked against the stored CRC32s."""
    import zlib
    z = np.load(os.path.join(golden_dir, "fullsize_cfg2_one_utterance.npz"))
    B, T, U, H, V = (int(x) for x in z["shape"])
    d = make_inputs(B, T, U, H, V, seed=int(z["seed"]), ragged=False)
    for name, crc in zip(z["crc_names"], z["crc_values"]):
        assert zlib.crc32(np.ascontiguousarray(d[str(name)]).tobytes()) == int(crc), "regenerated input differs from the fixture's: " + str(name)
    amd.engine.release_workspaces()
    r = _run_fused(amd, d, route)
    assert_close_loss("costs", r["costs"], z["costs"])
    assert_close_loss("loss", r["loss"], float(z["loss"]))
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], z[k])
        print(k, "max err / max |ref| = %.2e" % (np.abs(r[k] - z[k].astype(np.float64)).max() / np.abs(z[k]).max()))
    amd.engine.release_workspaces()


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_fullsize_config4_config5_one_utterance_vs_oracle_digest(amd, route, golden_dir, cfg):
    """ORACLE gradients at BASELINE config 4's (T=4000, U=600, H=640, V=1024) and config 5's (T=800, U=150, H=512, V=16384) full sizes,
    one utterance each.  The fp64 oracle's gradients there are 5-35 MB, too large to commit whole, so the fixture
    (tests/golden/make_fullsize_digest.py, generated once in the build container) holds a DIGEST of each: 64 random +-1 projections of
    the whole array (an error anywhere moves them), 4 096 sampled entries, the largest magnitude and the 2-norm — and the cost and the
    (small) bias gradient whole.  Inputs are regenerated from the seed and checked against the stored CRC32s."""
    import zlib
    z = np.load(os.path.join(golden_dir, "fullsize_%s_one_utterance_digest.npz" % cfg))
    B, T, U, H, V = (int(x) for x in z["shape"])
    seed = int(z["seed"])
    d = make_inputs(B, T, U, H, V, seed=seed, ragged=False)
    for name, crc in zip(z["crc_names"], z["crc_values"]):
        assert zlib.crc32(np.ascontiguousarray(d[str(name)]).tobytes()) == int(crc), "regenerated input differs from the fixture's: " + str(name)
    amd.engine.release_workspaces()
    r = _run_fused(amd, d, route)
    assert_close_loss("costs", r["costs"], z["costs"])
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        dig = {f: z[k + "." + f] for f in ("proj", "sample", "sample_idx", "amax", "norm")}
        es, ep = assert_close_digest(k, r[k], dig, seed)
        print(cfg, k, "sampled max err / max |ref| = %.2e, projections / (max |ref| sqrt n) = %.2e" % (es, ep))
    if "grad_bias" in z.files and z["grad_bias"].size:  # stored whole where it is small
        assert_close_grad("grad_bias", r["grad_bias"], z["grad_bias"])
    amd.engine.release_workspaces()


def test_fullsize_config2_fused_vs_unfused_subset(amd, route):
    """Full T,U,H,V of config 2 on 2 utterances, dense random data."""
    _fused_vs_unfused(amd, make_inputs(2, 1000, 200, 512, 1024, seed=22), route)


def test_fullsize_config4_fused_vs_unfused_subset(amd, route):
    """Full T,U,H,V of config 4 (T=4000,U=600,H=640,V=1024) on 1 utterance, dense random W: the
    H > 512 backward (k_dhidden_gen for columns 0-511 + k_dhidden for 512-639), the barrier lattice
    kernel (mailboxes > 64 KB) and the odd 128-column dW tile multiply non-zero data."""
    _fused_vs_unfused(amd, make_inputs(1, 4000, 600, 640, 1024, seed=44), route)


def test_fullsize_config5_fused_vs_unfused_subset(amd, route):
    """Full T,U,H,V of config 5 (T=800,U=150,H=512,V=16384) on 2 ragged utterances, fp32 route,
    dense random W: 32 forward column passes and 2048-chunk dHidden K loops on non-degenerate data."""
    _fused_vs_unfused(amd, make_inputs(2, 800, 150, 512, 16384, seed=55), route)


def test_fused_vs_unfused_reference_joint_width(amd, route):
    """The reference's real joint width (hidden_features: 1024, rnnt/config/*.yaml) at BASELINE
    config 1's plumbing shape scaled to a training batch: B=4,T=208,U=50,H=1024,V=1024, ragged."""
    _fused_vs_unfused(amd, make_inputs(4, 208, 50, 1024, 1024, seed=11), route)


# ---- bf16 route (BASELINE config 3).  (B, T, U, H, V): ragged, several u-blocks / t-tiles /
# forward passes / dW tiles and splits, dead tiles (t0 >= T_b), H < 512
BF16_SHAPES = [(1, 1, 0, 128, 128), (2, 9, 4, 128, 128), (3, 23, 19, 256, 384), (2, 40, 33, 512, 1024),
               (4, 30, 12, 384, 256), (1, 70, 40, 128, 2048),
               # H > 512: one more k_dhidden_bf16 launch per further 512 columns (G read back in place)
               (2, 13, 20, 1024, 256), (3, 21, 9, 640, 128),
               # config 3's lattice length (T = 1000, U = 200: 1 200 sweep steps on fp16-rounded logits)
               (1, 1000, 200, 128, 128)]


@pytest.mark.parametrize("shape", BF16_SHAPES)
def test_bf16_fused_vs_rounding_point_oracle(amd, shape):
    """bf16 GEMM operands, fp32 accumulate: against the float64 oracle that rounds tanh(enc+pred),
    W and the logits gradient to bf16 at the same points, and (looser) against the unrounded one."""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=sum(shape) + 1)
    r = _run_fused(amd, d, "bf16")
    ref = oracle_fused_bf16(d)
    assert_close_loss("loss", r["loss"], ref["loss"], rtol=BF16_LOSS_RTOL)
    assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL)
    exact = oracle_fused(d)
    assert_close_loss("loss vs unrounded", r["loss"], exact["loss"], rtol=BF16_LOSS_RTOL_EXACT)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k + " vs unrounded", r[k], exact[k], rtol=BF16_GRAD_RTOL_EXACT)


@pytest.mark.parametrize("shape,ll,tl", [((2, 70, 0, 256, 128), None, None),            # U1 = 1: every cell its own time step
                                         ((3, 50, 2, 512, 256), [50, 7, 33], [2, 0, 1]),  # U1 = 3: 10 time steps per wave
                                         ((2, 200, 30, 128, 128), [200, 20], [30, 5]),    # whole tiles past T_b (hidden only)
                                         ((1, 37, 6, 1024, 384), None, None),             # H = 1024: one workgroup per CU
                                         ((5, 3, 40, 256, 128), [3, 1, 2, 3, 1], [40, 0, 13, 40, 7])])  # 123 cells each
def test_bf16_forward_register_resident_edges(amd, shape, ll, tl):
    """k_joint_fwd_bf16_ra (H = 128/256/512/1024): a wave's 32 cells spanning many time steps (small U1), tiles that
    straddle utterances or lie past an utterance's length, the last tile past the lattice."""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=sum(shape) + 7)
    if ll is not None:
        d["logit_lens"] = np.array(ll, dtype=np.int32); d["target_lens"] = np.array(tl, dtype=np.int32)
    r = _run_fused(amd, d, "bf16")
    ref = oracle_fused_bf16(d)
    assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL)


def test_bf16_rejects_unsupported_dims(amd):
    d = make_inputs(2, 5, 2, 64, 128, seed=3)
    with pytest.raises(RuntimeError, match="RNNT_DTYPE_BF16"):
        _run_fused(amd, d, "bf16")


# ---- greedy-decode scan (reference rnnt/model.py:108-125, joint.py:44-55)
@pytest.mark.parametrize("T,H,V,n,t0", [(50, 64, 40, 32, 7), (9, 512, 1024, 9, 0), (200, 128, 260, 128, 72)])
def test_greedy_scan_vs_torch(amd, T, H, V, n, t0):
    torch.manual_seed(T + V)
    enc_ct = torch.randn(H, T, device="cuda")         # encoder layout (C,T): the scan gets the permuted view
    enc = enc_ct.permute(1, 0)
    pred = torch.randn(H, device="cuda")
    W = torch.randn(V, H, device="cuda") / H ** 0.5
    bias = torch.randn(V, device="cuda") * 0.1
    blank = V - 1
    bias[blank] += 1.5  # make blank frequent, as in a trained model
    out = amd.engine.greedy_scan(enc, pred, W, bias, t0, n, blank).cpu().numpy()
    logits = torch.tanh(enc[t0:t0 + n].double() + pred.double()) @ W.double().T + bias.double()
    ref = logits.argmax(dim=-1).cpu().numpy()
    # ties/near-ties between fp32 MFMA and fp64: compare where the top-2 margin is clear
    top2 = logits.topk(2, dim=-1).values
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-4).cpu().numpy()
    assert clear.mean() > 0.9
    assert (out[2:][clear] == ref[clear]).all()
    assert clear.all(), "seeded inputs are expected to have clear margins"
    nb = np.nonzero(ref != blank)[0]
    if len(nb):
        assert out[0] == t0 + nb[0] and out[1] == ref[nb[0]]
    else:
        assert out[0] == t0 + n and out[1] == blank


def test_greedy_decode_scan_matches_per_frame_loop(amd):
    """RNNTModel.greedy_decode with the device-side scan == the reference's per-frame loop
    (rnnt/model.py:95-125: emit until blank or 10 symbols per frame)."""
    torch.manual_seed(3)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv1d(10, 24, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    class Pred(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.e = torch.nn.Embedding(16, 20)

        def forward(self, ids):
            return self.e(ids)

    for fa, ft, hid in ((-1, -1, 24), (24, 20, 32)):  # without / with audio_ln + text_ln
        pred_mod = Pred() if ft > 0 else torch.nn.Embedding(16, 24)
        model = amd.RNNTModel(pred_mod, Enc(), amd.JointNetwork(fa, ft, hid, 16)).cuda()
        with torch.no_grad():
            model.joint.joint_ln.bias[15] += 1.0
        mel = torch.randn(1, 10, 120, device="cuda")
        lens = torch.tensor([120], device="cuda")
        a = model.greedy_decode(mel, lens, max_length=80, scan_frames=16)
        b = model.greedy_decode(mel, lens, max_length=80, scan_frames=0)
        assert a == b and 0 < len(a) <= 79
        assert model.greedy_decode(mel, lens, max_length=80, scan_frames=128) == b


def test_greedy_decode_device_loop_matches_per_frame_loop(amd):
    """RNNTModel.greedy_decode(device_loop=True): the WHOLE loop of rnnt/model.py:108-125 on the device
    (rnnt_engine_greedy_decode: scan, argmax, bookkeeping, the ConvPredictor step per token computed incrementally
    from rings of the last 7 positions; one host sync per utterance) decodes exactly what the reference's per-frame
    loop decodes with the predictor re-run on the whole history — with and without audio_ln / text_ln, for block
    sizes that split the utterance differently, when max_length cuts the loop, and when a frame hits the
    10-symbols-per-frame cap."""
    torch.manual_seed(5)

    class Enc(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.c = torch.nn.Conv1d(10, c, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    V = 32
    for fa, ft, hid, E, O, bias_blank in ((-1, -1, 64, 48, 64, 1.0), (40, 56, 72, 32, 56, 1.0), (-1, -1, 64, 48, 64, -2.0)):
        pred = amd.ConvPredictor(V, O, E, 0.3)
        model = amd.RNNTModel(pred, Enc(fa if fa > 0 else hid), amd.JointNetwork(fa, ft, hid, V)).cuda().eval()
        with torch.no_grad():
            model.joint.joint_ln.bias[V - 1] += bias_blank  # -2: blank rarely wins -> the 10-per-frame cap and max_length cut in
        assert model._device_loop_ok(torch.zeros(1, 1, device="cuda"))
        mel = torch.randn(1, 10, 150, device="cuda")
        lens = torch.tensor([150], device="cuda")
        for max_length in (60, 9):
            want = model.greedy_decode(mel, lens, max_length=max_length, scan_frames=0)
            assert 0 < len(want) <= max_length - 1
            assert model.greedy_decode(mel, lens, max_length=max_length, scan_frames=16, device_loop=False) == want
            for scan in (16, 64, 7, 128):
                got = model.greedy_decode(mel, lens, max_length=max_length, scan_frames=scan, device_loop=True, persistent=False)
                assert got == want, (fa, bias_blank, max_length, scan, got, want)
        assert model.greedy_decode(mel, lens, max_length=60) == model.greedy_decode(mel, lens, max_length=60, scan_frames=0)  # default: the device loop
    # training mode with dropout: the device loop is not the module's arithmetic any more -> refused, default falls back
    model.train()
    with pytest.raises(RuntimeError, match="device_loop"):
        model.greedy_decode(mel, lens, max_length=20, device_loop=True)


def test_greedy_decode_persistent_matches_per_frame_loop(amd):
    """RNNTModel.greedy_decode(persistent=True): the loop of rnnt/model.py:108-125 with the ConvPredictor of rnnt/predictor.py:189-229
    as ONE persistent launch per utterance (rnnt_engine_greedy_decode_persistent: conv1 as table rows, conv2's old taps ahead of the
    token, joint.text_ln folded into the predictor's linear layer, 16-frame scans on 16x16x4 MFMAs, hand-offs through tagged words)
    decodes exactly what the reference's per-frame loop decodes: with and without audio_ln / text_ln; vocabularies smaller than the
    grid (idle workgroups), not a multiple of 16, and larger than 128 blocks (several blocks per workgroup); when max_length cuts the
    loop; when a frame hits the 10-symbols-per-frame cap; at the reference's own widths (E=512, O=H=V=1024); sizes it does not take
    (H % 64 != 0) fall back to the kernel-per-layer loop."""
    torch.manual_seed(7)

    class Enc(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.c = torch.nn.Conv1d(10, c, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    cases = (  # fa, ft, hid, E, O, V, bias on blank, frames
        (-1, -1, 64, 48, 64, 32, 1.0, 150),
        (40, 56, 128, 32, 56, 32, 1.0, 150),
        (-1, -1, 64, 48, 64, 32, -2.0, 150),
        (40, 56, 64, 36, 56, 300, 0.6, 90),
        (-1, -1, 192, 64, 192, 4000, 0.4, 90),
        (24, 1024, 256, 512, 1024, 1024, 0.4, 120),
        (-1, -1, 1024, 512, 1024, 1024, 0.6, 200),
    )
    counts = []
    for fa, ft, hid, E, O, V, bias_blank, nfr in cases:
        pred = amd.ConvPredictor(V, O, E, 0.3)
        model = amd.RNNTModel(pred, Enc(fa if fa > 0 else hid), amd.JointNetwork(fa, ft, hid, V)).cuda().eval()
        with torch.no_grad():
            model.joint.joint_ln.bias[V - 1] += bias_blank
        assert amd.engine.greedy_decode_persistent_supported(nfr // 2, V, E, O, hid, V, ft > 0)
        mel = torch.randn(1, 10, nfr, device="cuda")
        lens = torch.tensor([nfr], device="cuda")
        for max_length in (60, 9):
            want = model.greedy_decode(mel, lens, max_length=max_length, scan_frames=0)
            assert len(want) <= max_length - 1
            counts.append(len(want))
            got = model.greedy_decode(mel, lens, max_length=max_length, persistent=True)
            assert got == want, (fa, hid, V, bias_blank, max_length, got, want)
        assert model.greedy_decode(mel, lens, max_length=60) == model.greedy_decode(mel, lens, max_length=60, scan_frames=0)  # the default
        # the model's tables (conv1 tap tables, conv2 pack, folded text_ln) are cached between utterances: a weight changed in place
        # (an optimizer step, load_state_dict) must rebuild them
        with torch.no_grad():
            model.predictor.conv1.conv.weight.mul_(-1.3)
            model.predictor.linear.weight.add_(0.05)
        assert model.greedy_decode(mel, lens, max_length=60) == model.greedy_decode(mel, lens, max_length=60, scan_frames=0)
    # max_length beyond the 2 048 ids the kernel keeps in LDS: the ids go to memory as they are found (10 per frame here: 750 of them)
    with torch.no_grad():
        model.joint.joint_ln.bias[V - 1] -= 6.0  # blank almost never wins
    want = model.greedy_decode(mel, lens, max_length=2100, scan_frames=0)
    assert len(want) > 300 and model.greedy_decode(mel, lens, max_length=2100, persistent=True) == want
    print("tokens per case:", counts)
    assert sum(c > 3 for c in counts) >= 8 and any(c == 59 for c in counts)  # the cases decode something; one runs into max_length
    # a width the persistent loop does not take: the default falls back to the kernel-per-layer loop, persistent=True says why
    pred = amd.ConvPredictor(32, 72, 48, 0.3)
    model = amd.RNNTModel(pred, Enc(72), amd.JointNetwork(-1, -1, 72, 32)).cuda().eval()
    assert not amd.engine.greedy_decode_persistent_supported(75, 32, 48, 72, 72, 32, False)
    mel = torch.randn(1, 10, 150, device="cuda")
    lens = torch.tensor([150], device="cuda")
    assert model.greedy_decode(mel, lens, max_length=40) == model.greedy_decode(mel, lens, max_length=40, scan_frames=0)
    with pytest.raises(RuntimeError, match="persistent greedy decode"):
        model.greedy_decode(mel, lens, max_length=40, persistent=True)


def test_greedy_decode_many_matches_one_by_one(amd):
    """RNNTModel.greedy_decode_many: several utterances in flight on streams of their own (up to compute units // workgroups per decode
    persistent launches side by side) return, in order, exactly what greedy_decode returns for each — different lengths, more utterances than
    streams, concurrency 1 / default / more than fits (clamped); a model the persistent loop does not take decodes one by one."""
    torch.manual_seed(9)

    class Enc(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.c = torch.nn.Conv1d(10, c, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    for fa, ft, hid, E, O, V, bias_blank in ((-1, -1, 128, 64, 128, 256, 0.0), (24, 1024, 256, 512, 1024, 1024, 0.3), (-1, -1, 72, 48, 72, 32, 0.8)):
        pred = amd.ConvPredictor(V, O, E, 0.3)
        model = amd.RNNTModel(pred, Enc(fa if fa > 0 else hid), amd.JointNetwork(fa, ft, hid, V)).cuda().eval()
        with torch.no_grad():
            model.joint.joint_ln.bias[V - 1] += bias_blank
        mels = [torch.randn(1, 10, n, device="cuda") for n in (150, 61, 240, 33, 199, 120, 87, 176, 54)]
        want = [model.greedy_decode(m, torch.tensor([m.shape[-1]], device="cuda"), max_length=50) for m in mels]
        assert sum(len(w) > 3 for w in want) >= 3, [len(w) for w in want]
        for conc in (None, 1, 3, 64):
            assert model.greedy_decode_many(mels, max_length=50, concurrency=conc) == want, (hid, conc)
    assert model.greedy_decode_many([], max_length=50) == []


def test_greedy_decode_stateful_predictor_branch(amd):
    """RNNTModel.greedy_decode with a STATEFUL predictor — forward(ids, lens, state=None) ->
    (features, lens, state), fed the last token only (reference rnnt/model.py:45-87, the LSTMPredictor
    branch): a running-sum stand-in whose state is its last output decodes exactly what the stateless
    predictor computing the same prefix sums from the whole token list does, scan and per-frame loop alike."""
    torch.manual_seed(11)

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv1d(10, 24, 3, stride=2, padding=1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return (lens + 1) // 2

    class Stateless(torch.nn.Module):
        def __init__(self, emb):
            super().__init__()
            self.e = emb

        def forward(self, ids):
            return torch.cumsum(self.e(ids), dim=1)

    class Stateful(torch.nn.Module):
        def __init__(self, emb):
            super().__init__()
            self.e = emb
            self.calls = []

        def forward(self, ids, lens, state=None):
            self.calls.append((ids.shape[1], state is not None))
            f = torch.cumsum(self.e(ids), dim=1)
            if state is not None:
                f = f + state
            return f, lens, f[:, -1:, :]

    emb = torch.nn.Embedding(16, 24)
    with torch.no_grad():
        emb.weight.mul_(0.3)
    joint = amd.JointNetwork(-1, -1, 24, 16)
    enc = Enc()
    ref_model = amd.RNNTModel(Stateless(emb), enc, joint).cuda()
    sf = Stateful(emb)
    model = amd.RNNTModel(sf, enc, joint).cuda()
    assert model._predictor_is_stateful() and not ref_model._predictor_is_stateful()
    with torch.no_grad():
        joint.joint_ln.bias[15] += 1.0
    mel = torch.randn(1, 10, 120, device="cuda")
    lens = torch.tensor([120], device="cuda")
    want = ref_model.greedy_decode(mel, lens, max_length=60, scan_frames=0)
    assert 0 < len(want) <= 59
    for scan in (16, 0, 128):
        sf.calls.clear()
        assert model.greedy_decode(mel, lens, max_length=60, scan_frames=scan) == want
        # first call: the start blank without a state; afterwards ONE token at a time with the state
        assert sf.calls[0] == (1, False) and all(c == (1, True) for c in sf.calls[1:])
        assert len(sf.calls) == len(want) + 1


def _bf16_vs_fp32_fullsize(amd, B, T, U, H, V, seed):
    """Full-size ragged inputs through both routes: every bf16 kernel at BASELINE sizes (row
    offsets beyond 2^31 bytes, all tiles / passes / splits), held to bf16's error of the fp32 route."""
    d = make_inputs(B, T, U, H, V, seed)
    amd.engine.release_workspaces()
    ref = _run_fused(amd, d, "fp32")
    amd.engine.release_workspaces()
    r = _run_fused(amd, d, "bf16")
    amd.engine.release_workspaces()
    assert_close_loss("costs", r["costs"], ref["costs"], rtol=BF16_LOSS_RTOL_EXACT)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k], rtol=BF16_GRAD_RTOL_EXACT)


def test_fullsize_config2_bf16_vs_fp32(amd):
    """BASELINE config 3's per-step arithmetic at config 2's size (B=32,T=1000,U=200,H=512,V=1024)."""
    _bf16_vs_fp32_fullsize(amd, 32, 1000, 200, 512, 1024, seed=32)
    _full(amd, 32, 1000, 200, 512, 1024, seed=2, dtype="bf16")
    amd.engine.release_workspaces()


def test_fullsize_config5_bf16_vs_fp32(amd):
    """Large vocabulary (B=16,T=800,U=150,H=512,V=16384) on the bf16 route: 64 forward passes, 512
    k chunks per dHidden tile, 64 dW column tiles."""
    _bf16_vs_fp32_fullsize(amd, 16, 800, 150, 512, 16384, seed=35)


def test_fullsize_config2_softmax_shift_invariance(amd, route):
    """Size-independent property at BASELINE config 2's full size with random W: adding a constant
    to every bias entry shifts all logits of a cell equally, so costs and every gradient must not
    move (the bias gradient keeps summing to ~0 as well).  Exercises all kernels on dense,
    non-degenerate data at 6.4 M cells."""
    d = make_inputs(32, 1000, 200, 512, 1024, seed=77)
    amd.engine.release_workspaces()
    r0 = _run_fused(amd, d, route)
    d2 = dict(d)
    d2["bias"] = (d["bias"] + np.float32(2.5)).astype(np.float32)
    r1 = _run_fused(amd, d2, route)
    amd.engine.release_workspaces()
    assert_close_loss("costs", r1["costs"], r0["costs"], rtol=2e-5)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r1[k], r0[k], rtol=2e-4)
    assert abs(r0["grad_bias"].sum()) < 1e-3 * np.abs(r0["grad_bias"]).sum()
    # per-utterance costs of ragged full-size inputs are positive and ordered by lattice size
    assert (r0["costs"] > 0).all()


def test_fullsize_config4_softmax_shift_invariance(amd, route):
    """The same size-independent property at BASELINE config 4's FULL size (B=8,T=4000,U=600,H=640,
    V=1024; 144 GB workspace, row offsets >> 2^31, ragged lengths) with dense random W."""
    d = make_inputs(8, 4000, 600, 640, 1024, seed=78)
    amd.engine.release_workspaces()
    r0 = _run_fused(amd, d, route)
    d2 = dict(d)
    d2["bias"] = (d["bias"] - np.float32(1.75)).astype(np.float32)
    r1 = _run_fused(amd, d2, route)
    amd.engine.release_workspaces()
    assert_close_loss("costs", r1["costs"], r0["costs"], rtol=2e-5)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r1[k], r0[k], rtol=2e-4)
        assert np.abs(r0[k]).max() > 0
    assert abs(r0["grad_bias"].sum()) < 1e-3 * np.abs(r0["grad_bias"]).sum()
    assert (r0["costs"] > 0).all()


def _fused_outs(amd, g, outs=None, *, dtype):
    V = g["W"].shape[0]
    return amd.engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"],
                                         g["target_lens"], V - 1, 0.25, outs=outs, dtype=dtype)


@pytest.mark.parametrize("dtype,shape", [("fp32", (3, 40, 12, 256, 512)), ("fp32", (2, 30, 9, 1024, 260)),
                                         ("fp32", (2, 30, 9, 1024, 256)), ("bf16", (3, 40, 12, 256, 512)),
                                         ("bf16", (2, 30, 9, 1024, 256)), ("f16x2", (3, 40, 12, 256, 512)), ("f16x2", (2, 30, 9, 1024, 256))])
def test_fused_call_is_hip_graph_capturable(amd, dtype, shape):
    """The C-ABI call enqueues KERNELS on the caller's stream and nothing else (no memset / memcpy nodes —
    fills and copies are kernels too, tests/test_abi.py::test_engine_sources_only_enqueue_kernels — no
    allocation, no synchronisation, no host read-back): it can be captured into a HIP graph and
    replayed.  The replay reads its inputs at replay time — new contents in the same buffers give the
    new answer, bit for bit what the eager call gives."""
    B, T, U, H, V = shape
    d1, d2 = make_inputs(B, T, U, H, V, seed=501), make_inputs(B, T, U, H, V, seed=502)
    g = _dev(d1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        outs = tuple(torch.empty_like(o) for o in _fused_outs(amd, g, dtype=dtype))  # warm-up: sizes the (device, stream) workspace
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            _fused_outs(amd, g, outs=outs, dtype=dtype)
    torch.cuda.current_stream().wait_stream(s)
    for d in (d2, d1):
        for k, v in d.items():
            g[k].copy_(torch.from_numpy(v))
        for o in outs:
            o.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        want = _fused_outs(amd, g, dtype=dtype)
        torch.cuda.synchronize()
        for a_, b_ in zip(got, want):
            assert torch.equal(a_, b_)
        ref = oracle_fused_bf16(d) if dtype == "bf16" else oracle_fused(d)
        assert_close_loss("costs", got[0].cpu().numpy(), ref["costs"], rtol=BF16_LOSS_RTOL if dtype == "bf16" else LOSS_RTOL)


def test_two_streams_do_not_share_scratch(amd, route):
    """Two streams of one device, different inputs, enqueued back to back with no synchronisation in
    between: each (device, stream) has its own workspace, so both answers equal their serial runs."""
    da, db = make_inputs(4, 120, 30, 512, 1024, seed=511), make_inputs(4, 120, 30, 512, 1024, seed=512)
    ga, gb = _dev(da), _dev(db)
    want_a = [o.clone() for o in _fused_outs(amd, ga, dtype=route)]
    want_b = [o.clone() for o in _fused_outs(amd, gb, dtype=route)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for _ in range(3):
        with torch.cuda.stream(sa):
            res["a"] = _fused_outs(amd, ga, dtype=route)
        with torch.cuda.stream(sb):
            res["b"] = _fused_outs(amd, gb, dtype=route)
    torch.cuda.synchronize()
    for got, want in ((res["a"], want_a), (res["b"], want_b)):
        for x, y in zip(got, want):
            assert torch.equal(x, y)


def _guarded(shape, dtype, margin=4096, fill=-7.25):
    """A tensor of `shape` inside a larger buffer whose margins carry a sentinel value: an out-of-bounds
    WRITE of a kernel shows up as a changed margin."""
    n = int(np.prod(shape)) if len(shape) else 1
    buf = torch.full((n + 2 * margin,), fill, dtype=dtype, device="cuda")
    return buf, buf[margin:margin + n].view(shape)


@pytest.mark.parametrize("dtype,shape", [("fp32", (3, 37, 11, 128, 132)), ("fp32", (2, 50, 21, 1024, 64)), ("fp32", (2, 33, 9, 640, 260)),
                                         ("bf16", (3, 37, 11, 128, 128)), ("bf16", (2, 26, 17, 1024, 256)),
                                         ("bf16x3", (3, 37, 11, 128, 128)), ("bf16x3", (2, 26, 17, 1024, 256)), ("bf16x3", (2, 33, 9, 640, 256)),
                                         ("f16x2", (3, 37, 11, 128, 128)), ("f16x2", (2, 26, 17, 1024, 256)), ("f16x2", (2, 33, 9, 640, 256))])
def test_kernels_write_only_inside_outputs_and_workspace(amd, dtype, shape, monkeypatch):
    """Every output and the workspace sit inside larger buffers with sentinel margins (the workspace at the
    exact size rnnt_engine_workspace_bytes asks for): after a fused call on a ragged batch the margins are
    untouched and the result is the one computed in ordinary allocations."""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=4000 + H + V)
    g = _dev(d)
    want = [o.clone() for o in _fused_outs(amd, g, dtype=dtype)]
    torch.cuda.synchronize()
    bufs, outs = zip(*[_guarded(tuple(o.shape), torch.float32) for o in want])
    ws_holder = {}

    def guarded_workspace(device, nbytes):
        # 256-byte aligned start inside a byte buffer with 64 KiB sentinel margins
        buf = torch.full((int(nbytes) + 2 * 65536,), 0x5A, dtype=torch.uint8, device=device)
        off = 65536 + (-(buf.data_ptr() + 65536)) % 256
        ws_holder["buf"], ws_holder["off"], ws_holder["n"] = buf, off, int(nbytes)
        return buf[off:off + int(nbytes)]

    monkeypatch.setattr(amd.engine, "workspace", guarded_workspace)
    got = _fused_outs(amd, g, outs=outs, dtype=dtype)
    torch.cuda.synchronize()
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)
    for buf, o in zip(bufs, outs):
        n = o.numel()
        assert (buf[:4096] == -7.25).all() and (buf[4096 + n:] == -7.25).all()
    buf, off, n = ws_holder["buf"], ws_holder["off"], ws_holder["n"]
    assert (buf[:off] == 0x5A).all() and (buf[off + n:] == 0x5A).all()
