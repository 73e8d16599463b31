"""-m gpu: the RNNT_DTYPE_F32_BF16X3 route (rnnt_amd/csrc/x3.hip: fp32-accurate products as six bf16 MFMA
products of 3-way split operands) held to EXACTLY the fp32 route's bar — every fp32 parity case of
tests/test_gpu_parity.py at the unchanged 1e-4 tolerances (tests/helpers.py LOSS_RTOL / GRAD_RTOL) against the
plain fp64 oracle (no rounding-point oracle: the route claims fp32 accuracy) — plus what is specific to it:
each bf16x3 kernel checked in isolation (the other stages on the fp32 route's kernels, RNNT_VARIANT_X3_FP32_*),
and its measured error beside the fp32-MFMA route's on the same inputs."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import (GRAD_RTOL, LOSS_RTOL, assert_close_grad, assert_close_loss, make_inputs, oracle_fused)
from tests.test_gpu_parity import FUSED_SHAPES, _compare, _dev, _full, _random_case, _run_fused, _sd

pytestmark = pytest.mark.gpu
X3 = "bf16x3"


@pytest.fixture(scope="module")
def amd():
    import rnnt_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    rnnt_amd.engine.lib()
    return rnnt_amd


@pytest.fixture(scope="module")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("shape", FUSED_SHAPES)
def test_x3_fused_joint_loss_vs_oracle(amd, shape):
    """The fp32 route's own shape list (single cell, U = 0, T = 1, ragged, H % 8, V % 4, H > 512, config 2's and
    config 4's lattice lengths): the host side zero-pads H and V to multiples of 128 for this route."""
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=sum(shape))
    _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))


@pytest.mark.parametrize("variant", ["dw_only", "dw_dhidden", "all"])
@pytest.mark.parametrize("shape", [(2, 9, 4, 128, 128), (3, 23, 19, 256, 384), (2, 40, 33, 512, 1024), (2, 13, 20, 1024, 256),
                                   (3, 21, 9, 640, 128), (2, 130, 50, 512, 256)])
def test_x3_kernels_in_isolation(amd, variant, shape):
    """One bf16x3 kernel at a time: with RNNT_VARIANT_X3_FP32_FWD | _DH only k_dw_x3 runs (forward and dHidden
    on the fp32 route's kernels, plain splitting kernels in between), with _FWD alone k_dhidden_x3 + k_dw_x3,
    without a variant all three — each against the fp64 oracle at the fp32 tolerances."""
    e = amd.engine
    var = {"dw_only": e.VARIANT_X3_FP32_FWD | e.VARIANT_X3_FP32_DH, "dw_dhidden": e.VARIANT_X3_FP32_FWD, "all": 0}[variant]
    B, T, U, H, V = shape
    d = make_inputs(B, T, U, H, V, seed=sum(shape))
    g = _dev(d)
    outs = e.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
                                V - 1, 1.0 / B, dtype=X3, variant=var)
    torch.cuda.synchronize()
    ref = oracle_fused(d)
    assert_close_loss("costs", outs[0].cpu().numpy(), ref["costs"])
    for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias")):
        assert_close_grad(k, o.cpu().numpy(), ref[k])


def test_x3_very_ragged_batch_and_workspace_reuse(amd):
    """Utterances of 1, 2 and a few time steps next to a full one, empty and full targets; twice on one
    workspace (rows the first call left behind must not leak into the second)."""
    d = make_inputs(6, 37, 10, 128, 128, seed=77)
    d["logit_lens"] = np.array([37, 1, 2, 9, 36, 17], dtype=np.int32)
    d["target_lens"] = np.array([10, 0, 1, 10, 0, 5], dtype=np.int32)
    _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))
    d2 = dict(d)
    d2["logit_lens"] = np.array([3, 37, 30, 1, 1, 37], dtype=np.int32)
    d2["target_lens"] = np.array([2, 10, 0, 0, 10, 3], dtype=np.int32)
    _compare(_run_fused(amd, d2, dtype=X3), oracle_fused(d2))


def test_x3_random_shapes_and_lengths_vs_oracle(amd):
    """The seeded sweep of test_fused_random_shapes_and_lengths_vs_oracle (arbitrary lengths per utterance,
    H / V not multiples of anything) on this route."""
    d = make_inputs(5, 41, 4, 480, 192, seed=5)
    d["logit_lens"] = np.array([20, 8, 41, 19, 28], dtype=np.int32)
    d["target_lens"] = np.array([1, 4, 0, 4, 3], dtype=np.int32)
    _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))
    rng = np.random.default_rng(4242)
    for it in range(16):
        d = _random_case(rng, it % 2 == 1)  # odd cases: H, V multiples of 128 (no padding)
        _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))


@pytest.mark.parametrize("pattern", [0x7FA00000, 0xFFFFFFFF])
def test_x3_poisoned_workspace_does_not_leak(amd, pattern):
    """Whatever the caller-owned workspace holds (signalling / quiet NaNs) must not reach a result: slots no
    kernel writes are only ever dropped by a select or a range check, G rows of dead tiles are zero-filled."""
    dev = torch.device("cuda", 0)

    def poison():
        ws = amd.engine.workspace(dev, 1)
        ws.view(torch.int32)[: ws.numel() // 4].fill_(pattern - (1 << 32) if pattern >= (1 << 31) else pattern)

    for (B, T, U, H, V) in ((3, 41, 13, 128, 128), (2, 30, 9, 640, 256), (3, 61, 70, 128, 128)):
        d = make_inputs(B, T, U, H, V, seed=pattern & 0xffff)
        d["logit_lens"] = np.array(([T, 7, 23] if B == 3 else [11, T]), dtype=np.int32)
        d["target_lens"] = np.array(([4, U, 0] if B == 3 else [U, 2]), dtype=np.int32)
        if U >= 70:
            d["target_lens"] = np.array([U, 9, 33], dtype=np.int32)
        _run_fused(amd, d, dtype=X3)  # sizes the workspace
        poison()
        _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))


def test_x3_config1_plumbing_shape_vs_oracle(amd):
    """BASELINE.json configs[0]'s shape (B=2, T~200, U~50, H=1024, V=1024): two 512-column dHidden launches."""
    d = make_inputs(2, 208, 50, 1024, 1024, seed=208)
    _compare(_run_fused(amd, d, dtype=X3), oracle_fused(d))


@pytest.mark.parametrize("name", ["e2e_tiny", "e2e_mid", "e2e_proj", "e2e_v1024", "e2e_refmodules"])
def test_x3_golden_fixtures(amd, golden_dir, name):
    """The committed end-to-end fixtures (reference JointNetwork in fp64 + independent autograd loss;
    e2e_refmodules: inputs from the reference's own AudioEncoder / ConvPredictor, encoder output handed over as
    the permuted (N,C,L) view of rnnt/model.py:28) through fused_loss(dtype="bf16x3") at the fp32 tolerances."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = amd.JointNetwork(*[int(x) for x in z["ctor"]]).cuda()
    m.load_state_dict(_sd(z))
    t = torch.from_numpy(z["text"]).cuda().requires_grad_(True)
    if name == "e2e_refmodules":
        enc_ncl = torch.from_numpy(z["enc_ncl"]).cuda().requires_grad_(True)
        a = enc_ncl.permute(0, 2, 1)
    else:
        a = torch.from_numpy(z["audio"]).cuda().requires_grad_(True)
    loss = m.fused_loss(a, t, torch.from_numpy(z["targets"]).cuda(), torch.from_numpy(z["logit_lens"]).cuda(),
                        torch.from_numpy(z["target_lens"]).cuda(), dtype=X3)
    loss.backward()
    assert_close_loss("loss", loss.item(), float(z["loss"]))
    ga = enc_ncl.grad.permute(0, 2, 1) if name == "e2e_refmodules" else a.grad
    assert_close_grad("grad_audio", ga.cpu().numpy(), z["grad_audio"])
    assert_close_grad("grad_text", t.grad.cpu().numpy(), z["grad_text"])
    for k, p in m.named_parameters():
        assert_close_grad(k, p.grad.cpu().numpy(), z["grad__" + k.replace(".", "__")])


def test_x3_error_beside_the_fp32_mfma_route(amd):
    """Both routes against the fp64 oracle on the same inputs (a lattice of 10 k cells at config 2's H, V): the
    bf16x3 route stays inside the fp32 bar with the same two orders of magnitude to spare as the fp32-MFMA route
    (errors of a few 1e-7 of the largest gradient entry on both; the numbers go to stdout for DESIGN.md)."""
    d = make_inputs(4, 100, 24, 512, 1024, seed=1234)
    ref = oracle_fused(d)
    err = {}
    for dt in ("fp32", X3):
        r = _run_fused(amd, d, dtype=dt)
        err[dt] = {k: float(np.abs(r[k] - ref[k]).max() / np.abs(ref[k]).max()) for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias")}
        err[dt]["loss"] = abs(r["loss"] - ref["loss"]) / abs(ref["loss"])
    print("\nerror vs fp64 oracle:", {dt: {k: "%.1e" % v for k, v in e.items()} for dt, e in err.items()})
    for k, v in err[X3].items():
        assert v < 0.05 * GRAD_RTOL, (k, v)              # 20x inside the 1e-4 bar
        assert v < 4.0 * err["fp32"][k] + 2e-7, (k, v, err["fp32"][k])  # same error class as the exact-fp32 MFMA route


def test_x3_fullsize_config2_closed_form_and_vs_fp32(amd):
    """BASELINE config 2 at full size (B=32,T=1000,U=200,H=512,V=1024) on this route: known-answer loss, then
    dense ragged data against the fp32-MFMA route at the fp32 tolerances (row offsets beyond 2^31 bytes, every
    tile / pass / split)."""
    amd.engine.release_workspaces()
    _full(amd, 32, 1000, 200, 512, 1024, seed=2, dtype=X3)
    d = make_inputs(32, 1000, 200, 512, 1024, seed=32)
    amd.engine.release_workspaces()
    ref = _run_fused(amd, d)
    amd.engine.release_workspaces()
    r = _run_fused(amd, d, dtype=X3)
    amd.engine.release_workspaces()
    assert_close_loss("costs", r["costs"], ref["costs"], rtol=LOSS_RTOL)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k], rtol=GRAD_RTOL)


def test_x3_fullsize_config5_large_vocab_closed_form(amd):
    """BASELINE config 5 (B=16,T=800,U=150,H=512,V=16384): 32 forward passes, 1024 k-steps per dHidden tile, 64
    dW column blocks."""
    amd.engine.release_workspaces()
    _full(amd, 16, 800, 150, 512, 16384, seed=5, dtype=X3)
    amd.engine.release_workspaces()


def test_x3_reference_joint_width_vs_fp32(amd):
    """The reference's real joint width (hidden_features: 1024) at a training-sized ragged batch, against the
    fp32-MFMA route."""
    d = make_inputs(8, 500, 100, 1024, 1024, seed=11)
    ref = _run_fused(amd, d)
    r = _run_fused(amd, d, dtype=X3)
    assert_close_loss("costs", r["costs"], ref["costs"], rtol=LOSS_RTOL)
    for k in ("grad_enc", "grad_pred", "grad_W", "grad_bias"):
        assert_close_grad(k, r[k], ref[k], rtol=GRAD_RTOL)
    amd.engine.release_workspaces()


def test_x3_is_bitwise_reproducible_and_graph_capturable(amd):
    """Fixed summation orders: two eager calls agree bit for bit; a captured call replayed on new buffer
    contents gives the eager answer bit for bit (kernel nodes only, no memset / memcpy nodes)."""
    B, T, U, H, V = 3, 40, 12, 256, 512
    d1, d2 = make_inputs(B, T, U, H, V, seed=501), make_inputs(B, T, U, H, V, seed=502)
    g = _dev(d1)
    e = amd.engine
    run = lambda outs=None: e.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"],
                                                 g["target_lens"], V - 1, 0.25, outs=outs, dtype=X3)
    a = [o.clone() for o in run()]
    b = [o.clone() for o in run()]
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        outs = run()  # warm-up on the capture stream (sizes its workspace)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            run(outs)
    for k, v in _dev(d2).items():
        g[k].copy_(v)
    graph.replay()
    torch.cuda.synchronize()
    want = run()
    torch.cuda.synchronize()
    for x, y in zip(outs, want):
        assert torch.equal(x, y)
    ref = oracle_fused(d2)
    assert_close_loss("costs", want[0].cpu().numpy(), ref["costs"])


def test_x3_rejects_unsupported_dims_at_the_c_abi(amd):
    """include/rnnt_engine.h: RNNT_DTYPE_F32_BF16X3 needs H % 128 == 0 and V % 128 == 0 at the C boundary (the
    Python operator pads); a variant that runs a stage on the fp32 kernels needs the larger workspace."""
    d = make_inputs(2, 9, 4, 64, 128, seed=1)
    g = _dev(d)
    with pytest.raises(RuntimeError, match="RNNT_DTYPE_F32_BF16X3"):
        amd.engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"],
                                      g["target_lens"], 127, 0.5, dtype=X3)
