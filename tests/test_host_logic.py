"""CPU tests of the host-side mirror of the reference interface (no GPU, no compute)."""
import os

import numpy as np
import pytest
import torch

import rnnt_amd


def test_joint_network_mirrors_reference_interface(golden_dir):
    """Constructor args, attribute names and state-dict keys of reference rnnt/joint.py:4-20."""
    for name in ("joint_tiny", "joint_proj"):
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        Fa, Ft, H, V = (int(x) for x in z["ctor"])
        m = rnnt_amd.JointNetwork(Fa, Ft, H, V)
        ref_keys = sorted(k[4:].replace("__", ".") for k in z.files if k.startswith("sd__"))
        assert sorted(m.state_dict().keys()) == ref_keys
        assert m.blank_idx == V - 1
        assert hasattr(m, "audio_ln") == (Fa > 0) and hasattr(m, "text_ln") == (Ft > 0)
        m.load_state_dict({k: torch.from_numpy(z["sd__" + k.replace(".", "__")]) for k in ref_keys})
        # single_forward (decode/export path, plain torch by design) reproduces the reference's
        # logits frame by frame
        a, t = torch.from_numpy(z["audio"]), torch.from_numpy(z["text"])
        out = m.single_forward(a[:, 2, :], t[:, 1, :]).detach().numpy()
        np.testing.assert_allclose(out, z["logits_f32"][:, 2, 1, :], rtol=0, atol=2e-5)


def test_batch_forward_rejects_cpu_tensors():
    """The hot path has no CPU fallback: CPU tensors fail loudly."""
    m = rnnt_amd.JointNetwork(-1, -1, 16, 8)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.randn(2, 5, 16), torch.randn(2, 3, 16))
    with pytest.raises(RuntimeError, match="HIP device"):
        rnnt_amd.joint_rnnt_loss(torch.randn(2, 5, 16), torch.randn(2, 3, 16), torch.randn(8, 16),
                                 torch.randn(8), torch.zeros(2, 2, dtype=torch.int32),
                                 torch.tensor([5, 4], dtype=torch.int32),
                                 torch.tensor([2, 1], dtype=torch.int32))


def test_rnnt_loss_argument_checks_match_torchaudio_style():
    logits = torch.randn(2, 5, 3, 8)
    tg = torch.zeros(2, 2, dtype=torch.int32)
    ll = torch.tensor([5, 4], dtype=torch.int32)
    tl = torch.tensor([2, 1], dtype=torch.int32)
    with pytest.raises(ValueError):
        rnnt_amd.rnnt_loss(logits, tg, ll, tl, reduction="avg")
    with pytest.raises(RuntimeError, match="int32"):
        rnnt_amd.rnnt_loss(logits, tg.long(), ll, tl)
    with pytest.raises(RuntimeError, match="int32"):
        rnnt_amd.rnnt_loss(logits, tg, ll.long(), tl)
    with pytest.raises(RuntimeError, match="4 dimensions"):
        rnnt_amd.rnnt_loss(logits[0], tg, ll, tl)
    with pytest.raises(RuntimeError, match="float32"):
        rnnt_amd.rnnt_loss(logits.double(), tg, ll, tl)
    with pytest.raises(RuntimeError, match="blank"):
        rnnt_amd.rnnt_loss(logits, tg, ll, tl, blank=8)
    with pytest.raises(RuntimeError, match="input length mismatch"):
        rnnt_amd.rnnt_loss(logits, tg, torch.tensor([4, 4], dtype=torch.int32), tl)
    with pytest.raises(RuntimeError, match="output length mismatch"):
        rnnt_amd.rnnt_loss(logits, tg, ll, torch.tensor([1, 1], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="batch"):
        rnnt_amd.rnnt_loss(logits, tg[:1], ll, tl)
    with pytest.raises(NotImplementedError):
        rnnt_amd.joint_rnnt_loss(torch.randn(2, 5, 16), torch.randn(2, 3, 16), torch.randn(8, 16),
                                 torch.randn(8), tg, ll, tl, reduction="none")


def test_model_container_interface():
    """RNNTModel(predictor, encoder, joint) / .device / .greedy_decode (reference model.py:7-139)."""
    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv1d(4, 8, 1)

        def forward(self, x):
            return self.c(x)

        def calc_output_lens(self, lens):
            return lens

    class Pred(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.e = torch.nn.Embedding(6, 8)

        def forward(self, ids):
            return self.e(ids)

    torch.manual_seed(0)
    m = rnnt_amd.RNNTModel(Pred(), Enc(), rnnt_amd.JointNetwork(-1, -1, 8, 6))
    assert m.device.type == "cpu"
    assert sorted(dict(m.named_children())) == ["encoder", "joint", "predictor"]
    toks = m.greedy_decode(torch.randn(1, 4, 7), torch.tensor([7]), max_length=5)
    assert isinstance(toks, list) and len(toks) <= 4 and all(0 <= t < 5 for t in toks)


def test_padding_helper_keeps_zero_probability_columns():
    from rnnt_amd.functional import _pad_hv
    enc, pred = torch.randn(1, 2, 6), torch.randn(1, 2, 6)
    W, b = torch.randn(5, 6), torch.randn(5)
    e, p, Wp, bp, H, V = _pad_hv(enc, pred, W, b)
    assert e.shape[-1] == 8 and Wp.shape == (8, 8) and (H, V) == (6, 5)
    assert (Wp[5:] == 0).all() and (Wp[:, 6:] == 0).all() and (bp[5:] < -1e29).all()


def test_conv_predictor_mirrors_reference_interface_and_has_no_cpu_path(golden_dir):
    """rnnt_amd.ConvPredictor: the reference's constructor and state-dict keys
    (rnnt/predictor.py:189-209: embedding, input_layer_norm, conv1.conv, conv2.conv, linear,
    output_layer_norm); CPU tensors fail loudly (no fallback)."""
    z = np.load(os.path.join(golden_dir, "predictor_small.npz"))
    S, O, E, B, U1 = (int(v) for v in z["dims"])
    m = rnnt_amd.ConvPredictor(S, O, E, dropout=0.3)
    ref_keys = sorted(k[4:].replace("__", ".") for k in z.files if k.startswith("sd__"))
    assert sorted(m.state_dict().keys()) == ref_keys
    assert m.conv1.conv.weight.shape == (E, E, 3) and m.conv2.conv.weight.shape == (E, E, 5)
    m.load_state_dict({k: torch.from_numpy(z["sd__" + k.replace(".", "__")]) for k in ref_keys})
    with pytest.raises(RuntimeError, match="HIP device"):
        m(torch.from_numpy(z["ids"]))
    with pytest.raises(NotImplementedError):
        from rnnt_amd.predictor import CausalConv1d
        CausalConv1d(4, 4, 3, stride=2, dilation=1)


def test_optim_host_logic():
    """clip_grad_norm_ on the reference's exhausted generator (rnnt/train.py:95,104,136) is a no-op
    returning 0; AdamW validates hyper-parameters and refuses CPU parameters (no fallback)."""
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.ones(4) * 3
    gen = (q for q in [p])
    list(gen)
    assert float(rnnt_amd.optim.clip_grad_norm_(gen, 0.1)) == 0.0 and torch.equal(p.grad, torch.ones(4) * 3)
    assert float(rnnt_amd.optim.clip_grad_norm_([torch.nn.Parameter(torch.ones(2))], 1.0)) == 0.0  # no .grad
    with pytest.raises(ValueError):
        rnnt_amd.optim.AdamW([p], lr=-1.0)
    opt = rnnt_amd.optim.AdamW([p], lr=3e-4, betas=(0.95, 0.9999), eps=1e-8, weight_decay=0.01)
    assert opt.defaults["betas"] == (0.95, 0.9999)
    with pytest.raises(RuntimeError, match="HIP device"):
        opt.step()


def test_projected_joint_single_forward_stays_plain_torch_on_cpu(golden_dir):
    """The export / decode path (rnnt/joint.py:44-55) with audio_ln / text_ln runs on CPU tensors
    through torch's own Linear (the engine's projection kernels only take HIP tensors)."""
    z = np.load(os.path.join(golden_dir, "joint_proj.npz"))
    Fa, Ft, H, V = (int(x) for x in z["ctor"])
    m = rnnt_amd.JointNetwork(Fa, Ft, H, V)
    m.load_state_dict({k[4:].replace("__", "."): torch.from_numpy(z[k]) for k in z.files if k.startswith("sd__")})
    a, t = torch.from_numpy(z["audio"]), torch.from_numpy(z["text"])
    out = m.single_forward(a[:, 1, :], t[:, 2, :]).detach().numpy()
    np.testing.assert_allclose(out, z["logits_f32"][:, 1, 2, :], rtol=0, atol=2e-5)


@pytest.mark.skipif(not os.path.isdir("/root/reference/rnnt"), reason="needs the reference checkout (build container only)")
def test_overlay_package_swaps_joint_model_and_conv_predictor_only():
    """integration/rnnt placed before the reference on PYTHONPATH: `rnnt.joint.JointNetwork` (the hydra target, train.py:63),
    `rnnt.model.RNNTModel` (train.py:19) and `rnnt.predictor.ConvPredictor` (the hydra target of config/basic_sp_convjs*.yaml:20-25,
    train.py:61) become the engine's classes — rnnt.model imports without torchaudio — while rnnt.predictor.LSTMPredictor IS the
    reference's class and rnnt.jasper / rnnt.lr_sched / rnnt.causalconv still come from the reference's own files.  With the engine's
    ConvPredictor in place the model's evaluation decode is eligible for the device loop (the rest of `_device_loop_ok` needs a GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import rnnt, rnnt.joint, rnnt.model, rnnt.predictor, rnnt.jasper, rnnt.lr_sched, rnnt.causalconv, rnnt_amd\n"
        "assert rnnt.joint.JointNetwork is rnnt_amd.JointNetwork\n"
        "assert rnnt.model.RNNTModel is rnnt_amd.RNNTModel\n"
        "assert rnnt.predictor.ConvPredictor is rnnt_amd.ConvPredictor\n"
        "assert rnnt.predictor.LSTMPredictor.__module__ == 'rnnt._reference_predictor'\n"
        "assert rnnt.predictor.ReferenceConvPredictor is not rnnt_amd.ConvPredictor\n"
        "import inspect; assert inspect.getsourcefile(rnnt.predictor.LSTMPredictor).startswith('/root/reference/')\n"
        "assert rnnt.jasper.__file__.startswith('/root/reference/') and rnnt.causalconv.__file__.startswith('/root/reference/')\n"
        "from rnnt.predictor import LSTMPredictor, ConvPredictor\n"  # (the reference's own import line, rnnt/model.py:3)
        "m = rnnt.model.RNNTModel(rnnt.predictor.ConvPredictor(16, 32, 8, 0.1), None, rnnt.joint.JointNetwork(-1, -1, 32, 16))\n"
        "assert sorted(k for k in m.state_dict() if k.startswith('joint.')) == ['joint.joint_ln.bias', 'joint.joint_ln.weight']\n"
        "ref = rnnt.predictor.ReferenceConvPredictor(16, 32, 8, 0.1)\n"
        "assert list(ref.state_dict()) == list(m.predictor.state_dict())\n"  # checkpoints move both ways
        "lstm = rnnt.predictor.LSTMPredictor(16, 32, 8, 1, 16, 0.1, 0.1)\n"
        "assert rnnt.model.RNNTModel(lstm, None, rnnt.joint.JointNetwork(-1, -1, 32, 16))._predictor_is_stateful()\n"
        "try:\n"
        "    rnnt.predictor.no_such_name\n"
        "    raise SystemExit('missing attribute did not raise')\n"
        "except AttributeError:\n"
        "    pass\n"
        "print('overlay ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "integration"), root, "/root/reference"]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "overlay ok" in out.stdout, out.stderr[-1500:]


def test_capturable_adamw_resume_keeps_the_loaded_step_count():
    """AdamW(capturable=True).load_state_dict after the first step — the device step counter
    continues from the LOADED count (not the pre-load one), in place, and the loaded learning rate lands
    in the tensor a captured graph points at.  Host logic only: no engine call, CPU tensors."""
    import torch
    from rnnt_amd.optim import AdamW
    ps = [torch.zeros(4, requires_grad=True), torch.zeros(3, requires_grad=True)]
    opt = AdamW(ps, lr=1e-3, capturable=True)
    for p in ps:
        opt.state[p].update(step=torch.tensor(5.0), exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
    step_dev, _ = opt._seed_dev(0, ps, torch.device("cpu"))
    assert int(step_dev) == 5
    lr_dev = torch.tensor(1e-3)
    opt.param_groups[0]["lr"] = lr_dev
    for p in ps:
        opt.state[p]["step"] = step_dev  # what _step_capturable does: one shared counter
    sd = opt.state_dict()
    for st in sd["state"].values():
        st["step"] = torch.tensor(12.0)  # a checkpoint written 7 steps later
    sd["param_groups"][0]["lr"] = 2.5e-4
    opt.load_state_dict(sd)
    assert opt._dev[0][0] is step_dev and int(step_dev) == 12       # same tensor, loaded count
    assert opt.param_groups[0]["lr"] is lr_dev and abs(float(lr_dev) - 2.5e-4) < 1e-10
    assert opt._state_step(ps) == 12
    # a fresh optimizer that loads before its first step seeds its counter from the loaded state
    opt2 = AdamW([torch.zeros(4, requires_grad=True), torch.zeros(3, requires_grad=True)], lr=1e-3, capturable=True)
    opt2.load_state_dict(sd)
    assert int(opt2._seed_dev(0, opt2.param_groups[0]["params"], torch.device("cpu"))[0]) == 12


def test_workspace_of_a_captured_graph_is_pinned(monkeypatch):
    """a workspace handed out while the stream is capturing is baked into the graph's kernel
    arguments — it must never be replaced (growth raises) or dropped by release_workspaces()."""
    import torch
    from rnnt_amd import engine
    monkeypatch.setattr(engine, "_workspaces", {})
    monkeypatch.setattr(engine, "_captured", {})
    state = {"capturing": False}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: state["capturing"])

    class _S:
        cuda_stream = 7
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: _S())
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda n, dtype=None, device=None: real_empty(n, dtype=dtype))
    dev = torch.device("cuda", 0)
    a = engine.workspace(dev, 100)
    assert engine.workspace(dev, 50) is a            # grow-only cache
    b = engine.workspace(dev, 200)                   # not captured yet: growth replaces
    assert b is not a and b.numel() == 200
    state["capturing"] = True
    assert engine.workspace(dev, 150) is b           # handed out under capture -> pinned
    state["capturing"] = False
    engine.release_workspaces()
    assert engine.workspace(dev, 10) is b            # still there after a release
    import pytest
    with pytest.raises(RuntimeError, match="captured HIP graph"):
        engine.workspace(dev, 400)


def test_decode_and_projection_host_decisions():
    """Host-side decisions of round 5, no device needed: which projections "auto" hands to the f16x2 kernels (work threshold, shapes they
    take), which decode sizes the persistent launch takes (the C ABI's own answer: workspace_bytes fails for the others), and that a
    give-up reported in state[7] is raised, never read as a decode."""
    import pytest
    from rnnt_amd import engine
    assert engine.linear_x2_preferred(6432, 1024, 1024) and engine.linear_x2_preferred(32000, 512, 1024)
    assert not engine.linear_x2_preferred(8000, 512, 1024) and not engine.linear_x2_preferred(808, 1024, 1024)
    assert not engine.linear_x2_preferred(32000, 1000, 1024)  # (K not a multiple of 128: the fp32-MFMA kernels or the library)
    assert engine.greedy_decode_persistent_supported(1000, 1024, 512, 1024, 1024, 1024, False)
    assert engine.greedy_decode_persistent_supported(1000, 1024, 512, 1024, 512, 1024, True)
    assert not engine.greedy_decode_persistent_supported(1000, 1024, 512, 72, 72, 32, False)      # H % 64
    assert not engine.greedy_decode_persistent_supported(1000, 5000, 512, 1024, 1024, 5000, False)  # > 4096 symbols: the conv1 tables
    assert not engine.greedy_decode_persistent_supported(1000, 1024, 512, 1024, 512, 1024, False)   # no text_ln: O must equal H
    engine.check_decode_state([10, 0, 3, 1, 1, 7, 64, 0])
    with pytest.raises(RuntimeError, match="gave up waiting for hand-off 2"):
        engine.check_decode_state([10, 0, 3, 0, 1, 7, 64, 2])
