"""-m gpu: every greedy-decode path of rnnt_amd.RNNTModel — per-frame host loop, device scan (rnnt_engine_greedy_scan), the
kernel-per-layer device loop (rnnt_engine_greedy_decode), the persistent launch (rnnt_engine_greedy_decode_persistent) and
greedy_decode_many — against (1) token lists the REFERENCE's own ConvPredictor + JointNetwork decoded
(tests/golden/decode_*.npz; reference rnnt/model.py:90-128, rnnt/predictor.py:189-229, rnnt/joint.py:44-55) and (2) the
numpy oracle (oracle/decode_oracle.py, itself pinned by those fixtures) on freshly seeded inputs."""
import numpy as np
import pytest
import torch

from oracle import decode_oracle
from tests.helpers import DECODE_CASES, decode_case_arrays, load_decode_case

pytestmark = pytest.mark.gpu


class PassThroughEncoder(torch.nn.Module):
    """The encoder is not on the decode path's arithmetic: the case's frames are handed over as the (N, C, L) tensor an encoder
    would return, so RNNTModel.greedy_decode's own permute(0, 2, 1) (reference rnnt/model.py:93) produces the (N, L, C) view."""

    def forward(self, x):
        return x

    def calc_output_lens(self, lens):
        return lens


def build_model(spec, pred_sd, joint_sd):
    import rnnt_amd
    rnnt_amd.engine.lib()  # fail loudly if the HIP extension is missing
    pred = rnnt_amd.ConvPredictor(spec["V"], spec["O"], spec["E"], 0.3)
    joint = rnnt_amd.JointNetwork(spec["fa"], spec["ft"], spec["H"], spec["V"])
    for mod, sd in ((pred, pred_sd), (joint, joint_sd)):  # the reference's state-dict keys, strictly
        res = mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    return rnnt_amd.RNNTModel(pred, PassThroughEncoder(), joint).cuda().eval()


def all_paths(model, mel, lens, max_length, want, tag):
    import rnnt_amd
    spec_ok = model._device_loop_ok(torch.zeros(1, 1, device="cuda"))
    assert spec_ok, tag
    got = {"per_frame_host_loop": model.greedy_decode(mel, lens, max_length=max_length, scan_frames=0),
           "scan16": model.greedy_decode(mel, lens, max_length=max_length, scan_frames=16, device_loop=False),
           "scan128": model.greedy_decode(mel, lens, max_length=max_length, scan_frames=128, device_loop=False),
           "device_loop16": model.greedy_decode(mel, lens, max_length=max_length, scan_frames=16, device_loop=True, persistent=False),
           "device_loop7": model.greedy_decode(mel, lens, max_length=max_length, scan_frames=7, device_loop=True, persistent=False),
           "default": model.greedy_decode(mel, lens, max_length=max_length)}
    p = model.predictor
    S, E = p.embedding.weight.shape
    H, V = model.joint.joint_ln.in_features, model.joint.joint_ln.out_features
    if rnnt_amd.engine.greedy_decode_persistent_supported(mel.shape[-1], S, E, p.linear.out_features, H, V, hasattr(model.joint, "text_ln")):
        got["persistent"] = model.greedy_decode(mel, lens, max_length=max_length, persistent=True)
    got["many"] = model.greedy_decode_many([mel], max_length=max_length)[0]
    for k, v in got.items():
        assert v == want, (tag, k, max_length, v, want)
    return set(got)


@pytest.mark.parametrize("name", list(DECODE_CASES))
def test_decode_paths_match_reference_token_lists(golden_dir, name):
    c = load_decode_case(golden_dir, name)
    model = build_model(c["spec"], c["pred_sd"], c["joint_sd"])
    mel = torch.from_numpy(np.ascontiguousarray(c["frames"].T))[None].cuda()  # (1, C, T)
    lens = torch.tensor([mel.shape[-1]], device="cuda")
    paths = set()
    for ml, want in c["tokens"].items():
        paths |= all_paths(model, mel, lens, ml, want, name)
    assert "persistent" in paths  # every fixture has sizes the persistent launch takes
    # several utterances in flight: the same utterance, a truncated one and the full one again, in order
    half = mel[..., : mel.shape[-1] // 2].contiguous()
    ml = max(c["tokens"])
    many = model.greedy_decode_many([mel, half, mel, half, mel], max_length=ml)
    assert many[0] == many[2] == many[4] == c["tokens"][ml] and many[1] == many[3]
    want_half, margins = decode_oracle.greedy_decode(c["frames"][: half.shape[-1]], c["pred_sd"], c["joint_sd"], max_length=ml,
                                                     window=7 if c["spec"]["E"] > 64 else None)
    assert many[1] == want_half and margins.min() > 1e-3


@pytest.mark.parametrize("name,seeds", [("decode_small", range(7)), ("decode_small_proj", range(7)), ("decode_ref_widths_proj", range(2)),
                                        ("decode_ref_widths", range(2))])
def test_decode_paths_match_oracle_on_fresh_seeds(golden_dir, name, seeds):
    """Beyond the committed lists: new weights and frames per seed (the case's shapes and blank bias), the numpy oracle as the checker.
    A seed whose smallest top-2 gap is at rounding level proves nothing either way and is skipped — most must count."""
    spec = DECODE_CASES[name]
    bias = float(np.load(f"{golden_dir}/{name}.npz")["blank_bias"])
    counted = 0
    for s in seeds:
        frames, pred_sd, joint_sd = decode_case_arrays(spec, 4200 + s, bias)
        ml = spec["max_lengths"][0]
        want, margins = decode_oracle.greedy_decode(frames, pred_sd, joint_sd, max_length=ml, window=7 if spec["E"] > 64 else None)
        if margins.min() < 1e-3:
            continue
        counted += 1
        model = build_model(spec, pred_sd, joint_sd)
        mel = torch.from_numpy(np.ascontiguousarray(frames.T))[None].cuda()
        all_paths(model, mel, torch.tensor([mel.shape[-1]], device="cuda"), ml, want, (name, s))
    assert counted >= (len(seeds) + 1) // 2


@pytest.mark.parametrize("capturable", [False, True])
def test_decode_follows_engine_optimizer_steps(golden_dir, capturable):
    """The reference flow evaluates between optimizer steps (rnnt/train.py:164-201).  rnnt_amd.optim.AdamW writes parameters through raw
    pointers and a replayed HIP graph of the step does so without any host code at all: the persistent decode's model tables (conv1 tap
    tables, conv2's pack, the folded text_ln) must be those of the LIVE weights after every step — decode, step, decode again, replay,
    decode again, each time against the per-frame host loop and the numpy oracle on the stepped weights.  (Round-5 advice, high.)"""
    import copy

    import rnnt_amd
    c = load_decode_case(golden_dir, "decode_small_proj")
    model = build_model(c["spec"], c["pred_sd"], c["joint_sd"])
    mel = torch.from_numpy(np.ascontiguousarray(c["frames"].T))[None].cuda()
    lens = torch.tensor([mel.shape[-1]], device="cuda")
    assert model.greedy_decode(mel, lens, max_length=60, persistent=True) == c["tokens"][60]
    twin = copy.deepcopy(model)  # (the round-5 cache held a CUDA event in the module's __dict__: deepcopy / torch.save raised)
    assert twin.greedy_decode(mel, lens, max_length=60) == c["tokens"][60]
    params = list(model.predictor.parameters()) + list(model.joint.parameters())
    opt = rnnt_amd.optim.AdamW(params, lr=0.05, capturable=capturable)
    gen = torch.Generator(device="cuda").manual_seed(5)

    def fill_grads():
        for p in params:
            p.grad = torch.randn(p.shape, device="cuda", generator=gen) * (0.02 if p.dim() > 1 else 0.0)  # (vectors: keep the blank bias)

    def check(tag):
        sd_p = {k: v.detach().cpu().numpy() for k, v in model.predictor.state_dict().items()}
        sd_j = {k: v.detach().cpu().numpy() for k, v in model.joint.state_dict().items()}
        want, margins = decode_oracle.greedy_decode(c["frames"], sd_p, sd_j, max_length=60)
        got = {"persistent": model.greedy_decode(mel, lens, max_length=60, persistent=True),
               "many": model.greedy_decode_many([mel, mel], max_length=60)[1],
               "host": model.greedy_decode(mel, lens, max_length=60, scan_frames=0)}
        assert got["persistent"] == got["many"] == got["host"], (tag, got)
        if margins.min() > 1e-3:
            assert got["host"] == want, tag
        return got["host"]

    seen = [c["tokens"][60]]
    for i in range(2):
        fill_grads()
        opt.step()
        seen.append(check(f"eager step {i}"))
    if capturable:
        fill_grads()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            opt.step()
        for i in range(2):
            graph.replay()
            seen.append(check(f"replayed step {i}"))
    assert len({tuple(s) for s in seen}) >= 3, "the steps were meant to change what the model decodes"


@pytest.mark.parametrize("where", ["frame", "text"])
def test_persistent_decode_leaves_activations_beyond_its_factored_tanh_to_the_other_loop(golden_dir, where):
    """The persistent launch computes tanh(e + p) as 1 - 2 / (1 + exp 2e * exp 2p), exact while |e|, |p| <= 30.  Beyond that the two
    factors cannot simply be clamped (enc = 40, text = -35: the clamped product says tanh(0), the truth is tanh(5)): the launch reports
    state[7] = 10 (an audio frame) / 11 (a text vector) and decodes nothing; RNNTModel.greedy_decode / _many then take the kernel-per-layer
    loop (tanh of the sum) without a warning, direct callers get a RuntimeError.  Tokens equal the numpy oracle's.  (Round-5 advice.)"""
    import warnings

    import rnnt_amd
    c = load_decode_case(golden_dir, "decode_small")
    frames, pred_sd, joint_sd = c["frames"].copy(), dict(c["pred_sd"]), dict(c["joint_sd"])
    if where == "frame":
        frames[3::7, 5] = 40.0    # with text ~ -2 the truth stays tanh(38) = 1; clamped factors would still agree here ...
        frames[3::7, 6] = -41.0
        pred_sd["output_layer_norm.bias"] = pred_sd["output_layer_norm.bias"].copy()
        pred_sd["output_layer_norm.bias"][5] = -36.0   # ... but not here: 40 - 36 = 4 -> tanh(4), clamped: tanh(30 - 30) = 0
    else:
        w = pred_sd["output_layer_norm.weight"].copy()
        w[::11] *= 40.0  # every 11th text feature up to ~100 in magnitude
        pred_sd["output_layer_norm.weight"] = w
    want, margins = decode_oracle.greedy_decode(frames, pred_sd, joint_sd, max_length=60)
    assert margins.min() > 1e-3 and len(want) > 3
    model = build_model(c["spec"], pred_sd, joint_sd)
    mel = torch.from_numpy(np.ascontiguousarray(frames.T))[None].cuda()
    lens = torch.tensor([mel.shape[-1]], device="cuda")
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the range fallback is an exact path, not a degraded one: no warning
        assert model.greedy_decode(mel, lens, max_length=60, persistent=True) == want
        assert model.greedy_decode(mel, lens, max_length=60) == want
        assert model.greedy_decode_many([mel, mel], max_length=60) == [want, want]
    assert model.greedy_decode(mel, lens, max_length=60, scan_frames=0) == want
    tl = getattr(model.joint, "text_ln", None)
    state, _ = rnnt_amd.engine.greedy_decode_persistent(
        mel[0].T.contiguous(), model.predictor._params(), 1e-5, None if tl is None else tl.weight, None if tl is None else tl.bias,
        model.joint.joint_ln.weight, model.joint.joint_ln.bias, model.joint.blank_idx, 60)
    st = state.tolist()
    assert st[7] == (10 if where == "frame" else 11), st
    with pytest.raises(RuntimeError, match="beyond"):
        rnnt_amd.engine.check_decode_state(st)


def test_decode_with_different_layer_norm_eps(golden_dir):
    """input_layer_norm.eps != output_layer_norm.eps (the reference constructs both with the default, a checkpoint or a subclass need
    not): every device path takes the two values separately (round 5: one eps, or a fallback to the host loop) and decodes what the
    oracle decodes with the same two values."""
    from oracle import predictor_oracle as po
    c = load_decode_case(golden_dir, "decode_small_proj")
    model = build_model(c["spec"], c["pred_sd"], c["joint_sd"])
    model.predictor.input_layer_norm.eps = 5.0
    model.predictor.output_layer_norm.eps = 0.05
    assert model._device_loop_ok(torch.zeros(1, 1, device="cuda"))
    mel = torch.from_numpy(np.ascontiguousarray(c["frames"].T))[None].cuda()
    lens = torch.tensor([mel.shape[-1]], device="cuda")
    # the oracle with two eps values: predictor_oracle.forward takes one, so its two LayerNorms are evaluated here
    orig = po._ln
    calls = {"n": 0}

    def ln_two_eps(x, w, b, eps):
        calls["n"] += 1
        return orig(x, w, b, 5.0 if calls["n"] % 2 == 1 else 0.05)  # forward(): input LayerNorm first, output LayerNorm second

    po._ln = ln_two_eps
    try:
        want, margins = decode_oracle.greedy_decode(c["frames"], c["pred_sd"], c["joint_sd"], max_length=60)
    finally:
        po._ln = orig
    assert margins.min() > 1e-3 and len(want) > 3 and want != c["tokens"][60]  # (the eps values matter)
    all_paths(model, mel, lens, 60, want, "two eps")
