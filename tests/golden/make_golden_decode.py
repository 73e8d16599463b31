"""Generate tests/golden/decode_*.npz: greedy-decode token lists produced by the REFERENCE's own modules —
`rnnt.predictor.ConvPredictor` (reference rnnt/predictor.py:189-229) and `rnnt.joint.JointNetwork.single_forward`
(reference rnnt/joint.py:44-55) — driven by the loop of `RNNTModel._greedy_decode_conv` (reference rnnt/model.py:90-128).

Run in the build container only (the reference never travels to the GPU box):
    PYTHONPATH=/root/reference python tests/golden/make_golden_decode.py

`rnnt.model` itself cannot be imported here (`import torchaudio`, absent from the image; not stubbed), so the loop is driven
from this script: the two modules are the reference's, the control flow around them is the dozen statements of model.py:95-125
(tokens = [blank]; argmax of single_forward(audio[:, t], features[:, -1]); blank or 10 symbols on this frame -> next frame;
otherwise append and re-run the predictor on the whole history).  The encoder is out of the decode path's arithmetic: a case's
"encoder output" is a seeded (T, C) array.

Per case (tests/helpers.py DECODE_CASES) the script searches (seed, blank bias) until
  * the fp32 reference modules and float64 copies of them decode the same tokens,
  * every decision's top-2 logit gap (float64) exceeds MIN_MARGIN — fp32 re-association in another implementation cannot flip a token,
  * the utterance emits a useful number of tokens (and, for `want_cap`, runs into the 10-per-frame cap),
  * oracle/decode_oracle.py (numpy) reproduces the list — the oracle is pinned at generation time as well as in tests/.
Stored: seed, blank bias, SHA-256 of the inputs, token lists and margins per max_length; the input arrays themselves for the
small cases (the reference's widths are regenerated from the seed by tests/helpers.decode_case_arrays and checked against the hash).
Fixtures are data; no reference source is copied.
"""
import os
import sys

import numpy as np
import torch

from rnnt.joint import JointNetwork      # reference, via PYTHONPATH=/root/reference
from rnnt.predictor import ConvPredictor  # reference

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import decode_oracle  # noqa: E402
from tests.helpers import DECODE_CASES, decode_case_arrays, sha256_of_arrays  # noqa: E402

MIN_MARGIN = 1e-3


def build(spec, pred_sd, joint_sd, dtype):
    p = ConvPredictor(spec["V"], spec["O"], spec["E"], dropout=0.3)
    j = JointNetwork(spec["fa"], spec["ft"], spec["H"], spec["V"])
    p.load_state_dict({k: torch.from_numpy(v) for k, v in pred_sd.items()})
    j.load_state_dict({k: torch.from_numpy(v) for k, v in joint_sd.items()})
    return p.to(dtype).eval(), j.to(dtype).eval()


@torch.no_grad()
def reference_loop(predictor, joint, audio_features, max_length):
    """reference rnnt/model.py:95-125 around the reference's modules; audio_features (1, T, C) as after model.py:93."""
    tokens = [joint.blank_idx]
    margins = []
    cur_audio_time, max_audio_time = 0, audio_features.shape[1]
    cur_outputs_per_step, max_outputs_per_step = 0, 10
    predictor_features = predictor(torch.tensor([tokens], dtype=torch.int64))
    while cur_audio_time < max_audio_time and len(tokens) < max_length:
        joint_features = joint.single_forward(audio_features[:, cur_audio_time, :], predictor_features[:, -1, :])
        token_idx = joint_features.argmax(dim=-1).item()
        top2 = torch.topk(joint_features[0], 2).values
        margins.append(float(top2[0] - top2[1]))
        if token_idx == joint.blank_idx or cur_outputs_per_step >= max_outputs_per_step:
            cur_audio_time += 1
            cur_outputs_per_step = 0
        else:
            tokens.append(token_idx)
            predictor_features = predictor(torch.tensor([tokens], dtype=torch.int64))
            cur_outputs_per_step += 1
    return tokens[1:], np.asarray(margins)


def reference_encoder_frames(spec, seed):
    """(T, C) frames from the reference's AudioEncoder (reference rnnt/jasper.py) on a seeded mel, as rnnt/model.py:92-93 produces them."""
    from rnnt.jasper import AudioEncoder, JasperBlock
    torch.manual_seed(seed)
    enc = AudioEncoder(input_features=16, prologue_kernel_size=5, prologue_stride=2, prologue_dilation=1,
                       blocks=[JasperBlock(5, 32, 48, 0.0, 2, norm_type="instance")],
                       epilogue_features=64, epilogue_kernel_size=7, epilogue_stride=1, epilogue_dilation=2,
                       output_features=spec["H"], norm_type="instance").eval()
    with torch.no_grad():
        out = enc(torch.randn(1, 16, 2 * spec["T"]))  # (N, C, L)
    frames = out.permute(0, 2, 1)[0].contiguous().numpy().astype(np.float32)
    assert frames.shape == (spec["T"], spec["H"]), frames.shape
    return frames


def make(name, spec):
    ml0 = spec["max_lengths"][0]
    T = spec["T"]
    lo, hi = (max(spec["max_lengths"][1] + 3, T // 5), min(ml0 - 2, (3 * T) // 4)) if not spec.get("want_cap") else (10 * (T - 2), ml0 - 2)
    for seed in range(1000, 1040):
        frames, pred_sd, joint_sd = decode_case_arrays(spec, seed, 0.0)
        if spec.get("encoder"):
            frames = reference_encoder_frames(spec, seed)
        p32, j32 = build(spec, pred_sd, joint_sd, torch.float32)
        with torch.no_grad():
            probe = j32.single_forward(torch.from_numpy(frames[:1]), p32(torch.tensor([[spec["V"] - 1]]))[:, -1, :])
        sigma = float(probe.std())
        biases = [-40.0 * sigma] if spec.get("want_cap") else [round(float(b), 3) for b in sigma * np.linspace(5.0, -1.0, 25)]
        for bias in biases:
            frames, pred_sd, joint_sd = decode_case_arrays(spec, seed, bias)
            if spec.get("encoder"):
                frames = reference_encoder_frames(spec, seed)
            p32, j32 = build(spec, pred_sd, joint_sd, torch.float32)
            t32, _ = reference_loop(p32, j32, torch.from_numpy(frames)[None], ml0)
            if not lo <= len(t32) <= hi:
                continue
            p64, j64 = build(spec, pred_sd, joint_sd, torch.float64)
            out = {}
            ok = True
            for ml in spec["max_lengths"]:
                a, _ = reference_loop(p32, j32, torch.from_numpy(frames)[None], ml)
                b, m64 = reference_loop(p64, j64, torch.from_numpy(frames).double()[None], ml)
                c, mo = decode_oracle.greedy_decode(frames, pred_sd, joint_sd, max_length=ml, window=7 if spec["E"] > 64 else None)
                ok = ok and a == b and m64.min() > MIN_MARGIN
                if ok:
                    assert c == b, f"{name}: oracle/decode_oracle.py disagrees with the reference modules"
                    assert np.abs(mo - m64).max() < 1e-6 * max(1.0, np.abs(m64).max()), "oracle margins"
                    assert len(b) <= ml - 1
                    out[ml] = (b, m64)
            if not ok:
                continue
            if spec.get("want_cap"):
                assert len(out[ml0][0]) == 10 * T, "cap case: every frame should emit exactly 10 symbols"
                assert len(out[spec["max_lengths"][1]][0]) == spec["max_lengths"][1] - 1
            else:
                assert len(out[spec["max_lengths"][1]][0]) == spec["max_lengths"][1] - 1, "second max_length should cut the loop"
            save = dict(seed=np.int64(seed), blank_bias=np.float64(bias), sha256=np.str_(sha256_of_arrays(frames, pred_sd, joint_sd)),
                        max_lengths=np.asarray(spec["max_lengths"]), dims=np.asarray([spec[k] for k in ("V", "E", "O", "H", "fa", "ft", "T")]))
            for ml, (toks, m) in out.items():
                save[f"tokens_ml{ml}"] = np.asarray(toks, dtype=np.int64)
                save[f"margins_ml{ml}"] = m
            if spec["store"]:
                save["frames"] = frames
                save.update({"pred__" + k.replace(".", "__"): v for k, v in pred_sd.items()})
                save.update({"joint__" + k.replace(".", "__"): v for k, v in joint_sd.items()})
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **save)
            print(f"{name}: seed {seed} bias {bias:.3f} (sigma {sigma:.2f}) tokens {[len(v[0]) for v in out.values()]} "
                  f"min margin {min(v[1].min() for v in out.values()):.2e}", flush=True)
            return
    raise SystemExit(f"{name}: no (seed, bias) met the conditions")


if __name__ == "__main__":
    only = sys.argv[1:]
    for k, v in DECODE_CASES.items():
        if not only or k in only:
            make(k, v)
