"""numpy restatement (float64) of the reference's greedy decode with a ConvPredictor.

TEST INFRASTRUCTURE ONLY (see oracle/rnnt_oracle.c): the checker of rnnt_engine_greedy_scan /
rnnt_engine_greedy_decode / rnnt_engine_greedy_decode_persistent and of RNNTModel.greedy_decode(_many).
Follows /root/reference/rnnt/model.py:90-128 (`_greedy_decode_conv`) statement by statement —
    tokens = [blank]; t = 0; emitted = 0; features = predictor(tokens)
    while t < T and len(tokens) < max_length:
        logits = joint.single_forward(audio[:, t], features[:, -1])      (rnnt/joint.py:44-55)
        tok = argmax(logits)                                              (first index on ties: torch.argmax)
        blank or emitted >= 10  ->  t += 1, emitted = 0
        else                    ->  tokens.append(tok); features = predictor(tokens) on the WHOLE history; emitted += 1
    return tokens[1:]
with the predictor of /root/reference/rnnt/predictor.py:211-229 (oracle/predictor_oracle.py, eval mode) and
single_forward = [audio_ln] / [text_ln] -> add -> tanh -> joint_ln.  Pinned by tests/golden/decode_*.npz: token lists
the reference's own `rnnt.predictor.ConvPredictor` + `rnnt.joint.JointNetwork` produced in the build container
(tests/golden/make_golden_decode.py; `rnnt.model` itself does not import there — torchaudio is absent — so the
generating script drives the two reference modules with the loop above).

Besides the tokens the oracle reports, per decision, the gap between the two largest logits: a parity test may only
demand identical tokens where that gap is far above fp32 re-association noise.
"""
import numpy as np

from . import predictor_oracle as po

MAX_PER_FRAME = 10  # rnnt/model.py:101 max_outputs_per_step


def _linear(x, w, b):
    return x @ np.asarray(w, dtype=np.float64).T + np.asarray(b, dtype=np.float64)


def single_forward(audio_row, text_row, joint_sd):
    """rnnt/joint.py:44-55 for one audio frame and one predictor frame (1-D float64 rows)."""
    a, t = audio_row, text_row
    if "audio_ln.weight" in joint_sd:
        a = _linear(a, joint_sd["audio_ln.weight"], joint_sd["audio_ln.bias"])
    if "text_ln.weight" in joint_sd:
        t = _linear(t, joint_sd["text_ln.weight"], joint_sd["text_ln.bias"])
    return _linear(np.tanh(a + t), joint_sd["joint_ln.weight"], joint_sd["joint_ln.bias"])


def greedy_decode(frames, pred_sd, joint_sd, max_length=200, eps=1e-5, window=None):
    """frames [T, C]: the encoder output of ONE utterance after the permute of rnnt/model.py:93 (time major).
    Returns (tokens without the leading blank, margins): margins[i] = top-1 minus top-2 logit of the i-th decision.
    `window=None` re-runs the predictor on the whole history like the reference (rnnt/model.py:119-121).  `window=7` feeds only the
    last 7 tokens once the history is longer: the module is causal (k=3 then k=5 convolutions behind left zero padding,
    rnnt/causalconv.py:28-29), so its last frame is a function of exactly those — the same numbers at a cost that does not grow
    with the history (tests/test_decode_oracle.py checks the two against each other)."""
    frames = np.asarray(frames, dtype=np.float64)
    pred_sd = {k: np.asarray(v, dtype=np.float64) for k, v in pred_sd.items()}
    joint_sd = {k: np.asarray(v, dtype=np.float64) for k, v in joint_sd.items()}
    blank = joint_sd["joint_ln.weight"].shape[0] - 1  # rnnt/joint.py:20
    # the projected audio frame does not depend on the loop: same arithmetic, once
    if "audio_ln.weight" in joint_sd:
        frames = _linear(frames, joint_sd["audio_ln.weight"], joint_sd["audio_ln.bias"])
    jsd = {k: v for k, v in joint_sd.items() if not k.startswith("audio_ln")}

    def predictor_last(tokens):
        if window is not None and len(tokens) > window:
            assert window >= 7
            tokens = tokens[-window:]
        out, _ = po.forward(np.asarray([tokens], dtype=np.int64), pred_sd, eps=eps)
        return out[0, -1]

    tokens = [blank]
    margins = []
    t, emitted = 0, 0
    feat = predictor_last(tokens)
    T = frames.shape[0]
    while t < T and len(tokens) < max_length:
        logits = single_forward(frames[t], feat, jsd)
        tok = int(np.argmax(logits))
        top2 = np.partition(logits, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
        if tok == blank or emitted >= MAX_PER_FRAME:
            t += 1
            emitted = 0
        else:
            tokens.append(tok)
            feat = predictor_last(tokens)
            emitted += 1
    return tokens[1:], np.asarray(margins)
