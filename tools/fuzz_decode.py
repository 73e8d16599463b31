"""Randomised comparison of the device greedy decodes of one model — the persistent launch (rnnt_engine_greedy_decode_persistent), the
kernel-per-layer device loop (rnnt_engine_greedy_decode) and, every few cases, the per-frame host loop — with the NUMPY ORACLE
(oracle/decode_oracle.py: the reference's loop rnnt/model.py:90-128 around its ConvPredictor and JointNetwork.single_forward, pinned by
token lists the reference's own modules decoded, tests/golden/decode_*.npz) as the checker (round 6: until then the paths were only
compared with each other).  A case whose smallest top-2 logit gap is below 1e-3 in the oracle proves nothing either way about a token
and is counted as "tie" when the lists differ.  Cases with large activations (|frame| or |text| > 30: the persistent loop's factored
tanh hands them to the other loop) are drawn too:
random widths (E, O, H within what the persistent loop takes), vocabularies (a few entries to several blocks per workgroup), utterance lengths,
blank biases (from "blank almost never" — the 10-per-frame cap and max_length cut in — to "blank almost always"), with and without
audio_ln / text_ln.  Prints one line per case; exit code 1 on any difference.   argv: cases (40), seed (0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rnnt_amd
from oracle import decode_oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


class Enc(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c = torch.nn.Conv1d(10, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.c(x)

    def calc_output_lens(self, lens):
        return (lens + 1) // 2


bad = 0
for case in range(n_cases):
    torch.manual_seed(1000 + case)
    H = int(rng.choice([64, 128, 192, 256, 320, 512, 1024]))
    E = int(rng.choice([16, 32, 48, 100, 256, 512, 768]))
    V = int(rng.choice([20, 32, 100, 256, 1000, 1024, 2500, 5000]))  # (the device loops need V % 4 == 0)
    proj = bool(rng.integers(0, 2))
    O = int(rng.choice([64, 128, 256, 1024])) if proj else H
    fa = int(rng.choice([24, 40])) if proj else -1
    nfr = int(rng.integers(20, 400))
    max_length = int(rng.choice([5, 30, 100]))
    bias = float(rng.choice([-3.0, 0.0, 0.5, 1.0, 1.5, 2.5]))
    pred = rnnt_amd.ConvPredictor(V, O, E, 0.3)
    model = rnnt_amd.RNNTModel(pred, Enc(fa if proj else H), rnnt_amd.JointNetwork(fa, O if proj else -1, H, V)).cuda().eval()
    with torch.no_grad():
        model.joint.joint_ln.bias[V - 1] += bias
    big = bool(rng.integers(0, 5) == 0)  # large activations: encoder output x 12 (|frame| well beyond 30)
    mel = torch.randn(1, 10, nfr, device="cuda") * (12.0 if big else 1.0)
    lens = torch.tensor([nfr], device="cuda")
    ok = rnnt_amd.engine.greedy_decode_persistent_supported((nfr + 1) // 2, V, E, O, H, V, proj)
    chain = model.greedy_decode(mel, lens, max_length=max_length, scan_frames=int(rng.choice([7, 16, 32, 128])), device_loop=True, persistent=False)
    pers = model.greedy_decode(mel, lens, max_length=max_length, persistent=True) if ok else None
    ref = model.greedy_decode(mel, lens, max_length=max_length, scan_frames=0) if case % 4 == 0 else None
    with torch.no_grad():
        frames = model.encoder(mel).permute(0, 2, 1)[0].double().cpu().numpy()
    sd_p = {k: v.detach().cpu().numpy() for k, v in model.predictor.state_dict().items()}
    sd_j = {k: v.detach().cpu().numpy() for k, v in model.joint.state_dict().items()}
    want, margins = decode_oracle.greedy_decode(frames, sd_p, sd_j, max_length=max_length, window=7)
    same = (pers is None or pers == chain) and (ref is None or ref == chain) and chain == want
    tie = (not same) and margins.min() < 1e-3
    bad += (not same) and not tie
    print(f"case {case:3d}: H={H:4d} E={E:3d} O={O:4d} V={V:4d} proj={int(proj)} frames={(nfr + 1) // 2:3d} max_length={max_length:3d} bias={bias:4.1f} "
          f"persistent={'yes' if ok else 'no '} big={int(big)} tokens={len(chain):3d} min gap {margins.min():.1e} {'equal' if same else 'tie' if tie else 'DIFFERENT'}"
          + ("" if same else f"\n   oracle {want}\n   chain  {chain}\n   pers   {pers}\n   host   {ref}"), flush=True)
print(f"{n_cases} cases, {bad} with token lists that differ from the oracle's beyond a rounding-level tie")
sys.exit(1 if bad else 0)
