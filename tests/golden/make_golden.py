"""Generate tests/golden/*.npz by importing the REFERENCE's own rnnt.joint.JointNetwork.

Run in the build container only (the reference never travels to the GPU box):
    PYTHONPATH=/root/reference python tests/golden/make_golden.py

What is pinned
  * joint_*.npz — inputs, state_dict, logits (fp32 run and fp64 run of the reference
    module, reference rnnt/joint.py:25-39) and torch-autograd gradients of
    sum(logits * G) for a fixed random upstream G.
  * e2e_*.npz — same reference module (fp64) followed by the independent torch log-space
    alpha recursion (oracle/torch_check.py; argument meaning of reference
    rnnt/model.py:35-41) with ragged lengths; loss, per-utterance costs and
    autograd gradients w.r.t. audio, text and every joint parameter.
torchaudio is absent from this image, so the loss half is NOT produced by the reference's
third-party dependency: see oracle/rnnt_oracle.c ("parity unpinned" for torchaudio itself).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.torch_check import rnnt_loss_torch  # noqa: E402

from rnnt.joint import JointNetwork  # noqa: E402  (reference, via PYTHONPATH=/root/reference)

OUT = os.path.dirname(os.path.abspath(__file__))

JOINT_CASES = {
    # name: (Fa, Ft, H, V, B, T, U1, seed)
    "joint_tiny": (-1, -1, 16, 8, 2, 5, 3, 11),
    "joint_mid": (-1, -1, 64, 32, 3, 17, 9, 12),
    "joint_proj": (24, 20, 32, 16, 2, 7, 4, 13),
    "joint_v1024": (-1, -1, 128, 1024, 2, 10, 6, 14),
}

E2E_CASES = {
    # name: (Fa, Ft, H, V, B, T, U, seed)
    "e2e_tiny": (-1, -1, 16, 8, 2, 5, 3, 21),
    "e2e_mid": (-1, -1, 64, 32, 3, 17, 8, 22),
    "e2e_proj": (24, 20, 32, 16, 2, 9, 4, 23),
    "e2e_v1024": (-1, -1, 128, 1024, 2, 12, 5, 24),
}


def _np(sd):
    return {"sd__" + k.replace(".", "__"): v.detach().cpu().numpy() for k, v in sd.items()}


def make_joint(name, Fa, Ft, H, V, B, T, U1, seed):
    torch.manual_seed(seed)
    m = JointNetwork(Fa, Ft, H, V)
    audio = torch.randn(B, T, Fa if Fa > 0 else H)
    text = torch.randn(B, U1, Ft if Ft > 0 else H)
    G = torch.randn(B, T, U1, V)
    out = {"audio": audio.numpy(), "text": text.numpy(), "G": G.numpy(),
           "ctor": np.array([Fa, Ft, H, V], dtype=np.int64)}
    out.update(_np(m.state_dict()))
    with torch.no_grad():
        out["logits_f32"] = m(audio, text).numpy()
    md = JointNetwork(Fa, Ft, H, V).double()
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    a64 = audio.double().requires_grad_(True)
    t64 = text.double().requires_grad_(True)
    logits = md(a64, t64)
    (logits * G.double()).sum().backward()
    out["logits_f64"] = logits.detach().numpy()
    out["grad_audio"] = a64.grad.numpy()
    out["grad_text"] = t64.grad.numpy()
    for k, p in md.named_parameters():
        out["grad__" + k.replace(".", "__")] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def make_e2e(name, Fa, Ft, H, V, B, T, U, seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    m = JointNetwork(Fa, Ft, H, V)
    U1 = U + 1
    audio = torch.randn(B, T, Fa if Fa > 0 else H)
    text = torch.randn(B, U1, Ft if Ft > 0 else H)
    targets = torch.randint(0, V - 1, (B, U), generator=g)
    # utterance 0 is full length (torchaudio requires max(len) == T / U); the rest ragged
    logit_lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g)
    target_lens = torch.randint(max(0, U // 2), U + 1, (B,), generator=g)
    logit_lens[0] = T
    target_lens[0] = U
    md = JointNetwork(Fa, Ft, H, V).double()
    md.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    a64 = audio.double().requires_grad_(True)
    t64 = text.double().requires_grad_(True)
    logits = md(a64, t64)
    loss, costs = rnnt_loss_torch(logits, targets, logit_lens, target_lens, blank=-1)
    loss.backward()
    out = {"audio": audio.numpy(), "text": text.numpy(),
           "targets": targets.numpy().astype(np.int32),
           "logit_lens": logit_lens.numpy().astype(np.int32),
           "target_lens": target_lens.numpy().astype(np.int32),
           "ctor": np.array([Fa, Ft, H, V], dtype=np.int64),
           "loss": np.float64(loss.item()), "costs": costs.detach().numpy(),
           "grad_audio": a64.grad.numpy(), "grad_text": t64.grad.numpy()}
    out.update(_np(m.state_dict()))
    for k, p in md.named_parameters():
        out["grad__" + k.replace(".", "__")] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


if __name__ == "__main__":
    for k, v in JOINT_CASES.items():
        make_joint(k, *v)
    for k, v in E2E_CASES.items():
        make_e2e(k, *v)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
