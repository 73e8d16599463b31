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
  * e2e_refmodules.npz — the joint's INPUTS come from the reference's own producer modules
    (rnnt.jasper.AudioEncoder + JasperBlock, rnnt.predictor.ConvPredictor; the call sequence of
    reference rnnt/model.py:20-29 incl. the (N,C,L)->(N,L,C) permute and calc_output_lens) at
    reduced dims on seeded synthetic mels; stored: the encoder output in its native (N,C,L)
    layout, the predictor output, targets, lengths, and loss / gradients w.r.t. both feature
    tensors and the joint parameters (fp64 reference JointNetwork + torch alpha recursion).
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


def make_refmodules(name="e2e_refmodules", seed=31):
    from rnnt.jasper import AudioEncoder, JasperBlock
    from rnnt.predictor import ConvPredictor
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    H, V, B, U = 128, 128, 3, 9
    enc = AudioEncoder(input_features=16, prologue_kernel_size=5, prologue_stride=2, prologue_dilation=1,
                       blocks=[JasperBlock(5, 32, 48, 0.0, 2, norm_type="instance")],
                       epilogue_features=64, epilogue_kernel_size=7, epilogue_stride=1, epilogue_dilation=2,
                       output_features=H, norm_type="instance").eval()
    predm = ConvPredictor(num_symbols=V, output_dim=H, symbol_embedding_dim=40, dropout=0.0).eval()
    joint = JointNetwork(-1, -1, H, V)
    mel = torch.randn(B, 16, 70)
    mel_lens = torch.tensor([70, 55, 41])
    target_lens = torch.tensor([U, 6, 3])
    input_ids = torch.randint(0, V - 1, (B, U), generator=g)
    for b in range(B):
        input_ids[b, target_lens[b]:] = 0  # zero padding as dataset.py:76-80
    with torch.no_grad():
        blank = torch.full((B, 1), V - 1, dtype=input_ids.dtype)
        text = predm(torch.cat([blank, input_ids], dim=1))          # model.py:20-21
        enc_ncl = enc(mel)                                          # (N,C,L), model.py:27
        logit_lens = enc.calc_output_lens(mel_lens)                 # model.py:29
    assert int(logit_lens.max()) == enc_ncl.shape[2]
    jd = JointNetwork(-1, -1, H, V).double()
    jd.load_state_dict({k: v.double() for k, v in joint.state_dict().items()})
    a64 = enc_ncl.double().permute(0, 2, 1).contiguous().requires_grad_(True)  # model.py:28
    t64 = text.double().requires_grad_(True)
    logits = jd(a64, t64)
    loss, costs = rnnt_loss_torch(logits, input_ids, logit_lens, target_lens, blank=-1)
    loss.backward()
    out = {"enc_ncl": enc_ncl.numpy(), "text": text.numpy(),
           "targets": input_ids.numpy().astype(np.int32),
           "logit_lens": logit_lens.numpy().astype(np.int32),
           "target_lens": target_lens.numpy().astype(np.int32),
           "ctor": np.array([-1, -1, H, V], dtype=np.int64),
           "loss": np.float64(loss.item()), "costs": costs.detach().numpy(),
           "grad_audio": a64.grad.numpy(), "grad_text": t64.grad.numpy()}
    out.update(_np(joint.state_dict()))
    for k, p in jd.named_parameters():
        out["grad__" + k.replace(".", "__")] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


if __name__ == "__main__":
    make_refmodules()
    for k, v in JOINT_CASES.items():
        make_joint(k, *v)
    for k, v in E2E_CASES.items():
        make_e2e(k, *v)
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
