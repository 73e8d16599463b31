"""Greedy decode of one synthetic utterance (T=1000 frames, H=512, V=1024): the reference's
per-frame loop (rnnt/model.py:108-125: single_forward + argmax().item() per frame) against the
device-side scan (rnnt_engine_greedy_scan).  Stand-in encoder / predictor modules: only the
joint + argmax + control flow is what is being timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rnnt_amd

torch.manual_seed(0)
T, H, V = 1000, 512, 1024


class Enc(torch.nn.Module):
    def forward(self, x):
        return x  # (1, H, T) already

    def calc_output_lens(self, lens):
        return lens


pred = torch.nn.Embedding(V, H)
model = rnnt_amd.RNNTModel(pred, Enc(), rnnt_amd.JointNetwork(-1, -1, H, V)).cuda()
with torch.no_grad():
    # blank wins ~4 of 5 frames (BASELINE's T/U is 5 frames per token)
    model.joint.joint_ln.bias[V - 1] += float(sys.argv[1]) if len(sys.argv) > 1 else 1.6
mel = torch.randn(1, H, T, device="cuda")
lens = torch.tensor([T], device="cuda")
for scan in (0, 16, 32, 64, 128):
    model.greedy_decode(mel, lens, max_length=400, scan_frames=scan)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = model.greedy_decode(mel, lens, max_length=400, scan_frames=scan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"scan_frames={scan:3d}: {dt * 1e3:8.1f} ms, {len(toks)} tokens, {dt / T * 1e6:7.1f} us per audio frame", flush=True)
