"""N>1 path on CPU: world_size-2 gloo processes shard the batch, each computes its shard's
joint+loss gradients (the CPU oracle stands in for the engine call — no GPU here), and ONE
all-reduce of the flat [dW | db | loss] buffer must reproduce the full-batch result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rdzv(tmp_path):
    """file:// rendezvous inside the test's own temporary directory (one node): no TCP port to pick, so no bind / close / reuse window."""
    return f"file://{tmp_path}/rdzv"


def _worker(rank, world, rdzv, out):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from oracle import cpu_oracle
    from rnnt_amd.parallel import FlatGrad, shard_bounds
    from tests.helpers import make_inputs
    B, T, U, H, V = 5, 9, 4, 16, 8
    d = make_inputs(B, T, U, H, V, seed=3)
    lo, hi = shard_bounds(B, world, rank)
    r = cpu_oracle.joint_loss_fwd_bwd(d["enc"][lo:hi], d["pred"][lo:hi], d["W"], d["bias"],
                                      d["targets"][lo:hi], d["logit_lens"][lo:hi],
                                      d["target_lens"][lo:hi], dtype=np.float64)
    w = (hi - lo) / B  # the oracle returns shard-mean gradients; the engine gets grad_scale=1/B
    fg = FlatGrad(V, H, "cpu")
    fg.grad_W.copy_(torch.from_numpy(r["grad_W"] * w))
    fg.grad_bias.copy_(torch.from_numpy(r["grad_bias"] * w))
    fg.set_loss(torch.from_numpy(r["costs"]).float(), 1.0 / B)
    fg.all_reduce()
    if rank == 0:
        np.savez(out, flat=fg.flat.numpy(), lo=lo, hi=hi)
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_full_batch(tmp_path):
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _rdzv(tmp_path), out), nprocs=2, join=True)
    z = np.load(out)
    from oracle import cpu_oracle
    from tests.helpers import make_inputs
    B, T, U, H, V = 5, 9, 4, 16, 8
    d = make_inputs(B, T, U, H, V, seed=3)
    full = cpu_oracle.joint_loss_fwd_bwd(d["enc"], d["pred"], d["W"], d["bias"], d["targets"],
                                         d["logit_lens"], d["target_lens"], dtype=np.float64)
    flat = z["flat"]
    np.testing.assert_allclose(flat[:V * H].reshape(V, H), full["grad_W"], atol=1e-6)
    np.testing.assert_allclose(flat[V * H:V * H + V], full["grad_bias"], atol=1e-6)
    assert abs(flat[V * H + V] - full["loss"]) < 1e-4 * abs(full["loss"])
    assert (int(z["lo"]), int(z["hi"])) == (0, 3)


def test_shard_bounds_cover_batch():
    from rnnt_amd.parallel import shard_bounds
    for B in (1, 5, 32):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(B, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


# ---- BucketGradNorm's communication hook and RcclComm's id hand-round with two real ranks (gloo)
def _hook_worker(rank, world, rdzv, out):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from rnnt_amd.optim import BucketGradNorm
    from rnnt_amd.parallel import RcclComm
    torch.manual_seed(0)  # identical replicas
    def make():
        return torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.Tanh(), torch.nn.Linear(64, 48), torch.nn.Tanh(),
                                   torch.nn.Linear(48, 5))
    net, twin = make(), make()
    twin.load_state_dict(net.state_dict())
    # tiny buckets: several all-reduces per backward, so the total really is a sum over buckets
    ddp = torch.nn.parallel.DistributedDataParallel(net, bucket_cap_mb=0.004)
    ref = torch.nn.parallel.DistributedDataParallel(twin, bucket_cap_mb=0.004)  # DDP's default hook
    norm = BucketGradNorm(ddp, norm_fn=lambda t: torch.linalg.vector_norm(t.double()).float())
    g = torch.Generator().manual_seed(100 + rank)  # every rank its own shard
    totals, nbuckets = [], []
    for it in range(3):
        x = torch.randn(6, 37, generator=g); y = torch.randn(6, 5, generator=g)
        for m in (ddp, ref):
            m.zero_grad()
            ((m(x) - y) ** 2).mean().backward()
        nbuckets.append(len(norm._parts))
        total = norm.total()
        want = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1e9)  # the global norm of the averaged gradients
        totals.append((float(total), float(want)))
        for p, q in zip(ddp.parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad)  # the hook reduces exactly as the default one does
        assert norm._parts == []  # total() starts the next iteration
    # RcclComm's id hand-round: rank 0 draws 128 bytes with zero bytes inside, every rank must hold the same 128
    drawn = bytes([7, 0, 0, 9] + list(range(124)))
    calls = []
    uid = RcclComm.share_unique_id(rank, world, lambda: calls.append(1) or drawn)
    assert uid == drawn and len(calls) == (1 if rank == 0 else 0)
    # an exchange that truncates at the first zero byte (a C-string hand-over) must be refused, not joined with
    try:
        RcclComm.share_unique_id(rank, world, lambda: drawn, exchange=lambda b: drawn.split(b"\0")[0])
        bad = False
    except RuntimeError:
        bad = True
    assert bad
    if rank == 0:
        np.savez(out, totals=np.array(totals), nbuckets=np.array(nbuckets))
    dist.destroy_process_group()


def test_bucket_grad_norm_hook_and_unique_id_exchange_two_ranks(tmp_path):
    """world_size 2, gloo: sqrt(sum of the buckets' norms^2) of BucketGradNorm's hook == the global norm
    clip_grad_norm_ computes on a twin under DDP's default hook (gradients bit-identical), over several
    iterations and several buckets per backward; RcclComm.share_unique_id hands rank 0's 128 bytes round intact."""
    out = str(tmp_path / "hook.npz")
    mp.spawn(_hook_worker, args=(2, _rdzv(tmp_path), out), nprocs=2, join=True)
    z = np.load(out)
    assert (z["nbuckets"][1:] >= 2).all(), z["nbuckets"]  # (DDP lays its buckets out for good after the first backward)
    for got, want in z["totals"]:
        assert want > 0 and abs(got - want) <= 1e-6 * want, (got, want)
