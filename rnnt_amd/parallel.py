"""Batch sharding of the joint+loss path over the GPUs of one node.

Utterances are independent (every lattice, every row of the joint GEMM, grad_enc and grad_pred
are per-utterance); only dW (V,H), db (V) and the loss couple them.  Each rank therefore runs
the fused engine call on its contiguous batch shard with grad_scale = 1/B_global and ONE
all-reduce (RCCL over xGMI on GPUs; "nccl" is RCCL on ROCm, gloo in the CPU tests) sums the
flat [dW | db | loss] buffer.  Under the reference's own DDP wrapper (rnnt/train.py:68) the
joint parameters are ordinary nn.Parameters and DDP's bucketed all-reduce covers them; this
module is for the standalone benchmark and for code that wants the global loss.
"""
import torch


def shard_bounds(batch: int, world: int, rank: int):
    """Contiguous shard [lo, hi) of `batch` utterances for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGrad:
    """One flat fp32 buffer [dW (V*H) | db (V) | loss, pad]; grad_W / grad_bias are views of it,
    so the engine writes straight into the all-reduce buffer (no packing copy)."""

    def __init__(self, V: int, H: int, device):
        self.V, self.H = V, H
        self.flat = torch.zeros(V * H + V + 4, dtype=torch.float32, device=device)
        self.grad_W = self.flat[:V * H].view(V, H)
        self.grad_bias = self.flat[V * H:V * H + V]

    def set_loss(self, costs: torch.Tensor, scale: float):
        self.flat[self.V * self.H + self.V] = costs.sum() * scale

    @property
    def loss(self) -> torch.Tensor:
        return self.flat[self.V * self.H + self.V]

    def all_reduce(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, group=group)  # SUM; shards already carry 1/B_global
        return self
