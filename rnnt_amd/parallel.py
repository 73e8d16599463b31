"""Batch sharding of the joint+loss path over the GPUs of one node.

Utterances are independent (every lattice, every row of the joint GEMM, grad_enc and grad_pred
are per-utterance); only dW (V,H), db (V) and the loss couple them.  Each rank therefore runs
the fused engine call on its contiguous batch shard with grad_scale = 1/B_global and ONE
all-reduce (RCCL over xGMI on GPUs; "nccl" is RCCL on ROCm, gloo in the CPU tests) sums the
flat [dW | db | loss] buffer.  Under the reference's own DDP wrapper (rnnt/train.py:68) the
joint parameters are ordinary nn.Parameters and DDP's bucketed all-reduce covers them; this
module is for the standalone benchmark and for code that wants the global loss.
"""
import ctypes
import os

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

    def all_reduce(self, group=None, comm=None):
        """SUM over ranks (shards already carry 1/B_global): `comm` (an RcclComm) sends the buffer through
        the engine's C entry rnnt_engine_allreduce, otherwise torch.distributed's backend does it."""
        if comm is not None:
            comm.all_reduce(self.flat)
            return self
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, group=group)
        return self


class _UniqueId(ctypes.Structure):  # rccl.h: ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
    _fields_ = [("internal", ctypes.c_ubyte * 128)]


class RcclComm:
    """An RCCL communicator of this process's own, one rank per GPU, for rnnt_engine_allreduce
    (include/rnnt_engine.h): rank 0 draws the unique id (ncclGetUniqueId), `exchange(bytes) -> bytes`
    hands it to the other ranks (default: torch.distributed.broadcast_object_list on the default
    group, any backend — it only carries 128 bytes), ncclCommInitRank joins.  The RCCL used is the one
    PyTorch-ROCm has loaded, the same copy the engine resolves.  Collectives run on the current stream.

    Verified with one rank on an MI355X (tests/test_train_step.py); no multi-GPU box was available to
    the build, so bench.py keeps torch.distributed ("nccl" = RCCL) as its default transport."""

    def __init__(self, rank: int, world: int, device, exchange=None):
        from . import engine
        self._engine = engine
        self.rank, self.world, self.device = rank, world, torch.device(device)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._rccl = ctypes.CDLL(path if os.path.exists(path) else "librccl.so.1")
        self._rccl.ncclGetErrorString.restype = ctypes.c_char_p
        uid = _UniqueId()

        def draw():
            self._ok(self._rccl.ncclGetUniqueId(ctypes.byref(uid)))
            return bytes(uid)  # all 128 bytes (not a C string)

        raw = RcclComm.share_unique_id(rank, world, draw, exchange)
        ctypes.memmove(ctypes.byref(uid), raw, 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._ok(self._rccl.ncclCommInitRank(ctypes.byref(self._comm), ctypes.c_int(world), uid, ctypes.c_int(rank)))

    @staticmethod
    def share_unique_id(rank: int, world: int, draw, exchange=None) -> bytes:
        """The id hand-round of the constructor, on its own (tests/test_dist_cpu.py runs it with two gloo ranks):
        rank 0 calls `draw()` for the 128 bytes of an ncclUniqueId, `exchange(bytes or None) -> bytes` carries them to
        every rank (default: torch.distributed.broadcast_object_list on the default group), and every rank returns
        the same 128 bytes — embedded zero bytes included; anything else raises."""
        raw = draw() if rank == 0 else None
        if world > 1:
            if exchange is None:
                import torch.distributed as dist

                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            raw = exchange(raw)
        if not isinstance(raw, (bytes, bytearray)) or len(raw) != 128:
            raise RuntimeError(f"RcclComm: the unique id must arrive as 128 bytes on every rank (rank {rank} got "
                               f"{type(raw).__name__} of length {len(raw) if hasattr(raw, '__len__') else '?'})")
        return bytes(raw)

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL: " + self._rccl.ncclGetErrorString(rc).decode())

    def all_reduce(self, flat: torch.Tensor):
        """In-place SUM of a contiguous fp32 device tensor over the ranks, on the current stream."""
        e = self._engine
        e._require_cuda(flat)
        e._require_dtype(torch.float32, flat=flat)
        e._require_contiguous(flat=flat)
        with torch.cuda.device(flat.device):
            e._check(e.lib().rnnt_engine_allreduce(e._p(flat), ctypes.c_size_t(flat.numel()), self._comm,
                                                   e._stream(flat.device)))
        return flat

    def destroy(self):
        if self._comm:
            self._rccl.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
