"""rnnt_amd — MI355X (gfx950) native RNN-T joint + transducer-loss engine.

Host-side mirror of the jakepoz/rnnt API for this path (rnnt.joint.JointNetwork,
rnnt.model.RNNTModel) over hand-written HIP kernels reached through a C ABI
(include/rnnt_engine.h -> rnnt_amd/csrc/librnnt_engine.so, bound with ctypes).
"""
from . import engine, optim  # noqa: F401
from .functional import joint_rnnt_loss, rnnt_loss, joint_logits, linear  # noqa: F401
from .joint import JointNetwork  # noqa: F401
from .predictor import ConvPredictor  # noqa: F401
from .model import RNNTModel  # noqa: F401

__all__ = ["engine", "optim", "joint_rnnt_loss", "rnnt_loss", "joint_logits", "JointNetwork", "RNNTModel", "ConvPredictor"]
