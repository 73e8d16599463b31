"""rnnt.model of the overlay: RNNTModel whose forward is the fused joint + transducer loss
(reference rnnt/model.py:7-139; no torchaudio import)."""
from rnnt_amd.model import RNNTModel  # noqa: F401
