"""rnnt.joint of the overlay: the engine-backed JointNetwork (reference rnnt/joint.py:5-55)."""
from rnnt_amd.joint import JointNetwork  # noqa: F401
