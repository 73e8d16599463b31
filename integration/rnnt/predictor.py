"""rnnt.predictor of the overlay: `ConvPredictor` is the engine-backed module (reference rnnt/predictor.py:189-229: same
constructor, attribute names and state-dict keys; forward + backward on rnnt_engine_conv_predictor_fwd / _bwd), so the hydra target
`rnnt.predictor.ConvPredictor` of rnnt/config/basic_sp_convjs*.yaml:20-25 (train.py:61) resolves to it with train.py and the yaml
unchanged — and `RNNTModel.greedy_decode` (train.py:170-201's evaluation) then takes the persistent device decode instead of the
per-token host loop.  Every other name of the module (`LSTMPredictor`, `_CustomLSTM`, ...) is the reference's own object, taken from
the reference's rnnt/predictor.py further down the package path."""
import importlib.util
import os
import sys

from rnnt_amd.predictor import ConvPredictor  # noqa: F401

import rnnt as _pkg


def _reference_module():
    here = os.path.dirname(os.path.abspath(__file__))
    for d in _pkg.__path__:  # pkgutil.extend_path: every rnnt/ directory on sys.path, this overlay first
        f = os.path.join(d, "predictor.py")
        if os.path.abspath(d) != here and os.path.isfile(f):
            spec = importlib.util.spec_from_file_location("rnnt._reference_predictor", f)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod  # (pickling / inspect of the reference's classes resolve their module by name)
            spec.loader.exec_module(mod)
            return mod
    return None


_ref = _reference_module()
ReferenceConvPredictor = getattr(_ref, "ConvPredictor", None)  # the reference's torch module, e.g. for CPU export scripts


def __getattr__(name):  # LSTMPredictor and whatever else the reference's module defines
    if _ref is not None and hasattr(_ref, name):
        return getattr(_ref, name)
    raise AttributeError(f"module 'rnnt.predictor' has no attribute {name!r}")
