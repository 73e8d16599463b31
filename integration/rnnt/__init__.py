"""Overlay of the reference's `rnnt` package (the no-edit route).

Put this directory's parent BEFORE the reference checkout on PYTHONPATH:

    PYTHONPATH=<this repo>/integration:<this repo>:<reference checkout> python -m rnnt.train ...

`rnnt.joint`, `rnnt.model` and `rnnt.predictor.ConvPredictor` then resolve to the engine-backed classes below (same names,
constructor arguments, state-dict keys: rnnt/joint.py:5-55, rnnt/model.py:7-139, rnnt/predictor.py:189-229), so the
reference's `from rnnt.model import RNNTModel` (rnnt/train.py:19) and its hydra targets
`rnnt.joint.JointNetwork` / `rnnt.predictor.ConvPredictor` (rnnt/config/*.yaml, train.py:61-63) pick up the HIP path, while every
other submodule (train, dataset, featurizer, jasper, causalconv, util, ...) and every other name of `rnnt.predictor`
(`LSTMPredictor`, ...) is still the reference's own: the package path is extended over all `rnnt/` directories on sys.path.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
