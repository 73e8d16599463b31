"""CPU: the numpy greedy-decode oracle (oracle/decode_oracle.py) against the token lists the REFERENCE's own ConvPredictor + JointNetwork
decoded in the build container (tests/golden/decode_*.npz, made by tests/golden/make_golden_decode.py; loop of rnnt/model.py:90-128)."""
import numpy as np
import pytest

from oracle import decode_oracle
from tests.helpers import DECODE_CASES, load_decode_case


@pytest.mark.parametrize("name", list(DECODE_CASES))
def test_decode_oracle_matches_reference_token_lists(golden_dir, name):
    c = load_decode_case(golden_dir, name)  # (regenerated inputs are checked against the fixture's SHA-256 in there)
    big = c["spec"]["E"] > 64
    for ml, want in c["tokens"].items():
        got, margins = decode_oracle.greedy_decode(c["frames"], c["pred_sd"], c["joint_sd"], max_length=ml, window=7 if big else None)
        assert got == want, (name, ml)
        assert len(got) <= ml - 1
        np.testing.assert_allclose(margins, c["margins"][ml], rtol=0, atol=1e-6 * max(1.0, np.abs(c["margins"][ml]).max()))
        assert margins.min() > 1e-3  # the fixtures' promise: no decision is a rounding-level tie


def test_decode_oracle_window_equals_whole_history(golden_dir):
    """The 7-token window is the same function as the reference's re-run on the whole history (causal convolutions, k = 3 then 5)."""
    for name in ("decode_small_proj", "decode_cap"):
        c = load_decode_case(golden_dir, name)
        ml = max(c["tokens"])
        full, m_full = decode_oracle.greedy_decode(c["frames"], c["pred_sd"], c["joint_sd"], max_length=ml)
        win, m_win = decode_oracle.greedy_decode(c["frames"], c["pred_sd"], c["joint_sd"], max_length=ml, window=7)
        assert full == win == c["tokens"][ml] and len(full) > 8
        np.testing.assert_allclose(m_full, m_win, rtol=0, atol=1e-9)


def test_decode_fixtures_cover_the_loop_edges(golden_dir):
    cap = load_decode_case(golden_dir, "decode_cap")
    T = cap["spec"]["T"]
    assert len(cap["tokens"][200]) == 10 * T            # every frame hit max_outputs_per_step (rnnt/model.py:101,113)
    assert len(cap["margins"][200]) == 11 * T           # 10 emissions + the forced advance per frame
    assert len(cap["tokens"][37]) == 36                 # len(tokens) < max_length cuts the loop mid-frame (rnnt/model.py:108)
    for name in DECODE_CASES:
        c = load_decode_case(golden_dir, name)
        short = min(c["tokens"])
        assert len(c["tokens"][short]) == short - 1 and c["tokens"][max(c["tokens"])][:short - 1] == c["tokens"][short]
