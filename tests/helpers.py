"""Shared helpers for the parity tests (seeded inputs, oracle calls, tolerances)."""
import numpy as np

from oracle import cpu_oracle

# Parity bar (north_star: "within 1e-4 fp32"): the loss is compared RELATIVELY (fp32 cannot
# hold 1e-4 absolute on a cost of ~1e4), gradients against the largest reference entry.
LOSS_RTOL = 1e-4
GRAD_RTOL = 1e-4
GRAD_ATOL = 1e-7


def make_inputs(B, T, U, H, V, seed, ragged=True):
    """Synthetic inputs as SURVEY.md §8d prescribes (unit-scale enc/pred, Linear-default W)."""
    rng = np.random.default_rng(seed)
    k = 1.0 / np.sqrt(H)
    d = dict(
        enc=rng.standard_normal((B, T, H)).astype(np.float32),
        pred=rng.standard_normal((B, U + 1, H)).astype(np.float32),
        W=rng.uniform(-k, k, (V, H)).astype(np.float32),
        bias=rng.uniform(-k, k, (V,)).astype(np.float32),
        targets=rng.integers(0, V - 1, (B, max(U, 0))).astype(np.int32),
    )
    if ragged and B > 1:
        ll = rng.integers(max(1, T // 2), T + 1, B)
        tl = rng.integers(U // 2, U + 1, B)
        ll[0], tl[0] = T, U
    else:
        ll, tl = np.full(B, T), np.full(B, U)
    d["logit_lens"] = ll.astype(np.int32)
    d["target_lens"] = tl.astype(np.int32)
    return d


def oracle_fused(d):
    return cpu_oracle.joint_loss_fwd_bwd(d["enc"], d["pred"], d["W"], d["bias"], d["targets"],
                                         d["logit_lens"], d["target_lens"], blank=-1,
                                         dtype=np.float64)


def assert_close_grad(name, got, ref, rtol=GRAD_RTOL, atol=GRAD_ATOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    err = np.abs(got - ref).max() if got.size else 0.0
    bound = rtol * (np.abs(ref).max() if ref.size else 0.0) + atol
    assert err <= bound, f"{name}: max abs err {err:.3e} > {bound:.3e}"


def assert_close_loss(name, got, ref, rtol=LOSS_RTOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert np.isfinite(got).all(), f"{name}: non-finite"
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() <= rtol, f"{name}: rel err {rel.max():.3e} > {rtol:.1e} (got {got}, ref {ref})"


def lgamma_paths_cost(T, U, lp_blank, lp_emit_sum):
    """Closed form when every lattice cell has the same log-probs: all C(T+U-1, U) alignments
    have probability exp(T*lp_blank + sum_u lp_emit[u])."""
    from math import lgamma
    log_paths = lgamma(T + U) - lgamma(U + 1) - lgamma(T)
    return -(log_paths + T * lp_blank + lp_emit_sum)


# ---- bf16 route (BASELINE config 3): the oracle with the SAME rounding points as the engine
# (include/rnnt_engine.h RNNT_DTYPE_BF16): tanh(enc+pred), W and the logits gradient are rounded
# to bf16 (nearest-even) before each matrix product; everything else is float64.
BF16_LOSS_RTOL = 1e-3       # vs the rounding-point oracle (fp32 accumulation, rare 1-ulp flips)
BF16_GRAD_RTOL = 1e-2
BF16_LOSS_RTOL_EXACT = 2e-2  # vs the unrounded float64 oracle: what bf16 operands cost
BF16_GRAD_RTOL_EXACT = 6e-2


def bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def oracle_fused_bf16(d):
    enc, pred, W, bias = d["enc"], d["pred"], d["W"], d["bias"]
    B, T, H = enc.shape
    U1, V = pred.shape[1], W.shape[0]
    hidden = bf16_round(np.tanh(enc[:, :, None, :].astype(np.float64) +
                                pred[:, None, :, :].astype(np.float64)).astype(np.float32)).astype(np.float64)
    Wb = bf16_round(W).astype(np.float64)
    logits = (hidden.reshape(-1, H) @ Wb.T + bias.astype(np.float64)).astype(np.float32)
    # the route stores its logits in fp16 and computes everything downstream from the stored values
    logits = logits.astype(np.float16).astype(np.float32).reshape(B, T, U1, V)
    costs, G = cpu_oracle.rnnt_loss(logits, d["targets"], d["logit_lens"], d["target_lens"], blank=-1,
                                    dtype=np.float64)
    Gb = bf16_round((G / B).astype(np.float32)).astype(np.float64).reshape(-1, V)
    dpre = (Gb @ Wb).reshape(B, T, U1, H) * (1.0 - hidden * hidden)
    return dict(loss=costs.mean(), costs=costs, grad_enc=dpre.sum(2), grad_pred=dpre.sum(1),
                grad_W=Gb.T @ hidden.reshape(-1, H), grad_bias=Gb.sum(0))


# ---- published transducer known-answer vectors (tests/golden/published_transducer_kat.json):
# third-party published unit-test data (warp-transducer / torchaudio), the nearest thing to a pin
# of torchaudio.functional.rnnt_loss (reference rnnt/model.py:35-41) this image allows.
def published_kat_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "published_transducer_kat.json")
    out = []
    for c in json.load(open(path))["cases"]:
        shape = tuple(c["logits_shape"])
        V = shape[-1]
        case = dict(name=c["name"], blank=c["blank"], V=V,
                    logits=np.asarray(c["logits"], dtype=np.float32).reshape(shape),
                    targets=np.asarray(c["targets"], dtype=np.int32),
                    logit_lens=np.asarray(c["logit_lengths"], dtype=np.int32),
                    target_lens=np.asarray(c["target_lengths"], dtype=np.int32),
                    costs=np.asarray(c["costs"], dtype=np.float64),
                    grads=None if c["grads"] is None else np.asarray(c["grads"], dtype=np.float64).reshape(shape),
                    grad_atol=c["grad_atol"])
        out.append(case)
        if c["blank"] == 0:
            # the same case in the reference's convention (blank = -1 = V-1, rnnt/model.py:39): the
            # loss is invariant under a relabelling of the vocabulary, here v -> v-1 (mod V)
            r = dict(case)
            r["name"] = c["name"] + "_rolled_to_blank_last"
            r["blank"] = -1
            r["logits"] = np.ascontiguousarray(np.roll(case["logits"], -1, axis=-1))
            r["targets"] = (case["targets"] - 1).astype(np.int32)
            r["grads"] = None if case["grads"] is None else np.roll(case["grads"], -1, axis=-1)
            out.append(r)
    return out


# ---- digests of large gradients (tests/golden/make_fullsize_digest.py): random +-1 projections, sampled entries, max |.|, 2-norm
def digest_of(g, seed, nproj=64, nsample=4096):
    """A small, seed-determined digest of an array that is too large to commit: `proj` = nproj sums of the flattened array against
    random +-1 vectors, `sample` = nsample entries at random positions (`sample_idx`), `amax`, `norm`.  float64 throughout."""
    x = np.ascontiguousarray(g, dtype=np.float64).ravel()
    rng = np.random.default_rng(seed + x.size)
    idx = rng.integers(0, x.size, nsample)
    proj = np.zeros(nproj)
    CH = 1 << 18
    for c0 in range(0, x.size, CH):  # a chunk of sign vectors at a time: 128 MB of temporaries
        xc = x[c0:c0 + CH]
        sgn = rng.integers(0, 2, (xc.size, nproj), dtype=np.int8).astype(np.float64) * 2.0 - 1.0
        proj += xc @ sgn
    return dict(proj=proj, sample=x[idx], sample_idx=idx, amax=np.array(np.abs(x).max()), norm=np.array(np.sqrt((x * x).sum())))


def assert_close_digest(name, got, dig, seed, rtol=GRAD_RTOL):
    """`got` against a digest made by digest_of(reference, seed): every sampled entry within rtol * amax, every projection within
    rtol * amax * sqrt(size) (independent per-entry errors of rtol * amax add up to that), the norm within rtol."""
    mine = digest_of(got, seed, nproj=len(dig["proj"]), nsample=len(dig["sample"]))
    assert np.array_equal(mine["sample_idx"], dig["sample_idx"]), name + ": digest positions differ (numpy Generator stream?)"
    amax = float(dig["amax"])
    n = np.asarray(got).size
    assert np.isfinite(np.asarray(got)).all(), name + ": non-finite values"
    es = np.abs(mine["sample"] - dig["sample"]).max()
    ep = np.abs(mine["proj"] - dig["proj"]).max()
    assert es <= rtol * amax + GRAD_ATOL, f"{name}: sampled entries off by {es:.3e} > {rtol * amax:.3e}"
    assert ep <= rtol * amax * np.sqrt(n) + GRAD_ATOL, f"{name}: projections off by {ep:.3e} > {rtol * amax * np.sqrt(n):.3e}"
    assert abs(float(mine["norm"]) - float(dig["norm"])) <= rtol * float(dig["norm"]) + GRAD_ATOL, name + ": 2-norm"
    return es / max(amax, 1e-300), ep / max(amax * np.sqrt(n), 1e-300)


# ---- greedy-decode cases (tests/golden/decode_*.npz; tests/golden/make_golden_decode.py) --------------------------------
# name: V (= num_symbols = num_classes), E, O, H, fa, ft (audio_ln / text_ln inputs, -1 = none), T frames, max_lengths,
#       w_scale (joint_ln.weight multiplier: spreads the logits so that top-2 gaps sit far above rounding), store (arrays in
#       the fixture; False: regenerated from the seed below and pinned by SHA-256 — the reference's widths are 16-30 MB of weights)
DECODE_CASES = {
    "decode_small": dict(V=32, E=48, O=64, H=64, fa=-1, ft=-1, T=75, max_lengths=(60, 9), w_scale=12.0, store=True),
    "decode_small_proj": dict(V=300, E=36, O=56, H=64, fa=40, ft=56, T=45, max_lengths=(60, 9), w_scale=20.0, store=True),
    "decode_cap": dict(V=32, E=48, O=64, H=64, fa=-1, ft=-1, T=14, max_lengths=(200, 37), w_scale=12.0, store=True,
                       want_cap=True),  # blank almost never wins: every frame runs into the 10-symbols-per-frame cap
    # frames = the output of the REFERENCE's own AudioEncoder (rnnt/jasper.py: prologue, a JasperBlock, epilogue; instance norm) on a seeded mel,
    # permuted as rnnt/model.py:93 does — what the decode sees behind a real encoder instead of N(0,1) frames (stored: they are data)
    "decode_ref_encoder": dict(V=32, E=48, O=64, H=64, fa=-1, ft=-1, T=75, max_lengths=(60, 9), w_scale=12.0, store=True, encoder="jasper"),
    "decode_wide_vocab": dict(V=4000, E=64, O=192, H=192, fa=-1, ft=-1, T=90, max_lengths=(60, 9), w_scale=30.0, store=False),
    "decode_ref_widths": dict(V=1024, E=512, O=1024, H=1024, fa=-1, ft=-1, T=200, max_lengths=(200, 25), w_scale=30.0,
                              store=False),  # config/basic_sp_convjs_fullcausal.yaml:20-25,60-65
    "decode_ref_widths_proj": dict(V=1024, E=512, O=1024, H=1024, fa=1024, ft=1024, T=120, max_lengths=(200, 25), w_scale=30.0,
                                   store=False),  # config/basic_sp_conv.yaml:65-70 (audio_ln + text_ln enabled)
}


def decode_case_arrays(spec, seed, blank_bias):
    """(frames [T, C], predictor state dict, joint state dict) of a decode case, float32, from numpy's PCG64 stream `seed`:
    torch-default-like scales (Linear / Conv1d: U(+-1/sqrt(fan_in)); embedding N(0,1); LayerNorm affine 1 + 0.2 N, 0.2 N)."""
    rng = np.random.default_rng(seed)
    V, E, O, H, fa, ft = (spec[k] for k in ("V", "E", "O", "H", "fa", "ft"))

    def uni(shape, fan_in):
        k = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-k, k, shape).astype(np.float32)

    def nrm(shape, scale=1.0, shift=0.0):
        return (shift + scale * rng.standard_normal(shape)).astype(np.float32)

    pred = {"embedding.weight": nrm((V, E)),
            "input_layer_norm.weight": nrm(E, 0.2, 1.0), "input_layer_norm.bias": nrm(E, 0.2),
            "conv1.conv.weight": uni((E, E, 3), 3 * E), "conv1.conv.bias": uni(E, 3 * E),
            "conv2.conv.weight": uni((E, E, 5), 5 * E), "conv2.conv.bias": uni(E, 5 * E),
            "linear.weight": uni((O, E), E), "linear.bias": uni(O, E),
            "output_layer_norm.weight": nrm(O, 0.2, 1.0), "output_layer_norm.bias": nrm(O, 0.2)}
    joint = {}
    if fa > 0:
        joint["audio_ln.weight"], joint["audio_ln.bias"] = uni((H, fa), fa), uni(H, fa)
    if ft > 0:
        assert ft == O
        joint["text_ln.weight"], joint["text_ln.bias"] = uni((H, ft), ft), uni(H, ft)
    joint["joint_ln.weight"] = (uni((V, H), H) * np.float32(spec["w_scale"])).astype(np.float32)
    joint["joint_ln.bias"] = uni(V, H)
    joint["joint_ln.bias"][V - 1] += np.float32(blank_bias)
    frames = nrm((spec["T"], fa if fa > 0 else H))
    return frames, pred, joint


def sha256_of_arrays(*dicts_or_arrays):
    import hashlib
    h = hashlib.sha256()
    for d in dicts_or_arrays:
        items = sorted(d.items()) if isinstance(d, dict) else [("", d)]
        for k, a in items:
            a = np.ascontiguousarray(a)
            h.update(k.encode() + str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    return h.hexdigest()


def load_decode_case(golden_dir, name):
    """-> dict(frames, pred_sd, joint_sd, tokens {max_length: list}, margins {max_length: array}, spec).  Cases stored as seed +
    SHA-256 regenerate their arrays and fail loudly if numpy's stream ever changes."""
    import os
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    spec = DECODE_CASES[name]
    seed, bias = int(z["seed"]), float(z["blank_bias"])
    if spec["store"]:
        frames = z["frames"]
        pred = {k[6:].replace("__", "."): z[k] for k in z.files if k.startswith("pred__")}
        joint = {k[7:].replace("__", "."): z[k] for k in z.files if k.startswith("joint__")}
    else:
        frames, pred, joint = decode_case_arrays(spec, seed, bias)
    got = sha256_of_arrays(frames, pred, joint)
    assert got == str(z["sha256"]), f"{name}: inputs regenerated from seed {seed} do not hash to the fixture's SHA-256"
    toks = {int(m): z[f"tokens_ml{int(m)}"].tolist() for m in z["max_lengths"]}
    margins = {int(m): z[f"margins_ml{int(m)}"] for m in z["max_lengths"]}
    return dict(frames=frames, pred_sd=pred, joint_sd=joint, tokens=toks, margins=margins, spec=spec)
