"""CPU study for DESIGN.md §7 ("fp32 through 3 x bf16"): split each fp32 operand into three bf16
pieces (a = a1 + a2 + a3 exactly, 8 significant bits each), multiply with the six products
a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 and accumulate in fp32 — against plain fp32 products with
fp32 accumulation (what v_mfma_f32_32x32x2_f32 does) and the fp64 result.  Logits-like dot
products of length H = 512 (hidden in [-1,1], W ~ U(+-1/sqrt(H)))."""
import numpy as np


def bf16_trunc_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    x1 = bf16_trunc_round(x)
    r1 = (x - x1).astype(np.float32)
    x2 = bf16_trunc_round(r1)
    x3 = bf16_trunc_round((r1 - x2).astype(np.float32))
    return x1, x2, x3


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    M, H, V = 2048, 512, 1024
    A = np.tanh(rng.standard_normal((M, H)) * 1.4).astype(np.float32)
    W = rng.uniform(-1 / np.sqrt(H), 1 / np.sqrt(H), (V, H)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    fp32 = A @ W.T  # fp32 products, fp32 accumulation (BLAS order differs from the MFMA's; same error class)
    a1, a2, a3 = split3(A)
    w1, w2, w3 = split3(W)
    six = np.zeros((M, V), np.float32)
    for x, y in ((a1, w1), (a1, w2), (a2, w1), (a1, w3), (a2, w2), (a3, w1)):
        six += x @ y.T
    three = a1 @ w1.T + a1 @ w2.T + a2 @ w1.T
    one = a1 @ w1.T
    scale = np.abs(ref).max()
    for name, got in (("fp32 products          ", fp32), ("bf16 x 6 products      ", six),
                      ("bf16 x 3 products      ", three), ("bf16 x 1 (bf16 route)  ", one)):
        err = np.abs(got.astype(np.float64) - ref)
        print(f"{name} max abs err {err.max():.3e}  rms {np.sqrt((err ** 2).mean()):.3e}  (max |logit| {scale:.2f})")
