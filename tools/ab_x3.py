"""A/B of the bf16x3 route's kernel forms in ONE process (interleaved rounds: cdna_hip_programming.md §5.4 rule 24):
the forward's two-waves-per-SIMD forms (RNNT_VARIANT_X3_FWD_2WG / _8W) against the default kernels, stage by stage with HIP
events, after checking on a small ragged batch that every form agrees with the fp64 oracle.
    python3 tools/ab_x3.py [config] [rounds] [reps]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from rnnt_amd import engine
import bench

ST = {"fwd": 1 << 1, "dh": 1 << 4, "dw": 1 << 6}


def check(variant):
    from tests.helpers import make_inputs, oracle_fused, assert_close_grad, assert_close_loss
    for shape in ((3, 23, 19, 256, 384), (2, 40, 33, 512, 1024), (2, 13, 20, 1024, 256), (3, 21, 9, 640, 128)):
        B, T, U, H, V = shape
        d = make_inputs(B, T, U, H, V, seed=sum(shape))
        g = {k: torch.from_numpy(v).cuda() for k, v in d.items()}
        outs = engine.joint_loss_fwd_bwd(g["enc"], g["pred"], g["W"], g["bias"], g["targets"], g["logit_lens"], g["target_lens"],
                                         V - 1, 1.0 / B, dtype="bf16x3", variant=variant)
        torch.cuda.synchronize()
        ref = oracle_fused(d)
        assert_close_loss("costs", outs[0].cpu().numpy(), ref["costs"])
        for o, k in zip(outs[1:], ("grad_enc", "grad_pred", "grad_W", "grad_bias")):
            assert_close_grad(k, o.cpu().numpy(), ref[k])
    print("variant", variant, "agrees with the oracle", flush=True)


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda", 0)
    V1 = engine.VARIANT_X3_FWD_2WG
    check(0)
    check(V1)
    check(engine.VARIANT_X3_FWD_8W)
    B, T, U, H, V = bench.CONFIGS[cfg]
    enc, pred, W, bias, targets, ll, tl = bench.synth(B, T, U, H, V, 1234, dev)
    outs = engine.alloc_fused_outputs(enc, pred, W)

    def run(mask, variant):
        engine.joint_loss_fwd_bwd(enc, pred, W, bias, targets, ll, tl, V - 1, 1.0 / B, outs=outs, dtype="bf16x3",
                                  stage_mask=mask, variant=variant)

    def timed(mask, variant):
        run(mask, variant)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(mask, variant)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps

    run(engine.STAGES_ALL, 0)
    res = {}
    for r in range(rounds):
        for name, mask in list(ST.items()) + [("step", engine.STAGES_ALL)]:
            for vn, var in (("default", 0), ("2wg", V1), ("8w", engine.VARIANT_X3_FWD_8W)):
                if name in ("dw", "dh") and vn != "default":
                    continue
                res.setdefault((name, vn), []).append(timed(mask, var))
    for (name, vn), v in res.items():
        v = sorted(v)
        print(f"{cfg} {name:5s} {vn:4s} median {v[len(v) // 2]:8.3f} ms  min {v[0]:8.3f}  max {v[-1]:8.3f}", flush=True)
