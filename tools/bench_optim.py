"""rnnt_engine_grad_norm / rnnt_engine_adamw_step (reference rnnt/train.py:136,164) at the
reference model's size: 47.8 M fp32 parameters in ~150 tensors.  Both kernels are HBM-bound: the
step reads p, g, m, v and writes p, m, v = 28 B per element, the norm reads g = 4 B per element; reported as
GB/s against the 8 TB/s peak, beside torch.optim.AdamW(foreach) + clip_grad_norm_ on the same tensors.
   python tools/bench_optim.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rnnt_amd

torch.manual_seed(0)
# the reference's parameter count split like its model: a few big matrices (joint 1024x1024, predictor, the
# Jasper blocks' 1x1 / depthwise convs) and many small vectors (norm scales, biases)
shapes = [(1024, 1024)] * 30 + [(1024, 512)] * 20 + [(512, 512, 3)] * 6 + [(1024,)] * 60 + [(512,)] * 30 + [(1024, 64)] * 4
n_el = sum(torch.Size(s).numel() for s in shapes)
ps = [torch.randn(*s, device="cuda").requires_grad_(True) for s in shapes]
for p in ps:
    p.grad = torch.randn_like(p)
hp = dict(lr=3e-4, betas=(0.95, 0.9999), eps=1e-8, weight_decay=0.01)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[n // 2]

opt = rnnt_amd.optim.AdamW(ps, max_grad_norm=1.0, **hp)
opt_plain = rnnt_amd.optim.AdamW(ps, **hp)
t_norm = timeit(lambda: rnnt_amd.optim.grad_norm([p.grad for p in ps]))
t_step = timeit(opt_plain.step)
t_fused = timeit(opt.step)
ref = torch.optim.AdamW(ps, foreach=True, **hp)
t_ref_step = timeit(ref.step)
t_ref_clip = timeit(lambda: torch.nn.utils.clip_grad_norm_(ps, 1.0))
out = {"parameters": n_el, "tensors": len(ps),
       "engine_grad_norm_ms": t_norm, "engine_grad_norm_GBs": 4 * n_el / t_norm / 1e6,
       "engine_adamw_step_ms": t_step, "engine_adamw_step_GBs": 28 * n_el / t_step / 1e6, "engine_adamw_step_frac_of_8TBs": 28 * n_el / t_step / 1e6 / 8000,
       "engine_clip_plus_step_ms": t_fused, "engine_clip_plus_step_GBs": 32 * n_el / t_fused / 1e6,
       "torch_foreach_adamw_step_ms": t_ref_step, "torch_clip_grad_norm_ms": t_ref_clip,
       "torch_clip_plus_step_ms": t_ref_step + t_ref_clip}
print(json.dumps(out))
