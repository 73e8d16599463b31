"""-m gpu: rnnt_amd.optim (C ABI rnnt_engine_grad_norm / rnnt_engine_adamw_step) against
torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW — the statements at reference
rnnt/train.py:136,164 with the hyper-parameters of rnnt/config/basic_sp_convjs_fullcausal.yaml:80-87."""
import pytest
import torch

pytestmark = pytest.mark.gpu

HP = dict(lr=3e-4, betas=(0.95, 0.9999), eps=1e-8, weight_decay=0.01)  # the reference's optimizer block
# > 40 tensors (two launches), sizes around the 16384-element chunk and the float4 tail
SIZES = [(1024, 512), (1024,), (3,), (16384,), (16385,), (7, 9, 5), (1,), (40000,)] + [(33, 17)] * 45


def _params(seed, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g).to(device).requires_grad_(True) for s in SIZES]


def _set_grads(ps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = (torch.randn(*p.shape, generator=g) * scale).to(p.device)


def test_clip_grad_norm_matches_torch():
    import rnnt_amd
    a, b = _params(1), _params(1)
    for scale, max_norm in ((1.0, 10.0), (1e-3, 10.0), (50.0, 0.5)):
        _set_grads(a, 7, scale)
        _set_grads(b, 7, scale)
        n_ref = torch.nn.utils.clip_grad_norm_(a, max_norm)
        n_amd = rnnt_amd.optim.clip_grad_norm_(b, max_norm)
        assert abs(n_amd.item() - n_ref.item()) <= 1e-6 * n_ref.item()
        for x, y in zip(a, b):
            assert torch.allclose(x.grad, y.grad, rtol=2e-6, atol=0)
    # the reference's exhausted generator (rnnt/train.py:95,104,136): nothing clipped, norm 0
    gen = (p for p in b)
    list(gen)
    before = [p.grad.clone() for p in b]
    assert float(rnnt_amd.optim.clip_grad_norm_(gen, 1e-9)) == 0.0
    assert all(torch.equal(x, p.grad) for x, p in zip(before, b))
    # a misaligned view (4-byte aligned only) goes through the scalar path
    base = torch.randn(1001, device="cuda")
    v = base[1:].requires_grad_(True)
    v.grad = torch.randn(1001, device="cuda")[1:]
    ref = v.grad.norm().item()
    assert abs(rnnt_amd.optim.clip_grad_norm_([v], 1e9).item() - ref) <= 1e-6 * ref


@pytest.mark.parametrize("clip", [None, 2.0])
def test_adamw_matches_torch_over_steps(clip):
    import rnnt_amd
    a, b = _params(3), _params(3)
    ref = torch.optim.AdamW(a, foreach=False, fused=False, **HP)
    opt = rnnt_amd.optim.AdamW(b, max_grad_norm=clip, **HP)
    for step in range(12):
        _set_grads(a, 100 + step, scale=3.0 if step % 3 == 0 else 0.2)
        _set_grads(b, 100 + step, scale=3.0 if step % 3 == 0 else 0.2)
        if step == 5:  # a parameter without a gradient this step keeps its state and step count
            a[3].grad = None
            b[3].grad = None
        if clip is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(a, clip)
        ref.step()
        opt.step()
        if clip is not None:
            assert abs(opt.last_grad_norm.item() - n_ref.item()) <= 1e-6 * n_ref.item()
            for x, y in zip(a, b):  # clip_grad_norm_ is in place: so is the fused clip
                if x.grad is not None:
                    assert torch.allclose(x.grad, y.grad, rtol=2e-6, atol=0)
        ref.zero_grad()
        opt.zero_grad()
    for i, (x, y) in enumerate(zip(a, b)):
        err = (x.detach() - y.detach()).abs().max().item()
        assert err <= 2e-6 * max(1.0, x.detach().abs().max().item()), (i, err)
        sx, sy = ref.state[x], opt.state[y]
        assert int(sx["step"]) == sy["step"]
        assert torch.allclose(sx["exp_avg"], sy["exp_avg"], rtol=1e-5, atol=1e-9)
        assert torch.allclose(sx["exp_avg_sq"], sy["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_adamw_rejects_what_it_cannot_do():
    import rnnt_amd
    p = torch.randn(8, device="cuda", dtype=torch.float64, requires_grad=True)
    p.grad = torch.randn_like(p)
    with pytest.raises(RuntimeError, match="float32"):
        rnnt_amd.optim.AdamW([p]).step()
    q = torch.randn(8, requires_grad=True)
    q.grad = torch.randn(8)
    with pytest.raises(RuntimeError, match="HIP device"):
        rnnt_amd.optim.AdamW([q]).step()
    with pytest.raises(ValueError):
        rnnt_amd.optim.AdamW([torch.zeros(1, device="cuda", requires_grad=True)], betas=(1.0, 0.9))


def test_optimizer_checkpoint_moves_between_torch_and_engine(tmp_path):
    """The reference checkpoints `optimizer.state_dict()` (rnnt/util.py:7-12, train.py:204-210).  A
    state dict written by torch.optim.AdamW loads into the engine's AdamW (torch keeps `step` as a
    tensor) and the other way round; training continues on the same trajectory."""
    import rnnt_amd
    a, b = _params(5), _params(5)
    ref = torch.optim.AdamW(a, foreach=False, fused=False, **HP)
    opt = rnnt_amd.optim.AdamW(b, **HP)
    for step in range(3):  # ref trains, then hands over through a file
        _set_grads(a, 200 + step)
        ref.step()
    torch.save({"optimizer_state_dict": ref.state_dict(), "completed_steps": 3}, tmp_path / "ck.pt")
    with torch.no_grad():
        for x, y in zip(a, b):
            y.copy_(x)
    opt.load_state_dict(torch.load(tmp_path / "ck.pt")["optimizer_state_dict"])
    for step in range(3, 6):
        _set_grads(a, 200 + step)
        _set_grads(b, 200 + step)
        ref.step()
        opt.step()
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=2e-6, atol=1e-7)
    # and back: the engine's state dict into a fresh torch optimizer
    c = _params(5)
    with torch.no_grad():
        for y, z in zip(b, c):
            z.copy_(y)
    back = torch.optim.AdamW(c, foreach=False, fused=False, **HP)
    torch.save(opt.state_dict(), tmp_path / "ck2.pt")  # through a file: load_state_dict does not copy same-device tensors
    back.load_state_dict(torch.load(tmp_path / "ck2.pt"))
    for step in range(6, 8):
        for ps in (a, b, c):
            _set_grads(ps, 200 + step)
        ref.step(); opt.step(); back.step()
    for x, y, z in zip(a, b, c):
        assert torch.allclose(x, y, rtol=3e-6, atol=1e-7) and torch.allclose(x, z, rtol=3e-6, atol=1e-7)


def test_capturable_adamw_matches_torch_with_a_scheduler():
    """capturable=True: step count and learning rate on the device (rnnt_engine_adamw_step_dev); an LR
    scheduler fills the lr tensor in place (reference train.py:164-166: optimizer.step(); lr_scheduler.step())."""
    import rnnt_amd
    a, b = _params(9), _params(9)
    ref = torch.optim.AdamW(a, foreach=False, fused=False, **HP)
    opt = rnnt_amd.optim.AdamW(b, capturable=True, max_grad_norm=1.5, **HP)
    lam = lambda s: min(1.0, (s + 1) / 4) * (0.9 ** s)
    sched_ref = torch.optim.lr_scheduler.LambdaLR(ref, lam)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    for step in range(10):
        _set_grads(a, 300 + step, scale=2.0)
        _set_grads(b, 300 + step, scale=2.0)
        torch.nn.utils.clip_grad_norm_(a, 1.5)
        ref.step(); sched_ref.step()
        opt.step(); sched.step()
    assert torch.is_tensor(opt.param_groups[0]["lr"]) and opt.param_groups[0]["lr"].is_cuda
    assert abs(float(opt.param_groups[0]["lr"]) - ref.param_groups[0]["lr"]) <= 1e-9
    assert int(opt.state[b[0]]["step"]) == 10
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=3e-6, atol=1e-7)
    b[3].grad = None
    with pytest.raises(RuntimeError, match="every parameter"):
        opt.step()
