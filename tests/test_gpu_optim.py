"""bg_adam_step / branchgan.optim.ArenaAdam against torch.optim.Adam (the optimizer the reference builds,
Training/Optimizers.py:6-16), the single-tensor implementation whose order of operations the kernel follows.

Floating point: parameters and moments agree to rtol 2e-6 / atol 1e-7 per step chain (contraction of a*b+c into one FMA is the
only difference; moments near zero are cancellations and are held to 2 ulp of the tensor's scale); the bf16 weight copy is BIT-exact against rounding the kernel's own fp32 result; gradients come back zero."""
import pytest
import torch

from branchgan import _capi as C

pytestmark = pytest.mark.gpu


def _adam_call(p, g, m, v, lp, n, lr, b1, b2, eps, wd, step, scale, zero):
    C.check(C.lib().bg_adam_step(C.ptr(p), C.ptr(g), C.ptr(m), C.ptr(v), C.ptr(lp), C.BG_BF16 if lp is not None else 0, n,
                                 lr, b1, b2, eps, wd, step, scale, zero, C.stream_ptr()), "bg_adam_step")


@pytest.mark.parametrize("n", [1, 3, 8, 1003, (1 << 20) + 5, 3 * (1 << 22)])
@pytest.mark.parametrize("wd,scale", [(0.0, 1.0), (0.01, 0.5)])
def test_adam_step_matches_torch(n, wd, scale):
    torch.manual_seed(n % 1000)
    dev = "cuda"
    lr, b1, b2, eps = 3e-3, 0.5, 0.999, 1e-8
    p = torch.randn(n, device=dev)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd, foreach=False, fused=False)
    m, v, lp = torch.zeros_like(p), torch.zeros_like(p), torch.empty(n, dtype=torch.bfloat16, device=dev)
    for step in range(1, 6):
        g = torch.randn(n, device=dev) * (10.0 ** (step - 3))
        ref.grad = g * scale
        opt.step()
        gk = g.clone()
        _adam_call(p, gk, m, v, lp, n, lr, b1, b2, eps, wd, step, scale, 1)
        st = opt.state[ref]
        torch.testing.assert_close(p, ref.detach(), rtol=2e-6, atol=1e-7)
        # a moment close to zero is a cancellation of terms of the tensor's scale: absolute tolerance of 2 ulp of that scale
        torch.testing.assert_close(m, st["exp_avg"], rtol=2e-6, atol=2.4e-7 * float(st["exp_avg"].abs().max()))
        torch.testing.assert_close(v, st["exp_avg_sq"], rtol=2e-6, atol=2.4e-7 * float(st["exp_avg_sq"].abs().max()))
        assert torch.equal(lp, p.to(torch.bfloat16))
        assert not gk.any()
    # without the low-precision copy and without zeroing: the same update, gradient untouched
    p2, m2, v2, g = p.clone(), m.clone(), v.clone(), torch.randn(n, device=dev)
    gk = g.clone()
    _adam_call(p, g.clone(), m, v, lp, n, lr, b1, b2, eps, wd, 6, scale, 1)
    _adam_call(p2, gk, m2, v2, None, n, lr, b1, b2, eps, wd, 6, scale, 0)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2) and torch.equal(gk, g)


def test_adam_step_rejects_bad_arguments():
    p = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError):
        _adam_call(p, p, p, p, None, 16, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, 1)          # steps count from 1
    with pytest.raises(RuntimeError):
        _adam_call(p[1:], p, p, p, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 1)       # misaligned arena
    with pytest.raises(RuntimeError):
        _adam_call(p, p, p, p, None, 16, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, 1.0, 1)          # beta1 out of range
    _adam_call(p, p, p, p, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 1)               # empty: nothing to do


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(37, 53)
        self.b = torch.nn.Linear(53, 3)
        self.unused = torch.nn.Linear(5, 3)        # never gets a gradient: stays outside the arenas, like torch skips it

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_arena_adam_matches_torch_adam_through_the_wrapper():
    from branchgan import fused, optim, training
    torch.manual_seed(0)
    net, ref = _Net().cuda(), _Net().cuda()
    ref.load_state_dict(net.state_dict())
    opt = training.make_optimizer(net.parameters(), "adam", 1e-2, betas=(0.5, 0.999))
    assert isinstance(opt, optim.ArenaAdam)
    wrap = training.OptimizerWrapper(net, opt, gradAccum=2)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.5, 0.999), foreach=False)
    x = torch.randn(6, 11, 37, device="cuda")
    for step in range(6):
        wrap.performUpdateStep({"loss": net(x[step]).square().mean()}, step)
        ref(x[step]).square().mean().backward()
        if step % 2 == 0:
            ref_opt.step()
            ref_opt.zero_grad()
        for (k, a), b in zip(net.named_parameters(), ref.parameters()):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=lambda m: f"step {step} {k}: {m}")
            if a.grad is not None and b.grad is not None:
                torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-6)
    assert net.unused.weight.grad is None
    # parameters now live in one arena, and the fused path reads the bf16 copies the optimizer keeps current
    assert net.a.weight.data_ptr() == opt.param_arena.data_ptr()
    lp = fused.lowp(net.b.weight, torch.bfloat16)
    assert lp.data_ptr() >= opt.lowp_arena.data_ptr() and torch.equal(lp, net.b.weight.detach().to(torch.bfloat16))
    # torch.optim.Adam's state layout: a checkpoint of one loads into the other
    sd = opt.state_dict()
    ref_sd = ref_opt.state_dict()
    assert len(sd["state"]) == len(ref_sd["state"]) == 4
    for i in sd["state"]:
        assert float(sd["state"][i]["step"]) == float(ref_sd["state"][i]["step"]) == 3.0
        torch.testing.assert_close(sd["state"][i]["exp_avg"], ref_sd["state"][i]["exp_avg"], rtol=1e-4, atol=1e-7)
    opt.load_state_dict(ref_sd)
    assert opt.steps == 3
    torch.testing.assert_close(opt.exp_avg[:37 * 53].view(53, 37), ref_sd["state"][0]["exp_avg"])
    # hand edits + invalidate_weights() re-round the copies
    with torch.no_grad():
        net.b.weight.mul_(2.0)
    fused.invalidate_weights()
    assert torch.equal(fused.lowp(net.b.weight, torch.bfloat16), net.b.weight.detach().to(torch.bfloat16))


def test_resume_from_checkpoint_before_the_first_step(tmp_path):
    """The reference's resume flow (SpawnedExperiment.py:27-34): build the model and the optimizer wrappers, load the start
    weights and the optimizer state, THEN train.  ArenaAdam has no arenas at that point: the loaded moments and step count
    are kept and copied in when the first step lays the arenas out.  A run resumed that way continues exactly like the
    uninterrupted one."""
    from branchgan import checkpoint, configs, gan, sampling, training

    def build():
        torch.manual_seed(0)
        model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", device="cuda").cuda()
        g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-3, betas=(0.5, 0.999)))
        d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
        return model, g_opt, d_opt

    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(2)]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=3)
    conf.sparseIndex = torch.tensor([1, 5, 9, 13, 17, 21, 25, 29], device="cuda")
    data = torch.randint(0, 2048, (4, 4, 32), generator=torch.Generator().manual_seed(2)).cuda()
    model, g_opt, d_opt = build()
    for i in range(3):
        training.train_step(model, data[i], conf, g_opt, d_opt, i, 9999, None)
    checkpoint.saveGannSetup(model, "run", "Mid", d_opt, g_opt, mainSavePath=str(tmp_path), saveOptimizers=True)
    training.train_step(model, data[3], conf, g_opt, d_opt, 3, 9999, None)
    want = {k: v.detach().clone() for k, v in model.state_dict().items()}

    fresh, g2, d2 = build()
    checkpoint.loadGannSetup(fresh, str(tmp_path / "run" / "Mid"), d2, g2)          # no step has run: the arenas do not exist yet
    assert g2.optimizer.param_arena is None and g2.optimizer._pending_state
    training.train_step(fresh, data[3], conf, g2, d2, 3, 9999, None)
    assert g2.optimizer.steps == 4 and d2.optimizer.steps == 4
    for k, v in fresh.state_dict().items():
        torch.testing.assert_close(v, want[k], rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")
