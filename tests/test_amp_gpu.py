"""GPU: the fp16-operand mode's loss scaling (alpro_amd/amp.py, FlatAdamW, alpro_adamw_step / alpro_loss_scale_update) -- what apex.amp
does for the reference under `fp16: 1` (run_pretrain_sparse.py:441,596-634), restated on the device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, n=3):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in [(64, 48), (48,), (7, 5)][:n]]


def _ref_adamw(p, g, m, v, t, lr, b1, b2, eps, max_norm, total_norm):
    """src/optimization/adamw.py:77-101 after clip_grad_norm_ (run_pretrain_sparse.py:633), fp64."""
    g = g * min(1.0, max_norm / (total_norm + 1e-6))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    step = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return p - step * m / (np.sqrt(v) + eps), m, v


def test_scaled_gradients_give_the_unscaled_update_and_the_device_counter_drives_bias_correction():
    from alpro_amd import config as rt
    from alpro_amd.optim import FlatAdamW
    with rt.use_compute_dtype("fp16"):
        ps = _params(0)
        opt = FlatAdamW(ps, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, max_grad_norm=0.5, allreduce=False)
        assert opt.scaler is not None
        opt.scaler.to("cuda").state[0] = 1024.0
        ref = [(p.detach().double().cpu().numpy(), 0.0, 0.0) for p in ps]
        gen = torch.Generator().manual_seed(1)
        for t in range(1, 4):
            grads = [torch.randn(p.shape, generator=gen) for p in ps]
            tot = float(np.sqrt(sum((g.double() ** 2).sum() for g in grads)))
            for p, g in zip(ps, grads):
                if opt.flat is None:
                    p.grad = (g * 1024.0).cuda()
                else:
                    p.grad.copy_(g * 1024.0)
            opt._grads_scaled = True
            opt.step()
            ref = [_ref_adamw(r[0], g.double().numpy(), r[1], r[2], t, 1e-2, 0.9, 0.98, 1e-6, 0.5, tot) for r, g in zip(ref, grads)]
            for p, r in zip(ps, ref):
                assert np.abs(p.detach().cpu().double().numpy() - r[0]).max() < 2e-6, t
        st = opt.scaler.state.tolist()
        assert st[0] == 1024.0 and st[2] == 3.0 and st[3] == 0.0 and st[1] == 3.0


def test_overflow_skips_the_update_halves_the_scale_and_growth_doubles_it():
    from alpro_amd import config as rt
    from alpro_amd.optim import FlatAdamW
    with rt.use_compute_dtype("fp16"):
        ps = _params(2)
        opt = FlatAdamW(ps, lr=1e-2, max_grad_norm=None, allreduce=False)
        sc = opt.scaler.to("cuda")
        sc.window = 2
        for p in ps:
            p.grad = torch.ones_like(p) * 65536.0
        opt._grads_scaled = True
        opt.step()                                   # clean step 1 (builds the flat buffers)
        before = [p.detach().clone() for p in ps]
        m0, v0 = opt.flat["m"].clone(), opt.flat["v"].clone()
        ps[0].grad[3, 5] = float("inf")
        opt._grads_scaled = True
        opt.step()                                   # overflow: nothing moves, S halves, the tracker restarts
        assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps)) and torch.equal(m0, opt.flat["m"]) and torch.equal(v0, opt.flat["v"])
        assert sc.state.tolist() == [32768.0, 0.0, 1.0, 1.0]
        ps[0].grad.fill_(float("nan"))
        opt._grads_scaled = True
        opt.step()
        assert sc.state.tolist() == [16384.0, 0.0, 1.0, 2.0]
        for _ in range(2):                           # two clean steps == the window: S doubles
            for p in ps:
                p.grad.fill_(16384.0)
            opt._grads_scaled = True
            opt.step()
        assert sc.state.tolist() == [32768.0, 0.0, 3.0, 2.0]
        assert not torch.equal(before[0], ps[0].detach()) and torch.isfinite(ps[0]).all()


def test_fp16_backward_without_a_scaled_loss_is_refused_and_scale_loss_leaves_true_scale_gradients(bert_cfg):
    """The hand-written backward refuses fp16 gradient operands under an unscaled loss (loud, not silently flushed to zero); the apex-style
    context (delay_unscale=False, what the reference's drivers use) hands back TRUE-scale gradients, equal to the fp32-mode ones within
    fp16's operand rounding, and the FlatAdamW-fused form (optimizer.backward) leaves them scaled by S until step()."""
    from alpro_amd import amp, config as rt
    from alpro_amd.modeling.timesformer.vit import TimeSformer
    from alpro_amd.optim import FlatAdamW
    from tests.test_host_cpu import VENC
    torch.manual_seed(3)
    enc = TimeSformer(dict(VENC, num_frm=2, drop_path_rate=0.0), input_format="RGB").cuda().train()
    x = torch.randn(2, 3, 2, 224, 224, device="cuda")

    def loss_of():
        return enc.forward_features(x).float().square().mean()

    with rt.use_compute_dtype("fp32"):
        loss_of().backward()
    want = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    enc.zero_grad(set_to_none=True)
    with rt.use_compute_dtype("fp16"):
        with pytest.raises(RuntimeError, match="scaled loss"):
            loss_of().backward()
        enc.zero_grad(set_to_none=True)
        opt = FlatAdamW([p for p in enc.parameters()], lr=0.0, allreduce=False)
        assert opt.scaler.state_dict()["loss_scale"] == 65536.0          # apex's initial dynamic scale
        opt.scaler.to("cuda").state[0] = 1024.0                         # (no step() here to back off an overflow: pick a safe fixed scale)
        with amp.scale_loss(loss_of(), opt) as scaled:
            scaled.backward()
        assert opt._grads_scaled is False
        worst = 0.0
        for n, g in want.items():
            got = dict(enc.named_parameters())[n].grad
            worst = max(worst, float((got - g).norm() / (g.norm() + 1e-12)))
        assert worst < 1e-2, worst
        enc.zero_grad(set_to_none=True)
        opt.backward(loss_of())
        S = opt.scaler.loss_scale()
        assert opt._grads_scaled is True and S == 1024.0
        n0 = "model.blocks.0.attn.qkv.weight"
        got = dict(enc.named_parameters())[n0].grad
        assert float((got / S - want[n0]).norm() / want[n0].norm()) < 1e-2
    assert set(amp.state_dict()) >= {"loss_scaler0"}
