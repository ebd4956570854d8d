"""SURVEY.md 8f row f2 on the device: `losses.DeferredLogs` with device scalars -- no synchronisation per step, ONE
readback per fetch, the reference's numbers (model.py:111-134), and usable inside a step captured by GraphedStep."""
import numpy as np
import pytest
import torch

from test_gpu_parity import build_vae, dev
from oracle import bottleneck_oracle as orc

pytestmark = pytest.mark.gpu


def test_deferred_logs_device_scalars_one_readback(monkeypatch):
    from transformer_vae_amd import losses
    logs = losses.DeferredLogs()
    vals = [(2.0, 0.5, 0.1), (4.0, 1.5, 0.3), (9.0, 4.0, 0.2)]
    for ce, reg, w in vals:
        logs.update(decoder_ce=torch.tensor(ce, device=dev()), reg_loss=torch.tensor(reg, device=dev(), dtype=torch.bfloat16),
                    reg_loss_w=w)
    calls = []
    real_cpu = torch.Tensor.cpu
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *a, **k: (calls.append(1), real_cpu(self, *a, **k))[1])
    out = logs.get_latest_logs()
    monkeypatch.undo()
    assert len(calls) == 1                                               # one device -> host transfer for all keys
    assert out == pytest.approx({"decoder_ce": 5.0, "reg_loss": 2.0, "reg_loss_w": 0.2})
    assert logs.latest_logs == pytest.approx({"decoder_ce": 15.0, "reg_loss": 6.0, "reg_loss_w": 0.6})   # running totals (model.py:59-67)
    assert logs.get_latest_logs() == {}
    logs.update(decoder_ce=torch.tensor(1.0, device=dev()), reg_loss=torch.tensor(0.25, device=dev()), reg_loss_w=0.5)
    assert logs.get_latest_logs() == pytest.approx({"decoder_ce": 1.0, "reg_loss": 0.25, "reg_loss_w": 0.5})
    # a long window keeps its precision: 20 000 updates of a value that fp32 running sums would smear
    for _ in range(20000):
        logs.update(reg_loss=torch.tensor(1.0001, device=dev()))
    assert logs.get_latest_logs()["reg_loss"] == pytest.approx(float(np.float32(1.0001)), rel=1e-9)


def test_deferred_logs_inside_a_captured_step():
    import transformer_vae_amd as amd
    from transformer_vae_amd import losses
    B, S, d, n, L = 8, 12, 64, 5, 77
    _, params, _, _ = orc.synth_inputs(2, 2, d, n, L, 1, "")
    vae = build_vae(d, L, n, "", params)
    gen = torch.Generator(device=dev()).manual_seed(3)
    h = torch.randn((B, S, d), generator=gen, device=dev()).requires_grad_(True)
    prior = torch.randn((B * n, L), generator=gen, device=dev())
    logs = losses.DeferredLogs()
    leaves = [h] + list(vae.parameters())

    def step(hh, pp):
        out = vae(input_encoding=hh, true_samples=pp)
        ce = out.reconstructed_encoding.square().mean()          # stands in for the decoder's cross entropy
        loss = losses.combine_loss(ce, out.reg_loss, 0.25)
        loss.backward()
        logs.update(decoder_ce=ce, reg_loss=out.reg_loss)        # captured: in-place adds on the device
        return out.reg_loss, ce

    gs = amd.GraphedStep(step, (h, prior), zero_grads=leaves)
    logs.get_latest_logs()                                       # drop what warm-up and capture accumulated
    want_reg, want_ce = [], []
    for i in range(5):
        hn = torch.randn((B, S, d), generator=gen, device=dev())
        pn = torch.randn((B * n, L), generator=gen, device=dev())
        reg, ce = gs(hn, pn)
        want_reg.append(reg.clone())
        want_ce.append(ce.clone())
    out = logs.get_latest_logs()                                 # the replays advanced the device sums and counter
    assert out["reg_loss"] == pytest.approx(float(torch.stack(want_reg).double().mean()), rel=1e-6)
    assert out["decoder_ce"] == pytest.approx(float(torch.stack(want_ce).double().mean()), rel=1e-6)
    assert len(set(float(r) for r in want_reg)) == 5             # five different steps, not one replayed number
