"""CPU: host-side mirror of transformer_vae/autoencoders.py — construction, registry keys, parameter
names, error behaviour — and the refusal to compute anywhere but on the GPU."""
from types import SimpleNamespace

import pytest
import torch

import transformer_vae_amd as amd
from transformer_vae_amd import functional as F


def cfg(d=32, L=11, n=3):
    return SimpleNamespace(t5=SimpleNamespace(d_model=d), funnel=SimpleNamespace(d_model=d, layer_norm_eps=1e-9),
                           latent_size=L, n_latent_tokens=n)


def test_registry_keys_and_classes():
    assert set(amd.VAE_ENCODER_MODELS) == {""}                              # autoencoders.py:70-72
    assert set(amd.VAE_DECODER_MODELS) == {"", "t5_norm", "no_norm"}        # autoencoders.py:73-77
    assert amd.VAE_DECODER_MODELS[""] is amd.LatentDecoderFunnelNorm
    assert amd.VAE_DECODER_MODELS["no_norm"] is amd.LatentDecoderNTokens
    assert amd.LatentEncoderLargeTanh is amd.LatentEncoderNTokens and amd.LatentDecoder is amd.LatentDecoderFunnelNorm


@pytest.mark.parametrize("dec,expected", [
    ("", {"encoder.token_to_latent.weight", "encoder.token_to_latent.bias", "decoder.latent_to_token.weight",
          "decoder.latent_to_token.bias", "decoder.norm.weight", "decoder.norm.bias"}),
    ("no_norm", {"encoder.token_to_latent.weight", "encoder.token_to_latent.bias", "decoder.latent_to_token.weight",
                 "decoder.latent_to_token.bias"}),
])
def test_parameter_names_match_the_reference(dec, expected):
    c = cfg()
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](c), amd.VAE_DECODER_MODELS[dec](c))
    sd = vae.state_dict()
    assert set(sd) == expected
    assert tuple(sd["encoder.token_to_latent.weight"].shape) == (11, 32)
    assert tuple(sd["decoder.latent_to_token.weight"].shape) == (32, 11)
    assert vae.use_reg_loss is True and vae.encoder.n_tokens == 3


def test_identity_decoder_has_no_parameters():
    c = cfg(d=32, L=32)
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](c), amd.VAE_DECODER_MODELS[""](c))
    assert set(vae.state_dict()) == {"encoder.token_to_latent.weight", "encoder.token_to_latent.bias",
                                     "decoder.norm.weight", "decoder.norm.bias"}
    plain = amd.VAE_DECODER_MODELS["no_norm"](c)
    x = torch.randn(2, 3, 32)
    assert plain(x) is x            # `lambda x: x` in the reference (autoencoders.py:32-34)


def test_errors_and_no_cpu_fallback():
    c = cfg()
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](c), amd.VAE_DECODER_MODELS[""](c))
    with pytest.raises(ValueError, match="Both `input_encoding` and `latent` sent to VAE are None."):
        vae()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae(input_encoding=torch.randn(2, 5, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae(latent=torch.randn(2, 3, 11))
    with pytest.raises(ValueError):
        amd.EncoderDecoderVAE(None, None, mmd_scope="everywhere")


def test_mmd_scales_fold_the_reference_scaling():
    # reg = MMD / batch * n  (autoencoders.py:112);  d reg / d z carries 4 n / (B N^2 D^2)
    ls, gs = F.mmd_scales(batch_size=8, n_latents=5, n_total=40, dim=77)
    assert ls == pytest.approx(5 / (8 * 40 ** 2))
    assert gs == pytest.approx(4 * 5 / (8 * 40 ** 2 * 77 ** 2))


def test_oracle_is_not_imported_by_the_product():
    """oracle/ is test infrastructure: no module of the package may import it (source check over the whole package)."""
    import glob
    import os
    pkg = os.path.dirname(amd.__file__)
    files = glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True) + glob.glob(os.path.join(pkg, "csrc", "*"))
    assert len(files) > 8
    for f in files:
        if os.path.isfile(f) and not f.endswith((".o", ".so")):
            src = open(f, errors="ignore").read()
            assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_reg_weight_schedule_and_deferred_logs():
    import os
    import numpy as np
    from transformer_vae_amd import losses
    s = np.load(os.path.join(os.path.dirname(__file__), "golden", "reg_weight_schedule.npz"))
    for step, w in zip(s["steps"], s["w"]):                      # values from the reference's torch.sigmoid(...).item()
        assert abs(losses.regulariser_loss_weight(int(step)) - float(w)) <= 1e-6 * float(w)
    assert losses.regulariser_loss_weight(None) == 0 and losses.regulariser_loss_weight(10, use_reg_loss=False) == 0
    assert float(losses.combine_loss(torch.tensor(2.0), torch.tensor(0.5), 0.25)) == 2.125
    # same numbers as the reference's _update_logs/get_latest_logs bookkeeping (model.py:111-134)
    logs = losses.DeferredLogs()
    assert logs.get_latest_logs() == {}
    logs.update(decoder_ce=torch.tensor(2.0), reg_loss=torch.tensor(0.5), reg_loss_w=0.1)
    logs.update(decoder_ce=torch.tensor(4.0), reg_loss=torch.tensor(1.5), reg_loss_w=0.3)
    out = logs.get_latest_logs()
    assert out == pytest.approx({"decoder_ce": 3.0, "reg_loss": 1.0, "reg_loss_w": 0.2})
    logs.update(decoder_ce=torch.tensor(1.0), reg_loss=torch.tensor(0.0), reg_loss_w=0.5)
    assert logs.get_latest_logs() == pytest.approx({"decoder_ce": 1.0, "reg_loss": 0.0, "reg_loss_w": 0.5})
    # the reference's trainer writes running totals from outside (trainer.py:341-376); a key that did not move reports 0
    logs.update(decoder_ce=torch.tensor(2.0))
    logs.latest_logs["cycle_loss"] = logs.latest_logs.get("cycle_loss", 0) + 0.75
    logs.update(decoder_ce=torch.tensor(6.0))
    assert logs.get_latest_logs() == pytest.approx({"decoder_ce": 4.0, "reg_loss": 0.0, "reg_loss_w": 0.0, "cycle_loss": 0.375})
    assert logs.get_latest_logs() == {}                          # no update since: nothing to report, no division by zero


def test_trainer_arguments_and_loss_grad():
    """the MMD scope surfaces through the reference's argument system (train.py:33-100)
    and `get_loss_grad` keeps the reference's reduction order (trainer.py:258-281)."""
    from dataclasses import dataclass, fields
    from transformer_vae_amd import trainer as T

    @dataclass
    class FakeTrainingArguments:          # stands in for VAE_TrainingArguments (HF TrainingArguments subclass)
        save_steps: int = 1000
        interpolate_all_at_once: bool = False

    Args = T.extend_training_arguments(FakeTrainingArguments, "VAE_TrainingArguments")
    assert {f.name for f in fields(Args)} == {"save_steps", "interpolate_all_at_once", "mmd_scope",
                                              "mmd_compensate_ddp_mean"}
    a = Args()
    assert a.mmd_scope == "local" and a.mmd_compensate_ddp_mean is True and a.save_steps == 1000
    a.validate_mmd()
    with pytest.raises(ValueError, match="Unknown `mmd_scope`"):
        Args(mmd_scope="planet").validate_mmd()
    from transformers import HfArgumentParser
    parsed, = HfArgumentParser(Args).parse_args_into_dataclasses(["--mmd_scope", "global", "--save_steps", "5"])
    assert parsed.mmd_scope == "global" and parsed.save_steps == 5

    c = cfg()
    vae = amd.EncoderDecoderVAE(amd.VAE_ENCODER_MODELS[""](c), amd.VAE_DECODER_MODELS[""](c))
    T.configure_vae(vae, parsed, process_group="pg")
    assert vae.mmd_scope == "global" and vae.process_group == "pg" and vae.compensate_ddp_mean is True
    with pytest.raises(ValueError):
        T.configure_vae(vae, SimpleNamespace(mmd_scope="nope"))

    # reduction order of trainer.py:265-281: mean over replicas, / accumulation steps, backward, detached result
    w = torch.tensor([1.0, 3.0], requires_grad=True)
    out = T.get_loss_grad(w * 2.0, n_gpu=2, gradient_accumulation_steps=4)
    assert float(out) == 1.0 and not out.requires_grad
    assert torch.allclose(w.grad, torch.tensor([0.25, 0.25]))
    seen = []
    w2 = torch.tensor(2.0, requires_grad=True)
    T.get_loss_grad(w2 * 3.0, backward=lambda l: (seen.append(float(l)), l.backward()))
    assert seen == [6.0] and float(w2.grad) == 3.0
    assert T.get_loss({"loss": 7}) == 7 and T.get_loss((9, 1)) == 9
