"""Shim surface that needs no GPU: state_dict keys/shapes/order equal the reference inventory, load_state_dict
accepts reference-style checkpoints (both weight-norm flavours, posterior_encoder ignored) and rejects mismatches."""
import numpy as np
import pytest
import torch

from stylish_tts_amd import modules, params


def test_state_dict_matches_reference_inventory(cfg):
    mods = modules.build_inference_modules(cfg)
    for name, m in mods.items():
        spec = params.module_spec(name, cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == [n for n, _, _ in spec]
        assert all(tuple(sd[n].shape) == tuple(s) for n, s, _ in spec)
    sp = mods["speech_predictor"].state_dict()
    # the two weight-norm flavours the reference checkpoints carry (SURVEY.md §5 checkpoint/resume)
    assert tuple(sp["decoder.encode.conv1.parametrizations.weight.original1"].shape) == (512, 130, 3)
    assert tuple(sp["flow.flows.0.enc.in_layers.0.weight_v"].shape) == (256, 128, 5)
    assert tuple(sp["generator.convnext.0.grn.gamma"].shape) == (1, 1, 1536)
    assert sum(v.numel() for v in sp.values()) == 42808288  # 44.7 M minus the training-only posterior encoder


def test_load_state_dict_contract(cfg):
    m = modules.Decoder(dim_in=128, style_dim=64, dim_out=512, hidden_dim=512, residual_dim=64, cfg=cfg)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    sd["posterior_encoder.pre_spec.weight"] = torch.zeros(3)  # training-only keys are ignored
    m.load_state_dict(sd)
    assert torch.equal(m.state_dict()["F0_conv.bias"], sd["F0_conv.bias"])
    bad = dict(sd)
    bad["encode.conv1.bias"] = torch.zeros(7)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        m.load_state_dict(bad)
    del sd["N_conv.bias"]
    with pytest.raises(RuntimeError, match="missing"):
        m.load_state_dict(sd)
    m.load_state_dict(sd, strict=False)
    for p in m.parameters():
        p.requires_grad = False  # what ExportModel.__init__ does (export_model.py:27-28)
    assert m.eval() is m and m.to("cpu") is m


def test_synthetic_weights_equal_fixture_weights(cfg, weights):
    m = modules.Generator(style_dim=64, n_fft=2048, win_length=1200, hop_length=75, config=cfg.generator, cfg=cfg).load_synthetic(0)
    w = weights["speech_predictor"]
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), w["generator." + k])


def test_no_cpu_fallback(cfg):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = modules.Decoder(dim_in=128, style_dim=64, dim_out=512, hidden_dim=512, residual_dim=64, cfg=cfg).load_synthetic(0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 128, 16), torch.zeros(1, 16), torch.zeros(1, 16), torch.zeros(1, 64))
