"""Pin the oracle (oracle/stylish_oracle.py) to golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import stylish_oracle as O
from stylish_tts_amd import params, synth


def hint(g):
    return (g["cut_idx"].astype(np.int64), g["cut_phase"].astype(np.float32))


def close(a, b, rtol=2e-4, atol=None, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6)
    tol = atol if atol is not None else rtol * scale
    err = np.abs(a - b).max()
    assert err <= tol, f"{what}: max-abs err {err:.3e} > {tol:.3e} (scale {scale:.3e})"
    return err


def test_decoder(weights):
    g = load_golden("decoder")
    w = weights["speech_predictor"]
    enc = O.adaptive_decoder_block(g["enc_in"], g["style"], w, "decoder.encode")
    close(enc, g["enc_out"], what="AdaptiveDecoderBlock")
    x = O.decoder_forward(g["asr"], g["pitch"], g["energy"], g["style"], w)
    close(x, g["x"], what="Decoder")


def test_prior_flow(weights):
    g = load_golden("flow")
    w = weights["speech_predictor"]
    nz = synth.path_noise("frame64", 1, 64)
    z, _, _ = O.prior_encoder(g["x"], nz["prior_noise"], w)
    close(z, g["z"], what="PriorEncoder")
    z2 = O.flow_reverse(g["z"], g["style"][:, :, None], w)
    close(z2, g["z_out"], what="flow reverse")
    close(O.post_flow(g["z_out"], w), g["mel"], what="post_flow")


def test_generator(weights):
    g = load_golden("generator")
    w = weights["speech_predictor"]
    nz = synth.path_noise("frame64", 1, 64)
    audio, la, ph, mid = O.generator_forward(
        g["mel"], g["style"], g["pitch"], nz["src_noise"], nz["init_phase"], w, return_intermediates=True, branch_hint=hint(g)
    )
    close(mid["prior_signal"], g["prior_signal"], atol=2e-6, what="generate_pcph")
    close(mid["har_spec"], g["har_spec"], atol=2e-5, what="STFT magnitude")
    # phase is discontinuous at ±pi: compare on the circle, and only where the bin is not ~0
    d = np.angle(np.exp(1j * (mid["har_phase"].astype(np.float64) - g["har_phase"])))
    strong = g["har_spec"] > 1e-3
    assert np.abs(d[strong]).max() < 5e-3
    # after choosing the reference's branch at the ±pi ties, the raw values agree too
    assert np.abs(mid["har_phase"].astype(np.float64) - g["har_phase"])[strong].max() < 5e-3
    close(la, g["logamp"], atol=2e-3, what="logamp")
    close(ph, g["phase"], atol=2e-3, what="phase")
    close(audio, g["audio"], atol=1e-4, what="audio")


@pytest.mark.parametrize("case", ["unvoiced", "low", "high", "transition", "batch2"])
def test_pcph_edges(case):
    g = load_golden("pcph_edges")
    f0 = g[f"{case}_f0"]
    nz = synth.path_noise("pcph." + case, f0.shape[0], f0.shape[1])
    out = O.generate_pcph(f0[:, None, :], nz["src_noise"], nz["init_phase"])
    close(out, g[f"{case}_out"], atol=2e-6, what=case)


def test_pcph_raises_when_voiced_but_nothing_above_20hz():
    f0 = np.full((1, 1, 8), 15.0, np.float32)
    nz = synth.path_noise("x", 1, 8)
    with pytest.raises(RuntimeError):
        O.generate_pcph(f0, nz["src_noise"], nz["init_phase"])


def test_duration_predictor(weights, cfg):
    g = load_golden("duration")
    w = weights["duration_predictor"]
    logits, mid = O.duration_predictor(g["texts"], g["lengths"], w, cfg, return_intermediates=True)
    close(mid["text_mu"], g["text_mu"], what="TextEncoder mu")
    close(mid["text_x"], g["text_x"], what="TextEncoder x")
    close(mid["style"], g["style"], what="TextStyleEncoder")
    close(mid["prosody"], g["prosody"], what="ProsodyEncoder")
    close(logits, g["logits"], what="duration logits")
    dur = O.prediction_to_duration(g["logits"][0])
    assert np.array_equal(dur, g["duration"])
    assert O.duration_to_alignment(dur).shape == tuple(g["alignment_shape"])


def test_duration_predictor_ragged_batch(weights, cfg):
    g = load_golden("duration_b2")
    logits, mid = O.duration_predictor(g["texts"], g["lengths"], weights["duration_predictor"], cfg, return_intermediates=True)
    close(mid["text_mu"], g["text_mu"], what="mu b2")
    close(mid["style"], g["style"], what="style b2")
    close(logits, g["logits"], what="logits b2")


def test_duration_processor_both_branches():
    g = load_golden("duration_processor")
    dur = O.prediction_to_duration(g["logits"])
    assert np.array_equal(dur, g["duration"])
    hard = O.CLASS_TO_DUR[g["logits"].argmax(-1)]
    assert (hard < 7).any() and (hard >= 7).any()


def test_pitch_energy(weights, cfg):
    g = load_golden("pitch_energy")
    al = synth.alignment_from_durations(g["durations"])[None]
    enc, _, _ = O.text_encoder(g["texts"], g["lengths"], weights["pe_text_encoder"], cfg)
    close(enc, g["pe_text"], what="pe_text_encoder")
    sty = O.text_style_encoder(enc, g["lengths"], weights["pe_text_style_encoder"], cfg)
    close(sty, g["pe_style"], what="pe_text_style_encoder")
    f0, n, mid = O.pitch_energy_predictor(
        g["pe_text"], g["lengths"], al, g["pe_style"], weights["pitch_energy_predictor"], cfg, return_intermediates=True
    )
    close(mid["prosody"], g["prosody"], what="pe prosody")
    close(mid["cross"], g["cross"], what="compute_cross (inverted band mask)")
    close(f0, g["f0"], what="F0")
    close(n, g["energy"], rtol=5e-4, what="N")


@pytest.mark.parametrize("name,case", [("speech_predictor", "sp1"), ("speech_predictor_b2", "sp2")])
def test_speech_predictor(weights, cfg, name, case):
    g = load_golden(name)
    d = np.atleast_2d(g["durations"])
    al = np.stack([synth.alignment_from_durations(x) for x in d])
    T = al.shape[2]
    nz = synth.path_noise(case, al.shape[0], 4 * T)
    audio, _, _ = O.speech_predictor_forward(g["texts"], g["lengths"], al, g["pitch"], g["energy"], nz, weights["speech_predictor"], cfg, hint(g))
    close(audio, g["audio"], atol=1e-3, what=name)


def test_export_model_end_to_end(weights, cfg):
    """ExportModel composition.  The harmonic source integrates pitch over the whole utterance and
    har_phase = atan2(.) feeds a linear conv, so audio is discontinuous in pitch: a 3e-3 Hz pitch error
    (fp32 rounding through the predictor) moves ~500 STFT bins across the ±pi cut and changes the audio by
    ~0.1.  Parity is therefore staged: predicted pitch/energy within tolerance, then audio with the
    reference's own pitch/energy fed to the speech predictor."""
    g = load_golden("export_model")
    al = O.duration_to_alignment(g["duration"])[None]
    nz = synth.path_noise("export", 1, 4 * al.shape[2])
    _, pitch, energy = O.export_model_forward(g["texts"], g["lengths"], al, nz, weights, cfg, hint(g))
    close(pitch, g["pitch"], atol=2e-2, what="pitch [Hz]")
    close(energy, g["energy"], atol=2e-3, what="energy")
    audio, _, _ = O.speech_predictor_forward(
        g["texts"], g["lengths"], al, g["pitch"], g["energy"], nz, weights["speech_predictor"], cfg, hint(g)
    )
    close(audio[0, 0], g["audio"], atol=1e-3, what="export audio (teacher-forced pitch)")


def test_frame_path_3s(weights):
    g = load_golden("frame_path_3s")
    w = weights["speech_predictor"]
    T4 = 960
    asr = synth.normal("g3.asr", (1, 128, T4))
    pitch = synth.pitch_curve("g3.pitch", 1, T4)
    energy = synth.uniform("g3.energy", (1, T4)) * 2.0 + 2.0
    style = synth.normal("g3.style", (1, 64)) * 0.7
    nz = synth.path_noise("frame960", 1, T4)
    audio, _, _ = O.frame_path(asr, pitch, energy.astype(np.float32), style.astype(np.float32), nz, w, hint(g))
    close(audio, g["audio"], atol=1e-3, what="3 s audio")


def test_mrf_block():
    g = load_golden("mrf_block")
    w = params.synth_state_dict(params.adaptive_generator_block_spec("", 128, 7, 64), 0, prefix="mrf.")
    y = O.adaptive_generator_block(g["x"], g["style"], w, "")
    close(y, g["y"], what="AdaptiveGeneratorBlock")


def test_conv_stft_module():
    """models/stft.py STFT (the ONNX export's conv-form STFT) standalone at the generator's geometry, and the recorded
    outcome of the reference's own export wiring (it raises, so there is no ONNX-path waveform to pin)."""
    import json
    import os

    g = load_golden("conv_stft")
    mag, x, y = O.conv_stft_transform(g["wave"])
    close(mag, g["mag"], rtol=2e-5, what="conv STFT magnitude")
    strong = g["mag"] > 1e-3
    assert np.abs(x - g["x"])[strong].max() < 2e-3 and np.abs(y - g["y"])[strong].max() < 2e-3
    close(O.conv_stft_inverse(g["mag"], g["x"], g["y"]), g["back"], rtol=2e-5, what="conv iSTFT of the transform")
    close(O.conv_stft_inverse(g["m2"], g["x2"], g["y2"]), g["inv2"], rtol=2e-5, what="conv iSTFT of an arbitrary spectrum")
    ev = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "onnx_stft_wiring_evidence.json")))
    assert ev["outcome"] == "raised" and ev["hop_length"] == 4 * ev["generator_hop"] and "generator.py" in ev["frames"][-1]


def narrow_cfg():
    """The non-default model.yml of tests/golden/gen_golden.py:narrow_golden (overrides recorded in the fixture)."""
    import copy

    import yaml
    from stylish_tts_amd.config import load_model_config

    g = load_golden("frame_path_narrow")
    over = yaml.safe_load(bytes(g["config_overrides"]).decode())
    base = copy.deepcopy(dict(load_model_config()))
    for sec, kv in over.items():
        base[sec] = dict(base[sec], **kv)
    return load_model_config(base), g


def test_frame_path_non_default_widths():
    """decoder / generator width 384, residual 32, ConvNeXt intermediate 1152 (flow on 96 channels): the oracle follows the
    config through the weights' shapes."""
    cfg, g = narrow_cfg()
    w = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
    T4 = 64
    asr = synth.normal("nw.asr", (1, 128, T4))
    pitch = synth.pitch_curve("nw.pitch", 1, T4)
    energy = (synth.uniform("nw.energy", (1, T4)) * 2.0 + 2.0).astype(np.float32)
    style = (synth.normal("nw.style", (1, 64)) * 0.7).astype(np.float32)
    nz = synth.path_noise("narrow64", 1, T4, flow_dim=cfg.decoder.hidden_dim // 4)
    x = O.decoder_forward(asr, pitch, energy, style, w)
    close(x, g["x"], what="Decoder (hidden 384)")
    z, _, _ = O.prior_encoder(g["x"], nz["prior_noise"], w)
    close(z, g["z"], what="PriorEncoder (96 channels)")
    close(O.flow_reverse(g["z"], style[:, :, None], w), g["z_out"], what="reverse flow (96 channels)")
    close(O.post_flow(g["z_out"], w), g["mel"], what="post_flow")
    audio, _, _ = O.frame_path(asr, pitch, energy, style, nz, w, hint(g))
    close(audio, g["audio"], atol=1e-3, what="waveform (non-default widths)")
