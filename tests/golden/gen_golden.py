#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE in the build container.

Runs only where ``/root/reference`` exists (never on the GPU box, never from tests).
It imports the reference's own modules from ``/root/reference/src`` (three
``sys.modules`` stubs for absent third-party packages the hot path never calls:
``munch``, ``pynvml``, ``torchaudio`` — SURVEY.md §8c), loads the name-keyed
synthetic weights of ``stylish_tts_amd.params`` into them, records the three RNG
draws on the inference path (``models/flow.py:314``, ``models/generator.py:272,306``)
and stores inputs + expected outputs as small ``.npz`` fixtures next to this
script.  Nothing of the reference (source, bytecode, traced graphs) is written.

    python tests/golden/gen_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"


def _install_stubs():
    m = types.ModuleType("munch")

    class Munch(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    m.Munch = Munch
    sys.modules["munch"] = m
    p = types.ModuleType("pynvml")
    for n in ("nvmlInit", "nvmlDeviceGetHandleByIndex", "nvmlDeviceGetMemoryInfo"):
        setattr(p, n, lambda *a, **k: None)
    sys.modules["pynvml"] = p
    ta = types.ModuleType("torchaudio")
    tam = types.ModuleType("torchaudio.models")
    tat = types.ModuleType("torchaudio.transforms")
    tam.Conformer = object
    tat.Spectrogram = object
    ta.models, ta.transforms = tam, tat
    sys.modules.update({"torchaudio": ta, "torchaudio.models": tam, "torchaudio.transforms": tat})


_install_stubs()
sys.path.insert(0, REF_SRC)
import torch  # noqa: E402

from stylish_tts.lib.config_loader import load_model_config_yaml  # noqa: E402
from stylish_tts.train.models.ada_norm import AdaptiveGeneratorBlock  # noqa: E402
from stylish_tts.train.models.duration_predictor import DurationPredictor  # noqa: E402
from stylish_tts.train.models.export_model import ExportModel  # noqa: E402
from stylish_tts.train.models.pitch_energy_predictor import PitchEnergyPredictor  # noqa: E402
from stylish_tts.train.models.speech_predictor import SpeechPredictor  # noqa: E402
from stylish_tts.train.models.text_encoder import TextEncoder  # noqa: E402
from stylish_tts.train.models.text_style_encoder import TextStyleEncoder  # noqa: E402
from stylish_tts.train.utils import DurationProcessor  # noqa: E402

from stylish_tts_amd import params, synth  # noqa: E402
from stylish_tts_amd.config import load_model_config  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(1234)
SEED = 0


class Replay:
    """Feed predetermined tensors to torch.randn / rand / randn_like inside the block.

    The three draws on the inference path (flow.py:314 randn_like; generator.py:272 randn,
    :306 rand) are replaced by hash-generated tensors that the tests regenerate from the
    same names, so fixtures need not store them."""

    def __init__(self, randn=None, rand=None, randn_like=None):
        self.q = {"randn": list(randn or []), "rand": list(rand or []), "randn_like": list(randn_like or [])}

    def __enter__(self):
        self._orig = (torch.randn, torch.rand, torch.randn_like)

        def feed(tag):
            def inner(*a, **k):
                assert self.q[tag], f"unexpected extra torch.{tag} draw"
                return torch.from_numpy(np.ascontiguousarray(self.q[tag].pop(0)))

            return inner

        torch.randn, torch.rand, torch.randn_like = feed("randn"), feed("rand"), feed("randn_like")
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand, torch.randn_like = self._orig
        if exc[0] is None:
            assert not any(self.q.values()), "unused injected draws"


class CutTape:
    """Record the reference's har_phase where it is ill-conditioned, so that a run being compared can adopt the
    reference's value there (oracle.align_branch):
      (a) on the atan2 branch cut: |phase| > pi - 2e-3 (all of frame 0's negative-real bins: the reflect-padded
          first frame is even-symmetric so its spectrum is real up to FFT rounding);
      (b) bins of negligible magnitude (< 1e-4) whose phase is rounding noise.
    Stored as flat indices into [B, bins, T4] (after the last frame is dropped) + the reference's phase values."""

    stft = None  # set to the generator's TorchSTFT instance

    def __enter__(self):
        self._orig = CutTape.stft.transform
        self.mag = self.phase = None

        def inner(x):
            mag, cx, sy = self._orig(x)
            self.mag, self.phase = mag, torch.atan2(sy, cx)
            return mag, cx, sy

        CutTape.stft.transform = inner
        return self

    def __exit__(self, *exc):
        CutTape.stft.transform = self._orig

    def hints(self):
        ph = self.phase[:, :, :-1].contiguous().numpy().reshape(-1)
        mg = self.mag[:, :, :-1].contiguous().numpy().reshape(-1)
        idx = np.nonzero((np.abs(ph) > np.pi - 2e-3) | (mg < 1e-4))[0].astype(np.int32)
        return dict(cut_idx=idx, cut_phase=ph[idx].astype(np.float32))


def load_synth(module, name, cfg):
    spec = params.module_spec(name, cfg)
    sd = params.synth_state_dict(spec, SEED, prefix=name + ".")
    ref_sd = module.state_dict()
    ref_keys = [k for k in ref_sd if not k.startswith("posterior_encoder.")]
    assert ref_keys == list(sd.keys()), f"{name}: key/order mismatch vs reference"
    for k, v in sd.items():
        assert tuple(ref_sd[k].shape) == v.shape, (name, k, ref_sd[k].shape, v.shape)
    missing, unexpected = module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert all(k.startswith("posterior_encoder.") for k in missing), missing
    assert not unexpected
    return module.eval()


def t(x, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(x)) if dtype is None else torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    mc = load_model_config_yaml(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    cfg = load_model_config()
    meta = dict(torch_version=torch.__version__, seed=SEED)

    sp = load_synth(SpeechPredictor(mc), "speech_predictor", cfg)
    dp = load_synth(
        DurationPredictor(
            style_dim=mc.style_dim,
            inter_dim=mc.inter_dim,
            text_config=mc.text_encoder,
            style_config=mc.style_encoder,
            duration_config=mc.duration_predictor,
        ),
        "duration_predictor",
        cfg,
    )
    pe = load_synth(
        PitchEnergyPredictor(
            style_dim=mc.style_dim,
            inter_dim=mc.pitch_energy_predictor.inter_dim,
            text_config=mc.text_encoder,
            style_config=mc.style_encoder,
            duration_config=mc.duration_predictor,
            pitch_energy_config=mc.pitch_energy_predictor,
        ),
        "pitch_energy_predictor",
        cfg,
    )
    pte = load_synth(TextEncoder(inter_dim=mc.pitch_energy_predictor.inter_dim, config=mc.text_encoder), "pe_text_encoder", cfg)
    ptse = load_synth(
        TextStyleEncoder(mc.pitch_energy_predictor.inter_dim, mc.style_dim, mc.style_encoder), "pe_text_style_encoder", cfg
    )
    durproc = DurationProcessor(mc.duration_predictor.duration_classes, mc.duration_predictor.max_duration)
    CutTape.stft = sp.generator.stft

    # ---------------- frame-rate modules at a reduced shape (T4 = 64) ----------------
    B, T4 = 1, 64
    asr = t(synth.normal("g.asr", (B, 128, T4)))
    pitch4 = t(synth.pitch_curve("g.pitch4", B, T4))
    energy4 = t(synth.uniform("g.energy4", (B, T4)) * 2.0 + 2.0)
    style = t(synth.normal("g.style", (B, 64)) * 0.7)
    nz = synth.path_noise("frame64", B, T4)

    print("decoder")
    x_dec, _ = sp.decoder(asr, pitch4, energy4, style)
    # one AdaIN res-block alone (row 9): the encode block on its concatenated input
    F0c = sp.decoder.F0_conv(pitch4.unsqueeze(1))
    Nc = sp.decoder.N_conv(energy4.unsqueeze(1))
    enc_in = torch.cat([asr, F0c, Nc], dim=1)
    enc_out = sp.decoder.encode(enc_in, style)
    save("decoder", asr=asr, pitch=pitch4, energy=energy4, style=style, enc_in=enc_in, enc_out=enc_out, x=x_dec, **meta)

    print("prior + reverse flow + post_flow")
    with Replay(randn_like=[nz["prior_noise"]]):
        z, mean, logstd = sp.prior_encoder(x_dec)
    z2, _, _ = sp.flow(z, mean, logstd, 1, style.unsqueeze(-1), reverse=True)
    mel = sp.post_flow(z2.mT).mT
    save("flow", x=x_dec, style=style, z=z, z_out=z2, mel=mel, **meta)

    print("generator")
    with Replay(randn=[nz["src_noise"]], rand=[nz["init_phase"]]), CutTape() as cut:
        pred = sp.generator(mel=mel, style=style, pitch=pitch4, energy=energy4)
    cut_gen = cut.hints()
    with Replay(randn=[nz["src_noise"]], rand=[nz["init_phase"]]):
        p1 = pitch4.unsqueeze(1)
        prior_sig = sp.generator.prior_generator(p1, (p1 > 10.0).float()).squeeze(1)
    har_spec, hx, hy = sp.generator.stft.transform(prior_sig)
    har_phase = torch.atan2(hy, hx)
    save(
        "generator",
        mel=mel,
        style=style,
        pitch=pitch4,
        energy=energy4,
        prior_signal=prior_sig,
        har_spec=har_spec[:, :, :-1],
        har_phase=har_phase[:, :, :-1],
        logamp=pred.magnitude,
        phase=pred.phase,
        audio=pred.audio,
        **cut_gen,
        **meta,
    )

    # pcph edge cases (row 13): all unvoiced; min f0 just above 20 Hz; K < 16 (every f0 > 750 Hz);
    # transition frames in (10, 20] Hz (voiced but excluded from the min); a batch of two whose
    # shared harmonic count comes from the other utterance
    print("pcph edge cases")
    edge = {}
    Te = 24
    ar = np.arange(Te)
    cases = {
        "unvoiced": np.zeros((1, Te), np.float32),
        "low": np.where(ar % 5 == 0, 0.0, 20.5 + ar * 0.25).astype(np.float32)[None],
        "high": np.where(ar < 4, 0.0, 800.0 + 10.0 * ar).astype(np.float32)[None],
        "transition": np.concatenate([np.zeros(6), [12.0, 15.0, 18.0], 120 + np.arange(Te - 9) * 3.0]).astype(np.float32)[None],
        "batch2": np.stack([np.full(Te, 900.0), np.where(ar < 8, 0.0, 140.0)]).astype(np.float32),
    }
    for nm, f0 in cases.items():
        f0t = t(f0).unsqueeze(1)
        nze = synth.path_noise("pcph." + nm, f0.shape[0], Te)
        rand = [] if nm == "unvoiced" else [nze["init_phase"]]  # early return draws no phase (generator.py:275-276)
        with Replay(randn=[nze["src_noise"]], rand=rand):
            sig = sp.generator.prior_generator(f0t, (f0t > 10.0).float())
        edge[f"{nm}_f0"] = f0
        edge[f"{nm}_out"] = sig.numpy()
    save("pcph_edges", **edge, **meta)

    # ---------------- phoneme-rate modules ----------------
    print("text encoder / style encoder / duration")
    P = 12
    texts = synth.tokens("g.texts", 1, P, mc.text_encoder.tokens)
    lengths = np.array([P], np.int64)
    mu, xenc, xmask = dp.text_encoder(t(texts), t(lengths))
    sty = dp.style_encoder(mu, t(lengths))
    pros = dp.prosody_encoder(mu, sty, t(lengths))
    logits = dp(t(texts), t(lengths))
    dur = durproc.prediction_to_duration(logits[0], P)
    align = durproc(logits[0], P)
    save(
        "duration",
        texts=texts,
        lengths=lengths,
        text_mu=mu,
        text_x=xenc,
        style=sty,
        prosody=pros,
        logits=logits,
        duration=dur,
        alignment_shape=np.array(align.shape),
        **meta,
    )
    # B=2 ragged lengths through the text encoder + duration predictor (masking behaviour)
    texts2 = synth.tokens("g.texts2", 2, P, mc.text_encoder.tokens)
    len2 = np.array([P, 7], np.int64)
    texts2[1, 6] = 0
    texts2[1, 7:] = 0
    logits2 = dp(t(texts2), t(len2))
    mu2, _, _ = dp.text_encoder(t(texts2), t(len2))
    sty2 = dp.style_encoder(mu2, t(len2))
    save("duration_b2", texts=texts2, lengths=len2, text_mu=mu2, style=sty2, logits=logits2, **meta)

    # duration processor alone, both branches of prediction_to_duration (utils.py:472)
    lg = synth.normal("g.durlogits", (40, 16)) * 3.0
    d_ = durproc.prediction_to_duration(t(lg), 40)
    save("duration_processor", logits=lg, duration=d_, **meta)

    print("pitch/energy predictor")
    durs = synth.durations_for("g.durs", P, 40)
    T = int(durs.sum())
    align_m = synth.alignment_from_durations(durs)[None]
    pe_enc, _, _ = pte(t(texts), t(lengths))
    pe_sty = ptse(pe_enc, t(lengths))
    pe_pros = pe.prosody_encoder(pe_enc, pe_sty, t(lengths))
    from stylish_tts.train.utils import length_to_mask

    pe_cross = pe.compute_cross(pe_pros, t(align_m), pe_sty, length_to_mask(t(lengths), P))
    f0_pred, n_pred = pe(pe_enc, t(lengths), t(align_m), pe_sty)
    save(
        "pitch_energy",
        texts=texts,
        lengths=lengths,
        durations=durs,
        pe_text=pe_enc,
        pe_style=pe_sty,
        prosody=pe_pros,
        cross=pe_cross,
        f0=f0_pred,
        energy=n_pred,
        **meta,
    )

    print("speech predictor (tokens → audio), B=1 and B=2")
    pitchT = synth.pitch_curve("g.pitchT", 1, T)
    energyT = (synth.uniform("g.energyT", (1, T)) * 2.0 + 2.0).astype(np.float32)
    nzs = synth.path_noise("sp1", 1, 4 * T)
    with Replay(randn=[nzs["src_noise"]], rand=[nzs["init_phase"]], randn_like=[nzs["prior_noise"]]), CutTape() as cut:
        pred = sp(t(texts), t(lengths), t(align_m), t(pitchT), t(energyT))
    save("speech_predictor", texts=texts, lengths=lengths, durations=durs, pitch=pitchT, energy=energyT, audio=pred.audio, **cut.hints(), **meta)

    texts_b = np.concatenate([texts, synth.tokens("g.texts_b", 1, P, mc.text_encoder.tokens)])
    len_b = np.array([P, P], np.int64)
    durs_b = np.stack([durs, durs[::-1]])
    align_b = np.stack([synth.alignment_from_durations(d) for d in durs_b])
    pitch_b = np.concatenate([pitchT, synth.pitch_curve("g.pitchT2", 1, T)])
    energy_b = np.concatenate([energyT, energyT[:, ::-1].copy()])
    nzb = synth.path_noise("sp2", 2, 4 * T)
    with Replay(randn=[nzb["src_noise"]], rand=[nzb["init_phase"]], randn_like=[nzb["prior_noise"]]), CutTape() as cut:
        pred = sp(t(texts_b), t(len_b), t(align_b), t(pitch_b), t(energy_b))
    save("speech_predictor_b2", texts=texts_b, lengths=len_b, durations=durs_b, pitch=pitch_b, energy=energy_b, audio=pred.audio, **cut.hints(), **meta)

    print("export model end to end (duration → alignment → audio)")
    em = ExportModel(
        speech_predictor=sp,
        duration_predictor=dp,
        pitch_energy_predictor=pe,
        pe_text_encoder=pte,
        pe_text_style_encoder=ptse,
        device="cpu",
    )
    alignment_e = durproc(logits[0], P).unsqueeze(0)
    Te_ = alignment_e.shape[2]
    pe_f0, pe_n = pe(pe_enc, t(lengths), alignment_e, pe_sty)
    nze = synth.path_noise("export", 1, 4 * Te_)
    with Replay(randn=[nze["src_noise"]], rand=[nze["init_phase"]], randn_like=[nze["prior_noise"]]), CutTape() as cut:
        audio_e = em(t(texts), t(lengths), alignment_e)
    save("export_model", texts=texts, lengths=lengths, duration=dur, pitch=pe_f0, energy=pe_n, audio=audio_e, **cut.hints(), **meta)

    print("3 s utterance through decoder → flow → vocoder (audio only)")
    T4b = 960
    asr_b = t(synth.normal("g3.asr", (1, 128, T4b)))
    pitch_3 = t(synth.pitch_curve("g3.pitch", 1, T4b))
    energy_3 = t(synth.uniform("g3.energy", (1, T4b)) * 2.0 + 2.0)
    style_3 = t(synth.normal("g3.style", (1, 64)) * 0.7)
    nz3 = synth.path_noise("frame960", 1, T4b)
    with Replay(randn=[nz3["src_noise"]], rand=[nz3["init_phase"]], randn_like=[nz3["prior_noise"]]), CutTape() as cut:
        xd, _ = sp.decoder(asr_b, pitch_3, energy_3, style_3)
        z, mean, logstd = sp.prior_encoder(xd)
        z2, _, _ = sp.flow(z, mean, logstd, 1, style_3.unsqueeze(-1), reverse=True)
        mel3 = sp.post_flow(z2.mT).mT
        pred3 = sp.generator(mel=mel3, style=style_3, pitch=pitch_3, energy=energy_3)
    save("frame_path_3s", audio=pred3.audio, mel_probe=mel3[:, ::64, ::16], x_probe=xd[:, ::64, ::16], **cut.hints(), **meta)

    print("AdaptiveGeneratorBlock (MRF + Snake) standalone")
    blk = AdaptiveGeneratorBlock(128, 7, (1, 3, 5), 64)
    spec = params.adaptive_generator_block_spec("", 128, 7, 64)
    sdm = params.synth_state_dict(spec, SEED, prefix="mrf.")
    assert set(sdm) == set(blk.state_dict()), set(sdm) ^ set(blk.state_dict())
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in sdm.items()})
    blk.eval()
    xm = t(synth.normal("g.mrf_x", (1, 128, 96)))
    ym = blk(xm, style)
    save("mrf_block", x=xm, style=style, y=ym, **meta)


def frame_path_3s_more():
    """Two more distinct 3-s utterances through the reference's decoder -> prior / reverse flow -> vocoder (frame_path_3s_b / _c): with
    frame_path_3s they make a batch of DISTINCT utterances at the benchmarked size (tests/test_hip_benchmarked_path.py)."""
    mc = load_model_config_yaml(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    cfg = load_model_config()
    meta = dict(torch_version=torch.__version__, seed=SEED)
    sp = load_synth(SpeechPredictor(mc), "speech_predictor", cfg)
    CutTape.stft = sp.generator.stft
    T4b = 960
    for tag in ("b", "c"):
        asr_b = t(synth.normal(f"g3{tag}.asr", (1, 128, T4b)))
        pitch_3 = t(synth.pitch_curve(f"g3{tag}.pitch", 1, T4b))
        energy_3 = t(synth.uniform(f"g3{tag}.energy", (1, T4b)) * 2.0 + 2.0)
        style_3 = t(synth.normal(f"g3{tag}.style", (1, 64)) * 0.7)
        nz3 = synth.path_noise(f"frame960{tag}", 1, T4b)
        # the initial phase is ONE scalar per call (models/generator.py:306 draws rand(1, 1) for the whole batch): the three goldens share
        # frame_path_3s's, so that they can be batched into one call
        nz3["init_phase"] = synth.path_noise("frame960", 1, T4b)["init_phase"]
        with Replay(randn=[nz3["src_noise"]], rand=[nz3["init_phase"]], randn_like=[nz3["prior_noise"]]), CutTape() as cut:
            xd, _ = sp.decoder(asr_b, pitch_3, energy_3, style_3)
            z, mean, logstd = sp.prior_encoder(xd)
            z2, _, _ = sp.flow(z, mean, logstd, 1, style_3.unsqueeze(-1), reverse=True)
            mel3 = sp.post_flow(z2.mT).mT
            pred3 = sp.generator(mel=mel3, style=style_3, pitch=pitch_3, energy=energy_3)
        save(f"frame_path_3s_{tag}", audio=pred3.audio, mel_probe=mel3[:, ::64, ::16], x_probe=xd[:, ::64, ::16], **cut.hints(), **meta)


def text_golden():
    """Tokeniser and wav-writer vectors (lib/text_utils.py:8-41, train/test_onnx.py:49-53,79-90)."""
    import io, json
    from scipy.io.wavfile import write
    from stylish_tts.lib.text_utils import TextCleaner

    mc = load_model_config_yaml(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    tc = TextCleaner(mc.symbol)
    texts = [
        "ðə kwˈɪk bɹˈaʊn fˈɑːks dʒˈʌmps ˌoʊvɚ ðə lˈeɪzi dˈɑːɡ.",
        "hɛlˈoʊ, wˈɜːld! — “kwˈoʊt” (ænd) mˈɔːɹ…",
        "'ᵻ' ǀǁᵊǃ ↓↑→↗↘ $",
        "unknown: @#%^ 123 stays out",
        "",
    ]
    ids = [tc(x) for x in texts]
    rng = np.random.default_rng(SEED)
    wave = np.tanh(rng.standard_normal(1234).astype(np.float32))
    pcm = np.multiply(wave, 32768).astype(np.int16)
    buf = io.BytesIO()
    write(buf, 24000, pcm)
    path = os.path.join(HERE, "text_tokens.json")
    json.dump(dict(texts=texts, ids=ids, table_size=len(tc.word_index_dictionary), wave=wave.tolist(), pcm=pcm.tolist(),
                   wav_hex=buf.getvalue().hex()), open(path, "w"), ensure_ascii=False)
    print("  wrote text_tokens.json", os.path.getsize(path))


def cfm_golden():
    """CfmSampler.forward (models/cfm/cfm.py:44-84) with a closed-form estimator: pins the time grid and update order."""
    from stylish_tts.train.models.cfm.cfm import CfmSampler

    def estimator(x, t, mask, cond, gain):
        return (cond - x) * (0.5 + t.reshape(-1, 1, 1)) * gain + torch.sin(3.0 * x)

    out = {}
    for n in (1, 4, 7, 32):
        z = t(synth.normal(f"cfm.z{n}", (2, 80, 37)))
        cond = t(synth.normal(f"cfm.c{n}", (2, 80, 37)))
        y = CfmSampler(estimator)(z, None, n, temperature=0.8, cond=cond, gain=1.7)
        out[f"z{n}"], out[f"cond{n}"], out[f"y{n}"] = z, cond, y
    save("cfm_euler", **out)


CFM_SMALL = dict(feat_dim=80, asr_dim=96, spk_dim=48, hidden_dim=128, emb_dim=64, depth=2, enc_blocks=1, dec_blocks=2, prev_depth=1, post_depth=1)


def cfm_decoder_golden():
    """CfmMelDecoder._forward and .forward (models/cfm/cfm_mel_decoder.py:318-413) with synthetic weights: the estimator alone
    (one evaluation, B = 2) at the class defaults and at a small size, and the n-step Euler sampling through it (small size)."""
    from stylish_tts.train.models.cfm.cfm_mel_decoder import CfmMelDecoder

    out = {}
    for tag, dims, B, n, L in (("default", dict(params.CFM_DEFAULT_DIMS), 2, 53, 27), ("small", dict(params.CFM_DEFAULT_DIMS, **CFM_SMALL), 2, 70, 70)):
        m = CfmMelDecoder(feat_dim=dims["feat_dim"], asr_dim=dims["asr_dim"], spk_dim=dims["spk_dim"], hidden_dim=dims["hidden_dim"],
                          emb_dim=dims["emb_dim"], xut_depth=dims["depth"], xut_enc_blocks=dims["enc_blocks"], xut_dec_blocks=dims["dec_blocks"],
                          tread_config={"prev_trns_depth": dims["prev_depth"], "post_trns_depth": dims["post_depth"], "dropout_ratio": 0.5}).eval()
        spec = params.cfm_mel_decoder_spec(dims)
        sd = params.synth_state_dict(spec, SEED, prefix="cfm_mel_decoder.")
        ref_sd = m.state_dict()
        assert sorted(ref_sd.keys()) == sorted(sd.keys()), sorted(set(ref_sd) ^ set(sd))
        for k, v in sd.items():
            assert tuple(ref_sd[k].shape) == v.shape, (k, tuple(ref_sd[k].shape), v.shape)
        m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=True)
        x = synth.normal(f"cfmd.{tag}.x", (B, dims["feat_dim"], n))
        asr = synth.normal(f"cfmd.{tag}.asr", (B, dims["asr_dim"], n))
        f0 = synth.pitch_curve(f"cfmd.{tag}.f0", B, L)
        f0[:, L // 3 : L // 3 + 4] = 0.0  # an unvoiced stretch
        nc = (synth.uniform(f"cfmd.{tag}.n", (B, L)) * 2 + 2).astype(np.float32)
        spk = synth.normal(f"cfmd.{tag}.spk", (B, dims["spk_dim"]))
        tt = np.array([0.3, 0.85][:B], np.float32)
        nz = synth.normal(f"cfmd.{tag}.nz", (B, n, 1))
        with torch.no_grad(), Replay(rand=[np.zeros((B, 1), np.float32)], randn_like=[nz]):
            y = m._forward(t(x), t(asr), t(f0), t(nc), t(spk), t(tt))
        out.update({f"{tag}_x": x, f"{tag}_asr": asr, f"{tag}_f0": f0, f"{tag}_n": nc, f"{tag}_spk": spk, f"{tag}_t": tt, f"{tag}_nz": nz, f"{tag}_y": y})
        if tag == "small":  # the sampler: z is drawn by torch.rand inside forward (:402), then one estimator call (with its two draws) per step
            steps = 4
            z = synth.uniform("cfmd.z", (B, dims["feat_dim"], n))
            nzs = [synth.normal(f"cfmd.nz{i}", (B, n, 1)) for i in range(steps)]
            rand = [z]
            for _ in range(steps):
                rand.append(np.zeros((B, 1), np.float32))
            with Replay(rand=rand, randn_like=nzs):
                ys = m(t(asr), t(f0), t(nc), t(spk), steps, 0.9)
            out.update(dict(sample_z=z, sample_y=ys, sample_steps=steps, sample_temperature=0.9, **{f"sample_nz{i}": a for i, a in enumerate(nzs)}))
    save("cfm_decoder", **out)


NARROW = {"decoder": {"hidden_dim": 384, "residual_dim": 32}, "generator": {"input_dim": 384, "hidden_dim": 384, "conv_intermediate_dim": 1152}}


def narrow_golden():
    """A model.yml that is NOT the default one (decoder / generator width 384 instead of 512, residual 32, ConvNeXt
    intermediate 1152: the flow then runs on 96 channels): the reference's SpeechPredictor built from that config, decoder ->
    prior / flow -> vocoder on a 64-frame utterance.  Pins that the build follows lib/config_loader.py:369-414 rather than
    one set of constants."""
    import copy

    import yaml

    raw = yaml.safe_load(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    ours = copy.deepcopy(dict(load_model_config()))
    for sec, kv in NARROW.items():
        raw[sec].update(kv)
        ours[sec] = dict(ours[sec], **kv)
    import io

    mc = load_model_config_yaml(io.StringIO(yaml.safe_dump(raw)))
    cfg = load_model_config(ours)
    sp = load_synth(SpeechPredictor(mc), "speech_predictor", cfg)
    CutTape.stft = sp.generator.stft
    T4 = 64
    asr = t(synth.normal("nw.asr", (1, 128, T4)))
    pitch = t(synth.pitch_curve("nw.pitch", 1, T4))
    energy = t(synth.uniform("nw.energy", (1, T4)) * 2.0 + 2.0)
    style = t(synth.normal("nw.style", (1, 64)) * 0.7)
    nz = synth.path_noise("narrow64", 1, T4, flow_dim=NARROW["decoder"]["hidden_dim"] // 4)
    with Replay(randn=[nz["src_noise"]], rand=[nz["init_phase"]], randn_like=[nz["prior_noise"]]), CutTape() as cut:
        xd, _ = sp.decoder(asr, pitch, energy, style)
        z, mean, logstd = sp.prior_encoder(xd)
        z2, _, _ = sp.flow(z, mean, logstd, 1, style.unsqueeze(-1), reverse=True)
        mel = sp.post_flow(z2.mT).mT
        pred = sp.generator(mel=mel, style=style, pitch=pitch, energy=energy)
    save("frame_path_narrow", x=xd, z=z, z_out=z2, mel=mel, audio=pred.audio, config_overrides=np.frombuffer(yaml.safe_dump(NARROW).encode(), np.uint8),
         torch_version=torch.__version__, **cut.hints())


def conv_stft_golden():
    """STFT.transform / STFT.inverse of models/stft.py:98-187 (the conv1d / conv_transpose1d DFT-matrix STFT the ONNX export
    swaps into the generator), standalone, at the generator's own geometry (n_fft 2048, hop 75, window 1200), plus the
    evidence for SURVEY 8a row 17: what the export wiring of train/convert_to_onnx.py:31-36 does to the generator."""
    import json
    import traceback

    from stylish_tts.train.models.stft import STFT

    st = STFT(filter_length=2048, hop_length=75, win_length=1200).eval()
    wave = np.tanh(synth.normal("cstft.wave", (2, 1200)) * 0.5).astype(np.float32)
    wave[1, 300:500] = 0.0  # a silent stretch: zero-magnitude bins (x, y are 0/sqrt(1e-14) there)
    mag, x, y = st.transform(t(wave))
    back = st.inverse(mag, x, y)
    # inverse on an arbitrary (non-Hermitian-consistent) spectrum too: it is a plain linear map
    m2 = np.abs(synth.normal("cstft.m2", (1, 1025, 9))).astype(np.float32)
    ph = synth.normal("cstft.ph", (1, 1025, 9)).astype(np.float32) * 2.0
    inv2 = st.inverse(t(m2), t(np.cos(ph)), t(np.sin(ph)))
    save("conv_stft", wave=wave, mag=mag, x=x, y=y, back=back, m2=m2, x2=np.cos(ph), y2=np.sin(ph), inv2=inv2, torch_version=torch.__version__)

    # --- row 17 evidence: the reference's own export wiring on its own generator
    mc = load_model_config_yaml(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    cfg = load_model_config()
    sp = load_synth(SpeechPredictor(mc), "speech_predictor", cfg)
    sp.generator.stft = STFT(filter_length=mc.n_fft, hop_length=mc.hop_length, win_length=mc.win_length).eval()  # convert_to_onnx.py:31-36
    T = 16
    rec = dict(wiring="train/convert_to_onnx.py:31-36: generator.stft = STFT(filter_length=n_fft, hop_length=hop_length, win_length=win_length)",
               n_fft=int(mc.n_fft), hop_length=int(mc.hop_length), win_length=int(mc.win_length), generator_hop=int(mc.hop_length) // 4, mel_frames=T,
               torch_version=torch.__version__)
    try:
        out = sp.generator(mel=t(synth.normal("cstft.mel", (1, 512, 4 * T))), style=t(synth.normal("cstft.style", (1, 64))),
                           pitch=t(synth.pitch_curve("cstft.pitch", 1, 4 * T)), energy=t(synth.uniform("cstft.en", (1, 4 * T))))
        rec["outcome"] = "ran"
        rec["audio_shape"] = list(out.audio.shape)
    except Exception as e:  # noqa: BLE001
        tb = traceback.extract_tb(e.__traceback__)
        rec["outcome"] = "raised"
        rec["exception"] = f"{type(e).__name__}: {e}"
        # file:line:function of every frame (no source text)
        rec["frames"] = [f"{os.path.relpath(f.filename, REF_SRC) if f.filename.startswith(REF_SRC) else os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in tb]
    path = os.path.join(HERE, "onnx_stft_wiring_evidence.json")
    json.dump(rec, open(path, "w"), indent=1)
    print("  wrote onnx_stft_wiring_evidence.json:", rec["outcome"], rec.get("exception", ""))


def atan2_instability_golden(n_trials: int = 16):
    """Evidence that `har_phase = atan2(Im, Re)` (models/generator.py:405-413) is indeterminate IN THE REFERENCE at the bins the parity tests adopt:
    the reference's own Generator.forward on frame_path_3s's inputs, once as is and `n_trials` times with the source noise multiplied by
    (1 + e), e uniform in +-2^-22 (a couple of fp32 ulps: the size of a summation-order change inside torch's own FFT).  Recorded: the bins whose
    har_phase moves by more than 1 rad in at least one trial (`unstable_idx`), per trial the frames that hold such a bin and the per-frame
    max-abs difference of the reference's audio against its own unperturbed run, the bins of negligible magnitude (< 2e-4) of the base run, and the
    bins on the cut whose imaginary part is below 1e-6 of their frame's largest magnitude (`on_cut_idx`: the sign of an imaginary part below the accuracy of a 2048-point fp32 FFT)."""
    mc = load_model_config_yaml(open(os.path.join(REF_SRC, "stylish_tts/train/config/model.yml")))
    cfg = load_model_config()
    sp = load_synth(SpeechPredictor(mc), "speech_predictor", cfg)
    CutTape.stft = sp.generator.stft
    T4b = 960
    asr_b = t(synth.normal("g3.asr", (1, 128, T4b)))
    pitch_3 = t(synth.pitch_curve("g3.pitch", 1, T4b))
    energy_3 = t(synth.uniform("g3.energy", (1, T4b)) * 2.0 + 2.0)
    style_3 = t(synth.normal("g3.style", (1, 64)) * 0.7)
    nz3 = synth.path_noise("frame960", 1, T4b)
    with Replay(randn_like=[nz3["prior_noise"]]):
        xd, _ = sp.decoder(asr_b, pitch_3, energy_3, style_3)
        z, mean, logstd = sp.prior_encoder(xd)
        z2, _, _ = sp.flow(z, mean, logstd, 1, style_3.unsqueeze(-1), reverse=True)
        mel3 = sp.post_flow(z2.mT).mT

    def run(src_noise):
        with Replay(randn=[src_noise], rand=[nz3["init_phase"]]), CutTape() as cut:
            pred = sp.generator(mel=mel3, style=style_3, pitch=pitch_3, energy=energy_3)
        return pred.audio.numpy().reshape(-1), cut.phase[:, :, :-1].contiguous().numpy().reshape(-1), cut.mag[:, :, :-1].contiguous().numpy().reshape(-1)

    audio0, phase0, mag0 = run(nz3["src_noise"])
    # bins ON the cut within the accuracy of a 2048-point fp32 FFT: the imaginary part |Im| = |X| sin(pi - |har_phase|) is below 1e-6 x the largest |X| of
    # its frame (an FFT's rounding error is absolute, ~ eps x log2(N) x the size of the frame's spectrum; exact or nearly exact zeros on the
    # even-symmetric first frame, whose spectrum is real).  The sign of such an imaginary part comes from torch's butterfly order, not from the data: a
    # perturbation that keeps the frame's symmetry does not flip it, another FFT does
    frame_max = mag0.reshape(1025, T4b).max(0, keepdims=True)
    im_abs = mag0.reshape(1025, T4b) * np.sin(np.float64(np.pi) - np.abs(phase0.reshape(1025, T4b)).astype(np.float64))
    on_cut = np.nonzero(((im_abs < 1e-6 * frame_max) & (np.abs(phase0.reshape(1025, T4b)) > 3.0) & (mag0.reshape(1025, T4b) > 0)).reshape(-1))[0].astype(np.int32)
    rng = np.random.default_rng(20260405)
    unstable = np.zeros(phase0.shape, bool)
    moved_frames = np.zeros((n_trials, T4b), bool)
    audio_diff = np.zeros((n_trials, T4b), np.float32)
    moved_count = []
    for k in range(n_trials):
        e = rng.uniform(-2.0 ** -22, 2.0 ** -22, nz3["src_noise"].shape).astype(np.float32)
        a_k, p_k, _ = run((nz3["src_noise"] * (np.float32(1.0) + e)).astype(np.float32))
        mv = np.abs(p_k.astype(np.float64) - phase0) > 1.0
        unstable |= mv
        moved_count.append(int(mv.sum()))
        moved_frames[k] = mv.reshape(1025, T4b).any(0)
        audio_diff[k] = np.abs(a_k - audio0).reshape(T4b, 75).max(1)
        print(f"  trial {k}: {moved_count[-1]} bins moved > 1 rad, audio max-abs diff vs the unperturbed run {audio_diff[k].max():.3e}")
    save("atan2_instability", unstable_idx=np.nonzero(unstable)[0].astype(np.int32), tiny_idx=np.nonzero(mag0 < 2e-4)[0].astype(np.int32), on_cut_idx=on_cut,
         moved_frames=np.packbits(moved_frames, axis=1), audio_diff=audio_diff.astype(np.float16), moved_count=np.asarray(moved_count, np.int32),
         rel_perturbation=np.float32(2.0 ** -22), torch_version=torch.__version__, seed=SEED)


if __name__ == "__main__":
    if "--only-atan2" in sys.argv:
        atan2_instability_golden()
    elif "--only-3s-more" in sys.argv:
        frame_path_3s_more()
    elif "--only-conv-stft" in sys.argv:
        conv_stft_golden()
    elif "--only-narrow" in sys.argv:
        narrow_golden()
    elif "--only-cfm" in sys.argv:
        cfm_golden()
    elif "--only-cfm-decoder" in sys.argv:
        cfm_decoder_golden()
    elif "--only-text" in sys.argv:
        _ = text_golden()
    else:
        main()
        text_golden()
        cfm_golden()
        cfm_decoder_golden()
        conv_stft_golden()
        narrow_golden()
        frame_path_3s_more()
        atan2_instability_golden()
