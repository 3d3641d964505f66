"""Drop-in nn.Module shims: the reference's constructor arguments, forward() signatures and state_dict keys, with
every forward() executed by libstylish_hip.so.

Reference classes mirrored (paths relative to /root/reference/src/stylish_tts/train/):
  TextEncoder            models/text_encoder.py:397-462
  TextStyleEncoder       models/text_style_encoder.py:6-26
  DurationPredictor      models/duration_predictor.py:8-36
  DurationProcessor      utils.py:385-494
  PitchEnergyPredictor   models/pitch_energy_predictor.py:11-121
  Decoder                models/decoder.py:6-60
  Generator              models/generator.py:340-438
  SpeechPredictor        models/speech_predictor.py:13-129
  ExportModel            models/export_model.py:5-45

Differences, all additive: forward() of the stochastic modules takes an optional ``noise`` dict with the three draws
the reference takes from the global torch generator (``prior_noise`` [B,128,4T], ``src_noise`` [B,1,300T],
``init_phase`` [1,1]); when omitted they are drawn with torch on the device.  Weights live in a flat store keyed by
the reference's state_dict names (both weight-norm flavours), so ``load_state_dict`` accepts reference checkpoints;
the training-only ``posterior_encoder.*`` keys are ignored.  There is no CPU path: forward() needs the GPU library.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import params
from .config import Record, load_model_config
from .runtime import HipModel, Segments

W_DECODER, W_FLOW, W_GENERATOR, W_SPEECH_TEXT, W_DURATION, W_PE_TEXT, W_PE_STYLE, W_PITCH_ENERGY = 1, 2, 4, 8, 16, 32, 64, 128

_ENGINES: Dict[int, HipModel] = {}


def get_engine(cfg=None, device: int = 0) -> HipModel:
    """One stts_ctx per device, shared by every shim (weights are namespaced by module name)."""
    if device not in _ENGINES or _ENGINES[device].ctx is None:
        _ENGINES[device] = HipModel(cfg if cfg is not None else load_model_config(), device)
    return _ENGINES[device]


class DecoderPrediction:
    """utils.py:363-382 (inference fields only)."""

    def __init__(self, *, audio, magnitude, phase):
        self.audio, self.magnitude, self.phase = audio, magnitude, phase
        self.text_stats = self.text2mel_stats = self.mel_stats = self.mel2text_stats = None


class HipModule(torch.nn.Module):
    """Flat parameter store with the reference's keys + lazy binding to the engine."""

    module_name = ""   # namespace inside the stts_ctx
    key_prefix = ""    # prefix of this module's keys inside that namespace (standalone sub-modules)
    components = 0     # STTS_W_* mask

    def __init__(self, spec, cfg, engine: Optional[HipModel] = None):
        super().__init__()
        self.cfg = cfg
        self._spec = spec
        self._store = OrderedDict((n, torch.zeros(s, dtype=torch.float32)) for n, s, _ in spec)
        self._engine = engine
        self._dirty = True
        self._device_index = 0

    # ---- nn.Module surface the reference's callers touch (models/export_model.py:19-28)
    def state_dict(self, *args, prefix: str = "", **kwargs):
        return OrderedDict((prefix + k, v) for k, v in self._store.items())

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        missing = [k for k in self._store if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._store and not k.startswith("posterior_encoder.")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for k in self._store:
            if k in state_dict:
                v = torch.as_tensor(state_dict[k]).detach().to("cpu", torch.float32)
                if tuple(v.shape) != tuple(self._store[k].shape):
                    raise RuntimeError(f"load_state_dict: shape mismatch for {k}: {tuple(v.shape)} vs {tuple(self._store[k].shape)}")
                self._store[k] = v.clone()
        self._dirty = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def parameters(self, recurse: bool = True):
        return iter(self._store.values())

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        return iter((prefix + k, v) for k, v in self._store.items())

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)):
                d = torch.device(a)
                if d.type == "cuda":
                    self._device_index = d.index or 0
        return self

    def load_synthetic(self, seed: int = 0):
        """Name-keyed synthetic weights (params.synth_state_dict), as used by the golden fixtures."""
        sd = params.synth_state_dict(self._spec, seed, prefix=self.module_name + "." + self.key_prefix)
        self.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return self

    # ---- engine binding
    @property
    def engine(self) -> HipModel:
        if self._engine is None:
            self._engine = get_engine(self.cfg, self._device_index)
        self._bind()
        return self._engine

    def _bind(self):
        """Shims share one engine per device and several may map to the same component (two Decoder instances, a Decoder
        next to a SpeechPredictor, the reference's three TextEncoder instances): the engine remembers which shim packed
        each component last, and a shim whose weights are not the packed ones re-binds before it runs."""
        eng = self._engine
        owners = eng.__dict__.setdefault("_owners", {})
        bits = [1 << i for i in range(8) if self.components >> i & 1]
        if self._dirty or any(owners.get(b) is not self for b in bits):
            eng.load_state_dict(self.module_name, self._store, prefix=self.key_prefix)
            eng.finalize(self.components)  # releases the previous packing of these components (stts_finalize_weights)
            for b in bits:
                owners[b] = self
            self._dirty = False


# ------------------------------------------------------------------------------------------------ helpers
def _f(x: torch.Tensor, dev) -> torch.Tensor:
    return x.to(device=dev, dtype=torch.float32).contiguous()


def _pack_tokens(texts: torch.Tensor, lengths: torch.Tensor, dev):
    L = [int(v) for v in lengths.tolist()]
    toks = torch.cat([texts[b, : L[b]] for b in range(len(L))]).to(device=dev, dtype=torch.int64).contiguous()
    return toks, Segments(L, dev)


def _pack_rows(eng: HipModel, x_bcp: torch.Tensor, lengths) -> torch.Tensor:
    """[B,C,P] padded -> packed time-major [sum P, ld] (transpose on the GPU library, row selection is plumbing)."""
    B, C, P = x_bcp.shape
    tmaj = eng.to_time_major(_f(x_bcp, eng.device))  # [B*P, ld]
    idx = torch.cat([torch.arange(b * P, b * P + int(lengths[b]), device=eng.device) for b in range(B)])
    return tmaj.index_select(0, idx).contiguous()


def _unpack_rows(eng: HipModel, x: torch.Tensor, seg: Segments, C: int, P: int) -> torch.Tensor:
    """packed [sum P, ld] -> [B,C,P] zero padded."""
    B = seg.n
    padded = torch.zeros(B * P, x.shape[1], dtype=torch.float32, device=eng.device)
    idx = torch.cat([torch.arange(b * P, b * P + seg.lengths[b], device=eng.device) for b in range(B)])
    padded.index_copy_(0, idx, x)
    return eng.to_channel_major(padded, B, C, P)


def _durations_from_alignment(alignment: torch.Tensor, lengths, dev):
    """The reference passes the 0/1 matrix [B,P,T] (utils.py:476-489); the library takes integer durations."""
    d = alignment.sum(dim=2).round().to(torch.int32)
    L = [int(v) for v in lengths]
    dur = torch.cat([d[b, : L[b]] for b in range(len(L))]).to(dev).contiguous()
    T = [int(d[b, : L[b]].sum().item()) for b in range(len(L))]
    return dur, T


def draw_noise(B: int, T4: int, dev, flow_dim: int = 128):
    """The reference's three draws (models/flow.py:314; models/generator.py:272,306) with torch's generator."""
    return dict(prior_noise=torch.randn(B, flow_dim, T4, device=dev), src_noise=torch.randn(B, 1, 75 * T4, device=dev),
                init_phase=torch.rand(1, 1, device=dev))


# ------------------------------------------------------------------------------------------------ modules
class TextEncoder(HipModule):
    module_name, components, _which = "pe_text_encoder", W_PE_TEXT, 2

    def __init__(self, *, inter_dim, config, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        c2 = Record(cfg)
        c2["text_encoder"] = Record(config)
        self.inter_dim = inter_dim
        super().__init__(params.text_encoder_spec("", c2, inter_dim), cfg, engine)

    def forward(self, x, x_lengths, spks=None):
        eng = self.engine
        toks, seg = _pack_tokens(x, x_lengths, eng.device)
        mu, xh = eng.text_encoder(self._which, seg, toks, return_hidden=True)
        P = x.shape[1]
        mask = (torch.arange(P, device=eng.device)[None, :] < x_lengths.to(eng.device)[:, None]).unsqueeze(1).float()
        return _unpack_rows(eng, mu, seg, self.inter_dim, P), _unpack_rows(eng, xh, seg, self.cfg.text_encoder.hidden_dim, P), mask


class TextStyleEncoder(HipModule):
    module_name, components, _which = "pe_text_style_encoder", W_PE_STYLE, 2

    def __init__(self, inter_dim, style_dim, config, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        c2 = Record(cfg)
        c2["style_encoder"] = Record(config)
        c2["style_dim"] = style_dim
        super().__init__(params.text_style_encoder_spec("", c2, inter_dim), cfg, engine)

    def forward(self, x, lengths):
        eng = self.engine
        L = [int(v) for v in lengths.tolist()]
        return eng.text_style(self._which, Segments(L, eng.device), _pack_rows(eng, x, L))


class DurationPredictor(HipModule):
    module_name, components = "duration_predictor", W_DURATION

    def __init__(self, style_dim, inter_dim, text_config, style_config, duration_config, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        super().__init__(params.duration_predictor_spec(cfg), cfg, engine)

    def forward(self, texts, text_lengths):
        eng = self.engine
        toks, seg = _pack_tokens(texts, text_lengths, eng.device)
        logits, _ = eng.duration(seg, toks)
        B, P = texts.shape
        # padded positions: prosody is masked to 0 there (prosody_encoder.py:80), so the reference returns the bias
        out = self._store["duration_proj.linear_layer.bias"].to(eng.device).expand(B, P, -1).clone()
        for b in range(B):
            out[b, : seg.lengths[b]] = logits[seg.host[b] : seg.host[b + 1]]
        return out


class DurationProcessor(torch.nn.Module):
    """utils.py:385-494: logits [P,16] -> 0/1 alignment [P,T]."""

    def __init__(self, class_count, max_dur):
        super().__init__()
        self.class_count, self.max_dur = class_count, max_dur

    def prediction_to_duration(self, pred, text_length=None):
        from . import _lib
        from .runtime import _ptr, _stream

        lib = _lib.load()
        p = pred.to(dtype=torch.float32).contiguous()
        if not p.is_cuda:
            p = p.cuda()
        dur = torch.empty(p.shape[0], dtype=torch.int32, device=p.device)
        _lib.check(lib.stts_duration_decode(_stream(), _ptr(p), p.shape[1], p.shape[0], _ptr(dur)))
        return dur

    def duration_to_alignment(self, duration):
        from . import _lib
        from .runtime import _ptr, _stream

        lib = _lib.load()
        d = duration.to(dtype=torch.int32).contiguous()
        if not d.is_cuda:
            d = d.cuda()
        T = int(d.sum().item())  # the same host round trip the reference has (test_onnx.py:65-66)
        out = torch.empty(d.shape[0], T, dtype=torch.float32, device=d.device)
        _lib.check(lib.stts_duration_to_alignment(_stream(), _ptr(d), d.shape[0], T, _ptr(out)))
        return out

    def forward(self, pred, text_length):
        return self.duration_to_alignment(self.prediction_to_duration(pred, text_length))


class PitchEnergyPredictor(HipModule):
    module_name, components = "pitch_energy_predictor", W_PITCH_ENERGY

    def __init__(self, style_dim, inter_dim, text_config, style_config, duration_config, pitch_energy_config, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        super().__init__(params.pitch_energy_predictor_spec(cfg), cfg, engine)

    def forward(self, text_encoding, text_lengths, alignment, style):
        eng = self.engine
        L = [int(v) for v in text_lengths.tolist()]
        dur, T = _durations_from_alignment(alignment, L, eng.device)
        sp, st = Segments(L, eng.device), Segments(T, eng.device)
        f0, en = eng.pitch_energy(sp, st, dur, _pack_rows(eng, text_encoding, L), _f(style, eng.device))
        Tm = alignment.shape[2]
        F0 = torch.zeros(len(L), Tm, device=eng.device)
        N = torch.zeros(len(L), Tm, device=eng.device)
        for b in range(len(L)):
            F0[b, : T[b]] = f0[st.host[b] : st.host[b + 1]]
            N[b, : T[b]] = en[st.host[b] : st.host[b + 1]]
        return F0, N


class Decoder(HipModule):
    module_name, key_prefix, components = "speech_predictor", "decoder.", W_DECODER

    def __init__(self, *, dim_in, style_dim, dim_out, hidden_dim, residual_dim, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        super().__init__(params.decoder_spec("", dim_in, style_dim, hidden_dim, residual_dim), cfg, engine)
        self.hidden_dim = hidden_dim

    def forward(self, asr, F0_curve, N, s):
        eng = self.engine
        B, _, T4 = asr.shape
        seg = Segments([T4] * B, eng.device)
        x = eng.decoder(seg, eng.to_time_major(_f(asr, eng.device)), _f(F0_curve, eng.device).reshape(-1), _f(N, eng.device).reshape(-1),
                        _f(s, eng.device))
        return eng.to_channel_major(x, B, self.hidden_dim, T4), F0_curve


class Generator(HipModule):
    module_name, key_prefix, components = "speech_predictor", "generator.", W_GENERATOR

    def __init__(self, *, style_dim, n_fft, win_length, hop_length, config, cfg=None, engine=None):
        cfg = cfg or load_model_config()
        super().__init__(params.generator_spec("", cfg), cfg, engine)
        self.n_bins = n_fft // 2 + 1

    def forward(self, *, mel, style, pitch, energy=None, noise=None):
        eng = self.engine
        B, _, T4 = mel.shape
        seg = Segments([T4] * B, eng.device)
        nz = noise or draw_noise(B, T4, eng.device)
        spec, phase = eng.harmonic_stft(seg, _f(pitch, eng.device).reshape(-1), _f(torch.as_tensor(nz["src_noise"]), eng.device).reshape(-1),
                                        _f(torch.as_tensor(nz["init_phase"]), eng.device).reshape(-1), batch_scope=True)
        audio, la, ph = eng.vocoder(seg, eng.to_time_major(_f(mel, eng.device)), _f(style, eng.device), spec, phase, return_spec=True)
        eng.check_status()
        rep = lambda t: torch.cat([t, t[:, :, -1:]], dim=2)  # F.pad(..., mode="replicate") (generator.py:425-426)  # noqa: E731
        return DecoderPrediction(audio=audio.reshape(B, 1, 75 * T4), magnitude=rep(eng.to_channel_major(la, B, self.n_bins, T4)),
                                 phase=rep(eng.to_channel_major(ph, B, self.n_bins, T4)))


class SpeechPredictor(HipModule):
    module_name, components = "speech_predictor", W_DECODER | W_FLOW | W_GENERATOR | W_SPEECH_TEXT

    def __init__(self, model_config=None, engine=None):
        cfg = model_config if model_config is not None else load_model_config()
        super().__init__(params.speech_predictor_spec(cfg), cfg, engine)

    def forward(self, texts, text_lengths, alignment, pitch, energy, audio_gt=None, noise=None, return_spectra=True):
        if audio_gt is not None:
            raise NotImplementedError("audio_gt (posterior encoder, training only: speech_predictor.py:103-110) is outside the inference hot path")
        eng = self.engine
        toks, sp = _pack_tokens(texts, text_lengths, eng.device)
        dur, T = _durations_from_alignment(alignment, sp.lengths, eng.device)
        st = Segments(T, eng.device)
        st4 = st.scaled(4)
        B = sp.n
        enc = eng.text_encoder(1, sp, toks)
        style = eng.text_style(1, sp, enc)
        asr = eng.length_regulate(sp, st4, dur, 4, enc, self.cfg.inter_dim)
        pk = lambda t: torch.cat([_f(t, eng.device)[b, : T[b]] for b in range(B)])  # noqa: E731
        p4, e4 = eng.upsample4(st, st4, pk(pitch)), eng.upsample4(st, st4, pk(energy))
        equal = len(set(T)) == 1
        nz = noise or draw_noise(B, 4 * max(T), eng.device)
        pn = _f(torch.as_tensor(nz["prior_noise"]), eng.device)
        sn = _f(torch.as_tensor(nz["src_noise"]), eng.device)
        pn_tm = torch.cat([pn[b, :, : 4 * T[b]].t() for b in range(B)]).contiguous()
        sn_flat = torch.cat([sn[b, 0, : 300 * T[b]] for b in range(B)]).contiguous()
        ip = _f(torch.as_tensor(nz["init_phase"]), eng.device).reshape(-1)
        x = eng.decoder(st4, asr, p4, e4, style)
        mel = eng.prior_flow(st4, x, style, pn_tm)
        spec, phase = eng.harmonic_stft(st4, p4, sn_flat, ip, batch_scope=True)
        audio, la, ph = eng.vocoder(st4, mel, style, spec, phase, return_spec=True)
        eng.check_status()
        if equal:
            T4 = 4 * T[0]
            rep = lambda t: torch.cat([t, t[:, :, -1:]], dim=2)  # noqa: E731
            return DecoderPrediction(audio=audio.reshape(B, 1, 75 * T4), magnitude=rep(eng.to_channel_major(la, B, 1025, T4)),
                                     phase=rep(eng.to_channel_major(ph, B, 1025, T4)))
        return DecoderPrediction(audio=[audio[75 * st4.host[b] : 75 * st4.host[b + 1]] for b in range(B)], magnitude=None, phase=None)


class STFT(torch.nn.Module):
    """models/stft.py:6-187: the conv-form STFT of the ONNX export, same constructor and transform() / inverse() signatures.
    Only the model.yml geometry is built (filter_length 2048, window 1200, hann, center, replicate); any hop."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann", center=True, pad_mode="replicate", cfg=None, engine=None):
        super().__init__()
        if (filter_length, win_length, window, center, pad_mode) != (2048, 1200, "hann", True, "replicate"):
            raise NotImplementedError("the HIP conv-form STFT is built for filter_length 2048, win_length 1200, hann, center, replicate (model.yml)")
        self.filter_length, self.hop_length, self.win_length, self.n_fft = filter_length, hop_length, win_length, filter_length
        self.freq_bins = filter_length // 2 + 1
        self._engine = engine
        self._cfg = cfg

    @property
    def engine(self) -> HipModel:
        if self._engine is None:
            self._engine = get_engine(self._cfg, 0)
        return self._engine

    def transform(self, waveform: torch.Tensor):
        """waveform [B, T] -> magnitude, x, y [B, 1025, T // hop + 1] (stft.py:98-139), any T like the reference.
        A tail shorter than a hop: the reference pads by REPLICATION, so extending the waveform to the next multiple of the hop with copies of its
        last sample leaves every frame it produces unchanged and appends one frame, which is dropped."""
        eng, hop = self.engine, self.hop_length
        B, T = waveform.shape
        F = T // hop + 1
        wave = _f(waveform, eng.device)
        if T % hop:
            wave = torch.cat([wave, wave[:, -1:].expand(B, hop - T % hop)], dim=1)
        Fk = wave.shape[1] // hop + 1
        seg = Segments([Fk] * B, eng.device)
        mag, x, y = eng.conv_stft_transform(seg, wave.reshape(-1).contiguous(), hop)
        return tuple(eng.to_channel_major(v, B, self.freq_bins, Fk)[:, :, :F].contiguous() for v in (mag, x, y))

    def inverse(self, magnitude: torch.Tensor, x: torch.Tensor, y: torch.Tensor, length=None):
        """[B, 1025, F] x 3 -> waveform [B, 1, (F - 1) * hop] (stft.py:141-187)."""
        eng, hop = self.engine, self.hop_length
        B, _, F = magnitude.shape
        seg = Segments([F] * B, eng.device)
        tmaj = [eng.to_time_major(_f(v, eng.device), 1056) for v in (magnitude, x, y)]
        wave = eng.conv_stft_inverse(seg, *tmaj, hop).reshape(B, 1, (F - 1) * hop)
        return wave if length is None else wave[..., :length]


class ExportModel(torch.nn.Module):
    """models/export_model.py:5-45: the inference composition (B = 1 in the reference: '1 1 l -> l')."""

    def __init__(self, *, speech_predictor, duration_predictor=None, pitch_energy_predictor, pe_text_encoder, pe_text_style_encoder, device=None,
                 **kwargs):
        super().__init__()
        self.speech_predictor, self.pitch_energy_predictor = speech_predictor, pitch_energy_predictor
        self.pe_text_encoder, self.pe_text_style_encoder = pe_text_encoder, pe_text_style_encoder

    def forward(self, texts, text_lengths, alignment, noise=None):
        pe_text_encoding, _, _ = self.pe_text_encoder(texts, text_lengths)
        pe_text_style = self.pe_text_style_encoder(pe_text_encoding, text_lengths)
        pitch, energy = self.pitch_energy_predictor(pe_text_encoding, text_lengths, alignment, pe_text_style)
        prediction = self.speech_predictor(texts, text_lengths, alignment, pitch, energy, noise=noise)
        assert prediction.audio.shape[0] == 1, "ExportModel.forward returns a single waveform (export_model.py:44)"
        return prediction.audio.reshape(-1)


def build_inference_modules(cfg=None, engine=None, synthetic_seed: Optional[int] = None):
    """The five modules of the inference composition (models/models.py:32-63, :79-101), optionally with synthetic weights."""
    cfg = cfg or load_model_config()
    m = dict(
        speech_predictor=SpeechPredictor(cfg, engine=engine),
        duration_predictor=DurationPredictor(cfg.style_dim, cfg.inter_dim, cfg.text_encoder, cfg.style_encoder, cfg.duration_predictor, cfg=cfg,
                                             engine=engine),
        pitch_energy_predictor=PitchEnergyPredictor(cfg.style_dim, cfg.pitch_energy_predictor.inter_dim, cfg.text_encoder, cfg.style_encoder,
                                                    cfg.duration_predictor, cfg.pitch_energy_predictor, cfg=cfg, engine=engine),
        pe_text_encoder=TextEncoder(inter_dim=cfg.pitch_energy_predictor.inter_dim, config=cfg.text_encoder, cfg=cfg, engine=engine),
        pe_text_style_encoder=TextStyleEncoder(cfg.pitch_energy_predictor.inter_dim, cfg.style_dim, cfg.style_encoder, cfg=cfg, engine=engine),
    )
    if synthetic_seed is not None:
        for mod in m.values():
            mod.load_synthetic(synthetic_seed)
    return m
