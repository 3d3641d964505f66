"""Host-side handle on the HIP library: context, weights, and the stage calls on torch device tensors.

PyTorch is plumbing here (device memory, streams); all arithmetic runs in libstylish_hip.so.
Tensors at this level are TIME-MAJOR packed rows (see include/stylish_hip.h); the nn.Module shims in
``modules.py`` convert from/to the reference's ``[B, C, T]`` layout at the module boundary.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import load_model_config

N_BINS_LD = 1056  # 1025 bins padded to a multiple of 32 floats (spectrum rows of an fp32 engine; HipModel.har_ld is the engine's own stride)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Segments:
    """Utterance row offsets (host int32 array + device copy).

    capacity=True (``Segments.capacity``): the host array holds UPPER BOUNDS - cumulative capacities that size every buffer and
    grid - and the device tensor the real offsets, written on the device by ``HipModel.frame_offsets`` (include/stylish_hip.h,
    STTS_SEG_CAPACITY): no host read of the predicted durations is needed before the frame-rate stages."""

    def __init__(self, lengths: Sequence[int], device, dev: Optional[torch.Tensor] = None, capacity: bool = False):
        lengths = [int(x) for x in lengths]
        self.lengths = lengths
        self.host = np.zeros(len(lengths) + 1, np.int32)
        self.host[1:] = np.cumsum(lengths)
        self.dev = dev if dev is not None else torch.from_numpy(self.host).to(device)
        self.n = len(lengths)
        self.rows = int(self.host[-1])
        self.max_len = max(lengths)
        self.is_capacity = capacity

    @classmethod
    def capacity(cls, caps: Sequence[int], device) -> "Segments":
        """Capacity layout: `caps` rows per utterance at most; the device offsets are uninitialised until frame_offsets fills them."""
        return cls(caps, device, dev=torch.empty(len(caps) + 1, dtype=torch.int32, device=device), capacity=True)

    @property
    def flags(self) -> int:
        return 1 if self.is_capacity else 0  # STTS_SEG_CAPACITY

    @property
    def host_ptr(self):
        return self.host.ctypes.data_as(C.c_void_p)

    def scaled(self, k: int, dev: Optional[torch.Tensor] = None) -> "Segments":
        if self.is_capacity:
            assert dev is not None, "a scaled capacity layout needs its own device offsets"
            return Segments([x * k for x in self.lengths], self.dev.device, dev=dev, capacity=True)
        return Segments([x * k for x in self.lengths], self.dev.device)


class CapacityOverflow(RuntimeError):
    """An utterance's predicted frame count exceeded the capacity its buffers were sized for (stts_frame_offsets)."""


class HipModel:
    """Owns a stts_ctx.  weights: {module name: {state_dict key: array}} for the five inference modules."""

    PRECISIONS = {"f32": 0, "bf16": 1, "f16": 2, "f32_native": 3}

    def __init__(self, cfg=None, device: int = 0, precision: str = "f32"):
        """precision: operand precision of the contractions ("f32" = the reference's arithmetic, fp32 products formed from exact
        three-term bf16 splits on the bf16 matrix cores; "f32_native" = the same on the f32 matrix cores; "bf16" / "f16" round
        the matrix-core operands, fp32 accumulate; include/stylish_hip.h:stts_set_precision)."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        if not torch.cuda.is_available():
            raise RuntimeError("HipModel needs a GPU (MI355X); there is no CPU fallback in the product path")
        self.lib = _lib.load()
        self.cfg = cfg if cfg is not None else load_model_config()
        self.device = torch.device("cuda", device)
        self._dims = _lib.dims_from_config(self.cfg)
        h = C.c_void_p()
        _lib.check(self.lib.stts_ctx_create(C.byref(self._dims), device, C.byref(h)))
        self.ctx = h
        self.precision = precision
        _lib.check(self.lib.stts_set_precision(self.ctx, self.PRECISIONS[precision]))
        # row stride of the harmonic spectra = the prior convs' packed input width, asked from the library (1025 bins padded to 32 in fp32, 64 in the 16-bit modes)
        self.har_ld = int(self.lib.stts_har_ld(self.ctx))
        # grow-only workspaces, one per launch stream (stages issued on different streams may run concurrently)
        self._ws: Dict[int, torch.Tensor] = {}
        self._pws: Dict[int, torch.Tensor] = {}
        self._ws_lock = threading.Lock()

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.stts_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Mapping[str, Mapping[str, "np.ndarray | torch.Tensor"]], which: int = 7):
        for mod, sd in weights.items():
            self.load_state_dict(mod, sd)
        self.finalize(which)

    def load_state_dict(self, module: str, sd: Mapping[str, "np.ndarray | torch.Tensor"], prefix: str = ""):
        for k, v in sd.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(v, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            name = (module + "." if module else "") + prefix + k
            _lib.check(self.lib.stts_load_weight(self.ctx, name.encode(), a.ctypes.data_as(C.c_void_p), shape, max(a.ndim, 1)))

    def finalize(self, which: int = 7):
        _lib.check(self.lib.stts_finalize_weights(self.ctx, which))

    def check_status(self):
        _lib.check(self.lib.stts_check_status(self.ctx, _stream()))

    def frame_offsets(self, seg_p: Segments, dur: torch.Tensor, caps: Sequence[int]):
        """Device-side DurationProcessor bookkeeping: -> (mel-rate Segments, vocoder-rate Segments, need) as CAPACITY layouts whose
        device offsets are the real ones (cumulative predicted frames).  `caps`: mel frames each utterance may have at most.
        need [n_utt] int32 (device): the frames each utterance asked for; need[u] > caps[u] = overflow: the utterance was truncated
        to its capacity (every stage stays in bounds) and the call must be repeated with larger capacities (CapacityOverflow is
        what Synthesizer raises internally when it reads `need` with the audio)."""
        st = Segments.capacity(caps, self.device)
        off4 = torch.empty(seg_p.n + 1, dtype=torch.int32, device=self.device)
        need = torch.empty(seg_p.n, dtype=torch.int32, device=self.device)
        cap_dev = torch.from_numpy(st.host).to(self.device, non_blocking=True)
        _lib.check(self.lib.stts_frame_offsets(self.ctx, _stream(), seg_p.n, _ptr(seg_p.dev), _ptr(dur), _ptr(cap_dev), _ptr(st.dev), _ptr(off4), _ptr(need)))
        return st, st.scaled(4, dev=off4), need

    # ------------------------------------------------------------------ workspace
    def workspace(self, seg: Segments) -> torch.Tensor:
        need = int(self.lib.stts_frame_workspace_bytes(self.ctx, seg.rows, seg.n, seg.max_len))
        key = torch.cuda.current_stream(self.device).cuda_stream
        with self._ws_lock:
            ws = self._ws.get(key)
            if ws is None or ws.numel() < need:
                ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def _f32(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ stages (time-major tensors)
    def decoder(self, seg: Segments, asr, pitch, energy, style):
        dh = self.cfg.decoder.hidden_dim
        x = self._f32(seg.rows, dh)
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_decoder_forward(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(asr), asr.shape[1], _ptr(pitch),
                                                 _ptr(energy), _ptr(style), _ptr(x), dh, _ptr(ws), ws.numel()))
        return x

    def prior_flow(self, seg: Segments, x, style, prior_noise, return_z=False):
        dh = self.cfg.decoder.hidden_dim
        mel = self._f32(seg.rows, dh)
        zp = self._f32(seg.rows, dh // 4) if return_z else None
        zf = self._f32(seg.rows, dh // 4) if return_z else None
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_prior_flow_forward(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], _ptr(style),
                                                    _ptr(prior_noise), _ptr(mel), dh, _ptr(zp), _ptr(zf), _ptr(ws), ws.numel()))
        return (mel, zp, zf) if return_z else mel

    def harmonic_stft(self, seg: Segments, pitch, src_noise, init_phase, batch_scope=True, return_signal=False):
        spec = self._f32(seg.rows, self.har_ld)
        phase = self._f32(seg.rows, self.har_ld)
        sig = self._f32(seg.rows * 75) if return_signal else None
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_harmonic_stft(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(pitch), _ptr(src_noise),
                                               _ptr(init_phase), int(batch_scope), _ptr(sig), _ptr(spec), _ptr(phase), self.har_ld, _ptr(ws),
                                               ws.numel()))
        return (spec, phase, sig) if return_signal else (spec, phase)

    def vocoder(self, seg: Segments, mel, style, har_spec, har_phase, return_spec=False):
        audio = self._f32(seg.rows * 75)
        la = self._f32(seg.rows, N_BINS_LD) if return_spec else None
        ph = self._f32(seg.rows, N_BINS_LD) if return_spec else None
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_vocoder_forward(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(mel), mel.shape[1], _ptr(style),
                                                 _ptr(har_spec), _ptr(har_phase), har_spec.shape[1], _ptr(audio), _ptr(la), _ptr(ph), N_BINS_LD,
                                                 _ptr(ws), ws.numel()))
        return (audio, la, ph) if return_spec else audio

    def frame_path(self, seg: Segments, asr, pitch, energy, style, prior_noise, src_noise, init_phase, batch_scope=True, out=None):
        audio = out if out is not None else self._f32(seg.rows * 75)
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_frame_path(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(asr), asr.shape[1], _ptr(pitch),
                                            _ptr(energy), _ptr(style), _ptr(prior_noise), _ptr(src_noise), _ptr(init_phase), int(batch_scope),
                                            _ptr(audio), _ptr(ws), ws.numel(), seg.flags))
        return audio

    # ------------------------------------------------------------------ layout bridge + single ops
    def to_time_major(self, x_bct: torch.Tensor, ld: Optional[int] = None) -> torch.Tensor:
        B, Cc, T = x_bct.shape
        ld = ld or ((Cc + 31) // 32 * 32)
        y = self._f32(B * T, ld)
        _lib.check(self.lib.stts_to_time_major(_stream(), _ptr(x_bct.contiguous()), B, Cc, T, _ptr(y), ld))
        return y

    def to_channel_major(self, x: torch.Tensor, B: int, Cc: int, T: int) -> torch.Tensor:
        y = self._f32(B, Cc, T)
        _lib.check(self.lib.stts_to_channel_major(_stream(), _ptr(x), x.shape[1], B, Cc, T, _ptr(y)))
        return y

    def op_conv1d(self, seg: Segments, x, cin, w: np.ndarray, bias: Optional[np.ndarray], dil=1, act=0, force_tile=0, precision: Optional[str] = None):
        cout, _, k = w.shape
        ldy = (cout + 31) // 32 * 32
        y = torch.zeros(seg.rows, ldy, dtype=torch.float32, device=self.device)
        w = np.ascontiguousarray(w, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        _lib.check(self.lib.stts_op_conv1d(_stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], cin, w.ctypes.data_as(C.c_void_p),
                                           None if b is None else b.ctypes.data_as(C.c_void_p), cout, k, dil, act, _ptr(y), ldy, force_tile,
                                           self.PRECISIONS[precision or self.precision]))
        return y

    def op_adain_block(self, prefix: str, seg: Segments, x, cin, cout, style):
        y = self._f32(seg.rows, cout)
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_op_adain_block(self.ctx, _stream(), prefix.encode(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], cin,
                                                cout, _ptr(style), _ptr(y), cout, _ptr(ws), ws.numel()))
        return y

    def op_attention(self, seg_q: Segments, seg_k: Segments, q, k, v, heads: int, kc: int, band_centre=None, window: int = 0, kernel: int = 0):
        """Packed multi-head attention (test surface): q [q rows, heads * kc], k / v [k rows, heads * kc] -> [q rows, heads * kc]."""
        o = self._f32(seg_q.rows, heads * kc)
        _lib.check(self.lib.stts_op_attention(_stream(), seg_q.n, seg_q.host_ptr, _ptr(seg_q.dev), seg_k.host_ptr, _ptr(seg_k.dev), _ptr(q), _ptr(k), _ptr(v),
                                              _ptr(o), heads, kc, None if band_centre is None else _ptr(band_centre), window, kernel))
        return o

    def op_mrf_block(self, prefix: str, seg: Segments, x, channels, kernel, style):
        y = self._f32(seg.rows, channels)
        ws = self.workspace(seg)
        _lib.check(self.lib.stts_op_mrf_block(self.ctx, _stream(), prefix.encode(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], channels,
                                              kernel, _ptr(style), _ptr(y), channels, _ptr(ws), ws.numel()))
        return y

    # ------------------------------------------------------------------ conv-form STFT of the ONNX export (models/stft.py)
    def conv_stft_transform(self, seg_frames: Segments, wave: torch.Tensor, hop: int):
        """wave: packed samples, utterance u has (frames_u - 1) * hop of them -> mag, x, y [frames, 1056] time-major."""
        mag, x, y = (self._f32(seg_frames.rows, N_BINS_LD) for _ in range(3))
        _lib.check(self.lib.stts_conv_stft_transform(self.ctx, _stream(), seg_frames.n, seg_frames.host_ptr, _ptr(seg_frames.dev), _ptr(wave), hop,
                                                     _ptr(mag), _ptr(x), _ptr(y), N_BINS_LD))
        return mag, x, y

    def conv_stft_inverse(self, seg_frames: Segments, mag, x, y, hop: int):
        out = self._f32((seg_frames.rows - seg_frames.n) * hop)
        ws = self._f32(seg_frames.rows * 1200)
        _lib.check(self.lib.stts_conv_stft_inverse(self.ctx, _stream(), seg_frames.n, seg_frames.host_ptr, _ptr(seg_frames.dev), _ptr(mag), _ptr(x), _ptr(y),
                                                   mag.shape[1], hop, _ptr(out), _ptr(ws), ws.numel() * 4))
        return out

    # ------------------------------------------------------------------ phoneme-rate stages (packed tokens)
    def _ph_ws(self, n_tok: int, n_frames: int, n_utt: int) -> torch.Tensor:
        """Grow-only phoneme-stage workspace, one per launch stream (stages on different streams may run concurrently)."""
        need = int(self.lib.stts_phoneme_workspace_bytes(self.ctx, n_tok, n_frames, n_utt))
        key = torch.cuda.current_stream(self.device).cuda_stream
        with self._ws_lock:
            ws = self._pws.get(key)
            if ws is None or ws.numel() < need:
                ws = self._pws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def text_encoder(self, which: int, seg: Segments, tokens: torch.Tensor, return_hidden=False):
        """tokens int64 [n_tok] (packed) -> mu [n_tok, inter] (+ last hidden [n_tok, 128])."""
        inter = self.cfg.pitch_energy_predictor.inter_dim if which == 2 else self.cfg.inter_dim
        mu = self._f32(seg.rows, inter)
        xh = self._f32(seg.rows, self.cfg.text_encoder.hidden_dim) if return_hidden else None
        ws = self._ph_ws(seg.rows, 0, seg.n)
        _lib.check(self.lib.stts_text_encoder_forward(self.ctx, _stream(), which, seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(tokens), _ptr(mu), inter,
                                                      _ptr(xh), _ptr(ws), ws.numel()))
        return (mu, xh) if return_hidden else mu

    def text_style(self, which: int, seg: Segments, x: torch.Tensor):
        style = self._f32(seg.n, self.cfg.style_dim)
        ws = self._ph_ws(seg.rows, 0, seg.n)
        _lib.check(self.lib.stts_text_style_forward(self.ctx, _stream(), which, seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], _ptr(style),
                                                    _ptr(ws), ws.numel()))
        return style

    def duration(self, seg: Segments, tokens: torch.Tensor, taps=False):
        """-> logits [n_tok,16], dur int32 [n_tok] (+ dict of taps)."""
        logits = self._f32(seg.rows, 16)
        dur = torch.empty(seg.rows, dtype=torch.int32, device=self.device)
        t = None
        if taps:
            t = dict(text_mu=self._f32(seg.rows, self.cfg.inter_dim), style=self._f32(seg.n, self.cfg.style_dim),
                     prosody=self._f32(seg.rows, self.cfg.inter_dim + self.cfg.style_dim))
        ws = self._ph_ws(seg.rows, 0, seg.n)
        _lib.check(self.lib.stts_duration_forward(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(tokens), _ptr(logits), _ptr(dur),
                                                  _ptr(t["text_mu"]) if t else None, _ptr(t["style"]) if t else None,
                                                  _ptr(t["prosody"]) if t else None, _ptr(ws), ws.numel()))
        return (logits, dur, t) if taps else (logits, dur)

    def pitch_energy(self, seg_p: Segments, seg_t: Segments, dur: torch.Tensor, pe_enc: torch.Tensor, pe_style: torch.Tensor, taps=False):
        """dur int32 [n_tok]; -> f0, energy [n_frames] at the mel-frame rate."""
        f0 = self._f32(seg_t.rows)
        en = self._f32(seg_t.rows)
        C = self.cfg.pitch_energy_predictor.inter_dim + self.cfg.style_dim
        t = dict(prosody=self._f32(seg_p.rows, C), cross=self._f32(seg_t.rows, C)) if taps else None
        ws = self._ph_ws(seg_p.rows, seg_t.rows, seg_p.n)
        _lib.check(self.lib.stts_pitch_energy_forward(self.ctx, _stream(), seg_p.n, seg_p.host_ptr, _ptr(seg_p.dev), seg_t.host_ptr, _ptr(seg_t.dev),
                                                      _ptr(dur), _ptr(pe_enc), pe_enc.shape[1], _ptr(pe_style), _ptr(f0), _ptr(en),
                                                      _ptr(t["prosody"]) if t else None, _ptr(t["cross"]) if t else None, _ptr(ws), ws.numel(), seg_t.flags))
        return (f0, en, t) if taps else (f0, en)

    def length_regulate(self, seg_p: Segments, seg_f: Segments, dur: torch.Tensor, rep: int, enc: torch.Tensor, C: int):
        """enc [n_tok, ld] -> [n_frames, C] rows gathered by the duration alignment at rate rep (1 or 4)."""
        out = self._f32(seg_f.rows, C)
        idx = torch.empty(seg_f.rows, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.stts_length_regulate(self.ctx, _stream(), seg_p.n, _ptr(dur), _ptr(seg_p.dev), _ptr(seg_f.dev), seg_f.rows, rep, _ptr(enc),
                                                 enc.shape[1], C, _ptr(out), C, _ptr(idx)))
        return out

    def upsample4(self, seg_t: Segments, seg_t4: Segments, x: torch.Tensor):
        y = self._f32(seg_t4.rows)
        _lib.check(self.lib.stts_upsample4(self.ctx, _stream(), seg_t.n, seg_t.host_ptr, _ptr(seg_t.dev), _ptr(seg_t4.dev), _ptr(x), _ptr(y)))
        return y
