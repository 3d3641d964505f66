"""CfmMelDecoder: drop-in for the reference's flow-matching mel decoder at inference (``models/cfm/cfm_mel_decoder.py:190-413``).

Same constructor keywords, ``load_state_dict`` with the reference's keys, ``_forward(x, asr, F0, N, spk_emb, t)`` (one estimator
evaluation, ``[B, C, n]`` tensors) and ``forward(asr, F0, N, spk_emb, n_timesteps, temperature)`` (Euler sampling, ``CfmSampler``).
All arithmetic runs in libstylish_hip.so (``stts_cfm_estimator``); torch provides device memory and the random draws the reference
makes with ``torch.rand`` / ``torch.randn_like`` (the initial state ``z`` and the SineGenerator's additive noise, ``:99,402``), both
injectable for reproducible comparisons.  Training-only pieces (TREAD token dropout, ``compute_pred_target``) are not part of it.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import numpy as np
import torch

from . import _lib
from .euler_sampler import CfmSampler
from .runtime import HipModel, Segments, _ptr, _stream


class CfmMelDecoder:
    def __init__(self, feat_dim=80, asr_dim=768, spk_dim=1024, hidden_dim=256, emb_dim=256, xut_depth=4, xut_heads=8, xut_enc_blocks=1,
                 xut_dec_blocks=2, tread_config=None, device: int = 0, precision: str = "f32"):
        tread_config = {"prev_trns_depth": 1, "post_trns_depth": 3, "dropout_ratio": 0.5} if tread_config is None else tread_config
        # (xut_heads is accepted and ignored like in the reference: XUTBackBone / TBackBone are built with dim_head = 64, so the
        #  number of heads is hidden_dim / 64 whatever `heads` says, xut/attention.py:14-21)
        self.dims = _lib.CfmDims(feat_dim, asr_dim, spk_dim, hidden_dim, emb_dim, xut_depth, xut_enc_blocks, xut_dec_blocks,
                                 int(tread_config["prev_trns_depth"]), int(tread_config["post_trns_depth"]), 64)
        self.model = HipModel(device=device, precision=precision)
        self.lib, self.ctx, self.device = self.model.lib, self.model.ctx, self.model.device
        self.feat_dim, self.hidden_dim = feat_dim, hidden_dim
        self.sampler = CfmSampler(self._forward, non_drop_conds=["spk_emb"])
        self._ws: Optional[torch.Tensor] = None
        self._ready = False

    @classmethod
    def from_model_config(cls, cfg=None, **kw):
        """The instance build_model makes (models/models.py:65-70): feat_dim = n_mels, asr_dim = hubert.hidden_dim,
        spk_dim = speaker_embedder.hidden_dim, hidden_dim = decoder.hidden_dim; everything else at the class defaults."""
        from .config import load_model_config

        cfg = cfg if cfg is not None else load_model_config()
        return cls(feat_dim=cfg.n_mels, asr_dim=cfg.hubert.hidden_dim, spk_dim=cfg.speaker_embedder.hidden_dim, hidden_dim=cfg.decoder.hidden_dim, **kw)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Mapping[str, "np.ndarray | torch.Tensor"]):
        self.model.load_state_dict("cfm_mel_decoder", sd)
        _lib.check(self.lib.stts_cfm_finalize(self.ctx, C.byref(self.dims)))
        self._ready = True
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ one estimator evaluation on packed rows
    def estimator_packed(self, seg: Segments, x, asr, f0, n_curve, curve_seg: Segments, spk_emb, t, sine_noise):
        """x [rows, >= feat], asr [rows, ld % 32 == 0], f0 / n_curve [sum L], spk_emb [n_utt, spk], t [n_utt], sine_noise [rows]."""
        if not self._ready:
            raise RuntimeError("CfmMelDecoder: load_state_dict first")
        out = torch.empty(seg.rows, self.feat_dim, dtype=torch.float32, device=self.device)
        need = int(self.lib.stts_cfm_workspace_bytes(self.ctx, seg.rows, seg.n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.stts_cfm_estimator(self.ctx, _stream(), seg.n, seg.host_ptr, _ptr(seg.dev), _ptr(x), x.shape[1], _ptr(asr), asr.shape[1],
                                               _ptr(f0), _ptr(n_curve), curve_seg.host_ptr, _ptr(curve_seg.dev), _ptr(spk_emb), _ptr(t), _ptr(sine_noise),
                                               _ptr(out), self.feat_dim, _ptr(self._ws), self._ws.numel()))
        return out

    # ------------------------------------------------------------------ the reference's interface ([B, C, n] tensors)
    def _rows(self, x_bct, ld):
        B, Cc, n = x_bct.shape
        y = torch.zeros(B * n, ld, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.stts_to_time_major(_stream(), _ptr(x_bct.contiguous()), B, Cc, n, _ptr(y), ld))
        return y

    @torch.no_grad()
    def _forward(self, x, asr, F0, N, spk_emb, t, mask=None, sine_noise=None):
        f32 = lambda a: a.detach().to(self.device, torch.float32).contiguous()  # noqa: E731
        x, asr, F0, N, spk_emb, t = map(f32, (x, asr, F0, N, spk_emb, t))
        B, _, n = x.shape
        L = F0.shape[-1]
        seg, cseg = Segments([n] * B, self.device), Segments([L] * B, self.device)
        if sine_noise is None:  # SineGenerator: noise_amp * torch.randn_like(sine_waves)
            sine_noise = torch.randn(B, n, 1, device=self.device)
        ld_asr = (asr.shape[1] + 31) // 32 * 32
        out = self.estimator_packed(seg, self._rows(x, self.feat_dim), self._rows(asr, ld_asr), F0.reshape(-1), N.reshape(-1), cseg, spk_emb,
                                    t.reshape(-1), f32(sine_noise).reshape(-1))
        y = torch.empty(B, self.feat_dim, n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.stts_to_channel_major(_stream(), _ptr(out), self.feat_dim, B, self.feat_dim, n, _ptr(y)))
        return y

    @torch.no_grad()
    def forward(self, asr, F0, N, spk_emb, n_timesteps, temperature, z=None, sine_noise=None, graph=False):
        """sine_noise: optional list of per-step draws ([B, n, 1] each) replacing the estimator's own randn.
        graph=True: the ~250 launches of one estimator evaluation are captured once into a HIP graph and replayed per Euler step
        (the loop is launch-bound at small batches); same kernels, same arithmetic."""
        b, _, n = asr.shape
        if z is None:
            z = torch.rand((b, self.feat_dim, n), device=self.device)  # (uniform, as the reference draws it, :402)
        z = z.to(self.device, torch.float32)
        if graph:
            return self._forward_graph(asr, F0, N, spk_emb, n_timesteps, temperature, z, sine_noise)
        if sine_noise is None:
            return self.sampler(z, None, n_timesteps, temperature, asr=asr, F0=F0, N=N, spk_emb=spk_emb)
        it = iter(sine_noise)
        est = lambda x, t, mask=None, **kw: self._forward(x, t=t, mask=mask, sine_noise=next(it), **kw)  # noqa: E731
        return CfmSampler(est, non_drop_conds=["spk_emb"])(z, None, n_timesteps, temperature, asr=asr, F0=F0, N=N, spk_emb=spk_emb)

    def _forward_graph(self, asr, F0, N, spk_emb, n_timesteps, temperature, z, sine_noise):
        f32 = lambda a: a.detach().to(self.device, torch.float32).contiguous()  # noqa: E731
        asr, F0, N, spk_emb = map(f32, (asr, F0, N, spk_emb))
        B, _, n = asr.shape
        seg, cseg = Segments([n] * B, self.device), Segments([F0.shape[-1]] * B, self.device)
        x = self._rows(f32(z) * temperature, self.feat_dim)            # the state stays in time-major rows for the whole solve
        asr_r = self._rows(asr, (asr.shape[1] + 31) // 32 * 32)
        f0, nc = F0.reshape(-1), N.reshape(-1)
        t_buf = torch.zeros(B, dtype=torch.float32, device=self.device)
        nz_buf = torch.zeros(B * n, dtype=torch.float32, device=self.device)
        draws = None if sine_noise is None else iter(sine_noise)

        def next_noise():
            if draws is None:
                nz_buf.normal_()
            else:
                nz_buf.copy_(f32(next(draws)).reshape(-1))

        # one eager evaluation first, ON THE CAPTURE STREAM: lazily created library state (the zero page, the split-K scratch, which is
        # kept per launch stream) must exist before capture - an allocation during capture invalidates it
        next_noise()
        caller = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(caller)
        with torch.cuda.stream(side):
            v = self.estimator_packed(seg, x, asr_r, f0, nc, cseg, spk_emb, t_buf, nz_buf)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            v_static = self.estimator_packed(seg, x, asr_r, f0, nc, cseg, spk_emb, t_buf, nz_buf)
        caller.wait_stream(side)
        v.record_stream(caller)
        ts = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)  # CfmSampler.solve_euler's grid and update order (cfm.py:65-84)
        t, dt = ts[0].clone(), ts[1] - ts[0]
        for step in range(1, len(ts)):
            t_buf.fill_(float(t))
            if step == 1:
                v_use = v  # the eager evaluation above already is step 1 (t = 0, the first draw)
            else:
                next_noise()
                g.replay()
                v_use = v_static
            _lib.check(self.lib.stts_euler_step(_stream(), _ptr(x), _ptr(v_use), C.c_float(float(dt)), x.numel()))
            t = t + dt
            if step < len(ts) - 1:
                dt = ts[step + 1] - t
        y = torch.empty(B, self.feat_dim, n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.stts_to_channel_major(_stream(), _ptr(x), self.feat_dim, B, self.feat_dim, n, _ptr(y)))
        return y

    __call__ = forward
