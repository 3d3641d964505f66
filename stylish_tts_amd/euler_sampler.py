"""CfmSampler: the fixed-step Euler ODE solver of the reference's flow-matching sampler (``models/cfm/cfm.py:24-84``).

Only the inference side is mirrored (``forward`` / ``solve_euler``); the estimator is any callable with the reference's
contract ``estimator(x, t=[b], mask=mask, **args) -> dphi/dt`` on device tensors.  The reference's own estimator
(``CfmMelDecoder._forward``: XUT transformer on HuBERT / wespeaker features) is ``stylish_tts_amd.cfm_decoder.CfmMelDecoder``
(``csrc/cfm.hip.h``), whose ``forward`` drives this solver; the solver itself is pinned separately: time grid
and update order are reproduced exactly (fp32 ``linspace``, ``t += dt``, ``dt = t_span[step + 1] - t``), the update
``x += dt * v`` runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class CfmSampler:
    def __init__(self, estimator, guidance_w: float = 0.7, cond_drop_prob: float = 0.0, non_drop_conds=(), sigma_min: float = 1e-4):
        self.estimator = estimator
        self.guidance_w = guidance_w          # training-time model guidance: unused at inference, kept for signature parity
        self.cond_drop_prob = cond_drop_prob
        self.non_drop_conds = list(non_drop_conds)
        self.sigma_min = sigma_min
        self.lib = _lib.load()

    @torch.no_grad()
    def forward(self, z, mask, n_timesteps, temperature=1.0, **estimator_args):
        z = z * temperature
        t_span = torch.linspace(0, 1, n_timesteps + 1, device=z.device)
        return self.solve_euler(z, t_span=t_span, mask=mask, **estimator_args)

    __call__ = forward

    @torch.no_grad()
    def solve_euler(self, x, t_span, mask, **estimator_args):
        if x.device.type != "cuda":
            raise RuntimeError("CfmSampler: tensors must live on the GPU (no CPU fallback)")
        x = x.detach().to(torch.float32).contiguous().clone()
        ts = t_span.detach().to("cpu", torch.float32)  # the grid is tiny: step sizes are formed on the host in fp32
        t = ts[0].clone()
        dt = ts[1] - ts[0]
        for step in range(1, len(ts)):
            _t = t.to(x.device).reshape(1).expand(x.shape[0])
            v = self.estimator(x, t=_t, mask=mask, **estimator_args).to(torch.float32).contiguous()
            if v.shape != x.shape:
                raise ValueError(f"estimator returned {tuple(v.shape)}, expected {tuple(x.shape)}")
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(self.lib.stts_euler_step(stream, C.c_void_p(x.data_ptr()), C.c_void_p(v.data_ptr()), C.c_float(float(dt)), x.numel()))
            t = t + dt
            if step < len(ts) - 1:
                dt = ts[step + 1] - t
        return x
