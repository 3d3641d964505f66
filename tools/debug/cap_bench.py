"""Sensitivity of the full chain to the capacity slack (frames-per-token bound) of the host-sync-free Synthesizer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.pipeline import Synthesizer
from stylish_tts_amd.runtime import HipModel
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
for B, P in ((1, 14), (8, 14), (64, 14), (8, 50)):
    toks = [synth.tokens(f"fc.{B}.{i}", 1, P, 178)[0].tolist() for i in range(B)]
    for r in (0.0, 1.0, 1.1, 1.25, 1.6, 2.2):
        syn = Synthesizer(eng, frames_per_token=46, adapt=False)
        _, det = syn(toks, return_details=True)
        T = det["frames"]
        mx = max(t / P for t in T)
        if r == 0.0:
            caps_fn = lambda L, T=T: list(T)  # exact-fit capacities
        else:
            caps_fn = lambda L, r=r, mx=mx: [int(np.ceil(r * mx * n)) + 8 for n in L]
        syn._capacities = caps_fn
        for _ in range(3): syn(toks)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
        for _ in range(n): syn(toks)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        caps = caps_fn([P] * B)
        print(f"B={B:3d} P={P:3d} capacity = {'exact' if r == 0 else f'{r:.2f} x max ratio'}: rows real {4*sum(T):7d} cap {4*sum(caps):7d} ({sum(caps)/sum(T):.2f}x)  {dt*1e3:8.2f} ms/call  retries {syn.capacity_retries}", flush=True)
