"""Side experiment: tokens -> waveform (duration + pitch/energy predictors + speech predictor) throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.pipeline import Synthesizer
from stylish_tts_amd.runtime import HipModel
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
syn = Synthesizer(eng, adapt=True)
for B, P in ((1, 14), (8, 14), (64, 14), (8, 50)):
    toks = [synth.tokens(f"fc.{B}.{i}", 1, P, 178)[0].tolist() for i in range(B)]
    waves, det = syn(toks, return_details=True)
    secs = sum(w_.numel() for w_ in waves) / 24000
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 10
    for _ in range(n): syn(toks)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d} P={P:3d}: {dt*1e3:8.2f} ms/call  {B/dt:8.1f} utt/s  audio {secs:7.1f} s/call  {secs/dt:8.0f}x real time", flush=True)

# several batches in flight (Synthesizer.map): the phoneme-rate stages and the host read of one batch overlap another's frame path
for B, P in ((1, 14), (8, 14), (8, 50)):
    batches = [[synth.tokens(f"fm.{B}.{j}.{i}", 1, P, 178)[0].tolist() for i in range(B)] for j in range(12)]
    for workers in (1, 2, 3):
        syn.map(batches[:workers * 2], workers=workers)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = syn.map(batches, workers=workers)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(batches)
        secs = sum(w_.numel() for o in out for w_ in o) / 24000 / len(batches)
        print(f"map B={B:3d} P={P:3d} workers={workers}: {dt*1e3:8.2f} ms/batch  {B/dt:8.1f} utt/s  {secs/dt:8.0f}x real time", flush=True)
