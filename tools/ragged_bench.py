"""Side experiment: frame path on a mixed-length batch (cfg4 shape: 0.25-10 s utterances) vs the same audio as equal lengths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
cfg = load_model_config()
w = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
eng = HipModel(cfg, 0, precision=os.environ.get("PREC", "f32")); eng.load_weights({"speech_predictor": w}, which=7)
dev = eng.device
rng = np.random.default_rng(0)
n = int(os.environ.get("N", 32))
secs = rng.uniform(0.25, 10.0, n)
T4r = [max(16, int(s * 320)) for s in secs]          # vocoder frames (hop 75 @ 24 kHz = 320 per second)
T4u = [int(round(sum(T4r) / n))] * n
for name, T4 in (("ragged", T4r), ("ragged, sorted", sorted(T4r)), ("uniform", T4u)):
    seg = Segments(T4, dev); R = seg.rows
    g = torch.Generator(device="cpu").manual_seed(1)
    asr = torch.randn(R, 128, generator=g).to(dev); pitch = (torch.rand(R, generator=g) * 100 + 120).to(dev); energy = (torch.rand(R, generator=g) * 2 + 2).to(dev)
    style = (torch.randn(n, 64, generator=g) * 0.7).to(dev); pn = torch.randn(R, 128, generator=g).to(dev); sn = torch.randn(R * 75, generator=g).to(dev); ph = torch.rand(1, generator=g).to(dev)
    for _ in range(2): eng.frame_path(seg, asr, pitch, energy, style, pn, sn, ph, batch_scope=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.frame_path(seg, asr, pitch, energy, style, pn, sn, ph, batch_scope=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:16s} n={n} rows={R} audio={R*75/24000:6.1f} s  {dt*1e3:7.2f} ms/step  {R*75/24000/dt:7.0f}x real time  {R/dt/1e6:6.2f} Mframes/s", flush=True)
