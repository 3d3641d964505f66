"""CfmMelDecoder on the GPU box: parity figures against the committed reference vectors and time per estimator evaluation / sampling."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.cfm_decoder import CfmMelDecoder

dims = dict(params.CFM_DEFAULT_DIMS)
m = CfmMelDecoder().load_state_dict(params.synth_state_dict(params.cfm_mel_decoder_spec(dims), 0, prefix="cfm_mel_decoder."))
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfm_decoder.npz"))
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
y = m._forward(d(g["default_x"]), d(g["default_asr"]), d(g["default_f0"]), d(g["default_n"]), d(g["default_spk"]), d(g["default_t"]), sine_noise=d(g["default_nz"])).cpu().numpy()
print("estimator vs reference (class defaults, B=2 x 53 frames): max-abs err %.2e of max %.2f" % (np.abs(y - g["default_y"]).max(), np.abs(g["default_y"]).max()))
shapes = ((1, 240), (8, 240), (32, 240), (8, 800))
if os.environ.get("CFM_ONLY"):  # e.g. CFM_ONLY=8x800: one shape (for a rocprofv3 trace of it)
    shapes = (tuple(int(v) for v in os.environ["CFM_ONLY"].split("x")),)
for B, n in shapes:
    x = d(synth.normal("cb.x", (B, 80, n))); asr = d(synth.normal("cb.a", (B, 768, n))); f0 = d(synth.pitch_curve("cb.f", B, n))
    nc = d((synth.uniform("cb.n", (B, n)) * 2 + 2).astype(np.float32)); spk = d(synth.normal("cb.s", (B, 1024))); t = d(np.full((B,), 0.4, np.float32))
    nz = d(synth.normal("cb.z", (B, n, 1)))
    for _ in range(3):
        m._forward(x, asr, f0, nc, spk, t, sine_noise=nz)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m._forward(x, asr, f0, nc, spk, t, sine_noise=nz)
    torch.cuda.synchronize(); ev = (time.perf_counter() - t0) / 10
    steps = 10
    m(asr, f0, nc, spk, steps, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(asr, f0, nc, spk, steps, 1.0)
    torch.cuda.synchronize(); sm = time.perf_counter() - t0
    m(asr, f0, nc, spk, steps, 1.0, graph=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(asr, f0, nc, spk, steps, 1.0, graph=True)
    torch.cuda.synchronize(); sg = time.perf_counter() - t0
    steps2 = 32
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(asr, f0, nc, spk, steps2, 1.0, graph=True)
    torch.cuda.synchronize(); sg2 = time.perf_counter() - t0
    print(f"B={B:3d} x {n} frames ({B * n / 80:.0f} s of mel at 80 frames/s): {ev * 1e3:7.2f} ms per evaluation, {sm * 1e3:7.1f} ms per {steps}-step sampling = {B * n / 80 / sm:7.0f}x real time;"
          f" HIP graph: {sg * 1e3:7.1f} ms ({B * n / 80 / sg:7.0f}x), {steps2} steps {sg2 * 1e3:7.1f} ms = {(sg2 - sg) / (steps2 - steps) * 1e3:.2f} ms per replayed step")
