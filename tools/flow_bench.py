"""Time the prior + reverse-flow stage (stts_prior_flow_forward: 32 fused WaveNet-layer launches + 3 small contractions)
for a list of batch sizes; STTS_WN_M=2|4 forces the F(2,5) / F(4,5) block shape, STTS_NO_WN_FUSED=1 the staged kernel.
usage: python tools/flow_bench.py [batch ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from stylish_tts_amd import params, synth  # noqa: E402
from stylish_tts_amd.config import load_model_config  # noqa: E402
from stylish_tts_amd.runtime import HipModel, Segments  # noqa: E402

cfg = load_model_config()
sd = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
m = HipModel(cfg, 0)
m.load_weights({"speech_predictor": sd}, which=7)
T4 = int(os.environ.get("T4", "960"))
for B in [int(v) for v in sys.argv[1:]] or [1, 4, 8, 16, 64]:
    seg = Segments([T4] * B, m.device)
    R = B * T4
    x = torch.from_numpy(synth.normal("fb.x", (R, 512))).cuda()
    style = torch.from_numpy((synth.normal("fb.s", (B, 64)) * 0.7).astype(np.float32)).cuda()
    pn = torch.from_numpy(synth.normal("fb.pn", (R, 128))).cuda()
    for _ in range(3):
        m.prior_flow(seg, x, style, pn)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        m.prior_flow(seg, x, style, pn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gflop = R * 12.71e6 / 1e9
    print(f"B={B:3d} rows={R:6d}  flow {dt * 1e3:7.3f} ms  ({dt * 1e6 / 32:6.1f} us per WaveNet layer, {gflop / dt / 1e3:6.1f} TFLOP/s of direct-conv flops) "
          f"M={os.environ.get('STTS_WN_M', 'auto')} fused={'no' if os.environ.get('STTS_NO_WN_FUSED') else 'yes'}", flush=True)
