"""Diagnostic: the benchmarked frame path (B x 3 s, side streams on) run N times on the same inputs must give the same bits every time."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stylish_tts_amd import modules, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments

cfg = load_model_config()
prec = os.environ.get("PREC", "f32")
mods = modules.build_inference_modules(cfg, engine=HipModel(cfg, 0, precision=prec), synthetic_seed=0)
eng = mods["speech_predictor"].engine
for m in mods.values():
    m.engine
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 240))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
seg = Segments([4 * T] * B, eng.device)
R = seg.rows
inp = dict(asr=dev(synth.normal("rp.asr", (R, cfg.inter_dim))), pitch=dev(np.abs(synth.normal("rp.f0", (R,))) * 60 + 120), energy=dev(synth.normal("rp.en", (R,))),
           style=dev(synth.normal("rp.sty", (B, cfg.style_dim))), pn=dev(synth.normal("rp.pn", (R, 128))), sn=dev(synth.normal("rp.sn", (R * 75,))), ph=dev(synth.uniform("rp.ph", (1,))))
run = lambda: eng.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["pn"], inp["sn"], inp["ph"], batch_scope=False).clone()
ref = run()
torch.cuda.synchronize()
bad = 0
for i in range(int(os.environ.get("REPS", 100))):
    y = run()
    if not torch.equal(y, ref):
        bad += 1
        d = (y - ref).abs()
        print(f"run {i}: {int((d > 0).sum())} samples differ, max {d.max().item():.3e}")
print(f"{prec} B={B}: {bad} of the runs differ from the first")
