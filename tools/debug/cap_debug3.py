import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
from stylish_tts_amd.pipeline import Synthesizer
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
dev = eng.device
syn = Synthesizer(eng)
toks = [synth.tokens("syo.a", 1, 11, 178)[0].tolist(), synth.tokens("syo.b", 1, 17, 178)[0].tolist(), synth.tokens("syo.c", 1, 29, 178)[0].tolist()]
_, det = syn(toks, return_details=True)
T = det["frames"]; R4 = 4 * sum(T)
print("T", T, "caps", det["capacities"], "ratio", syn._ratio, "retries", syn.capacity_retries)
g = torch.Generator(device=dev); g.manual_seed(1)
noise = dict(prior_noise=torch.randn(R4, 128, generator=g, device=dev), src_noise=torch.randn(R4 * 75, generator=g, device=dev), init_phase=torch.rand(1, generator=g, device=dev))
outs = []
for k in range(3):
    wv, d = syn(toks, noise=noise, return_details=True)
    outs.append((torch.cat(wv).clone(), d["pitch"].clone(), d["energy"].clone(), d["capacities"]))
    print(k, "caps", d["capacities"], "ratio", syn._ratio)
for k in (1, 2):
    print("call", k, "vs 0: wave diff", float((outs[k][0] - outs[0][0]).abs().max()), "pitch diff", float((outs[k][1] - outs[0][1]).abs().max()), "energy diff", float((outs[k][2] - outs[0][2]).abs().max()))
# exact composition, teacher-forced with call 0's pitch / energy
st = Segments(T, dev); st4 = st.scaled(4)
sp = Segments([len(t) for t in toks], dev)
tk = torch.tensor([v for t in toks for v in t], dtype=torch.int64, device=dev)
enc = eng.text_encoder(1, sp, tk); style = eng.text_style(1, sp, enc)
asr = eng.length_regulate(sp, st4, d["durations"], 4, enc, cfg.inter_dim)
p4, e4 = eng.upsample4(st, st4, outs[0][1].contiguous()), eng.upsample4(st, st4, outs[0][2].contiguous())
ex = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False)
print("capacity call 0 vs exact composition (teacher-forced pitch): wave diff", float((outs[0][0] - ex).abs().max()))
