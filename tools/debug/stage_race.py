"""Diagnostic: which frame-rate stage gives different results when several calls run concurrently on their own streams / host threads."""
import os, sys
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stylish_tts_amd import modules, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import Segments

cfg = load_model_config()
from stylish_tts_amd.runtime import HipModel
mods = modules.build_inference_modules(cfg, engine=HipModel(cfg, 0, precision=os.environ.get("PREC", "f32")), synthetic_seed=0)
eng = mods["speech_predictor"].engine
for m in mods.values():
    m.engine
devid = eng.device
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
lens = [[128], [195, 264], [264, 320, 130], [320], [120, 186], [197, 155, 330], [264]]
if os.environ.get("BIG"):  # cfg3-sized batches
    lens = [[300] * 16, [250] * 12, [320] * 16, [200] * 8, [304] * 24]
cases = []
for j, L in enumerate(lens):
    L4 = [4 * n for n in L]
    seg = Segments(L4, devid)
    R = seg.rows
    inp = dict(asr=dev(synth.normal(f"sr.asr{j}", (R, cfg.inter_dim))), pitch=dev(np.abs(synth.normal(f"sr.f0{j}", (R,))) * 60 + 120), energy=dev(synth.normal(f"sr.en{j}", (R,))),
               style=dev(synth.normal(f"sr.sty{j}", (len(L), cfg.style_dim))), pn=dev(synth.normal(f"sr.pn{j}", (R, 128))), sn=dev(synth.normal(f"sr.sn{j}", (R * 75,))), ph=dev(synth.uniform(f"sr.ph{j}", (1,))))
    cases.append((seg, inp))

ONLY = os.environ.get("ONLY", "").split(",") if os.environ.get("ONLY") else None
def stages(seg, inp, ref=None):
    out = {}
    want = lambda k: ref is None or ONLY is None or k in ONLY
    if want("decoder"):
        out["decoder"] = eng.decoder(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"])
    x = out["decoder"] if ref is None else ref["decoder"]
    if want("prior_flow"):
        out["prior_flow"] = eng.prior_flow(seg, x, inp["style"], inp["pn"])
    if want("harmonic_stft"):
        hs, hp = eng.harmonic_stft(seg, inp["pitch"], inp["sn"], inp["ph"], batch_scope=False)
        out["har_spec"], out["har_phase"] = hs, hp
    mel = out["prior_flow"] if ref is None else ref["prior_flow"]
    hs0, hp0 = (hs, hp) if ref is None else (ref["har_spec"], ref["har_phase"])
    if want("vocoder"):
        out["vocoder"], la, ph = eng.vocoder(seg, mel, inp["style"], hs0, hp0, return_spec=True)
        out["voc_la"], out["voc_ph"] = la[:, :1025].contiguous(), ph[:, :1025].contiguous()
    if want("frame_path"):
        out["frame_path"] = eng.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["pn"], inp["sn"], inp["ph"], batch_scope=False)
    torch.cuda.current_stream().synchronize()
    return out

refs = [stages(s, i) for s, i in cases]
refs2 = [stages(s, i, r) for (s, i), r in zip(cases, refs)]
for r, r2 in zip(refs, refs2):
    for k in r2:
        assert torch.equal(r[k], r2[k]), ("sequential differs", k)
workers = 3
streams = [torch.cuda.Stream(device=devid) for _ in range(workers)]
def run(k):
    torch.cuda.set_device(devid)
    res = []
    with torch.cuda.stream(streams[k]):
        for j in range(k, len(cases), workers):
            res.append((j, stages(cases[j][0], cases[j][1], refs[j])))
    return res
pool = ThreadPoolExecutor(max_workers=workers)
from collections import Counter
bad = Counter()
for rep in range(int(os.environ.get("REPS", 10))):
    for f in [pool.submit(run, k) for k in range(workers)]:
        for j, out in f.result():
            for k in out:
                if not torch.equal(out[k], refs[j][k]):
                    bad[k] += 1
                    d = (out[k] - refs[j][k]).abs()
                    extra = ""
                    if d.dim() == 2:
                        nz = torch.nonzero(d > 0)
                        extra = f" rows {nz[:,0].min().item()}..{nz[:,0].max().item()} of {d.shape[0]} cols {nz[:,1].min().item()}..{nz[:,1].max().item()} count {nz.shape[0]} nan {torch.isnan(out[k]).sum().item()}"
                    if d.dim() == 1:
                        nz = torch.nonzero(d > 0).flatten()
                        idx = nz[:24].tolist()
                        extra = f" count {nz.numel()} idx {idx} got {[round(v, 4) for v in out[k][nz[:6]].tolist()]} ref {[round(v, 4) for v in refs[j][k][nz[:6]].tolist()]}"
                    print(f"rep {rep} case {j} {k}: max |d| {d.max().item():.3e}" + extra)
print("mismatches per stage:", dict(bad))
