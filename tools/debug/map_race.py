"""Diagnostic: Synthesizer.map vs sequential calls, repeated; prints where the two differ (tests/test_hip_modules.py::test_synthesizer_map_equals_sequential_calls)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stylish_tts_amd import modules, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.pipeline import Synthesizer

cfg = load_model_config()
mods = modules.build_inference_modules(cfg, synthetic_seed=0)
eng = mods["speech_predictor"].engine
for m in mods.values():
    m.engine
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
syn = Synthesizer(eng, frames_per_token=24, adapt=False)
batches = [[synth.tokens(f"map.{j}.{i}", 1, 6 + 3 * ((i + j) % 4), 178)[0].tolist() for i in range(1 + j % 3)] for j in range(7)]
noises = []
for j, b in enumerate(batches):
    _, det = syn(b, return_details=True)
    R4 = 4 * sum(det["frames"])
    print("batch", j, "tokens", [len(t) for t in b], "frames", det["frames"], "caps", det["capacities"])
    noises.append(dict(prior_noise=dev(synth.normal(f"map.pn{j}", (R4, 128))), src_noise=dev(synth.normal(f"map.sn{j}", (R4 * 75,))), init_phase=dev(synth.uniform(f"map.ph{j}", (1,)))))
seq = [syn(b, noise=nz) for b, nz in zip(batches, noises)]
seq2 = [syn(b, noise=nz) for b, nz in zip(batches, noises)]
for a, b in zip(seq, seq2):
    for x, y in zip(a, b):
        assert torch.equal(x, y), "sequential calls differ from each other"
print("retries so far", syn.capacity_retries)
if os.environ.get("SERIAL"):  # lanes keep their own streams / workspaces, but only one call runs at a time
    import threading
    _lk = threading.Lock()
    _orig = syn._run
    def _serial(*a, **k):
        with _lk:
            out = _orig(*a, **k)
            if os.environ["SERIAL"] == "2":
                torch.cuda.synchronize()
            return out
    syn._run = _serial
for rep in range(int(os.environ.get("REPS", 6))):
    for workers in (2, 3):
        par = syn.map(batches, workers=workers, noise=noises)
        torch.cuda.synchronize()
        for j, (a, b) in enumerate(zip(seq, par)):
            for i, (x, y) in enumerate(zip(a, b)):
                if not torch.equal(x, y):
                    d = (x - y).abs()
                    nz = torch.nonzero(d > 0).flatten()
                    print(f"rep {rep} workers {workers} batch {j} utt {i}: len {x.numel()} max |d| {d.max().item():.3e} first {nz[0].item()} last {nz[-1].item()} count {nz.numel()} nan {torch.isnan(y).sum().item()}")
print("done; retries", syn.capacity_retries)

# ---- which stage differs: concurrent calls with details
if os.environ.get("STAGES"):
    from concurrent.futures import ThreadPoolExecutor
    ref = [syn(b, noise=nz, return_details=True) for b, nz in zip(batches, noises)]
    workers = 3
    while len(syn._lanes) < workers:
        syn._lanes.append(syn._new_lane(own_stream=True))
    def run(i):
        lane = syn._lanes[i % workers]
        torch.cuda.set_device(eng.device)
        with torch.cuda.stream(lane["main"]):
            out = syn._run(batches[i], noises[i], True, lane)
            torch.cuda.current_stream().synchronize()
        return out
    pool = ThreadPoolExecutor(max_workers=workers)
    for rep in range(8):
        futs = [pool.submit(lambda k=k: [run(i) for i in range(k, len(batches), workers)]) for k in range(workers)]
        per = [f.result() for f in futs]
        got = [per[i % workers][i // workers] for i in range(len(batches))]
        for j, ((w0, d0), (w1, d1)) in enumerate(zip(ref, got)):
            msg = []
            for k in ("durations", "pitch", "energy", "style"):
                if not torch.equal(d0[k], d1[k]):
                    msg.append(f"{k} {(d0[k].float() - d1[k].float()).abs().max().item():.3e}")
            wd = max((x - y).abs().max().item() for x, y in zip(w0, w1))
            if msg or wd > 0:
                print(f"rep {rep} batch {j}: audio {wd:.3e}", *msg)
    print("stages done")
