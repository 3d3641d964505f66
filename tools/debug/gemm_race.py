"""Diagnostic: one contraction (op_conv1d) run concurrently from several host threads / streams vs the same call alone."""
import os, sys
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stylish_tts_amd import synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments

cfg = load_model_config()
eng = HipModel(cfg, 0, os.environ.get("PREC", "f32"))
devid = eng.device
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
shapes = [(512, 1536, 1), (1536, 512, 1), (512, 512, 3), (768, 1024, 7), (640, 512, 1)]
lens = [[512], [780, 1056], [1056, 1280, 520], [1280], [480, 744], [788, 620, 1320], [1056]]
cases = []
for j, L in enumerate(lens):
    seg = Segments(L, devid)
    cin, cout, k = shapes[j % len(shapes)]
    x = dev(synth.normal(f"gr.x{j}", (seg.rows, cin)))
    w = synth.normal(f"gr.w{j}", (cout, cin, k)) * 0.05
    b = synth.normal(f"gr.b{j}", (cout,))
    cases.append((seg, x, cin, w, b))
tile = int(os.environ.get("TILE", 0))
def run(c):
    seg, x, cin, w, b = c
    y = eng.op_conv1d(seg, x, cin, w, b, force_tile=tile)
    torch.cuda.current_stream().synchronize()
    return y
refs = [run(c) for c in cases]
for c, r in zip(cases, refs):
    assert torch.equal(run(c), r)
workers = int(os.environ.get("WORKERS", 3))
streams = [torch.cuda.Stream(device=devid) for _ in range(workers)]
def lane(k):
    torch.cuda.set_device(devid)
    out = []
    with torch.cuda.stream(streams[k]):
        for j in range(k, len(cases), workers):
            out.append((j, run(cases[j])))
    return out
pool = ThreadPoolExecutor(max_workers=workers)
bad = 0
for rep in range(int(os.environ.get("REPS", 30))):
    for f in [pool.submit(lane, k) for k in range(workers)]:
        for j, y in f.result():
            if not torch.equal(y, refs[j]):
                bad += 1
                d = (y - refs[j]).abs()
                nz = torch.nonzero(d > 0)
                print(f"rep {rep} case {j} shape {shapes[j % len(shapes)]}: max |d| {d.max().item():.3e} rows {nz[:,0].min().item()}..{nz[:,0].max().item()} cols {nz[:,1].min().item()}..{nz[:,1].max().item()} count {nz.shape[0]}")
print("mismatches:", bad)
