"""Per-kernel times of stts_frame_path on exact offsets vs capacity segments (slack 1.4x) for the same batch."""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stylish_tts_amd import params, synth, _lib
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
cfg = load_model_config()
w = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
eng = HipModel(cfg, 0); eng.load_weights({"speech_predictor": w}, which=7)
dev = eng.device
rng = np.random.default_rng(0)
T4 = [int(4 * rng.integers(230, 330)) for _ in range(64)]
R = sum(T4)
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
def run(seg, Rc, tag):
    grow = lambda x: torch.cat([x, torch.zeros((Rc - x.shape[0],) + tuple(x.shape[1:]), device=dev)]) if Rc > x.shape[0] else x
    asr, pitch, energy = grow(A), grow(P), grow(E)
    pn, sn = grow(PN), torch.cat([SN, torch.zeros(75 * Rc - SN.shape[0], device=dev)])
    out = torch.empty(Rc * 75, device=dev)
    for _ in range(2): eng.frame_path(seg, asr, pitch, energy, S, pn, sn, PH, batch_scope=False, out=out)
    lib = _lib.load(); lib.stts_profile_begin()
    for _ in range(3): eng.frame_path(seg, asr, pitch, energy, S, pn, sn, PH, batch_scope=False, out=out)
    buf = C.create_string_buffer(1 << 17)
    _lib.check(lib.stts_profile_report(C.c_void_p(torch.cuda.current_stream().cuda_stream), buf, len(buf)))
    recs = json.loads(buf.value.decode())
    return {r["kernel"]: (r["launches"] // 3, r["ms"] / 3) for r in recs}
A, P, E = rn(R, 128), 80 + 220 * torch.rand(R, generator=g, device=dev), 2 + 2 * torch.rand(R, generator=g, device=dev)
S, PN, SN, PH = 0.7 * rn(64, 64), rn(R, 128), rn(R * 75), torch.rand(1, generator=g, device=dev)
ex = run(Segments(T4, dev), R, "exact")
caps = [int(np.ceil(t * 1.4 / 4) * 4) for t in T4]
sc = Segments(caps, dev, dev=torch.from_numpy(np.concatenate([[0], np.cumsum(T4)]).astype(np.int32)).to(dev), capacity=True)
cp = run(sc, sc.rows, "cap")
print(f"rows real {R} cap {sc.rows}")
tot_e = tot_c = 0
for k in sorted(set(ex) | set(cp), key=lambda k: -(cp.get(k, (0, 0))[1])):
    e, c = ex.get(k, (0, 0.0)), cp.get(k, (0, 0.0))
    tot_e += e[1]; tot_c += c[1]
    print(f"{k:30s} exact n={e[0]:3d} {e[1]*1e3:9.1f} us | capacity n={c[0]:3d} {c[1]*1e3:9.1f} us  ({c[1]/e[1] if e[1] else 0:.2f}x)")
print(f"total exact {tot_e:.3f} ms  capacity {tot_c:.3f} ms")
