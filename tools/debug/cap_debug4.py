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
def d(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
syn = Synthesizer(eng, frames_per_token=24, adapt=False)
batches = [[synth.tokens(f"map.{j}.{i}", 1, 6 + 3 * ((i + j) % 4), 178)[0].tolist() for i in range(1 + j % 3)] for j in range(7)]
for j, toks in enumerate(batches):
    _, det = syn(toks, return_details=True)
    T = det["frames"]; R4 = 4 * sum(T)
    noise = dict(prior_noise=d(synth.normal(f"map.pn{j}", (R4, 128))), src_noise=d(synth.normal(f"map.sn{j}", (R4 * 75,))), init_phase=d(synth.uniform(f"map.ph{j}", (1,))))
    wv, dd = syn(toks, noise=noise, return_details=True)
    wc = torch.cat(wv)
    st = Segments(T, dev); st4 = st.scaled(4)
    sp = Segments([len(t) for t in toks], dev)
    tk = torch.tensor([v for t in toks for v in t], dtype=torch.int64, device=dev)
    enc = eng.text_encoder(1, sp, tk); style = eng.text_style(1, sp, enc)
    asr = eng.length_regulate(sp, st4, dd["durations"], 4, enc, cfg.inter_dim)
    p4, e4 = eng.upsample4(st, st4, dd["pitch"].contiguous()), eng.upsample4(st, st4, dd["energy"].contiguous())
    ex = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False)
    # exact pitch / energy
    pe_enc = eng.text_encoder(2, sp, tk); pe_style = eng.text_style(2, sp, pe_enc)
    f0e, ene = eng.pitch_energy(sp, st, dd["durations"], pe_enc, pe_style)
    print(j, "L", [len(t) for t in toks], "T", T, "caps", dd["capacities"], "| wave |max|", float(wc.abs().max()), "exact |max|", float(ex.abs().max()), "diff", float((wc - ex).abs().max()),
          "| pitch diff vs exact", float((dd["pitch"] - f0e).abs().max()), "pitch range", float(f0e.min()), float(f0e.max()), "energy diff", float((dd["energy"] - ene).abs().max()), "retries", syn.capacity_retries)
