import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
dev = eng.device
L = [12, 30, 7]
toks_l = [synth.tokens(f"ovf.{i}", 1, n, 178)[0].tolist() for i, n in enumerate(L)]
toks = torch.tensor([v for t in toks_l for v in t], dtype=torch.int64, device=dev)
sp = Segments(L, dev)
_, dur = eng.duration(sp, toks)
pe_enc = eng.text_encoder(2, sp, toks); pe_style = eng.text_style(2, sp, pe_enc)
enc = eng.text_encoder(1, sp, toks); style = eng.text_style(1, sp, enc)
d = dur.cpu().numpy()
T = [int(d[sp.host[i]:sp.host[i+1]].sum()) for i in range(len(L))]
print("T", T)
st_e = Segments(T, dev); st4_e = st_e.scaled(4)
f0_e, en_e, tp_e = eng.pitch_energy(sp, st_e, dur, pe_enc, pe_style, taps=True)
asr_e = eng.length_regulate(sp, st4_e, dur, 4, enc, cfg.inter_dim)
p4_e = eng.upsample4(st_e, st4_e, f0_e)
for mult in (1.0, 1.5, 46.0):
    caps = [max(t, int(np.ceil(t * mult))) if mult < 40 else 46 * n for t, n in zip(T, L)]
    st, st4, need = eng.frame_offsets(sp, dur, caps)
    eng.check_status()
    f0, en, tp = eng.pitch_energy(sp, st, dur, pe_enc, pe_style, taps=True)
    asr = eng.length_regulate(sp, st4, dur, 4, enc, cfg.inter_dim)
    p4 = eng.upsample4(st, st4, f0)
    R = sum(T)
    print(f"caps x{mult}: f0 diff {float((f0[:R]-f0_e).abs().max()):.3e} en diff {float((en[:R]-en_e).abs().max()):.3e} cross diff {float((tp['cross'][:R]-tp_e['cross']).abs().max()):.3e} "
          f"asr diff {float((asr[:4*R]-asr_e).abs().max()):.3e} p4 diff {float((p4[:4*R]-p4_e).abs().max()):.3e}  need {need.cpu().tolist()} offs {st.dev.cpu().tolist()}")
