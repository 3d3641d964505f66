import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import stylish_oracle as O
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
cfg = load_model_config()
w = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
m = HipModel(cfg, 0); m.load_weights({"speech_predictor": w}, 1)
g = load_golden("frame_path_3s")
T4 = 960
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def tm(x, ld=None):
    B, C, T = x.shape; ld = ld or (C + 31)//32*32
    y = np.zeros((B*T, ld), np.float32); y[:, :C] = x.transpose(0,2,1).reshape(B*T, C); return dev(y)
s = Segments([T4], m.device)
asr = synth.normal("g3.asr", (1, 128, T4)); pitch = synth.pitch_curve("g3.pitch", 1, T4)
energy = (synth.uniform("g3.energy", (1, T4)) * 2.0 + 2.0).astype(np.float32); style = (synth.normal("g3.style", (1, 64)) * 0.7).astype(np.float32)
nz = synth.path_noise("frame960", 1, T4)
x = m.decoder(s, tm(asr), dev(pitch[0]), dev(energy[0]), dev(style))
mel = m.prior_flow(s, x, dev(style), tm(nz["prior_noise"]))
spec, phase, sig = m.harmonic_stft(s, dev(pitch[0]), dev(nz["src_noise"].reshape(-1)), dev(nz["init_phase"].reshape(-1)), True, True)
# oracle intermediates
prior = O.generate_pcph(pitch[:, None, :], nz["src_noise"], nz["init_phase"])[:, 0]
print("prior sig err", np.abs(sig.cpu().numpy() - prior[0]).max())
hs, hx, hy = O.stft_transform(prior); hp = np.arctan2(hy, hx)[:, :, :-1]; hs = hs[:, :, :-1]
myspec = spec.cpu().numpy()[:, :1025].T[None]; myph = phase.cpu().numpy()[:, :1025].T[None]
print("spec err", np.abs(myspec - hs).max())
d = myph.astype(np.float64) - hp
fl = np.abs(d) > 1
print("flips vs oracle:", fl.sum(), "in frame0:", fl[..., 0].sum())
idx = np.argwhere(fl & (np.arange(T4)[None, None, :] > 0))
print("interior flips (bin, frame, mag, my, oracle):")
for _, b, f in idx[:20]: print(b, f, hs[0, b, f], myph[0, b, f], hp[0, b, f])
hint = (g["cut_idx"].astype(np.int64), g["cut_sign"].astype(np.float32))
ph_h = O.align_branch(myph.astype(np.float32), hint)
hp_h = O.align_branch(hp.astype(np.float32), hint)
d2 = ph_h.astype(np.float64) - hp_h
print("after hints: flips", (np.abs(d2) > 1).sum(), np.argwhere(np.abs(d2) > 1)[:10])
print("non-flip max phase diff where mag>1e-3:", np.abs(np.where(np.abs(d2) > 1, 0, d2))[hs > 1e-3].max())
audio = m.vocoder(s, mel, dev(style), spec, tm(ph_h, 1056)).cpu().numpy()
e = np.abs(audio - g["audio"][0, 0]).reshape(960, 75).max(1)
print("audio err by 40-frame chunk:", np.array2string(e.reshape(24, 40).max(1), precision=1, max_line_width=250))
# oracle with same hint
a_or, _, _ = O.frame_path(asr, pitch, energy, style, nz, w, branch_hint=hint)
e2 = np.abs(a_or[0, 0] - g["audio"][0, 0]).reshape(960, 75).max(1)
print("oracle err by chunk       :", np.array2string(e2.reshape(24, 40).max(1), precision=1, max_line_width=250))
