import os, sys, random
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
rnd = random.Random(7)
for it in range(24):
    B = rnd.choice([1, 2, 3, 5, 8])
    toks = [synth.tokens(f"soak.{it}.{i}", 1, rnd.randint(4, 40), 178)[0].tolist() for i in range(B)]
    waves, det = syn(toks, return_details=True)
    mx = [float(w_.abs().max()) for w_ in waves]
    if max(mx) >= 1.0:
        T = det["frames"]; R4 = 4 * sum(T)
        g = torch.Generator(device=dev); g.manual_seed(it)
        noise = dict(prior_noise=torch.randn(R4, 128, generator=g, device=dev), src_noise=torch.randn(R4 * 75, generator=g, device=dev), init_phase=torch.rand(1, generator=g, device=dev))
        wc, dc = syn(toks, noise=noise, return_details=True)
        # exact-offset composition with the same noise and the SAME pitch / energy (teacher-forced: pitch integrates over the utterance)
        st = Segments(T, dev); st4 = st.scaled(4)
        sp = Segments([len(t) for t in toks], dev)
        tk = torch.tensor([v for t in toks for v in t], dtype=torch.int64, device=dev)
        enc = eng.text_encoder(1, sp, tk); style = eng.text_style(1, sp, enc)
        asr = eng.length_regulate(sp, st4, dc["durations"], 4, enc, cfg.inter_dim)
        p4, e4 = eng.upsample4(st, st4, dc["pitch"].contiguous()), eng.upsample4(st, st4, dc["energy"].contiguous())
        ex = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False)
        print(it, "max", [round(m, 4) for m in mx], "| same noise: capacity max", float(torch.cat(wc).abs().max()), "exact max", float(ex.abs().max()), "diff", float((torch.cat(wc) - ex).abs().max()), "retries", syn.capacity_retries)
print("done; retries", syn.capacity_retries, "ratio", syn._ratio)
