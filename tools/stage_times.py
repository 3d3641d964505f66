"""Side experiment: per-stage GPU time of the tokens -> waveform chain (HIP events around each Synthesizer stage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stylish_tts_amd import params, synth
from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.runtime import HipModel, Segments
cfg = load_model_config()
w = {m: params.synth_state_dict(params.module_spec(m, cfg), 0, prefix=m + ".") for m in params.MODULE_SPECS}
eng = HipModel(cfg, 0); eng.load_weights(w, which=255)
dev = eng.device
for B, P in ((1, 14), (8, 14), (8, 50), (64, 14)):
    token_lists = [synth.tokens(f"fc.{B}.{i}", 1, P, 178)[0].tolist() for i in range(B)]
    L = [len(t) for t in token_lists]
    toks = torch.tensor([int(v) for t in token_lists for v in t], dtype=torch.int64, device=dev)
    sp = Segments(L, dev)
    acc = {}
    for rep in range(6):
        ev = []
        def mark(name):
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, e))
        mark("start")
        _, dur = eng.duration(sp, toks); mark("duration")
        csum = torch.cumsum(dur, 0)
        ends = csum[torch.as_tensor(sp.host[1:].astype(np.int64) - 1, device=dev)]
        T = [int(v) for v in torch.diff(ends, prepend=torch.zeros(1, dtype=ends.dtype, device=dev)).cpu().tolist()]; mark("host sync")
        st = Segments(T, dev); st4 = st.scaled(4)
        pe_enc = eng.text_encoder(2, sp, toks); mark("pe_text_encoder")
        pe_style = eng.text_style(2, sp, pe_enc); mark("pe_text_style")
        f0, en = eng.pitch_energy(sp, st, dur, pe_enc, pe_style); mark("pitch_energy")
        enc = eng.text_encoder(1, sp, toks); mark("text_encoder")
        style = eng.text_style(1, sp, enc); mark("text_style")
        asr = eng.length_regulate(sp, st4, dur, 4, enc, cfg.inter_dim)
        p4, e4 = eng.upsample4(st, st4, f0), eng.upsample4(st, st4, en); mark("regulate+upsample")
        R = st4.rows
        noise = dict(prior_noise=torch.randn(R, 128, device=dev), src_noise=torch.randn(R * 75, device=dev), init_phase=torch.rand(1, device=dev)); mark("noise")
        audio = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False); mark("frame_path")
        torch.cuda.synchronize()
        if rep >= 2:
            for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 4
    print(f"B={B} P={P} frames={sum(T)}: " + "  ".join(f"{k} {v:.2f}" for k, v in acc.items()) + f"  | total {sum(acc.values()):.2f} ms", flush=True)
