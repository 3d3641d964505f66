"""Phoneme-rate stages at the sizes BASELINE's configs and the reference's limits reach:

  * P = 160 tokens (cfg5: 10 s segments) and P = 510 (the reference's hard limit, train/dataloader.py:106-109),
    fp32, every stage against the numpy oracle on the same seeded inputs;
  * cfg3: 64 utterances of 50 tokens in ONE packed call in the bf16 mode: the phoneme-rate stages (always fp32) against the fp32
    oracle with integer-exact durations, the frame path against the oracle whose contraction operands are rounded at the same
    points (oracle.OPERAND_ROUND), teacher-forced stage by stage
    (predicted durations decide the frame count, and pitch is integrated over the utterance: DESIGN.md §5), and the
    `Synthesizer` (tokens -> waveforms in one pass) against that staged composition.

Tolerances: fp32 2e-4 of each tensor's max-abs (durations bit-equal); the bf16 waveform bars are stated next to the check.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def segs(lengths):
    from stylish_tts_amd.runtime import Segments

    return Segments(lengths, torch.device("cuda", 0))


def rel(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


@pytest.fixture(scope="module")
def hip(cfg, weights):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    m.load_weights(weights, which=255)
    yield m
    m.close()


@pytest.mark.parametrize("P,T", [(160, 800), (510, 1020)])
def test_long_token_sequences_vs_oracle(hip, weights, cfg, P, T):
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    toks = synth.tokens(f"long.{P}", 1, P, cfg.text_encoder.tokens)
    lengths = np.array([P], np.int64)
    sp = segs([P])
    t_dev = dev(toks[0])
    # duration predictor: text encoder -> style -> prosody encoder -> 16-class logits -> durations
    logits, dur, taps = hip.duration(sp, t_dev, taps=True)
    lo, mid = O.duration_predictor(toks, lengths, weights["duration_predictor"], cfg, return_intermediates=True)
    errs = dict(text_mu=rel(taps["text_mu"].cpu().numpy().T[None], mid["text_mu"]), style=rel(taps["style"], mid["style"]),
                prosody=rel(taps["prosody"].cpu().numpy()[None], mid["prosody"]), logits=rel(logits.cpu().numpy()[None], lo))
    want = O.prediction_to_duration(lo[0]).astype(np.int32)
    got = dur.cpu().numpy()
    # a logit within fp32 noise of a class boundary may legitimately land on the other side: allow it only there
    diff = np.nonzero(got != want)[0]
    for i in diff:
        srt = np.sort(lo[0, i])[::-1]
        assert srt[0] - srt[1] < 1e-3 * np.abs(lo).max() or abs(int(got[i]) - int(want[i])) <= 1, (i, got[i], want[i])
    assert len(diff) <= max(1, P // 100), diff
    # speech / pitch-energy text encoders and their style encoders
    for which, mod, sty_mod in ((1, "speech_predictor", None), (2, "pe_text_encoder", "pe_text_style_encoder")):
        mu, xh = hip.text_encoder(which, sp, t_dev, return_hidden=True)
        w_te = O.sub(weights[mod], "text_encoder.") if which == 1 else weights[mod]
        e_mu, e_x, _ = O.text_encoder(toks, lengths, w_te, cfg)
        errs[f"enc{which}.mu"], errs[f"enc{which}.x"] = rel(mu.cpu().numpy().T[None], e_mu), rel(xh.cpu().numpy().T[None], e_x)
        w_st = O.sub(weights[mod], "style_encoder.") if which == 1 else weights[sty_mod]
        errs[f"style{which}"] = rel(hip.text_style(which, sp, mu), O.text_style_encoder(e_mu, lengths, w_st, cfg))
        if which == 2:
            pe_enc, pe_mu = mu, e_mu
            pe_sty = O.text_style_encoder(e_mu, lengths, w_st, cfg)
    # pitch / energy at T mel frames (durations: a deterministic split of T over the P tokens), teacher-forced inputs
    d = synth.durations_for(f"long.d.{P}", P, T).astype(np.int32)
    st = segs([T])
    f0, en, t2 = hip.pitch_energy(sp, st, dev(d), dev(pe_mu[0].T.copy()), dev(pe_sty), taps=True)
    al = synth.alignment_from_durations(d)[None]
    o_f0, o_n, o_mid = O.pitch_energy_predictor(pe_mu, lengths, al, pe_sty, weights["pitch_energy_predictor"], cfg, return_intermediates=True)
    errs.update(pe_prosody=rel(t2["prosody"].cpu().numpy()[None], o_mid["prosody"]), cross=rel(t2["cross"].cpu().numpy().T[None], o_mid["cross"]),
                f0=rel(f0.cpu().numpy()[None], o_f0), energy=rel(en.cpu().numpy()[None], o_n))
    # length regulator at the vocoder rate: a gather, bit-equal to the reference's 0/1 matmul
    st4 = st.scaled(4)
    asr = hip.length_regulate(sp, st4, dev(d), 4, pe_enc, 256).cpu().numpy()
    assert np.array_equal(asr, np.repeat(pe_enc.cpu().numpy()[:, :256], 4 * d, axis=0))
    hip.check_status()
    print(f"\n[P={P}, T={T}] max-abs error / max-abs of the oracle's tensor:", {k: f"{v:.1e}" for k, v in errs.items()}, "duration mismatches:", len(diff))
    for k, v in errs.items():
        assert v < (5e-4 if k == "energy" else 2e-4), (k, v, errs)


# cfg3 (BASELINE configs[2]): the frame path runs with bf16 matrix-core operands, the phoneme-rate predictors ALWAYS run in fp32
# (include/stylish_hip.h, stts_set_precision): durations are integers and must equal the fp32 arithmetic's bit for bit, and pitch is
# integrated over the utterance.  So the phoneme-rate stages are held to the fp32 bars of the test above against the UNROUNDED
# oracle, the durations to equality, and only the waveform carries a 16-bit tolerance: vs the oracle whose contraction operands
# are rounded at the same points (oracle.OPERAND_ROUND) what remains is the fp32 summation order, which flips individual operand
# roundings (2^-9 relative each) from layer to layer, so that distance is of the size of the distance to the fp32
# oracle.  Audio bars are absolute (|audio| < 1), about twice the measured values (MI355X: 7.8e-3 / 1.0e-2 on ~14 s utterances).
CFG3_TOL = dict(text=2e-4, style=2e-4, prosody=2e-4, logits=2e-4, f0=2e-4, energy=5e-4, audio_rounded=1.6e-2, audio_fp32=2e-2)


def test_cfg3_chain_b64_bf16_vs_rounded_oracle(cfg, weights):
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth
    from stylish_tts_amd.pipeline import Synthesizer
    from stylish_tts_amd.runtime import HipModel

    B, P = 64, 50
    eng = HipModel(cfg, 0, precision="bf16")
    eng.load_weights(weights, which=255)
    toks = [synth.tokens(f"cfg3.{i}", 1, P, cfg.text_encoder.tokens)[0] for i in range(B)]
    sp = segs([P] * B)
    t_dev = dev(np.concatenate(toks))
    lengths1 = np.array([P], np.int64)
    check = (0, 31, 63)  # utterances the oracle restates (B = 1 semantics: utterances are independent)
    errs = {k: 0.0 for k in CFG3_TOL}
    try:
        logits, dur, taps = eng.duration(sp, t_dev, taps=True)
        # (a) the 16-bit mode leaves the integer part of the path untouched: every duration equals the fp32 engine's
        eng32 = HipModel(cfg, 0, precision="f32")
        eng32.load_weights({"duration_predictor": weights["duration_predictor"]}, which=16)
        logits32, dur32 = eng32.duration(sp, t_dev)
        assert torch.equal(dur, dur32) and torch.equal(logits, logits32), "bf16 mode changed the duration predictor's arithmetic"
        eng32.close()
        enc = eng.text_encoder(1, sp, t_dev)
        style = eng.text_style(1, sp, enc)
        pe_enc = eng.text_encoder(2, sp, t_dev)
        pe_style = eng.text_style(2, sp, pe_enc)
        csum = torch.cumsum(dur, 0).cpu().numpy()
        ends = csum[sp.host[1:] - 1]
        T = [int(v) for v in np.diff(ends, prepend=0)]
        st = segs(T)
        st4 = st.scaled(4)
        f0, en = eng.pitch_energy(sp, st, dur, pe_enc, pe_style)
        asr = eng.length_regulate(sp, st4, dur, 4, enc, cfg.inter_dim)
        p4, e4 = eng.upsample4(st, st4, f0), eng.upsample4(st, st4, en)
        R4 = st4.rows
        noise = dict(prior_noise=dev(synth.normal("cfg3.pn", (R4, 128))), src_noise=dev(synth.normal("cfg3.sn", (R4 * 75,))),
                     init_phase=dev(synth.uniform("cfg3.ph", (1,))))
        spec, phase = eng.harmonic_stft(st4, p4, noise["src_noise"], noise["init_phase"], batch_scope=False)
        audio = eng.frame_path(st4, asr, p4, e4, style, noise["prior_noise"], noise["src_noise"], noise["init_phase"], batch_scope=False)
        eng.check_status()
        dur_h = dur.cpu().numpy()
        for u in check:
            tk = toks[u][None]
            ps, pt, pt4 = slice(sp.host[u], sp.host[u + 1]), slice(st.host[u], st.host[u + 1]), slice(st4.host[u], st4.host[u + 1])
            O.OPERAND_ROUND = None  # phoneme-rate: fp32 arithmetic
            lo, mid = O.duration_predictor(tk, lengths1, weights["duration_predictor"], cfg, return_intermediates=True)
            errs["text"] = max(errs["text"], rel(taps["text_mu"][ps].cpu().numpy().T[None], mid["text_mu"]))
            errs["style"] = max(errs["style"], rel(taps["style"][u : u + 1], mid["style"]))
            errs["prosody"] = max(errs["prosody"], rel(taps["prosody"][ps].cpu().numpy()[None], mid["prosody"]))
            errs["logits"] = max(errs["logits"], rel(logits[ps].cpu().numpy()[None], lo))
            # (b) durations equal the fp32 oracle's (INT: bit-exact bar); the only licence is a logit pair within fp32 noise of a tie
            want = O.prediction_to_duration(lo[0]).astype(np.int32)
            for i in np.nonzero(dur_h[ps] != want)[0]:
                srt = np.sort(lo[0, i])[::-1]
                assert srt[0] - srt[1] < 1e-3 * np.abs(lo).max(), (u, i, dur_h[ps][i], want[i])
            e_mu, _, _ = O.text_encoder(tk, lengths1, O.sub(weights["speech_predictor"], "text_encoder."), cfg)
            errs["text"] = max(errs["text"], rel(enc[ps].cpu().numpy().T[None], e_mu))
            pe_mu, _, _ = O.text_encoder(tk, lengths1, weights["pe_text_encoder"], cfg)
            errs["text"] = max(errs["text"], rel(pe_enc[ps].cpu().numpy().T[None], pe_mu))
            # teacher-forced from here: the engine's own encoder outputs, styles and durations
            enc_h, sty_h = pe_enc[ps].cpu().numpy()[:, :256].T[None].copy(), pe_style[u : u + 1].cpu().numpy()
            errs["style"] = max(errs["style"], rel(pe_style[u : u + 1], O.text_style_encoder(enc_h, lengths1, weights["pe_text_style_encoder"], cfg)))
            al = synth.alignment_from_durations(dur_h[ps])[None]
            o_f0, o_n = O.pitch_energy_predictor(enc_h, lengths1, al, sty_h, weights["pitch_energy_predictor"], cfg)
            errs["f0"], errs["energy"] = max(errs["f0"], rel(f0[pt].cpu().numpy()[None], o_f0)), max(errs["energy"], rel(en[pt].cpu().numpy()[None], o_n))
            if u == check[1]:
                continue  # the frame path (seconds of oracle time per utterance) on the first and the last utterance only
            # frame path on the engine's own frame-rate inputs
            nz = dict(prior_noise=noise["prior_noise"][pt4].cpu().numpy().T[None].copy(), src_noise=noise["src_noise"][75 * pt4.start : 75 * pt4.stop].cpu().numpy()[None, None],
                      init_phase=noise["init_phase"].cpu().numpy().reshape(1, 1))
            hint = phase[pt4].cpu().numpy()[:, :1025].T[None]
            args = (asr[pt4].cpu().numpy()[:, :128].T[None].copy(), p4[pt4].cpu().numpy()[None], e4[pt4].cpu().numpy()[None], style[u : u + 1].cpu().numpy())
            got = audio[75 * pt4.start : 75 * pt4.stop].cpu().numpy()
            O.OPERAND_ROUND = "bf16"
            a_r, _, _ = O.frame_path(*args, nz, weights["speech_predictor"], branch_hint=hint)
            errs["audio_rounded"] = max(errs["audio_rounded"], float(np.abs(got - a_r[0, 0]).max()))
            O.OPERAND_ROUND = None
            a_f, _, _ = O.frame_path(*args, nz, weights["speech_predictor"], branch_hint=hint)
            errs["audio_fp32"] = max(errs["audio_fp32"], float(np.abs(got - a_f[0, 0]).max()))
    finally:
        O.OPERAND_ROUND = None
    print(f"\n[cfg3: B={B} x {P} tokens, bf16 operands; {sum(T)} mel frames = {sum(T) / 80:.0f} s of audio] vs the rounded oracle:",
          {k: f"{v:.1e}" for k, v in errs.items()})
    # Synthesizer (no host read between the duration predictor and the frame path: capacity segments) == this staged composition with
    # the exact offsets read on the host: same kernels, same noise, the same integer frame counts; the launch plans (tile shapes, split-K
    # factors) follow the sizes the host sees, so fp32 sums may be ordered differently and flip individual bf16 operand roundings
    syn = Synthesizer(eng)
    waves, det = syn([t.tolist() for t in toks], noise=noise, return_details=True)
    assert det["frames"] == T and syn.host_syncs_per_call == 0
    assert torch.equal(det["durations"], dur)
    gap = float((torch.cat(waves) - audio).abs().max())
    print(f"[cfg3] Synthesizer (capacity segments, {syn.capacity_retries} capacity retries) vs the staged composition: waveform max-abs difference {gap:.2e}")
    assert gap < CFG3_TOL["audio_rounded"]
    eng.close()
    for k, v in errs.items():
        assert v < CFG3_TOL[k], (k, v, errs)
