"""GPU parity tests of the phoneme-rate predictors (text encoders, style encoders, duration, pitch/energy, length
regulator) through the C-ABI against golden vectors produced by the reference."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip(cfg, weights):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    m.load_weights(weights, which=255)
    yield m
    m.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def segs(lengths):
    from stylish_tts_amd.runtime import Segments

    return Segments(lengths, torch.device("cuda", 0))


def close(a, b, rtol=2e-4, atol=None, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-6)
    tol = atol if atol is not None else rtol * scale
    err = np.abs(a - b).max()
    assert np.isfinite(a).all(), f"{what}: non-finite"
    assert err <= tol, f"{what}: max-abs err {err:.3e} > {tol:.3e} (scale {scale:.3e})"


def test_duration_predictor_golden(hip):
    g = load_golden("duration")
    P = int(g["lengths"][0])
    s = segs([P])
    logits, dur, t = hip.duration(s, dev(g["texts"][0]), taps=True)
    close(t["text_mu"].cpu().numpy().T[None], g["text_mu"], what="TextEncoder mu")
    close(t["style"].cpu().numpy(), g["style"], what="TextStyleEncoder")
    close(t["prosody"].cpu().numpy()[None], g["prosody"], what="ProsodyEncoder")
    close(logits.cpu().numpy()[None], g["logits"], what="duration logits")
    assert np.array_equal(dur.cpu().numpy(), g["duration"].astype(np.int32))


def test_text_encoder_hidden_golden(hip):
    g = load_golden("duration")
    s = segs([int(g["lengths"][0])])
    mu, xh = hip.text_encoder(0, s, dev(g["texts"][0]), return_hidden=True)
    close(xh.cpu().numpy().T[None], g["text_x"], what="TextEncoder x")
    close(mu.cpu().numpy().T[None], g["text_mu"], what="TextEncoder mu")


def test_ragged_batch_is_per_utterance(hip):
    """Packed ragged batch: utterance 0 (full length) matches the reference's padded-batch result everywhere; utterance 1
    (7 of 12 tokens) matches on the text encoder (masking == packing) and equals its own B=1 run downstream (the
    reference's padded statistics differ there by construction)."""
    g = load_golden("duration_b2")
    L = [int(x) for x in g["lengths"]]
    toks = np.concatenate([g["texts"][i, : L[i]] for i in range(2)])
    s = segs(L)
    logits, dur, t = hip.duration(s, dev(toks), taps=True)
    mu = t["text_mu"].cpu().numpy()
    close(mu[: L[0]].T, g["text_mu"][0], what="mu utt0")
    close(mu[L[0] :].T, g["text_mu"][1][:, : L[1]], what="mu utt1 (valid tokens)")
    close(t["style"].cpu().numpy()[0], g["style"][0], what="style utt0")
    close(logits.cpu().numpy()[: L[0]], g["logits"][0], what="logits utt0")
    s1 = segs([L[1]])
    l1, d1 = hip.duration(s1, dev(toks[L[0] :]))
    # the launcher may cut K differently for different batch shapes (split-K): same math, fp32 summation order only
    close(l1.cpu().numpy(), logits[L[0] :].cpu().numpy(), rtol=2e-5, what="packed == single-utterance run")
    assert torch.equal(d1, dur[L[0] :])


def test_duration_decode_both_branches(hip):
    """DurationProcessor.prediction_to_duration (train/utils.py:468-474) on the reference's golden logits, which reach
    both the hard (argmax class < 7 frames) and the soft (expected value, rounded half-to-even) branch:
    `stts_duration_decode` on the device, bit-equal to the reference's durations."""
    import ctypes as C

    from stylish_tts_amd import _lib
    from stylish_tts_amd.runtime import _ptr, _stream

    g = load_golden("duration_processor")
    want = g["duration"].astype(np.int32)
    hard = np.array([1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46])[g["logits"].argmax(-1)]
    assert (hard < 7).any() and (hard >= 7).any(), "the fixture must reach both branches"
    lg = dev(g["logits"].astype(np.float32))
    dur = torch.full((lg.shape[0],), -1, dtype=torch.int32, device=lg.device)
    _lib.check(_lib.load().stts_duration_decode(_stream(), _ptr(lg), lg.shape[1], lg.shape[0], _ptr(dur)))
    assert np.array_equal(dur.cpu().numpy(), want)
    # a wider leading dimension (the duration stage's own logits buffer may be padded)
    wide = torch.zeros(lg.shape[0], 32, device=lg.device)
    wide[:, :16] = lg
    wide[:, 16:] = 1e4  # must be ignored
    dur2 = torch.empty_like(dur)
    _lib.check(_lib.load().stts_duration_decode(_stream(), _ptr(wide), 32, lg.shape[0], _ptr(dur2)))
    assert torch.equal(dur, dur2)
    del C


def test_pitch_energy_golden(hip):
    g = load_golden("pitch_energy")
    P = int(g["lengths"][0])
    d = g["durations"].astype(np.int32)
    sp, st = segs([P]), segs([int(d.sum())])
    pe_enc = hip.text_encoder(2, sp, dev(g["texts"][0]))
    close(pe_enc.cpu().numpy().T[None], g["pe_text"], what="pe_text_encoder")
    pe_style = hip.text_style(2, sp, pe_enc)
    close(pe_style.cpu().numpy(), g["pe_style"], what="pe_text_style_encoder")
    f0, en, t = hip.pitch_energy(sp, st, dev(d), dev(g["pe_text"][0].T.copy()), dev(g["pe_style"]), taps=True)
    close(t["prosody"].cpu().numpy()[None], g["prosody"], what="pe prosody")
    close(t["cross"].cpu().numpy().T[None], g["cross"], what="compute_cross (inverted band mask)")
    close(f0.cpu().numpy()[None], g["f0"], what="F0")
    close(en.cpu().numpy()[None], g["energy"], rtol=5e-4, what="N")


def test_length_regulator_and_upsample(hip):
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    durs = [synth.durations_for("lr.a", 9, 31), synth.durations_for("lr.b", 5, 12)]
    sp = segs([9, 5])
    st = segs([31, 12])
    st4 = st.scaled(4)
    enc = synth.normal("lr.enc", (14, 128))
    dur = dev(np.concatenate(durs).astype(np.int32))
    out = hip.length_regulate(sp, st4, dur, 4, dev(enc), 128).cpu().numpy()
    off = 0
    for i, d in enumerate(durs):
        al = np.repeat(synth.alignment_from_durations(d), 4, axis=1)
        e = enc[sp.host[i] : sp.host[i + 1]].T
        ref = (e @ al).T  # text_encoding @ alignment (speech_predictor.py:93)
        assert np.array_equal(out[off : off + ref.shape[0]], ref)
        off += ref.shape[0]
    x = synth.normal("lr.x", (43,))
    y = hip.upsample4(st, st4, dev(x)).cpu().numpy()
    ref = np.concatenate([O.upsample_linear4(x[None, :31])[0], O.upsample_linear4(x[None, 31:])[0]])
    close(y, ref, atol=1e-6, what="upsample x4")


def test_length_regulator_has_no_token_limit(hip):
    """An utterance of 1 500 tokens next to a short one through the C-ABI gather (the chunked duration scan of
    frame_token_map_kernel; ADVICE r01: the LDS scan used to stop at 1 025 tokens)."""
    from stylish_tts_amd import synth

    P = [1500, 7]
    durs = [1 + (np.arange(P[0]) * 7 % 5), synth.durations_for("lrbig.b", P[1], 20)]
    T = [int(d.sum()) for d in durs]
    sp, st4 = segs(P), segs([4 * t for t in T])
    enc = synth.normal("lrbig.enc", (sum(P), 128))
    out = hip.length_regulate(sp, st4, dev(np.concatenate(durs).astype(np.int32)), 4, dev(enc), 128).cpu().numpy()
    want = np.repeat(enc, 4 * np.concatenate(durs), axis=0)
    assert np.array_equal(out, want)


def test_token_out_of_range_is_reported(hip):
    s = segs([4])
    hip.text_encoder(0, s, dev(np.array([0, 5, 9999, 0], np.int64)))
    with pytest.raises(RuntimeError, match="token id"):
        hip.check_status()
