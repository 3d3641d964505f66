"""16-bit operand modes of the contractions (BASELINE cfg3 bf16 / cfg5 fp16; include/stylish_hip.h:stts_set_precision).

Two bars per mode:
  * against the oracle with the SAME rounding points (operands of every matrix-core contraction rounded to nearest-even,
    fp32 products and sums): tight, only summation order differs;
  * against the fp32 reference goldens: the stated 16-bit tolerance (the reference itself only runs fp32).
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def segs(lengths):
    from stylish_tts_amd.runtime import Segments

    return Segments(lengths, torch.device("cuda", 0))


@pytest.fixture(scope="module")
def hip32(cfg):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    yield m
    m.close()


@pytest.fixture()
def rounded_oracle():
    from oracle import stylish_oracle as O

    def use(mode):
        O.OPERAND_ROUND = mode
        return O

    yield use
    O.OPERAND_ROUND = None


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize(
    "cin,cout,k,dil,lengths,tile",
    [
        (64, 128, 1, 1, [40], 3),
        (578, 512, 3, 1, [200, 37, 129], 0),
        (578, 512, 3, 1, [200, 37, 129], 2),
        (578, 512, 3, 1, [200, 37, 129], 5),
        (578, 512, 3, 1, [200, 37, 129], 6),
        (578, 512, 3, 1, [200, 37, 129], 8),
        (96, 1025, 7, 1, [50, 333], 0),
        (64, 130, 7, 3, [77, 5], 0),
        (1536, 512, 1, 1, [130], 0),
        # 16-bit ACTIVATION ROWS (force_tile 100 + tile): the contraction reads a pre-rounded copy of x - same values as
        # rounding at staging time - on the ordinary tiles and on the 256-row tiles of large batches
        (578, 512, 3, 1, [200, 37, 129], 105),
        (578, 512, 3, 1, [200, 37, 129], 106),
        (768, 1024, 7, 1, [300, 517, 2], 114),
        (512, 1536, 1, 1, [700, 1], 114),
        (578, 512, 3, 1, [260, 255, 257, 31], 115),
        (1024, 256, 7, 1, [513], 115),
        # ... and the LDS-DMA (global_load_lds) variants of the 256-row tiles: three stages, ragged tails, several K segments' worth of taps
        (768, 1024, 7, 1, [300, 517, 2], 116),
        (512, 1536, 1, 1, [700, 1], 116),
        (32, 256, 1, 1, [5, 256], 116),
        (578, 512, 3, 1, [260, 255, 257, 31], 117),
        (1024, 256, 7, 1, [513], 117),
        (64, 130, 7, 3, [77, 5], 117),
        # ... and the 128-row LDS-DMA tile with eight stages (more stages than K iterations, exactly as many, and many more iterations)
        (96, 256, 1, 1, [130, 5], 118),
        (256, 128, 1, 1, [128], 118),
        (578, 512, 3, 1, [200, 37, 129], 118),
        (768, 1024, 7, 1, [300, 17], 118),
        # ... and conv_gemm16_kernel (tile 19: persistent 256 x 256 blocks, LDS-DMA through buffer descriptors, K tiles of 64 channels): deep and
        # single K tiles, ragged tails, utterances shorter than a tile, taps reaching across both ends, dilation
        (768, 1024, 7, 1, [300, 517, 2], 119),
        (512, 1536, 1, 1, [700, 1], 119),
        (64, 256, 1, 1, [5, 256], 119),
        (1024, 256, 7, 1, [513], 119),
        (192, 512, 3, 1, [260, 255, 257, 31], 119),
        (64, 256, 7, 3, [77, 5], 119),
    ],
)
def test_conv1d_16bit_matches_rounded_oracle(hip32, rounded_oracle, prec, cin, cout, k, dil, lengths, tile):
    from stylish_tts_amd import synth

    O = rounded_oracle(prec)
    s = segs(lengths)
    ld = (cin + 31) // 32 * 32
    w = (synth.normal(f"p.w.{cin}.{cout}.{k}", (cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = synth.normal(f"p.b.{cout}", (cout,))
    xs = [synth.normal(f"p.x.{i}.{L}", (1, cin, L)) for i, L in enumerate(lengths)]
    x = np.zeros((s.rows, ld), np.float32)
    for i, xi in enumerate(xs):
        x[s.host[i] : s.host[i + 1], :cin] = xi[0].T
    y = hip32.op_conv1d(s, dev(x), cin, w, b, dil=dil, force_tile=tile, precision=prec).cpu().numpy()
    y32 = hip32.op_conv1d(s, dev(x), cin, w, b, dil=dil, force_tile=tile if tile < 100 else 0).cpu().numpy()
    for i, xi in enumerate(xs):
        ref = O.conv1d(xi, w, b, padding=(k - 1) // 2 * dil, dilation=dil)[0].T
        got = y[s.host[i] : s.host[i + 1], :cout]
        scale = np.abs(ref).max()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 2e-5 * scale, (prec, i, np.abs(got - ref).max(), scale)
        # and the mode really rounds: it differs from the fp32 contraction by about one operand ulp
        d32 = np.abs(got - y32[s.host[i] : s.host[i + 1], :cout]).max() / scale
        assert (1e-4 if prec == "f16" else 1e-3) < d32 < (3e-3 if prec == "f16" else 3e-2), (prec, d32)


def _ragged_case(tag, lens):
    from stylish_tts_amd import synth

    per = []
    for i, L in enumerate(lens):
        per.append(
            dict(
                asr=synth.normal(f"{tag}.asr{i}", (1, 128, L)),
                pitch=synth.pitch_curve(f"{tag}.p{i}", 1, L),
                energy=(synth.uniform(f"{tag}.e{i}", (1, L)) * 2 + 2).astype(np.float32),
                style=(synth.normal(f"{tag}.s{i}", (1, 64)) * 0.7).astype(np.float32),
                nz=synth.path_noise(f"{tag}{i}", 1, L),
            )
        )
    return per


# waveform max-abs tolerances of the 16-bit operand modes (fp32 accumulation; |audio| < 1):
#   vs the oracle with the same rounding points   vs the fp32 oracle (what rounding the operands costs)
TOL = {"bf16": (8e-3, 1.2e-2), "f16": (1.2e-3, 1.5e-3)}  # measured: bf16 3.8e-3 / 4.5e-3, f16 4.8e-4 / 5.8e-4


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_frame_path_16bit_ragged_batch(cfg, weights, rounded_oracle, prec):
    """cfg3 / cfg5 arithmetic on a mixed-length batch: decoder -> prior/flow -> vocoder with 16-bit matrix-core operands."""
    from stylish_tts_amd.runtime import HipModel

    hip = HipModel(cfg, 0, precision=prec)
    hip.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
    w = weights["speech_predictor"]
    lens = [40, 131, 76]
    s = segs(lens)
    per = _ragged_case("q", lens)
    init_phase = per[0]["nz"]["init_phase"]
    cat = lambda k: np.concatenate([p[k][0].T if p[k].ndim == 3 else p[k][0] for p in per])  # noqa: E731
    asr, pitch, energy = dev(cat("asr")), dev(cat("pitch")), dev(cat("energy"))
    style = dev(np.concatenate([p["style"] for p in per]))
    pn = dev(np.concatenate([p["nz"]["prior_noise"][0].T for p in per]))
    sn = dev(np.concatenate([p["nz"]["src_noise"].reshape(-1) for p in per]))
    x = hip.decoder(s, asr, pitch, energy, style)
    mel = hip.prior_flow(s, x, style, pn)
    spec, phase = hip.harmonic_stft(s, pitch, sn, dev(init_phase.reshape(-1)), batch_scope=False)
    audio = hip.vocoder(s, mel, style, spec, phase).cpu().numpy()
    fused = hip.frame_path(s, asr, pitch, energy, style, pn, sn, dev(init_phase.reshape(-1)), batch_scope=False).cpu().numpy()
    assert np.array_equal(fused, audio)
    ph_np, x_np, mel_np = phase.cpu().numpy(), x.cpu().numpy(), mel.cpu().numpy()
    assert np.isfinite(audio).all()
    errs = []
    for i, (L, p) in enumerate(zip(lens, per)):
        nz = dict(p["nz"], init_phase=init_phase)
        hint = ph_np[s.host[i] : s.host[i + 1], :1025].T[None]
        sl = slice(75 * s.host[i], 75 * s.host[i + 1])
        O = rounded_oracle(prec)
        xd = O.decoder_forward(p["asr"], p["pitch"], p["energy"], p["style"], w)
        ex = np.abs(x_np[s.host[i] : s.host[i + 1], :512].T - xd[0]).max() / np.abs(xd).max()
        a_r, _, _ = O.frame_path(p["asr"], p["pitch"], p["energy"], p["style"], nz, w, branch_hint=hint)
        O = rounded_oracle(None)
        a_f, _, _ = O.frame_path(p["asr"], p["pitch"], p["energy"], p["style"], nz, w, branch_hint=hint)
        errs.append((ex, np.abs(audio[sl] - a_r[0, 0]).max(), np.abs(audio[sl] - a_f[0, 0]).max()))
    print(prec, "decoder err / audio vs rounded oracle / audio vs fp32 oracle:", errs)
    hip.close()
    for ex, er, ef in errs:
        assert er < TOL[prec][0], (prec, errs)
        assert ef < TOL[prec][1], (prec, errs)


# The phoneme-rate predictors ALWAYS run in fp32 (include/stylish_hip.h, stts_set_precision): durations are integers (bar: bit-exact)
# and these stages are latency-bound, so the 16-bit operand modes must leave them untouched - every tensor bit-identical to the
# fp32 engine's, on a ragged batch.
@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_phoneme_stages_stay_fp32_in_16bit_modes(cfg, weights, prec):
    from stylish_tts_amd import synth
    from stylish_tts_amd.runtime import HipModel

    e32 = HipModel(cfg, 0)
    e32.load_weights(weights, which=255)
    e16 = HipModel(cfg, 0, precision=prec)
    e16.load_weights(weights, which=255)
    L = [23, 9, 41]
    toks = dev(np.concatenate([synth.tokens(f"h.{i}", 1, n, 178)[0] for i, n in enumerate(L)]).astype(np.int64))
    sp = segs(L)
    for which in (1, 2):
        enc32, enc16 = e32.text_encoder(which, sp, toks), e16.text_encoder(which, sp, toks)
        assert torch.equal(enc16, enc32), f"text encoder {which}"
        assert torch.equal(e16.text_style(which, sp, enc32), e32.text_style(which, sp, enc32)), f"style encoder {which}"
    lg32, dur32 = e32.duration(sp, toks)
    lg16, dur16 = e16.duration(sp, toks)
    assert torch.equal(lg16, lg32) and torch.equal(dur16, dur32)
    csum = torch.cumsum(dur32, 0).cpu().numpy()
    T = [int(v) for v in np.diff(csum[sp.host[1:] - 1], prepend=0)]
    st = segs(T)
    pe32 = e32.text_encoder(2, sp, toks)
    ps32 = e32.text_style(2, sp, pe32)
    f32_, n32 = e32.pitch_energy(sp, st, dur32, pe32, ps32)
    f16_, n16 = e16.pitch_energy(sp, st, dur32, pe32, ps32)
    assert torch.equal(f16_, f32_) and torch.equal(n16, n32)
    e32.check_status()
    e16.check_status()
    e32.close()
    e16.close()


@pytest.mark.parametrize("prec,tol", [("bf16", 2e-2), ("f16", 3e-3)])
def test_flow_16bit_fused_kernel_variants(cfg, weights, rounded_oracle, prec, tol, monkeypatch):
    """The reverse flow in the 16-bit operand modes: wn_fused16_kernel on 64- and 128-row blocks, wn_block16_kernel (one launch per coupling
    layer, 128-row blocks + 2 x 8 halo rows; all picked by batch size in production, forced here) and the staged wn_layer_kernel share their rounding points (h entering LDS, gated activations,
    `out` before post, the coupled half before pre), so all three must sit within the mode's noise of the oracle whose
    contraction operands are rounded at those points, on lengths that leave partial blocks and one-row utterances.
    Tolerance: relative to max |z|; 8 coupling layers x 4 WaveNet layers of operand roundings (2^-9 bf16, 2^-11 fp16)."""
    from stylish_tts_amd import synth
    from stylish_tts_amd.runtime import HipModel

    hip = HipModel(cfg, 0, precision=prec)
    hip.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
    w = weights["speech_predictor"]
    lens = [129, 1, 70, 260, 2, 63, 128, 400]
    s = segs(lens)
    xs = [synth.normal(f"wn16.x{i}", (1, 512, L)) for i, L in enumerate(lens)]
    st = (synth.normal("wn16.s", (len(lens), 64)) * 0.7).astype(np.float32)
    ns = [synth.normal(f"wn16.n{i}", (1, 128, L)) for i, L in enumerate(lens)]
    cat = lambda parts: dev(np.concatenate([p[0].T for p in parts]))  # noqa: E731
    O = rounded_oracle(prec)
    refs = []
    for i in range(len(lens)):
        z, _, _ = O.prior_encoder(xs[i], ns[i], w)
        refs.append(O.flow_reverse(z, st[i : i + 1, :, None], w))
    got = {}
    for rt in ("4", "8", "16", "-1"):
        monkeypatch.setenv("STTS_WN_RT", rt)
        _, _, zf = hip.prior_flow(s, cat(xs), dev(st), cat(ns), return_z=True)
        got[rt] = zf.cpu().numpy()
    hip.close()
    errs = {}
    for rt, zf in got.items():
        e = 0.0
        for i, L in enumerate(lens):
            ref = refs[i]
            e = max(e, float(np.abs(zf[s.host[i] : s.host[i + 1], :128].T[None] - ref).max() / np.abs(ref).max()))
        errs[rt] = e
    print(f"\n[{prec} flow vs rounded oracle] 64-row blocks {errs['4']:.1e}, 128-row blocks {errs['8']:.1e}, one launch per coupling layer {errs['16']:.1e}, "
          f"staged kernel {errs['-1']:.1e}")
    assert np.isfinite(got["4"]).all() and np.isfinite(got["8"]).all() and np.isfinite(got["16"]).all()
    for rt, e in errs.items():
        assert e < tol, (prec, rt, errs)
