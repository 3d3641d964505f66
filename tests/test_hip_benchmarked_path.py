"""The code path bench.py times, pinned directly.

At cfg2 (B = 8 x 3.0 s, R = 7 680 rows) the frame path takes other branches than the small parity cases do: the separate
AdaIN apply pass instead of the folded affine (`run_adain_block`, rows > 4 096), Winograd conv1 in the decoder, the
128-row / 16-wave contraction tiles and the 32-row fused WaveNet-layer kernel.  Here that very configuration is compared

  (1) with the REFERENCE's own waveform: the `frame_path_3s` golden tiled to B = 8 (every utterance of the batch must
      reproduce the golden's audio), with the atan2 branch ties adopted the reference's way between the STFT and the
      vocoder stage (oracle.align_branch; the number of adopted bins and the raw, un-adopted error are printed);
  (2) with THREE distinct reference waveforms batched as eight distinct rows of utterances (frame_path_3s, _b, _c: a fault at an
      utterance boundary inside a 128-row tile cannot hide behind identical neighbours), and the CONDITIONAL part of the parity
      claim itself: the raw product (no adoption) may differ from the reference only inside the vocoder's receptive field of an
      adopted bin - everywhere else it meets the 1e-3 bar on its own;
  (3) with the numpy oracle on utterances 0, 3 and 7 of bench.py's own synthetic cfg2 inputs;

and the fused entry point `stts_frame_path` (what bench.py calls) must be bit-identical to the staged calls.
Tolerance: waveform sample-wise max-abs < 1e-3, intermediates 2e-4 of their max-abs.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip(cfg, weights):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    m.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
    yield m
    m.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def segs(lengths):
    from stylish_tts_amd.runtime import Segments

    return Segments(lengths, torch.device("cuda", 0))


def adopt_reference_branches(phase_dev, spec_dev, g, B, T4):
    """oracle.align_branch per utterance with the golden's recorded cut bins -> (device phase [B*T4, 1056], #adopted, #bad)."""
    from oracle import stylish_oracle as O

    ph = phase_dev.cpu().numpy()[:, :1025].reshape(B, T4, 1025).transpose(0, 2, 1)
    sp = spec_dev.cpu().numpy()[:, :1025].reshape(B, T4, 1025).transpose(0, 2, 1)
    hint = (g["cut_idx"].astype(np.int64), g["cut_phase"].astype(np.float32))
    out = np.zeros((B * T4, 1056), np.float32)
    adopted = bad = 0
    for b in range(B):
        fixed, nb = O.align_branch(ph[b : b + 1], hint, sp[b : b + 1], return_bad=True)
        adopted += int((fixed != ph[b : b + 1]).sum())
        bad += nb
        out[b * T4 : (b + 1) * T4, :1025] = fixed[0].T
    return dev(out), adopted, bad


@pytest.mark.parametrize("B", [8, 3], ids=["B8", "B3"])
def test_cfg2_batch8_reproduces_the_reference_waveform(hip, B):
    """B = 8: the bench configuration.  B = 3 (2 880 rows): the smallest batches that take the large-batch branches (fp32: from 2 500 rows), without the side stream."""
    from stylish_tts_amd import synth

    g = load_golden("frame_path_3s")
    T4 = 960
    s = segs([T4] * B)
    assert s.rows > 2500  # model.hip.h fold_rows(): the large-batch branches (run_adain_block `fold` off, decoder convs in Winograd form)
    tile_rows = lambda a: dev(np.tile(a, (B, 1)))  # noqa: E731
    asr = tile_rows(synth.normal("g3.asr", (1, 128, T4))[0].T)
    pitch = dev(np.tile(synth.pitch_curve("g3.pitch", 1, T4)[0], B))
    energy = dev(np.tile((synth.uniform("g3.energy", (1, T4)) * 2.0 + 2.0).astype(np.float32)[0], B))
    style = dev(np.tile((synth.normal("g3.style", (1, 64)) * 0.7).astype(np.float32), (B, 1)))
    nz = synth.path_noise("frame960", 1, T4)
    pn = tile_rows(nz["prior_noise"][0].T)
    sn = dev(np.tile(nz["src_noise"].reshape(-1), B))
    ph0 = dev(nz["init_phase"].reshape(-1))

    x = hip.decoder(s, asr, pitch, energy, style)
    xs = x.cpu().numpy().reshape(B, T4, 512)
    mel = hip.prior_flow(s, x, style, pn)
    ms = mel.cpu().numpy().reshape(B, T4, 512)
    for b in range(B):
        for got, want, what in ((xs, g["x_probe"], "decoder"), (ms, g["mel_probe"], "mel")):
            probe = got[b].T[None][:, ::64, ::16]
            err = np.abs(probe - want).max()
            assert err <= 2e-4 * np.abs(want).max(), (what, b, err)
    spec, phase = hip.harmonic_stft(s, pitch, sn, ph0)
    phase_ref, adopted, bad = adopt_reference_branches(phase, spec, g, B, T4)
    assert bad == 0, f"{bad} hinted bins disagree with the reference by more than a branch choice"
    audio = hip.vocoder(s, mel, style, spec, phase_ref).cpu().numpy().reshape(B, -1)
    ref = g["audio"].reshape(-1)
    errs = np.abs(audio - ref[None]).max(axis=1)
    # the product path (no adoption): what `stts_frame_path` returns for the same inputs
    fused = hip.frame_path(s, asr, pitch, energy, style, pn, sn, ph0)
    staged_raw = hip.vocoder(s, mel, style, spec, phase)
    assert torch.equal(fused, staged_raw)
    raw = np.abs(fused.cpu().numpy().reshape(B, -1) - ref[None])
    frames = raw.reshape(B, T4, 75).max(axis=2)  # per vocoder frame
    late = frames[:, 64:].max()
    print(f"\n[cfg2 B=8 vs reference golden] adopted atan2 branch bins: {adopted} of {B * 1025 * T4} "
          f"({adopted // B} per utterance); waveform max-abs err with adoption {errs.max():.2e}; "
          f"raw product path (no adoption): {raw.max():.2e} over all samples, {late:.2e} after frame 64, "
          f"{(frames > 1e-3).sum() // B} of {T4} frames per utterance above 1e-3")
    assert errs.max() < 1e-3, errs


# frames of audio an adopted har_phase bin can reach: phase_prior_conv k7 (3) + ConvNeXt depthwise 31 / 15 / 7 / 3 (15 + 7 + 3 + 1) + output
# convs k7 (3) = 32 spectral frames, + 8 frames of overlap-add (a 1 200-sample window spans 16 hops of 75) = 40.  (GRN's norm over
# time couples all frames of an utterance, at the 1e-6 level: far inside the bar.)
RECEPTIVE_FRAMES = 40
# an adoption counts when it moves a frame's phases by this much in total (a branch flip is 2 pi; the phase of a ~zero-magnitude bin is
# rounding noise and can move by anything, but the prior conv's weights make a whole radian matter and a thousandth not)
ADOPTION_RAD = float(os.environ.get("STTS_ADOPTION_RAD", "1.0"))


def test_cfg2_distinct_utterances_and_the_scope_of_the_branch_adoption(hip):
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    gold = {"": load_golden("frame_path_3s"), "b": load_golden("frame_path_3s_b"), "c": load_golden("frame_path_3s_c")}
    order = ["", "b", "c", "b", "", "c", "c", "b"]  # eight utterances, neighbours always differ
    B, T4 = len(order), 960
    s = segs([T4] * B)
    cat = lambda f: np.concatenate([f(tag) for tag in order])  # noqa: E731
    asr = dev(cat(lambda tg: synth.normal(f"g3{tg}.asr", (1, 128, T4))[0].T))
    pitch = dev(cat(lambda tg: synth.pitch_curve(f"g3{tg}.pitch", 1, T4)[0]))
    energy = dev(cat(lambda tg: (synth.uniform(f"g3{tg}.energy", (1, T4)) * 2.0 + 2.0).astype(np.float32)[0]))
    style = dev(cat(lambda tg: (synth.normal(f"g3{tg}.style", (1, 64)) * 0.7).astype(np.float32)))
    pn = dev(cat(lambda tg: synth.path_noise(f"frame960{tg}", 1, T4)["prior_noise"][0].T))
    sn = dev(cat(lambda tg: synth.path_noise(f"frame960{tg}", 1, T4)["src_noise"].reshape(-1)))
    ph0 = dev(synth.path_noise("frame960", 1, T4)["init_phase"].reshape(-1))  # one scalar per call: the goldens share it
    x = hip.decoder(s, asr, pitch, energy, style)
    mel = hip.prior_flow(s, x, style, pn)
    spec, phase = hip.harmonic_stft(s, pitch, sn, ph0, batch_scope=False)  # harmonic count per utterance, as the reference's B = 1 runs
    ph = phase.cpu().numpy()[:, :1025].reshape(B, T4, 1025).transpose(0, 2, 1)
    sp_ = spec.cpu().numpy()[:, :1025].reshape(B, T4, 1025).transpose(0, 2, 1)
    fixed_rows = np.zeros((B * T4, phase.shape[1]), np.float32)
    moved = np.zeros((B, T4))  # per frame: total |phase change| of the adoption, radians (a branch flip moves a bin by 2 pi)
    for b, tg in enumerate(order):
        g = gold[tg]
        fx, nb = O.align_branch(ph[b : b + 1], (g["cut_idx"].astype(np.int64), g["cut_phase"].astype(np.float32)), sp_[b : b + 1], return_bad=True)
        assert nb == 0, (b, tg, nb)
        moved[b] = np.abs(fx[0].astype(np.float64) - ph[b]).sum(axis=0)
        fixed_rows[b * T4 : (b + 1) * T4, :1025] = fx[0].T
    adopted = hip.vocoder(s, mel, style, spec, dev(fixed_rows)).cpu().numpy().reshape(B, -1)
    raw = hip.frame_path(s, asr, pitch, energy, style, pn, sn, ph0, batch_scope=False).cpu().numpy().reshape(B, -1)
    worst, outside_max, n_inside, n_bad = 0.0, 0.0, 0, 0
    for b, tg in enumerate(order):
        ref = gold[tg]["audio"].reshape(-1)
        worst = max(worst, float(np.abs(adopted[b] - ref).max()))
        # audio frames within reach of a frame whose adoption moved the phases by at least ADOPTION_RAD in total
        reach = np.zeros(T4, bool)
        for f in np.nonzero(moved[b] >= ADOPTION_RAD)[0]:
            reach[max(0, f - RECEPTIVE_FRAMES) : f + RECEPTIVE_FRAMES + 1] = True
        err = np.abs(raw[b] - ref).reshape(T4, 75).max(axis=1)
        n_inside += int(reach.sum())
        n_bad += int((err >= 1e-3).sum())
        outside_max = max(outside_max, float(err[~reach].max()))
        assert (~reach).sum() > T4 // 2, (b, tg, int(reach.sum()))  # the claim is not vacuous: most of the utterance is out of reach
        assert (err[~reach] < 1e-3).all(), (b, tg, float(err[~reach].max()), int(np.argmax(np.where(reach, 0, err))))
    print(f"\n[cfg2, 8 distinct utterances vs 3 reference goldens] with adoption: max-abs {worst:.2e}; raw product: {n_bad // B} of {T4} frames per utterance differ by "
          f">= 1e-3, all of them among the {n_inside // B} frames within {RECEPTIVE_FRAMES} frames of an adoption of >= {ADOPTION_RAD} rad; everywhere else max-abs {outside_max:.2e}")
    assert worst < 1e-3, worst


def test_bench_inputs_match_the_oracle(hip, weights):
    """bench.py's own cfg2 batch (eight DISTINCT utterances at R = 7 680): staged B = 8 run vs the oracle on utterances 0, 3 and 7 (the
    first, one in the middle of a 128-row tile sequence, the last; the oracle takes ~1-2 s for each)."""
    from oracle import stylish_oracle as O

    sys.path.insert(0, ROOT)
    import bench

    a = bench.parse_args([])
    B, T4 = a.batch, 4 * a.mel_frames
    assert (B, T4) == (8, 960)
    inp = bench.cfg2_inputs(0, torch.device("cuda", 0), B, T4)
    s = segs([T4] * B)
    x = hip.decoder(s, inp["asr"], inp["pitch"], inp["energy"], inp["style"])
    mel = hip.prior_flow(s, x, inp["style"], inp["prior_noise"])
    spec, phase = hip.harmonic_stft(s, inp["pitch"], inp["src_noise"], inp["init_phase"], batch_scope=True)
    audio = hip.vocoder(s, mel, inp["style"], spec, phase)
    fused = hip.frame_path(s, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"],
                           batch_scope=True)
    hip.check_status()
    assert torch.equal(fused, audio)  # bench.py's call == the staged composition
    h = inp["host"]
    w = weights["speech_predictor"]
    for u in (0, 3, 7):
        sl = slice(u * T4, (u + 1) * T4)
        nz = dict(prior_noise=h["nz"]["prior_noise"][u : u + 1], src_noise=h["nz"]["src_noise"][u : u + 1], init_phase=h["nz"]["init_phase"])
        a_in, p_in, e_in, s_in = h["asr"][sl].T[None].copy(), h["pitch"][sl][None], h["energy"][sl][None], h["style"][u : u + 1]
        xd = O.decoder_forward(a_in, p_in, e_in, s_in, w)
        ex = np.abs(x.cpu().numpy()[sl, :512].T - xd[0]).max() / np.abs(xd).max()
        assert ex < 2e-4, ("decoder", ex)
        hint = phase.cpu().numpy()[sl, :1025].T[None]
        ref, _, _ = O.frame_path(a_in, p_in, e_in, s_in, nz, w, branch_hint=hint)
        err = np.abs(audio.cpu().numpy()[75 * u * T4 : 75 * (u + 1) * T4] - ref[0, 0]).max()
        print(f"\n[bench cfg2 inputs] utterance {u}: decoder rel err {ex:.2e}, waveform max-abs err vs oracle {err:.2e}")
        assert err < 1e-3, err


@pytest.mark.parametrize("lens", [[960] * 8, [100, 960, 37, 800, 1200, 20, 640]], ids=["cfg2", "ragged"])
def test_side_stream_equals_single_stream(hip, monkeypatch, lens):
    """fp32, batches of 3 000 rows and more: stts_frame_path runs the source -> STFT -> prior-conv chain on a side stream of the caller's stream (fork /
    join events).  Same kernels, same inputs: the waveform must be BIT-IDENTICAL to the single-stream order (STTS_NO_SIDE_STREAM=1), on the default
    stream, on a stream of the caller's, and when called again right away (the next call's fork must wait for this call's join); equal-length and
    ragged batches."""
    from stylish_tts_amd import synth

    B = len(lens)
    assert sum(lens) > 3000  # model.hip.h frame_path: side_min_rows
    s = segs(lens)
    asr = dev(np.concatenate([synth.normal(f"side.asr{b}", (t, 128)) for b, t in enumerate(lens)]))
    pitch = dev(np.concatenate([synth.pitch_curve(f"side.pitch{b}", 1, t)[0] for b, t in enumerate(lens)]))
    energy = dev(np.concatenate([(synth.uniform(f"side.energy{b}", (t,)) * 2 + 2).astype(np.float32) for b, t in enumerate(lens)]))
    style = dev((synth.normal("side.style", (B, 64)) * 0.7).astype(np.float32))
    nzs = [synth.path_noise(f"side{b}", 1, t) for b, t in enumerate(lens)]
    pn = dev(np.concatenate([n["prior_noise"][0].T for n in nzs]))
    sn = dev(np.concatenate([n["src_noise"].reshape(-1) for n in nzs]))
    ph0 = dev(nzs[0]["init_phase"].reshape(-1))
    run = lambda: hip.frame_path(s, asr, pitch, energy, style, pn, sn, ph0, batch_scope=True).clone()  # noqa: E731
    monkeypatch.setenv("STTS_NO_SIDE_STREAM", "1")
    ref = run()
    monkeypatch.delenv("STTS_NO_SIDE_STREAM")
    a, b = run(), run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref).all())
    assert torch.equal(a, ref) and torch.equal(b, ref) and torch.equal(c, ref)



def test_ragged_decoder_on_the_winograd_statistics_path_vs_the_oracle(hip, weights):
    """fp32, more than 2 500 rows: every AdaIN of the decoder takes its statistics from the preceding Winograd conv's output transform (chunks of 24 rows,
    winograd_output_kernel<N, true>) and the lane-parallel merge (adain_affine_lanes_kernel).  Ragged lengths that are multiples of neither 6 (a Winograd
    group) nor 24 (a statistics chunk), one shorter than a chunk: Decoder.forward (models/decoder.py:47-60, models/ada_norm.py:129-182) per utterance vs the oracle."""
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    lens = [997, 13, 1201, 613, 25]
    assert sum(lens) > 2500  # model.hip.h fold_rows(): the Winograd branch
    B = len(lens)
    s = segs(lens)
    asr_h = [synth.normal(f"wst.asr{b}", (t, 128)) for b, t in enumerate(lens)]
    pitch_h = [synth.pitch_curve(f"wst.pitch{b}", 1, t)[0] for b, t in enumerate(lens)]
    energy_h = [(synth.uniform(f"wst.energy{b}", (t,)) * 2 + 2).astype(np.float32) for b, t in enumerate(lens)]
    style_h = (synth.normal("wst.style", (B, 64)) * 0.7).astype(np.float32)
    x = hip.decoder(s, dev(np.concatenate(asr_h)), dev(np.concatenate(pitch_h)), dev(np.concatenate(energy_h)), dev(style_h))
    hip.check_status()
    x = x.cpu().numpy()
    assert np.isfinite(x).all()
    w = weights["speech_predictor"]
    lo = 0
    for u, t in enumerate(lens):
        xd = O.decoder_forward(asr_h[u].T[None].copy(), pitch_h[u][None], energy_h[u][None], style_h[u : u + 1], w)
        err = np.abs(x[lo : lo + t, :512].T - xd[0]).max() / np.abs(xd).max()
        print(f"\n[winograd statistics path] utterance {u} ({t} frames): decoder rel err {err:.2e}")
        assert err < 2e-4, (u, t, err)
        lo += t


def test_every_adopted_bin_is_indeterminate_in_the_reference_itself(hip):
    """The bins at which the HIP path's har_phase is more than 1 rad from the reference's recorded value (= the bins the tests above adopt) all lie in the set
    the reference itself cannot decide (tests/golden/atan2_instability.npz, tests/test_atan2_instability.py): moved by a two-ulp perturbation of the
    reference's own input, or an imaginary part below the accuracy of a 2048-point fp32 FFT, or a magnitude below 2e-4."""
    from stylish_tts_amd import synth

    g, tape = load_golden("atan2_instability"), load_golden("frame_path_3s")
    T4 = 960
    s = segs([T4])
    pitch = synth.pitch_curve("g3.pitch", 1, T4)
    nz = synth.path_noise("frame960", 1, T4)
    spec, phase = hip.harmonic_stft(s, dev(pitch[0]), dev(nz["src_noise"].reshape(-1)), dev(nz["init_phase"].reshape(-1)))
    hip.check_status()
    ph = phase.cpu().numpy()[:, :1025].T.reshape(-1)  # [bins, T4] flat = the tape's index space
    idx = tape["cut_idx"].astype(np.int64)
    dis = idx[np.abs(ph[idx].astype(np.float64) - tape["cut_phase"]) > 1.0]
    allowed = np.zeros(ph.size, bool)
    for k in ("unstable_idx", "tiny_idx", "on_cut_idx"):
        allowed[g[k]] = True
    stray = dis[~allowed[dis]]
    print(f"\n[atan2] HIP vs the reference's tape: {dis.size} bins on the other side of the cut ({(dis % T4 == 0).sum()} of them in frame 0), {stray.size} outside the reference's own indeterminate set")
    assert dis.size > 50 and stray.size == 0, stray[:10]
