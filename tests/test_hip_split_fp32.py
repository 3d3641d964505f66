"""Split-fp32 contractions (csrc/gemm.hip.h, PREC_X3) are fp32 arithmetic, not a reduced precision: every operand is the exact
sum of three bf16 terms and the six largest of the nine cross products run on the bf16 matrix cores with fp32 accumulation.

Evidence, per layer shape of the frame path: the error against a float64 contraction of the SAME fp32 operands, for the split
form and for the f32 matrix cores (v_mfma_f32_32x32x2_f32, `precision="f32_native"`), side by side.  The bar: the split form's
error is within 1.25 x the f32 matrix cores' (+ 2e-7 of the output scale), and both are ~1e-6 of the output scale - two to
three decimal digits below a bf16 / fp16 operand rounding (4e-3 / 5e-4).  Also: every split tile
agrees with the oracle, and the two forms agree with each other through a whole decoder block.
"""
import numpy as np
import pytest

from test_hip_frame_path import close, dev, hip, segs  # noqa: F401  (hip: module fixture)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def conv64(x, w, b, lengths, k, dil=1):
    """float64 'same' conv per utterance on time-major rows x [rows, cin], w [cout, cin, k]."""
    out, lo = [], 0
    pad = (k - 1) // 2 * dil
    w64 = w.astype(np.float64)
    for L in lengths:
        xi = np.zeros((L + 2 * pad, x.shape[1]), np.float64)
        xi[pad : pad + L] = x[lo : lo + L]
        y = np.zeros((L, w.shape[0]), np.float64)
        for t in range(k):
            y += xi[t * dil : t * dil + L] @ w64[:, :, t].T
        out.append(y + b.astype(np.float64))
        lo += L
    return np.concatenate(out)


SHAPES = [  # cin, cout, k, lengths: the layer shapes of the frame path on ragged batches
    (578, 512, 3, [300, 77, 129]),     # decoder conv1
    (512, 512, 3, [260, 131]),         # decoder conv2
    (512, 1536, 1, [200, 150]),        # ConvNeXt pwconv1
    (1536, 512, 1, [200, 150]),        # ConvNeXt pwconv2
    (768, 1024, 7, [140, 61]),         # output conv (direct form)
    (1025, 256, 7, [150]),             # prior conv (direct form)
    (128, 512, 1, [333]),              # post_flow
]


@pytest.mark.parametrize("cin,cout,k,lengths", SHAPES)
def test_split_fp32_error_equals_the_f32_matrix_cores_error(hip, cin, cout, k, lengths):
    from stylish_tts_amd import synth

    s = segs(lengths)
    ld = (cin + 31) // 32 * 32
    # weights ~ 1/sqrt(fan-in) like the model's; activations with a spread of magnitudes (post-AdaIN rows are O(1), log-amplitudes O(10))
    w = (synth.normal(f"x3.w.{cin}.{cout}.{k}", (cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = synth.normal(f"x3.b.{cout}", (cout,)).astype(np.float32)
    x = np.zeros((s.rows, ld), np.float32)
    x[:, :cin] = synth.normal(f"x3.x.{cin}.{s.rows}", (s.rows, cin)) * np.exp(synth.normal(f"x3.m.{cin}", (1, cin)))
    ref = conv64(x[:, :cin], w, b, lengths, k)
    scale = np.abs(ref).max()
    y_split = hip.op_conv1d(s, dev(x), cin, w, b, precision="f32").cpu().numpy()[:, :cout].astype(np.float64)
    y_native = hip.op_conv1d(s, dev(x), cin, w, b, precision="f32_native").cpu().numpy()[:, :cout].astype(np.float64)
    e_split, e_native = np.abs(y_split - ref), np.abs(y_native - ref)
    rms_split, rms_native = np.sqrt((e_split ** 2).mean()), np.sqrt((e_native ** 2).mean())
    print(f"\n[split fp32] {cin}->{cout} k{k}: max err split {e_split.max() / scale:.2e} native {e_native.max() / scale:.2e}; rms split {rms_split / scale:.2e} native {rms_native / scale:.2e} (of the output scale {scale:.2f})")
    assert e_split.max() <= 1.25 * e_native.max() + 2e-7 * scale
    assert rms_split <= 1.25 * rms_native + 5e-8 * scale
    assert e_split.max() <= 3e-6 * scale, "fp32-level agreement with float64"
    # a bf16 rounding of the operands would sit two to three digits above both
    y_bf16 = hip.op_conv1d(s, dev(x), cin, w, b, precision="bf16").cpu().numpy()[:, :cout].astype(np.float64)
    assert np.abs(y_bf16 - ref).max() > 50 * e_split.max()


@pytest.mark.parametrize("tile", [2, 3, 4, 5, 6, 8, 20, 21, 22])
def test_every_split_tile_matches_the_oracle(hip, tile):
    from oracle import stylish_oracle as O
    from stylish_tts_amd import synth

    cin, cout, k, lengths = 578, 512, 3, [300, 37, 129, 1]
    s = segs(lengths)
    ld = (cin + 31) // 32 * 32
    w = (synth.normal("x3t.w", (cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = synth.normal("x3t.b", (cout,)).astype(np.float32)
    xs = [synth.normal(f"x3t.x.{i}.{L}", (1, cin, L)) for i, L in enumerate(lengths)]
    x = np.full((s.rows, ld), 7.0, np.float32)  # finite garbage in the pad columns: the packed weight planes are zero there
    for i, xi in enumerate(xs):
        x[s.host[i] : s.host[i + 1], :cin] = xi[0].T
    y = hip.op_conv1d(s, dev(x), cin, w, b, force_tile=tile, precision="f32").cpu().numpy()
    for i, xi in enumerate(xs):
        ref = O.conv1d(xi, w, b, padding=(k - 1) // 2)[0].T
        close(y[s.host[i] : s.host[i + 1], :cout], ref, rtol=2e-6, what=f"tile {tile} utt {i}")


def test_frame_path_split_vs_native_waveforms(cfg, weights):
    """The whole frame path in both forms on bench.py's own cfg2 batch (8 x 3 s, Winograd branch): the waveforms agree to fp32 noise
    except where an atan2 branch cut flips on that noise (either form is pinned against the reference's goldens separately:
    test_hip_benchmarked_path.py runs the default, split, form)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from stylish_tts_amd.runtime import HipModel

    B, T4 = 8, 960
    inp = bench.cfg2_inputs(0, torch.device("cuda", 0), B, T4)
    s = segs([T4] * B)
    outs, mels = {}, {}
    for prec in ("f32", "f32_native"):
        m = HipModel(cfg, 0, precision=prec)
        m.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
        x = m.decoder(s, inp["asr"], inp["pitch"], inp["energy"], inp["style"])
        mels[prec] = m.prior_flow(s, x, inp["style"], inp["prior_noise"]).cpu().numpy()
        outs[prec] = m.frame_path(s, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["prior_noise"], inp["src_noise"], inp["init_phase"],
                                  batch_scope=True).cpu().numpy()
        m.check_status()
        m.close()
    dm = np.abs(mels["f32"] - mels["f32_native"]).max() / np.abs(mels["f32_native"]).max()
    d = np.abs(outs["f32"] - outs["f32_native"]).reshape(-1, 75).max(1)
    print(f"\n[split fp32] cfg2 frame path, split vs f32 matrix cores: mel rel diff {dm:.2e}; waveform max abs {d.max():.2e}, median frame {np.median(d):.2e}, frames above 1e-3: {(d > 1e-3).sum()} of {d.size}")
    assert dm < 2e-5      # decoder + flow: no branch cuts in between
    assert np.median(d) < 2e-5
