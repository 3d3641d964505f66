"""Size-independent properties at BASELINE.json's full sizes (the oracle is too slow there): determinism, batch ==
per-utterance (utterances are independent units), ragged sharding-shaped batches, long-form segments."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip(cfg, weights):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    m.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
    yield m
    m.close()


def make_inputs(lengths, tag):
    from stylish_tts_amd import synth

    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()  # noqa: E731
    R = int(sum(lengths))
    asr = synth.normal(tag + ".asr", (R, 128))
    pitch = np.concatenate([synth.pitch_curve(f"{tag}.p{i}", 1, L)[0] for i, L in enumerate(lengths)])
    energy = synth.uniform(tag + ".e", (R,)) * 2 + 2
    style = synth.normal(tag + ".s", (len(lengths), 64)) * 0.7
    pn = synth.normal(tag + ".pn", (R, 128))
    sn = synth.normal(tag + ".sn", (R * 75,))
    ph = synth.uniform(tag + ".ph", (1,))
    return dict(asr=d(asr), pitch=d(pitch), energy=d(energy), style=d(style), pn=d(pn), sn=d(sn), ph=d(ph))


def run(hip, lengths, x, batch_scope=False):
    from stylish_tts_amd.runtime import Segments

    seg = Segments(lengths, hip.device)
    return hip.frame_path(seg, x["asr"], x["pitch"], x["energy"], x["style"], x["pn"], x["sn"], x["ph"], batch_scope=batch_scope)


def slice_inputs(x, lo, hi, u):
    return dict(asr=x["asr"][lo:hi].contiguous(), pitch=x["pitch"][lo:hi].contiguous(), energy=x["energy"][lo:hi].contiguous(),
                style=x["style"][u : u + 1].contiguous(), pn=x["pn"][lo:hi].contiguous(), sn=x["sn"][75 * lo : 75 * hi].contiguous(), ph=x["ph"])


def test_cfg2_batch8_is_deterministic_and_equals_single_utterances(hip):
    """cfg2 (B = 8 x 3 s): two runs are bit-identical, and each utterance equals its own B = 1 run.  The contraction
    tiles never straddle utterances and every reduction has a fixed order, so the match is exact."""
    lengths = [960] * 8
    x = make_inputs(lengths, "full.cfg2")
    a = run(hip, lengths, x)
    b = run(hip, lengths, x)
    assert torch.equal(a, b)
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 1.0  # tanh output
    for u in (0, 3, 7):
        lo, hi = 960 * u, 960 * (u + 1)
        one = run(hip, [960], slice_inputs(x, lo, hi, u))
        err = float((one - a[75 * lo : 75 * hi]).abs().max())
        # the batch uses the Winograd form of the wide convs and other tiles than a single utterance does: fp32 rounding
        # differences only (F(6,7): ~3e-6 of the conv's scale), two decades inside the 1e-3 parity bar
        assert err < 1e-4, (u, err)


def test_cfg4_ragged_256_utterances(hip):
    """cfg4 shape: 256 utterances of 0.25-10 s in one call (one rank's worth and more): finite, deterministic per
    utterance, and a sample of utterances equals the B = 1 run."""
    rng = np.random.default_rng(4)
    lengths = [int(4 * round(80 * s)) for s in rng.uniform(0.25, 10.0, 256)]
    x = make_inputs(lengths, "full.cfg4")
    a = run(hip, lengths, x)
    assert bool(torch.isfinite(a).all())
    off = np.concatenate([[0], np.cumsum(lengths)])
    for u in (0, 17, 255, int(np.argmin(lengths)), int(np.argmax(lengths))):
        lo, hi = int(off[u]), int(off[u + 1])
        one = run(hip, [lengths[u]], slice_inputs(x, lo, hi, u))
        err = float((one - a[75 * lo : 75 * hi]).abs().max())
        assert err < 1e-4, (u, lengths[u], err)


def test_cfg5_long_form_10s_batch(hip):
    """cfg5 at its real per-GPU size in fp32: 64 segments of 10 s (T4 = 3200, R = 204 800 rows, ~11 GB of workspace; the
    row * leading-dimension products pass 2^27 floats, which is what the 32-bit offset arithmetic of the contraction
    kernels has to survive): deterministic, finite, and utterances from the start / middle / end equal their B = 1 run."""
    lengths = [3200] * 64
    x = make_inputs(lengths, "full.cfg5")
    a = run(hip, lengths, x)
    assert torch.equal(a, run(hip, lengths, x))
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 1.0
    for u in (0, 37, 63):
        one = run(hip, [3200], slice_inputs(x, 3200 * u, 3200 * (u + 1), u))
        err = float((one - a[75 * 3200 * u : 75 * 3200 * (u + 1)]).abs().max())
        assert err < 1e-4, (u, err)


def test_batch_scope_changes_only_the_harmonic_count(hip):
    """batch_scope selects the reference's batched-call semantics for generate_pcph (shared min f0): with ordinary speech
    pitch (< 750 Hz) the harmonic count is 16 either way and the two modes agree exactly."""
    lengths = [320, 480]
    x = make_inputs(lengths, "full.scope")
    assert torch.equal(run(hip, lengths, x, batch_scope=True), run(hip, lengths, x, batch_scope=False))


@pytest.mark.parametrize("prec,lengths,tag,tol", [("bf16", [960] * 64, "full.cfg3", 1.2e-2), ("f16", [3200] * 64, "full.cfg5h", 1.5e-3)])
def test_cfg3_cfg5_16bit_shapes(cfg, weights, hip, prec, lengths, tag, tol):
    """cfg3 (B = 64 x 3 s, bf16 operands) and cfg5 (64 segments of 10 s per GPU, fp16 operands): deterministic,
    finite, batch == per-utterance, and within the mode's stated tolerance of the fp32 engine on the same inputs."""
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0, precision=prec)
    m.load_weights({"speech_predictor": weights["speech_predictor"]}, which=7)
    x = make_inputs(lengths, tag)
    a = run(m, lengths, x)
    assert torch.equal(a, run(m, lengths, x))
    assert bool(torch.isfinite(a).all()) and float(a.abs().max()) <= 1.0
    L = lengths[0]
    for u in (0, len(lengths) // 2 + 5, len(lengths) - 1):
        one = run(m, [L], slice_inputs(x, L * u, L * (u + 1), u))
        # (tile shapes differ between the batch and the single run: a different fp32 summation order can flip individual
        #  16-bit operand roundings, so this is the mode's tolerance, not the fp32 one)
        assert float((one - a[75 * L * u : 75 * L * (u + 1)]).abs().max()) < tol
    ref = run(hip, lengths[:4], slice_inputs(x, 0, 4 * L, 0) | dict(style=x["style"][:4].contiguous()))
    assert float((ref - a[: 75 * 4 * L]).abs().max()) < tol
    m.close()
