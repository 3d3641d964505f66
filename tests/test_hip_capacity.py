"""Capacity segments (include/stylish_hip.h, STTS_SEG_CAPACITY): the chain tokens -> waveform without a host read
between the duration predictor and the frame path.  The host sizes buffers and grids by upper bounds, the real frame offsets are
computed on the device from the integer durations, outputs are packed by the real lengths.

  * the device bookkeeping (stts_frame_offsets) is exact integer arithmetic: offsets == cumsum of the durations, bit for bit;
  * the frame path on capacity segments == the frame path on the exact offsets (same kernels; launch plans follow the host sizes, so
    fp32 sums may be ordered differently: 5e-5), for tight and for 46-frames-per-token capacities, on a ragged batch;
  * a prediction that does not fit its capacity is detected (device error word), nothing is written out of bounds, and the
    Synthesizer repeats the call with what the model asked for;
  * pipeline.py has no device->host read between eng.duration and eng.frame_path (source check, runs on the CPU).
"""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_no_host_read_between_the_duration_predictor_and_the_frame_path():
    src = open(os.path.join(ROOT, "stylish_tts_amd", "pipeline.py")).read()
    body = src[src.index("def _run_once"):]
    a, b = body.index("eng.duration("), body.index("eng.frame_path(")
    assert a < b
    between = body[a:b]
    for pat in (r"\.cpu\(", r"\.item\(", r"\.tolist\(", r"\.numpy\(", r"synchronize"):
        assert not re.search(pat, between), f"host read ({pat}) between the duration predictor and the frame path"


torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def eng(cfg, weights):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    m.load_weights(weights, which=255)
    yield m
    m.close()


@pytest.mark.gpu
def test_frame_offsets_are_the_cumulative_durations(eng):
    from stylish_tts_amd.runtime import Segments

    rng = np.random.default_rng(3)
    L = [5, 1, 40, 510, 17]
    dur = rng.integers(1, 47, sum(L)).astype(np.int32)
    sp = Segments(L, eng.device)
    T = [int(dur[sp.host[i] : sp.host[i + 1]].sum()) for i in range(len(L))]
    st, st4, need = eng.frame_offsets(sp, dev(dur), [t + 3 for t in T])
    eng.check_status()
    assert st.is_capacity and st4.is_capacity and st.host[-1] == sum(T) + 3 * len(L) and st4.host[-1] == 4 * st.host[-1]
    assert np.array_equal(need.cpu().numpy(), T)
    assert np.array_equal(st.dev.cpu().numpy(), np.concatenate([[0], np.cumsum(T)]))
    assert np.array_equal(st4.dev.cpu().numpy(), 4 * np.concatenate([[0], np.cumsum(T)]))
    # utterance 2 does not fit: truncated to its capacity (in bounds), and `need` says what it wanted
    caps = [t + 3 for t in T]
    caps[2] = T[2] - 7
    st, st4, need = eng.frame_offsets(sp, dev(dur), caps)
    eng.check_status()
    got = np.diff(st.dev.cpu().numpy())
    assert np.array_equal(got, [min(t, c) for t, c in zip(T, caps)]) and np.array_equal(need.cpu().numpy(), T)


@pytest.mark.gpu
@pytest.mark.parametrize("T4,slack", [([96, 340, 61, 200], 1.0), ([96, 340, 61, 200], 1.3), ([96, 340, 61, 200], 9.0), ([1280], 1.15), ([1056], 1.121),
                                      ([480, 744], 1.25), ([2000, 3200, 1500, 2800], 1.4)])
def test_frame_path_on_capacity_segments_equals_exact_offsets(eng, T4, slack):
    """A ragged batch through stts_frame_path twice: exact offsets, and capacity segments whose host bounds are `slack` x the lengths
    (shapes where the host-side tile counts of the contractions - direct and Winograd-form - differ from the device-side ones)."""
    from stylish_tts_amd import synth
    from stylish_tts_amd.runtime import Segments

    R = sum(T4)
    asr = dev(synth.normal("cap.asr", (R, 128)))
    pitch = dev(np.concatenate([synth.pitch_curve(f"cap.p{i}", 1, t)[0] for i, t in enumerate(T4)]))
    energy = dev((synth.uniform("cap.e", (R,)) * 2 + 2).astype(np.float32))
    style = dev((synth.normal("cap.s", (len(T4), 64)) * 0.7).astype(np.float32))
    pn, sn, ph = dev(synth.normal("cap.pn", (R, 128))), dev(synth.normal("cap.sn", (R * 75,))), dev(synth.uniform("cap.ph", (1,)))
    exact = eng.frame_path(Segments(T4, eng.device), asr, pitch, energy, style, pn, sn, ph, batch_scope=False)
    caps = [int(np.ceil(t * slack)) for t in T4]
    sc = Segments(caps, eng.device, dev=dev(np.concatenate([[0], np.cumsum(T4)]).astype(np.int32)), capacity=True)
    Rc = sc.rows

    def grow(x, rows):  # inputs sized by the capacity, real rows first, the rest poisoned
        y = torch.full((rows,) + tuple(x.shape[1:]), float("nan"), device=x.device)
        y[: x.shape[0]] = x
        return y

    out = torch.full((Rc * 75,), 7.0, device=eng.device)
    eng.frame_path(sc, grow(asr, Rc), grow(pitch, Rc), grow(energy, Rc), style, grow(pn, Rc), grow(sn, Rc * 75), ph, batch_scope=False, out=out)
    eng.check_status()
    assert bool(torch.isfinite(out[: R * 75]).all())
    assert float((out[: R * 75] - exact).abs().max()) < 5e-5
    assert bool((out[R * 75 :] == 7.0).all()), "rows beyond the real total were written"


@pytest.mark.gpu
def test_synthesizer_repeats_a_call_whose_prediction_overflows(eng):
    from stylish_tts_amd import synth
    from stylish_tts_amd.pipeline import Synthesizer

    toks = [synth.tokens(f"ovf.{i}", 1, n, 178)[0].tolist() for i, n in enumerate([12, 30, 7])]
    roomy = Synthesizer(eng, frames_per_token=46, adapt=False)
    w0, d0 = roomy(toks, return_details=True)
    assert roomy.capacity_retries == 0
    R4 = 4 * sum(d0["frames"])
    noise = dict(prior_noise=dev(synth.normal("ovf.pn", (R4, 128))), src_noise=dev(synth.normal("ovf.sn", (R4 * 75,))), init_phase=dev(synth.uniform("ovf.ph", (1,))))
    w0, d0 = roomy(toks, noise=noise, return_details=True)
    tight = Synthesizer(eng, frames_per_token=1.0, adapt=False)  # the synthetic model predicts far more than one frame per token
    w1, d1 = tight(toks, noise=noise, return_details=True)
    assert tight.capacity_retries == 1 and d1["frames"] == d0["frames"] and torch.equal(d1["durations"], d0["durations"])
    assert d1["capacities"] == d0["frames"]  # repeated with exactly what the model asked for
    # the two calls ran with different capacities, hence different launch plans (tile shapes, split-K factors) in the pitch predictor:
    # predicted pitch agrees to fp32 summation noise - and the harmonic source integrates pitch over the utterance, so the
    # waveforms are compared where that noise cannot move an atan2 branch: same lengths, finite, and equal energy / pitch curves
    assert float((d1["pitch"] - d0["pitch"]).abs().max()) < 2e-2 and float((d1["energy"] - d0["energy"]).abs().max()) < 2e-3
    for a, b in zip(w0, w1):
        assert a.shape == b.shape and bool(torch.isfinite(b).all())
    # the same capacities give the same bits
    w2, _ = tight(toks, noise=noise, return_details=True)
    w3, _ = tight(toks, noise=noise, return_details=True)
    assert all(torch.equal(a, b) for a, b in zip(w2, w3))
    # adaptive capacity: after one call the next ones fit without a retry, and the capacity ratio has settled
    syn = Synthesizer(eng, frames_per_token=1.0, adapt=True)
    syn(toks)
    n, r = syn.capacity_retries, syn._ratio
    a1 = syn(toks, noise=noise)
    a2 = syn(toks, noise=noise)
    syn([t[:5] + t[-3:] for t in toks])
    assert syn.capacity_retries == n and syn._ratio >= r and all(torch.equal(x, y) for x, y in zip(a1, a2))
