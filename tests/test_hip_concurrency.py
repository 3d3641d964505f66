"""GPU: frame-rate stages give bit-identical results when several calls run at the same time on their own streams / host threads.

Round 4: with the split-fp32 contractions of other calls on the chip, `istft_frames_kernel` compiled with packed-fp32 instructions returned different
frames for identical inputs (csrc/signal.hip.h); `Synthesizer.map`'s test caught it about every second run.  This test runs
the stages of seven batches on three streams twenty times over: every output must equal the one of the same call made alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_stages_are_bit_stable_under_concurrent_streams(cfg, precision):
    from concurrent.futures import ThreadPoolExecutor

    from stylish_tts_amd import modules, synth
    from stylish_tts_amd.runtime import HipModel, Segments

    mods = modules.build_inference_modules(cfg, engine=HipModel(cfg, 0, precision=precision), synthetic_seed=0)
    eng = mods["speech_predictor"].engine
    for m in mods.values():
        m.engine
    devid = eng.device
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    lens = [[128], [195, 264], [264, 320, 130], [320], [120, 186], [197, 155, 330], [264]]
    cases = []
    for j, L in enumerate(lens):
        seg = Segments([4 * n for n in L], devid)
        R = seg.rows
        cases.append((seg, dict(asr=dev(synth.normal(f"cc.asr{j}", (R, cfg.inter_dim))), pitch=dev(np.abs(synth.normal(f"cc.f0{j}", (R,))) * 60 + 120),
                                energy=dev(synth.normal(f"cc.en{j}", (R,))), style=dev(synth.normal(f"cc.sty{j}", (len(L), cfg.style_dim))),
                                pn=dev(synth.normal(f"cc.pn{j}", (R, 128))), sn=dev(synth.normal(f"cc.sn{j}", (R * 75,))), ph=dev(synth.uniform(f"cc.ph{j}", (1,))))))

    def stages(seg, inp, ref=None):
        out = {}
        out["decoder"] = eng.decoder(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"])
        x = out["decoder"] if ref is None else ref["decoder"]
        out["prior_flow"] = eng.prior_flow(seg, x, inp["style"], inp["pn"])
        hs, hp = eng.harmonic_stft(seg, inp["pitch"], inp["sn"], inp["ph"], batch_scope=False)
        out["har_spec"], out["har_phase"] = hs, hp
        mel, hs0, hp0 = (out["prior_flow"], hs, hp) if ref is None else (ref["prior_flow"], ref["har_spec"], ref["har_phase"])
        out["vocoder"] = eng.vocoder(seg, mel, inp["style"], hs0, hp0)
        out["frame_path"] = eng.frame_path(seg, inp["asr"], inp["pitch"], inp["energy"], inp["style"], inp["pn"], inp["sn"], inp["ph"], batch_scope=False)
        torch.cuda.current_stream().synchronize()
        return out

    refs = [stages(s, i) for s, i in cases]
    workers = 3
    streams = [torch.cuda.Stream(device=devid) for _ in range(workers)]

    def lane(k):
        torch.cuda.set_device(devid)
        bad = []
        with torch.cuda.stream(streams[k]):
            for j in range(k, len(cases), workers):
                out = stages(cases[j][0], cases[j][1], refs[j])
                bad += [(j, name) for name in out if not torch.equal(out[name], refs[j][name])]
        return bad

    with ThreadPoolExecutor(max_workers=workers) as pool:
        bad = []
        for _ in range(20):
            for f in [pool.submit(lane, k) for k in range(workers)]:
                bad += f.result()
    assert not bad, f"{len(bad)} stage outputs changed under concurrency: {sorted(set(bad))[:12]}"


@pytest.mark.parametrize("batch", [8, 16])
def test_frame_path_with_its_side_streams_is_bit_stable_run_to_run(cfg, batch):
    """The benchmarked configuration runs the source / STFT / prior-conv chain on a side stream beside the decoder (B = 8 x 3 s) and, from 12 000 rows on, the decode blocks'
    learned shortcuts on a second one (B = 16): the same inputs must give the same bits every time."""
    from stylish_tts_amd import modules, synth
    from stylish_tts_amd.runtime import Segments

    mods = modules.build_inference_modules(cfg, synthetic_seed=0)
    eng = mods["speech_predictor"].engine
    for m in mods.values():
        m.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    seg = Segments([960] * batch, eng.device)
    R = seg.rows
    inp = [dev(synth.normal("fs.asr", (R, cfg.inter_dim))), dev(np.abs(synth.normal("fs.f0", (R,))) * 60 + 120), dev(synth.normal("fs.en", (R,))),
           dev(synth.normal("fs.sty", (batch, cfg.style_dim))), dev(synth.normal("fs.pn", (R, 128))), dev(synth.normal("fs.sn", (R * 75,))), dev(synth.uniform("fs.ph", (1,)))]
    ref = eng.frame_path(seg, *inp, batch_scope=False).clone()
    for i in range(25):
        y = eng.frame_path(seg, *inp, batch_scope=False)
        assert torch.equal(y, ref), f"run {i}: {int((y != ref).sum())} samples differ from the first run"
