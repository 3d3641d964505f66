"""A model.yml that is not the default one: decoder / generator width 384 instead of 512, residual 32,
ConvNeXt intermediate 1152, so the flow runs on 96 channels (the fused 128-channel WaveNet kernels do not apply and every
layer runs as two contractions).  The reference built from that config produced tests/golden/frame_path_narrow.npz
(gen_golden.py:narrow_golden); the HIP engine built from the same config must reproduce it.  fp32 tolerances: 2e-4 of each
intermediate's max-abs, 1e-3 sample-wise on the waveform (atan2 branch ties adopted the reference's way, oracle.align_branch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_frame_path_with_non_default_widths():
    from oracle import stylish_oracle as O
    from stylish_tts_amd import params, synth
    from stylish_tts_amd.runtime import HipModel, Segments
    from test_oracle_golden import narrow_cfg

    cfg, g = narrow_cfg()
    dh, fh = cfg.decoder.hidden_dim, cfg.decoder.hidden_dim // 4
    assert (dh, fh) == (384, 96)
    w = params.synth_state_dict(params.module_spec("speech_predictor", cfg), 0, prefix="speech_predictor.")
    eng = HipModel(cfg, 0)
    eng.load_weights({"speech_predictor": w}, which=7)
    T4 = 64
    s = Segments([T4], eng.device)
    asr = synth.normal("nw.asr", (1, 128, T4))
    pitch = synth.pitch_curve("nw.pitch", 1, T4)
    energy = (synth.uniform("nw.energy", (1, T4)) * 2.0 + 2.0).astype(np.float32)
    style = (synth.normal("nw.style", (1, 64)) * 0.7).astype(np.float32)
    nz = synth.path_noise("narrow64", 1, T4, flow_dim=fh)

    def rel(a, b):
        return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())

    x = eng.decoder(s, dev(asr[0].T), dev(pitch[0]), dev(energy[0]), dev(style))
    assert x.shape == (T4, dh) and rel(x.cpu().numpy().T[None], g["x"]) < 2e-4
    mel, zp, zf = eng.prior_flow(s, x, dev(style), dev(nz["prior_noise"][0].T), return_z=True)
    assert zp.shape == (T4, fh)
    errs = dict(z=rel(zp.cpu().numpy().T[None], g["z"]), z_out=rel(zf.cpu().numpy().T[None], g["z_out"]), mel=rel(mel.cpu().numpy().T[None], g["mel"]))
    spec, phase = eng.harmonic_stft(s, dev(pitch[0]), dev(nz["src_noise"].reshape(-1)), dev(nz["init_phase"].reshape(-1)))
    ph = phase.cpu().numpy()[:, :1025].T[None]
    sp = spec.cpu().numpy()[:, :1025].T[None]
    ph, bad = O.align_branch(ph, (g["cut_idx"].astype(np.int64), g["cut_phase"].astype(np.float32)), sp, return_bad=True)
    assert bad == 0
    phz = np.zeros((T4, 1056), np.float32)
    phz[:, :1025] = ph[0].T
    audio = eng.vocoder(s, mel, dev(style), spec, dev(phz)).cpu().numpy()
    errs["audio"] = float(np.abs(audio - g["audio"].reshape(-1)).max())
    # the fused entry point and a ragged batch (second utterance shorter) run the same widths
    fused = eng.frame_path(s, dev(asr[0].T), dev(pitch[0]), dev(energy[0]), dev(style), dev(nz["prior_noise"][0].T), dev(nz["src_noise"].reshape(-1)),
                           dev(nz["init_phase"].reshape(-1)))
    assert torch.equal(fused, eng.vocoder(s, mel, dev(style), spec, phase))
    print("\n[non-default model.yml: widths 384 / 96 / 1152]", {k: f"{v:.1e}" for k, v in errs.items()})
    eng.close()
    assert errs["z"] < 2e-4 and errs["z_out"] < 2e-4 and errs["mel"] < 2e-4 and errs["audio"] < 1e-3, errs
