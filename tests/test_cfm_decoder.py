"""CfmMelDecoder (models/cfm/cfm_mel_decoder.py:190-413): the XUT estimator of the flow-matching mel decoder and the Euler sampling
through it, against vectors produced by the reference itself with synthetic weights (tests/golden/gen_golden.py:cfm_decoder_golden)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import stylish_oracle as O
from stylish_tts_amd import params

SMALL = dict(feat_dim=80, asr_dim=96, spk_dim=48, hidden_dim=128, emb_dim=64, depth=2, enc_blocks=1, dec_blocks=2, prev_depth=1, post_depth=1)
CASES = {"default": dict(params.CFM_DEFAULT_DIMS), "small": dict(params.CFM_DEFAULT_DIMS, **SMALL)}


def _weights(dims):
    return params.synth_state_dict(params.cfm_mel_decoder_spec(dims), 0, prefix="cfm_mel_decoder.")


@pytest.mark.parametrize("tag", ["default", "small"])
def test_oracle_estimator_matches_reference(tag):
    g, dims = load_golden("cfm_decoder"), CASES[tag]
    y = O.cfm_mel_decoder_forward(g[tag + "_x"], g[tag + "_asr"], g[tag + "_f0"], g[tag + "_n"], g[tag + "_spk"], g[tag + "_t"], g[tag + "_nz"],
                                  _weights(dims), dims)
    ref = g[tag + "_y"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() < 3e-5 * np.abs(ref).max()  # measured 9e-6 / 3e-6 (16 / 6 transformer blocks in fp32 on both sides)


def test_oracle_sampling_matches_reference():
    g, dims = load_golden("cfm_decoder"), CASES["small"]
    sd, k = _weights(dims), [0]

    def estimator(x, t):
        nz = g[f"sample_nz{k[0]}"]
        k[0] += 1
        return O.cfm_mel_decoder_forward(x, g["small_asr"], g["small_f0"], g["small_n"], g["small_spk"], t, nz, sd, dims)

    y = O.cfm_solve_euler(g["sample_z"], int(g["sample_steps"]), estimator, float(g["sample_temperature"]))
    assert k[0] == int(g["sample_steps"])
    assert np.abs(y - g["sample_y"]).max() < 2e-5 * np.abs(g["sample_y"]).max()


def test_inventory_matches_reference_shapes():
    """The spec restates the reference state_dict (checked key by key against the live module when the fixture was generated);
    here: sizes follow the dims, every block of the U has its tensors, cross-attention only on the first decoder block of a level."""
    spec = params.spec_shapes(params.cfm_mel_decoder_spec())
    assert spec["in_proj.weight"] == (256, 80 + 256 + 256) and spec["backbone.dec_blocks.3.0.xattn.kv.weight"] == (512, 256)
    assert "backbone.dec_blocks.3.1.xattn.kv.weight" not in spec and "backbone.enc_blocks.0.0.xattn.q.weight" not in spec
    assert spec["post_tread_trns.blocks.2.attn.rope.freqs"] == (4, 32, 1)
    assert params.count_params(params.cfm_mel_decoder_spec()) == 23634097


# ----------------------------------------------------------------------------------------------------------------- HIP path
def _hip(dims):
    import torch  # noqa: F401

    from stylish_tts_amd.cfm_decoder import CfmMelDecoder

    m = CfmMelDecoder(feat_dim=dims["feat_dim"], asr_dim=dims["asr_dim"], spk_dim=dims["spk_dim"], hidden_dim=dims["hidden_dim"], emb_dim=dims["emb_dim"],
                      xut_depth=dims["depth"], xut_enc_blocks=dims["enc_blocks"], xut_dec_blocks=dims["dec_blocks"],
                      tread_config={"prev_trns_depth": dims["prev_depth"], "post_trns_depth": dims["post_depth"], "dropout_ratio": 0.5})
    return m.load_state_dict(_weights(dims))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["default", "small"])
def test_hip_estimator_matches_reference(tag):
    import torch

    g, dims = load_golden("cfm_decoder"), CASES[tag]
    m = _hip(dims)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    y = m._forward(d(g[tag + "_x"]), d(g[tag + "_asr"]), d(g[tag + "_f0"]), d(g[tag + "_n"]), d(g[tag + "_spk"]), d(g[tag + "_t"]),
                   sine_noise=d(g[tag + "_nz"])).cpu().numpy()
    ref = g[tag + "_y"]
    assert y.shape == ref.shape and np.isfinite(y).all()
    # fp32 on both sides; 16 (6) transformer blocks deep, different summation orders
    assert np.abs(y - ref).max() < 1e-4 * np.abs(ref).max(), np.abs(y - ref).max()


@pytest.mark.gpu
def test_hip_sampling_matches_reference():
    import torch

    g, dims = load_golden("cfm_decoder"), CASES["small"]
    m = _hip(dims)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    steps = int(g["sample_steps"])
    y = m(d(g["small_asr"]), d(g["small_f0"]), d(g["small_n"]), d(g["small_spk"]), steps, float(g["sample_temperature"]), z=d(g["sample_z"]),
          sine_noise=[d(g[f"sample_nz{i}"]) for i in range(steps)]).cpu().numpy()
    assert np.abs(y - g["sample_y"]).max() < 1e-4 * np.abs(g["sample_y"]).max()


@pytest.mark.gpu
def test_hip_packed_ragged_batch_is_per_utterance():
    """Utterances of different lengths in one packed call (the reference needs equal lengths in a batch) = each one alone, and
    = the oracle; F0 / N curves of another length are resampled per utterance."""
    import torch

    from stylish_tts_amd import synth
    from stylish_tts_amd.runtime import Segments

    dims = CASES["small"]
    m, sd = _hip(dims), _weights(dims)
    lens, clens = [41, 7, 96], [41, 20, 50]
    rng = lambda nm, shape: synth.normal("cfmr." + nm, shape)  # noqa: E731
    xs = [rng(f"x{i}", (1, dims["feat_dim"], n)) for i, n in enumerate(lens)]
    asrs = [rng(f"a{i}", (1, dims["asr_dim"], n)) for i, n in enumerate(lens)]
    f0s = [synth.pitch_curve(f"cfmr.f{i}", 1, L) for i, L in enumerate(clens)]
    ncs = [(synth.uniform(f"cfmr.n{i}", (1, L)) * 2 + 2).astype(np.float32) for i, L in enumerate(clens)]
    spk = rng("spk", (3, dims["spk_dim"]))
    t = np.array([0.1, 0.5, 0.9], np.float32)
    nzs = [rng(f"z{i}", (1, n, 1)) for i, n in enumerate(lens)]
    dev = m.device
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ld_asr = (dims["asr_dim"] + 31) // 32 * 32
    xr = d(np.concatenate([x[0].T for x in xs]))
    ar = torch.zeros(sum(lens), ld_asr, device=dev)
    ar[:, : dims["asr_dim"]] = d(np.concatenate([a[0].T for a in asrs]))
    seg, cseg = Segments(lens, dev), Segments(clens, dev)
    out = m.estimator_packed(seg, xr, ar, d(np.concatenate([f[0] for f in f0s])), d(np.concatenate([c[0] for c in ncs])), cseg, d(spk), d(t),
                             d(np.concatenate([z.reshape(-1) for z in nzs]))).cpu().numpy()
    for i, n in enumerate(lens):
        ref = O.cfm_mel_decoder_forward(xs[i], asrs[i], f0s[i], ncs[i], spk[i : i + 1], t[i : i + 1], nzs[i], sd, dims)[0].T
        got = out[seg.host[i] : seg.host[i + 1]]
        assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max(), (i, np.abs(got - ref).max())
        alone = m._forward(d(xs[i]), d(asrs[i]), d(f0s[i]), d(ncs[i]), d(spk[i : i + 1]), d(t[i : i + 1]), sine_noise=d(nzs[i])).cpu().numpy()[0].T
        assert np.abs(got - alone).max() < 2e-6 * np.abs(ref).max()


@pytest.mark.gpu
def test_hip_estimator_at_the_model_config_size():
    """The instance the reference's build_model constructs (models/models.py:65-70 with the default model.yml): hidden 512 (8 heads of
    64), HuBERT width 768, speaker-embedding width 10 240.  No reference vector at this size: against the oracle pinned above."""
    import torch

    from stylish_tts_amd import synth
    from stylish_tts_amd.cfm_decoder import CfmMelDecoder

    m = CfmMelDecoder.from_model_config()
    dims = dict(params.CFM_DEFAULT_DIMS, feat_dim=80, asr_dim=768, spk_dim=10240, hidden_dim=512)
    assert (m.dims.hidden_dim, m.dims.asr_dim, m.dims.spk_dim, m.dims.feat_dim) == (512, 768, 10240, 80)
    sd = _weights(dims)
    m.load_state_dict(sd)
    B, n, L = 2, 45, 31
    x, asr = synth.normal("cfmp.x", (B, 80, n)), synth.normal("cfmp.a", (B, 768, n))
    f0, nc = synth.pitch_curve("cfmp.f", B, L), (synth.uniform("cfmp.n", (B, L)) * 2 + 2).astype(np.float32)
    spk, t, nz = synth.normal("cfmp.s", (B, 10240)), np.array([0.2, 0.7], np.float32), synth.normal("cfmp.z", (B, n, 1))
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    y = m._forward(d(x), d(asr), d(f0), d(nc), d(spk), d(t), sine_noise=d(nz)).cpu().numpy()
    ref = O.cfm_mel_decoder_forward(x, asr, f0, nc, spk, t, nz, sd, dims)
    assert np.abs(y - ref).max() < 1e-4 * np.abs(ref).max(), np.abs(y - ref).max()


@pytest.mark.gpu
def test_hip_graph_sampling_equals_eager_sampling():
    """forward(graph=True): the estimator captured into a HIP graph and replayed per Euler step gives the eager result bit for bit."""
    import torch

    g, dims = load_golden("cfm_decoder"), CASES["small"]
    m = _hip(dims)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    steps = int(g["sample_steps"])
    args = (d(g["small_asr"]), d(g["small_f0"]), d(g["small_n"]), d(g["small_spk"]), steps, float(g["sample_temperature"]))
    nzs = [d(g[f"sample_nz{i}"]) for i in range(steps)]
    eager = m(*args, z=d(g["sample_z"]), sine_noise=nzs)
    graph = m(*args, z=d(g["sample_z"]), sine_noise=nzs, graph=True)
    assert torch.equal(eager, graph)
    assert np.abs(graph.cpu().numpy() - g["sample_y"]).max() < 1e-4 * np.abs(g["sample_y"]).max()


@pytest.mark.gpu
def test_hip_estimator_beyond_1024_frames():
    """An utterance of 1 100 frames (the matrix-core attention streams the keys: no 1 024-key limit with head_dim 64) next to a short one,
    against the oracle; also exercises the blocked double-precision prefix sums of the sine source over more than four frames per thread."""
    import torch

    from stylish_tts_amd import synth
    from stylish_tts_amd.runtime import Segments

    dims = CASES["small"]
    m, sd = _hip(dims), _weights(dims)
    lens = [1100, 37]
    rng = lambda nm, shape: synth.normal("cfml." + nm, shape)  # noqa: E731
    xs = [rng(f"x{i}", (1, dims["feat_dim"], n)) for i, n in enumerate(lens)]
    asrs = [rng(f"a{i}", (1, dims["asr_dim"], n)) for i, n in enumerate(lens)]
    f0s = [synth.pitch_curve(f"cfml.f{i}", 1, n) for i, n in enumerate(lens)]
    ncs = [(synth.uniform(f"cfml.n{i}", (1, n)) * 2 + 2).astype(np.float32) for i, n in enumerate(lens)]
    spk = rng("spk", (2, dims["spk_dim"]))
    t = np.array([0.3, 0.8], np.float32)
    nzs = [rng(f"z{i}", (1, n, 1)) for i, n in enumerate(lens)]
    dev = m.device
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ld_asr = (dims["asr_dim"] + 31) // 32 * 32
    ar = torch.zeros(sum(lens), ld_asr, device=dev)
    ar[:, : dims["asr_dim"]] = d(np.concatenate([a[0].T for a in asrs]))
    seg = Segments(lens, dev)
    out = m.estimator_packed(seg, d(np.concatenate([x[0].T for x in xs])), ar, d(np.concatenate([f[0] for f in f0s])), d(np.concatenate([c[0] for c in ncs])), seg,
                             d(spk), d(t), d(np.concatenate([z.reshape(-1) for z in nzs]))).cpu().numpy()
    for i, n in enumerate(lens):
        ref = O.cfm_mel_decoder_forward(xs[i], asrs[i], f0s[i], ncs[i], spk[i : i + 1], t[i : i + 1], nzs[i], sd, dims)[0].T
        got = out[seg.host[i] : seg.host[i + 1]]
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max(), (i, np.abs(got - ref).max(), np.abs(ref).max())
