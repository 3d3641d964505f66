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
