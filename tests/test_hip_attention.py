"""GPU parity tests of the two attention kernels behind `run_attention` (csrc/phoneme.hip.h), called directly through the C-ABI's
test surface (stts_op_attention): one wave per four queries (attention_kernel) and the matrix-core kernel (attention_mfma_kernel),
against a float64 numpy restatement of MultiHeadAttention.attention (models/text_encoder.py:233-277) on packed ragged batches.

Tolerance: 5e-6 of the output's max-abs (fp32 dot products of 64 terms + a softmax over up to ~1 500 keys; the matrix-core kernel keeps
a running maximum / sum, so it differs from the two-pass softmax by fp32 rounding only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip(cfg):
    from stylish_tts_amd.runtime import HipModel

    m = HipModel(cfg, 0)
    yield m
    m.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def segs(lengths):
    from stylish_tts_amd.runtime import Segments

    return Segments(lengths, torch.device("cuda", 0))


def reference(q, k, v, q_lens, k_lens, heads, kc, centre=None, window=0):
    """Per utterance and head: softmax(q k^T / sqrt(kc) [- 1e4 inside the band]) v, in float64."""
    out = np.zeros((q.shape[0], heads * kc), np.float64)
    qo = np.concatenate([[0], np.cumsum(q_lens)])
    ko = np.concatenate([[0], np.cumsum(k_lens)])
    for u in range(len(q_lens)):
        for h in range(heads):
            cs = slice(h * kc, (h + 1) * kc)
            Q = q[qo[u] : qo[u + 1], cs].astype(np.float64)
            K = k[ko[u] : ko[u + 1], cs].astype(np.float64)
            V = v[ko[u] : ko[u + 1], cs].astype(np.float64)
            s = Q @ K.T / np.sqrt(kc)
            if centre is not None:
                j = np.arange(K.shape[0])[None, :]
                c = centre[qo[u] : qo[u + 1], None]
                s = s + np.where((j >= c - window) & (j <= c + window), -1e4, 0.0)
            s = s - s.max(axis=1, keepdims=True)
            p = np.exp(s)
            out[qo[u] : qo[u + 1], cs] = (p / p.sum(axis=1, keepdims=True)) @ V
    return out


def make(tag, q_lens, k_lens, heads, kc, spread=1.0):
    from stylish_tts_amd import synth

    q = synth.normal(tag + ".q", (int(sum(q_lens)), heads * kc)) * spread
    k = synth.normal(tag + ".k", (int(sum(k_lens)), heads * kc))
    v = synth.normal(tag + ".v", (int(sum(k_lens)), heads * kc))
    return q.astype(np.float32), k.astype(np.float32), v.astype(np.float32)


CASES = [
    # (name, q lengths, k lengths, heads): self-attention unless the key lengths differ
    ("cfm_like", [240, 131, 97, 800], None, 8),
    ("short_and_one_row", [1, 31, 33, 64, 5], None, 2),
    ("cross", [200, 75, 333], [50, 129, 32], 4),
    ("tile_edges", [128, 129, 127, 256, 96], None, 2),
]


@pytest.mark.parametrize("name,q_lens,k_lens,heads", CASES)
@pytest.mark.parametrize("kernel", [1, 2, 3])  # 1: one wave per four queries; 2: matrix cores (keys split over two wave groups from 128 keys on); 3: matrix cores, never split
def test_attention_kernels_vs_float64(hip, name, q_lens, k_lens, heads, kernel):
    kc = 64
    k_lens = k_lens or q_lens
    q, k, v = make("att." + name, q_lens, k_lens, heads, kc, spread=2.0)  # (spread: softmax rows far from uniform)
    want = reference(q, k, v, q_lens, k_lens, heads, kc)
    got = hip.op_attention(segs(q_lens), segs(k_lens), dev(q), dev(k), dev(v), heads, kc, kernel=kernel).cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - want).max() / np.abs(want).max()
    print(f"\n[attention {name} kernel {kernel}] max-abs err {err:.1e} of the output's max-abs")
    assert err < 5e-6, (name, kernel, err)


@pytest.mark.parametrize("kernel", [1, 2])
def test_attention_band_mask(hip, kernel):
    """The pitch/energy predictor's cross-attention: scores are lowered by 1e4 INSIDE |key - centre| <= window (the reference's inverted mask,
    pitch_energy_predictor.py:194-212), frames attending to the tokens of their utterance."""
    heads, kc, window = 2, 64, 3
    q_lens, k_lens = [400, 150, 90], [60, 110, 9]  # (more than 2 * window + 1 keys: a row whose keys ALL sit in the band keeps only
    # the fp32 remainder of score - 1e4, quantised to ~1e-3, and cannot be compared at this tolerance)
    q, k, v = make("att.band", q_lens, k_lens, heads, kc)
    rng = np.random.RandomState(3)
    centre = np.concatenate([np.sort(rng.randint(0, kl, ql)) for ql, kl in zip(q_lens, k_lens)]).astype(np.int32)
    want = reference(q, k, v, q_lens, k_lens, heads, kc, centre, window)
    got = hip.op_attention(segs(q_lens), segs(k_lens), dev(q), dev(k), dev(v), heads, kc, band_centre=dev(centre), window=window, kernel=kernel).cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    print(f"\n[attention band kernel {kernel}] max-abs err {err:.1e}")
    assert np.isfinite(got).all() and err < 5e-6, (kernel, err)


def test_matrix_core_attention_beyond_1024_keys(hip):
    """attention_kernel keeps an utterance's scores in LDS (<= 1024 keys); the matrix-core kernel streams the keys and has no such limit."""
    heads, kc = 2, 64
    lens = [1500, 1025]
    q, k, v = make("att.long", lens, lens, heads, kc)
    want = reference(q, k, v, lens, lens, heads, kc)
    got = hip.op_attention(segs(lens), segs(lens), dev(q), dev(k), dev(v), heads, kc, kernel=2).cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    assert np.isfinite(got).all() and err < 5e-6, err
    with pytest.raises(RuntimeError, match="more than 1024 keys"):
        hip.op_attention(segs(lens), segs(lens), dev(q), dev(k), dev(v), heads, kc, kernel=1)


@pytest.mark.parametrize("kc", [16, 32, 40, 96, 128, 160])
@pytest.mark.parametrize("kernel", [1, 2])
def test_other_matrix_core_head_sizes(hip, kc, kernel):
    """The other matrix-core head sizes (text encoders: 8 x 16, prosody encoders: 2 x 96 / 2 x 160, pitch / energy cross-attention: 8 x 40) on both kernels."""
    heads = 2
    q_lens, k_lens = [50, 130, 7, 257], [50, 130, 40, 100]
    q, k, v = make(f"att.kc{kc}", q_lens, k_lens, heads, kc, spread=1.5)
    want = reference(q, k, v, q_lens, k_lens, heads, kc)
    got = hip.op_attention(segs(q_lens), segs(k_lens), dev(q), dev(k), dev(v), heads, kc, kernel=kernel).cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    print(f"\n[attention kc {kc} kernel {kernel}] max-abs err {err:.1e}")
    assert np.isfinite(got).all() and err < 5e-6, (kc, kernel, err)


def test_other_head_sizes_stay_on_the_wave_kernel(hip):
    heads, kc = 2, 48
    lens = [130, 40]
    q, k, v = make("att.kc48", lens, lens, heads, kc)
    want = reference(q, k, v, lens, lens, heads, kc)
    got = hip.op_attention(segs(lens), segs(lens), dev(q), dev(k), dev(v), heads, kc).cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 5e-6
    with pytest.raises(RuntimeError, match="matrix-core kernel needs heads of 16 / 32 / 40 / 64 / 96 / 128 / 160"):
        hip.op_attention(segs(lens), segs(lens), dev(q), dev(k), dev(v), heads, kc, kernel=2)
