"""Parameter inventory of the inference hot path + name-keyed synthetic weights.

The inventory restates, as data, the ``state_dict`` keys and shapes of the five
modules the reference's inference composition uses
(``models/export_model.py:35-45``): ``speech_predictor``, ``duration_predictor``,
``pitch_energy_predictor``, ``pe_text_encoder``, ``pe_text_style_encoder``
(factory: ``models/models.py:32-63``; module order on disk ``models.py:79-101``).
Two weight-norm flavours occur: the parametrization form
``….parametrizations.weight.original0/1`` (decoder, ``models/decoder.py:35-45``,
``models/ada_norm.py:158-163``) and the legacy ``weight_g/weight_v`` form (flow
``WN``, ``models/flow.py:40,52,60``).

No checkpoint ships with the reference, so parity fixtures use weights that both
sides can regenerate from a seed: every element is a counter hash
(splitmix64 of ``fnv1a64(name) + seed*K + index``) mapped to a uniform value and
scaled by the tensor's role.  Zero-initialised tensors of the reference
(``models/flow.py:193-194,279-280,308-309``, ``models/text_encoder.py:76-77``,
``models/generator.py:493-494``) get non-zero values so the flow, the prenet
residual and GRN are actually exercised.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np

Spec = List[Tuple[str, Tuple[int, ...], str]]


# ----------------------------------------------------------------------------
# inventory builders (one per reference module)
# ----------------------------------------------------------------------------
def _conv(p: str, cout: int, cin: int, k: int, bias: bool = True) -> Spec:
    s = [(p + ".weight", (cout, cin, k), "w")]
    if bias:
        s.append((p + ".bias", (cout,), "b"))
    return s


def _linear(p: str, cout: int, cin: int) -> Spec:
    return [(p + ".weight", (cout, cin), "w"), (p + ".bias", (cout,), "b")]


def _wn_conv_param(p: str, cout: int, cin: int, k: int, bias: bool = True) -> Spec:
    """torch.nn.utils.parametrizations.weight_norm(Conv1d) keys (bias first, as registered)."""
    s: Spec = []
    if bias:
        s.append((p + ".bias", (cout,), "b"))
    s.append((p + ".parametrizations.weight.original0", (cout, 1, 1), "wn_g"))
    s.append((p + ".parametrizations.weight.original1", (cout, cin, k), "wn_v"))
    return s


def _attn(p: str, c: int) -> Spec:
    """MultiHeadAttention 1x1 convs, registration order q, k, v, o (models/text_encoder.py:191-199)."""
    s: Spec = []
    for nm, kind in (("conv_q", "w_qk"), ("conv_k", "w_qk"), ("conv_v", "w"), ("conv_o", "w_attn_o")):
        s += [(p + nm + ".weight", (c, c, 1), kind), (p + nm + ".bias", (c,), "b")]
    return s


def _adain(p: str, style: int, c: int) -> Spec:
    # AdaptiveInstance / AdaptiveLayerNorm: fc = Linear(style, 2C)  (ada_norm.py:133,191)
    return [(p + ".fc.weight", (2 * c, style), "w_style"), (p + ".fc.bias", (2 * c,), "b_style")]


def adaptive_decoder_block_spec(p: str, cin: int, cout: int, style: int) -> Spec:
    """AdaptiveDecoderBlock (models/ada_norm.py:142-182)."""
    s: Spec = []
    s += _wn_conv_param(p + ".conv1", cout, cin, 3)
    s += _wn_conv_param(p + ".conv2", cout, cout, 3)
    s += _adain(p + ".norm1", style, cin)
    s += _adain(p + ".norm2", style, cout)
    if cin != cout:
        s += _wn_conv_param(p + ".conv1x1", cout, cin, 1, bias=False)
    return s


def text_encoder_spec(p: str, cfg, inter_dim: int) -> Spec:
    """TextEncoder (models/text_encoder.py:397-462)."""
    te = cfg.text_encoder
    c = te.hidden_dim
    s: Spec = [(p + "emb.weight", (te.tokens, c), "emb")]
    for i in range(3):
        s += _conv(p + f"prenet.conv_layers.{i}", c, c, 5)
    for i in range(3):
        s += [(p + f"prenet.norm_layers.{i}.gamma", (c,), "ln_g"), (p + f"prenet.norm_layers.{i}.beta", (c,), "ln_b")]
    s += [(p + "prenet.proj.weight", (c, c, 1), "w_small"), (p + "prenet.proj.bias", (c,), "b")]
    n = te.layers
    for i in range(n):
        s += _attn(p + f"encoder.attn_layers.{i}.", c)
    for i in range(n):
        s += [(p + f"encoder.norm_layers_1.{i}.gamma", (c,), "ln_g"), (p + f"encoder.norm_layers_1.{i}.beta", (c,), "ln_b")]
    for i in range(n):
        s += _conv(p + f"encoder.ffn_layers.{i}.conv_1", te.filter_channels, c, te.kernel_size)
        s += _conv(p + f"encoder.ffn_layers.{i}.conv_2", c, te.filter_channels, te.kernel_size)
    for i in range(n):
        s += [(p + f"encoder.norm_layers_2.{i}.gamma", (c,), "ln_g"), (p + f"encoder.norm_layers_2.{i}.beta", (c,), "ln_b")]
    s += _conv(p + "proj_m", inter_dim, c, 1)
    return s


def text_style_encoder_spec(p: str, cfg, inter_dim: int) -> Spec:
    """TextStyleEncoder + BasicConvNeXtBlock (models/text_style_encoder.py:6-26, models/conv_next.py:17-51)."""
    sd = cfg.style_dim
    s: Spec = _conv(p + "conv_in", sd, inter_dim, 7)
    for i in range(cfg.style_encoder.layers):
        q = p + f"blocks.{i}."
        s += [(q + "dwconv.weight", (sd, 1, 7), "w"), (q + "dwconv.bias", (sd,), "b")]
        s += [(q + "norm.weight", (sd,), "ln_g"), (q + "norm.bias", (sd,), "ln_b")]
        s += _linear(q + "pwconv1", 4 * sd, sd)
        s += [(q + "grn.gamma", (1, 1, 4 * sd), "grn"), (q + "grn.beta", (1, 1, 4 * sd), "grn")]
        s += _linear(q + "pwconv2", sd, 4 * sd)
    return s


def prosody_encoder_spec(p: str, style: int, d_model: int, nlayers: int) -> Spec:
    """ProsodyEncoder (models/prosody_encoder.py:10-81)."""
    c = d_model + style
    s: Spec = []
    for i in range(nlayers):
        s += _attn(p + f"attn_layers.{i}.", c)
    for i in range(nlayers):
        s += _adain(p + f"norm_layers_1.{i}", style, c)
    for i in range(nlayers):
        s += _conv(p + f"ffn_layers.{i}.conv_1", 2 * c, c, 1)
        s += _conv(p + f"ffn_layers.{i}.conv_2", c, 2 * c, 1)
    for i in range(nlayers):
        s += _adain(p + f"norm_layers_2.{i}", style, c)
    for i in range(nlayers):
        s += _conv(p + f"proj_layers.{i}", d_model, c, 1)
    return s


def duration_predictor_spec(cfg) -> Spec:
    """DurationPredictor (models/duration_predictor.py:8-36)."""
    s: Spec = []
    s += text_encoder_spec("text_encoder.", cfg, cfg.inter_dim)
    s += text_style_encoder_spec("style_encoder.", cfg, cfg.inter_dim)
    s += prosody_encoder_spec("prosody_encoder.", cfg.style_dim, cfg.inter_dim, cfg.duration_predictor.n_layer)
    c = cfg.inter_dim + cfg.style_dim
    s += [
        ("duration_proj.linear_layer.weight", (cfg.duration_predictor.duration_classes, c), "w_dur"),
        ("duration_proj.linear_layer.bias", (cfg.duration_predictor.duration_classes,), "b"),
    ]
    return s


def pitch_energy_predictor_spec(cfg) -> Spec:
    """PitchEnergyPredictor (models/pitch_energy_predictor.py:11-121)."""
    inter = cfg.pitch_energy_predictor.inter_dim
    st = cfg.style_dim
    c = inter + st
    s: Spec = prosody_encoder_spec("prosody_encoder.", st, inter, 3)
    s += _adain("query_norm", st, c)
    s += _adain("key_norm", st, c)
    s += _attn("cross_attention.", c)
    s += _wn_conv_param("cross_post.0", c, 1, 5)  # depthwise, groups=c
    s += _wn_conv_param("cross_post.2", c, c, 1)
    for br in ("F0", "N"):
        for i in range(3):
            s += adaptive_decoder_block_spec(f"{br}.{i}", c, c, st)
    s += [("F0_proj.weight", (1, c, 1), "w_f0"), ("F0_proj.bias", (1,), "b_f0")]
    s += [("N_proj.weight", (1, c, 1), "w"), ("N_proj.bias", (1,), "b")]
    return s


def decoder_spec(p: str, dim_in: int, style: int, hidden: int, residual: int) -> Spec:
    """Decoder (models/decoder.py:6-45)."""
    s: Spec = adaptive_decoder_block_spec(p + "encode", dim_in + 2, hidden, style)
    for i in range(4):
        s += adaptive_decoder_block_spec(p + f"decode.{i}", hidden + 2 + residual, hidden, style)
    s += _wn_conv_param(p + "F0_conv", 1, 1, 3)
    s += _wn_conv_param(p + "N_conv", 1, 1, 3)
    s += _wn_conv_param(p + "asr_res.0", residual, dim_in, 1)
    return s


def flow_spec(p: str, channels: int, hidden: int, k: int, n_layers: int, n_flows: int, gin: int) -> Spec:
    """ResidualCouplingBlock / ResidualCouplingLayer / WN (models/flow.py:17-218), legacy weight_g/_v keys.

    ``flows`` alternates coupling layers (even indices) and parameter-free ``Flip`` (odd)."""
    half = channels // 2
    s: Spec = []
    for f in range(n_flows):
        q = p + f"flows.{2 * f}."
        s += [(q + "pre.weight", (hidden, half), "w"), (q + "pre.bias", (hidden,), "b")]
        for i in range(n_layers):
            s += [
                (q + f"enc.in_layers.{i}.bias", (2 * hidden,), "b"),
                (q + f"enc.in_layers.{i}.weight_g", (2 * hidden, 1, 1), "wn_g"),
                (q + f"enc.in_layers.{i}.weight_v", (2 * hidden, hidden, k), "wn_v"),
            ]
        for i in range(n_layers):
            rs = 2 * hidden if i < n_layers - 1 else hidden
            s += [
                (q + f"enc.res_skip_layers.{i}.bias", (rs,), "b"),
                (q + f"enc.res_skip_layers.{i}.weight_g", (rs, 1), "wn_g_half"),
                (q + f"enc.res_skip_layers.{i}.weight_v", (rs, hidden), "wn_v"),
            ]
        s += [
            (q + "enc.cond_layer.bias", (2 * hidden * n_layers,), "b"),
            (q + "enc.cond_layer.weight_g", (2 * hidden * n_layers, 1), "wn_g_half"),
            (q + "enc.cond_layer.weight_v", (2 * hidden * n_layers, gin), "wn_v"),
        ]
        s += [(q + "proj_mean.weight", (half, hidden), "w_small"), (q + "proj_mean.bias", (half,), "b")]
        s += [(q + "proj_logstd.weight", (half, hidden), "w_tiny"), (q + "proj_logstd.bias", (half,), "b_tiny")]
    return s


def generator_spec(p: str, cfg) -> Spec:
    """Generator 'freegan' (models/generator.py:340-438) + ConvNeXtBlock/GRN (:441-499)."""
    g = cfg.generator
    st = cfg.style_dim
    nbin = cfg.n_fft // 2 + 1
    h = g.hidden_dim
    k = g.io_conv_kernel_size
    s: Spec = []
    s += [(p + "amp_output_conv.weight", (nbin, h + h // 2, k), "w"), (p + "amp_output_conv.bias", (nbin,), "b_logamp")]
    s += _conv(p + "phase_output_conv", nbin, h + h // 2, k)
    s += _adain(p + "amp_final_layer_norm", st, h)
    s += _adain(p + "phase_final_layer_norm", st, h)
    s += _conv(p + "projector", h, g.input_dim + h, 1)
    s += _conv(p + "amp_prior_conv", h // 2, nbin, 7)
    s += _conv(p + "phase_prior_conv", h // 2, nbin, 7)
    for i, kk in enumerate((31, 15, 7, 3)):
        q = p + f"convnext.{i}."
        s += [(q + "dwconv.weight", (h, 1, kk), "w"), (q + "dwconv.bias", (h,), "b")]
        s += _adain(q + "norm", st, h)
        s += _linear(q + "pwconv1", g.conv_intermediate_dim, h)
        s += [(q + "grn.gamma", (1, 1, g.conv_intermediate_dim), "grn"), (q + "grn.beta", (1, 1, g.conv_intermediate_dim), "grn")]
        s += _linear(q + "pwconv2", h, g.conv_intermediate_dim)
    return s


def speech_predictor_spec(cfg) -> Spec:
    """SpeechPredictor (models/speech_predictor.py:14-83) minus the training-only posterior encoder."""
    hid = cfg.decoder.hidden_dim
    fh = hid // 4
    s: Spec = []
    s += text_encoder_spec("text_encoder.", cfg, cfg.inter_dim)
    s += text_style_encoder_spec("style_encoder.", cfg, cfg.inter_dim)
    s += decoder_spec("decoder.", cfg.inter_dim, cfg.style_dim, hid, cfg.decoder.residual_dim)
    s += [("prior_encoder.proj_mean.weight", (fh, hid), "w"), ("prior_encoder.proj_mean.bias", (fh,), "b")]
    s += [("prior_encoder.proj_logstd.weight", (fh, hid), "w_tiny"), ("prior_encoder.proj_logstd.bias", (fh,), "b")]
    s += flow_spec("flow.", fh, fh, 5, 4, 8, cfg.style_dim)
    s += _linear("post_flow", hid, fh)
    s += generator_spec("generator.", cfg)
    return s


def adaptive_generator_block_spec(p: str, channels: int, k: int, style: int) -> Spec:
    """AdaptiveGeneratorBlock (MRF + Snake; models/ada_norm.py:11-120).  Not executed by any
    runnable reference path; inventoried for the standalone block only."""
    s: Spec = []
    for grp in ("convs1", "convs2"):
        for i in range(3):
            s += _wn_conv_param(p + f"{grp}.{i}", channels, channels, k)
    for grp in ("adain1", "adain2"):
        for i in range(3):
            s += _adain(p + f"{grp}.{i}", style, channels)
    for grp in ("alpha1", "alpha2"):
        for i in range(3):
            s.append((p + f"{grp}.{i}", (1, channels, 1), "alpha"))
    return s


MODULE_SPECS = {
    "speech_predictor": speech_predictor_spec,
    "duration_predictor": duration_predictor_spec,
    "pitch_energy_predictor": pitch_energy_predictor_spec,
    "pe_text_encoder": lambda cfg: text_encoder_spec("", cfg, cfg.pitch_energy_predictor.inter_dim),
    "pe_text_style_encoder": lambda cfg: text_style_encoder_spec("", cfg, cfg.pitch_energy_predictor.inter_dim),
}


# CfmMelDecoder (models/cfm/cfm_mel_decoder.py:190-245; blocks: models/xut/transformer.py:9-51).  Dimensions are not in the model
# YAML (the class is constructed with keyword defaults), so they travel as a plain dict.
CFM_DEFAULT_DIMS = dict(feat_dim=80, asr_dim=768, spk_dim=1024, hidden_dim=256, emb_dim=256, depth=4, enc_blocks=1, dec_blocks=2,
                        prev_depth=1, post_depth=3, head_dim=64)


def xut_block_spec(p: str, dim: int, mlp: int, head_dim: int, cross: bool) -> Spec:
    """TransformerBlock with shared AdaLN: SelfAttention [+ CrossAttention] + SwiGLU + RMSNorm weights (registration order)."""
    heads = dim // head_dim
    s: Spec = [(p + ".attn.qkv.weight", (3 * dim, dim), "w_qk"), (p + ".attn.out.weight", (dim, dim), "w_attn_o"), (p + ".attn.out.bias", (dim,), "b"),
               (p + ".attn.rope.freqs", (heads, head_dim // 2, 1), "rope_f")]
    if cross:
        s += [(p + ".xattn.q.weight", (dim, dim), "w_qk"), (p + ".xattn.kv.weight", (2 * dim, dim), "w_qk"),
              (p + ".xattn.out.weight", (dim, dim), "w_attn_o"), (p + ".xattn.out.bias", (dim,), "b"),
              (p + ".xattn.rope.freqs", (heads, head_dim // 2, 1), "rope_f")]
    s += [(p + ".mlp.w12.weight", (2 * mlp, dim), "w"), (p + ".mlp.w12.bias", (2 * mlp,), "b"),
          (p + ".mlp.w3.weight", (dim, mlp), "w_small"), (p + ".mlp.w3.bias", (dim,), "b"),
          (p + ".attn_pre_norm.norm.weight", (dim,), "ln_g"), (p + ".mlp_pre_norm.norm.weight", (dim,), "ln_g")]
    if cross:
        s.append((p + ".xattn_pre_norm.norm.weight", (dim,), "ln_g"))
    return s


def cfm_mel_decoder_spec(dims=None) -> Spec:
    d = dict(CFM_DEFAULT_DIMS, **(dims or {}))
    dim, emb, feat, mlp, hd = d["hidden_dim"], d["emb_dim"], d["feat_dim"], 4 * d["hidden_dim"], d["head_dim"]
    s: Spec = [("time_emb.freqs", (1, dim // 2), "time_freqs")]
    s += _linear("time_emb.proj.0", dim, dim)
    s += _linear("asr_emb.1", 4 * emb, d["asr_dim"]) + _linear("asr_emb.3", emb, 4 * emb)
    s += _linear("spk_emb.0", 4 * emb, d["spk_dim"]) + _linear("spk_emb.2", emb, 4 * emb)
    s += [("m_source.1.merge.0.weight", (1, 1), "w")]
    s += _conv("prior_generator.1", feat, 3, 7)
    for i in range(d["depth"]):
        for j in range(d["enc_blocks"]):
            s += xut_block_spec(f"backbone.enc_blocks.{i}.{j}", dim, mlp, hd, False)
    for i in range(d["depth"]):
        for j in range(d["dec_blocks"]):
            s += xut_block_spec(f"backbone.dec_blocks.{i}.{j}", dim, mlp, hd, j == 0)
    s += _linear("in_proj", dim, feat + 2 * emb) + _linear("out_proj.0", feat, dim)
    for nm in ("shared_adaln_attn", "shared_adaln_xattn", "shared_adaln_ffw"):
        s += [(nm + ".0.weight", (dim,), "ln_g"), (nm + ".0.bias", (dim,), "ln_b")]
        s += _linear(nm + ".1", 4 * dim, dim)
        s += [(nm + ".3.weight", (3 * dim, 4 * dim), "w_small"), (nm + ".3.bias", (3 * dim,), "b")]  # zero-initialised in the reference (:262-277)
    for i in range(d["prev_depth"]):
        s += xut_block_spec(f"prev_tread_trns.blocks.{i}", dim, mlp, hd, False)
    for i in range(d["post_depth"]):
        s += xut_block_spec(f"post_tread_trns.blocks.{i}", dim, mlp, hd, False)
    return s


def module_spec(module: str, cfg) -> Spec:
    return MODULE_SPECS[module](cfg)


# ----------------------------------------------------------------------------
# counter-hash weights
# ----------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n values in [-1, 1), exactly representable in fp32 (24-bit), keyed by (name, seed, index)."""
    base = (fnv1a64(name) + (seed * 0x632BE59BD9B4E019)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(base)
    h = _splitmix64(idx)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (2.0 * u - 1.0).astype(np.float32)


def hash_normal(name: str, n: int, seed: int = 0) -> np.ndarray:
    """Approximately N(0,1) values (sum of 4 uniforms, variance-normalised); deterministic, fp32."""
    acc = np.zeros(n, np.float64)
    for j in range(4):
        acc += hash_uniform(f"{name}#n{j}", n, seed).astype(np.float64)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def synth_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    u = hash_uniform(name, n, seed).astype(np.float64)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    if kind in ("w", "wn_v"):
        v = u * np.sqrt(3.0 / fan_in)
    elif kind == "w_qk":
        v = u * 2.5 * np.sqrt(3.0 / fan_in)
    elif kind == "w_attn_o":
        v = u * 0.35 * np.sqrt(3.0 / fan_in)
    elif kind == "w_small":
        v = u * 0.5 * np.sqrt(3.0 / fan_in)
    elif kind == "w_tiny":
        v = u * 0.15 * np.sqrt(3.0 / fan_in)
    elif kind == "w_style":
        v = u * 0.6 * np.sqrt(3.0 / fan_in)
    elif kind == "b_style":
        v = u * 0.2
    elif kind == "w_dur":
        v = u * 6.0 * np.sqrt(3.0 / fan_in)
    elif kind == "w_f0":
        v = u * 120.0 * np.sqrt(3.0 / fan_in)
    elif kind == "b_f0":
        v = 110.0 + 10.0 * u
    elif kind == "b_logamp":
        v = 1.5 + 0.1 * u
    elif kind == "b":
        v = u * 0.1
    elif kind == "b_tiny":
        v = u * 0.02
    elif kind == "wn_g":
        v = 0.9 + 0.3 * u
    elif kind == "wn_g_half":
        v = 0.6 + 0.2 * u
    elif kind == "ln_g":
        v = 1.0 + 0.2 * u
    elif kind == "ln_b":
        v = 0.1 * u
    elif kind == "grn":
        v = 0.3 * u
    elif kind == "alpha":
        v = 1.0 + 0.3 * u
    elif kind == "emb":
        v = u * np.sqrt(3.0) * (shape[1] ** -0.5)
    elif kind == "rope_f":  # log-frequencies in [log pi, log 5 pi] (xut/axial_rope.py:110-116 initialises a linspace over that range)
        v = np.log(np.pi) + 0.5 * (u + 1.0) * np.log(5.0)
    elif kind == "time_freqs":  # the TimestepEmbedding buffer (xut/time_emb.py:14-22), formed in fp32 like torch does
        half = shape[-1]
        v = np.exp((np.float32(-np.log(10000.0)) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)).astype(np.float32)
    else:  # pragma: no cover
        raise ValueError(kind)
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(spec: Spec, seed: int = 0, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """prefix keys the hash (e.g. the module name) so equal-named tensors of different modules differ."""
    return OrderedDict((name, synth_tensor(prefix + name, shape, kind, seed)) for name, shape, kind in spec)


def spec_shapes(spec: Spec) -> Dict[str, Tuple[int, ...]]:
    return {n: s for n, s, _ in spec}


def count_params(spec: Iterable) -> int:
    return int(sum(int(np.prod(s)) for _, s, _ in spec))
