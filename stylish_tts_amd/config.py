"""Model-shape contract: the same ``model.yml`` keys the reference validates.

The reference shapes its whole inference graph from ``ModelConfig``
(``/root/reference/src/stylish_tts/lib/config_loader.py:369-414``) loaded by
``load_model_config_yaml`` (``config_loader.py:442-455``) from
``train/config/model.yml``.  This module parses that YAML into plain
attribute-access records; only the keys the inference hot path reads are
required, every other key of the reference schema is carried through
untouched so a reference ``model.yml`` loads unchanged.
"""
from __future__ import annotations

import io
import json
from typing import Any, Mapping

import yaml


class Record(dict):
    """dict with attribute access (nested)."""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover - error path
            raise AttributeError(k) from e

    def __setattr__(self, k: str, v: Any) -> None:
        self[k] = v


def _wrap(x: Any) -> Any:
    if isinstance(x, Mapping):
        return Record({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


# keys the hot path reads (subset of ModelConfig, config_loader.py:369-414)
_REQUIRED = {
    "": ["n_fft", "win_length", "hop_length", "style_dim", "inter_dim", "sample_rate"],
    "decoder": ["hidden_dim", "residual_dim"],
    "generator": ["input_dim", "hidden_dim", "conv_intermediate_dim", "io_conv_kernel_size"],
    "text_encoder": ["tokens", "hidden_dim", "filter_channels", "heads", "layers", "kernel_size"],
    "style_encoder": ["layers"],
    "duration_predictor": ["n_layer", "duration_classes", "max_duration"],
    "pitch_energy_predictor": ["inter_dim"],
}

# The default model.yml of the reference (train/config/model.yml), restated as
# data: only the hot-path keys.  Used when no YAML is given (bench, tests).
DEFAULT_MODEL = {
    "multispeaker": False,
    "n_mels": 80,
    "sample_rate": 24000,
    "n_fft": 2048,
    "win_length": 1200,
    "hop_length": 300,
    "style_dim": 64,
    "inter_dim": 128,
    "decoder": {"hidden_dim": 512, "residual_dim": 64},
    "generator": {
        "type": "freegan",
        "input_dim": 512,
        "hidden_dim": 512,
        "conv_intermediate_dim": 1536,
        "io_conv_kernel_size": 7,
        "conformer_layers": 5,
        "conv_layers": 5,
    },
    "text_encoder": {
        "tokens": 178,
        "hidden_dim": 128,
        "filter_channels": 512,
        "heads": 8,
        "layers": 8,
        "kernel_size": 3,
        "dropout": 0.2,
    },
    "style_encoder": {"layers": 2},
    "duration_predictor": {
        "n_layer": 4,
        "duration_classes": 16,
        "max_duration": 50,
        "dropout": 0.2,
        "last_dropout": 0.5,
    },
    "pitch_energy_predictor": {"inter_dim": 256, "dropout": 0.2},
    # feature extractors whose OUTPUT WIDTHS size the flow-matching mel decoder (model.yml:68-74; the extractors themselves are given tensors)
    "hubert": {"hidden_dim": 768},
    "speaker_embedder": {"hidden_dim": 10240},
    # text symbols (model.yml:81-85); index = position in pad + punctuation + letters + letters_ipa
    "symbol": {
        "pad": '$',
        "punctuation": ';:,.!?¡¿—…"()“” ',
        "letters": 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz',
        "letters_ipa": "ɑɐɒæɓʙβɔɕçɗɖðʤəɘɚɛɜɝɞɟʄɡɠɢʛɦɧħɥʜɨɪʝɭɬɫɮʟɱɯɰŋɳɲɴøɵɸθœɶʘɹɺɾɻʀʁɽʂʃʈʧʉʊʋⱱʌɣɤʍχʎʏʑʐʒʔʡʕʢǀǁᵊǃˈˌːˑʼʴʰʱʲʷˠˤ˞↓↑→↗↘'̩'ᵻ",
    },
}


def validate(cfg: Record) -> Record:
    for section, keys in _REQUIRED.items():
        node = cfg if section == "" else cfg.get(section)
        if node is None:
            raise ValueError(f"model config: missing section '{section}'")
        for k in keys:
            if k not in node:
                raise ValueError(f"model config: missing key '{section}.{k}'".replace("'.", "'"))
    g = cfg.generator
    if g.get("type", "freegan") != "freegan":
        # GeneratorConfig.type is Literal["freegan"] (config_loader.py:203); the
        # ringformer generator cannot be instantiated in the reference either.
        raise ValueError("only generator.type == 'freegan' is executable in the reference")
    if cfg.hop_length % 4 != 0:
        raise ValueError("hop_length must be divisible by 4 (vocoder runs at hop/4)")
    return cfg


def load_model_config(src: Any = None) -> Record:
    """Accepts None (reference default), a path, an open file, a YAML/JSON string or a dict."""
    if src is None:
        data = DEFAULT_MODEL
    elif isinstance(src, Mapping):
        data = src
    elif isinstance(src, (io.IOBase,)) or hasattr(src, "read"):
        data = yaml.safe_load(src)
    elif isinstance(src, str) and ("\n" in src or src.lstrip().startswith("{")):
        data = yaml.safe_load(src)
    else:
        with open(src, "r", encoding="utf-8") as f:
            data = yaml.safe_load(f)
    return validate(_wrap(data))


def to_json(cfg: Record) -> str:
    return json.dumps(cfg, sort_keys=True)
