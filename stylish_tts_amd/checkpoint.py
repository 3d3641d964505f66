"""Reader for the reference's training checkpoints, and a packed single-file weight format.

The reference saves with ``accelerator.save_state(dir, safe_serialization=False)`` (``train/train.py:433-449``): one
``pytorch_model[_<i>].bin`` per prepared model, numbered in the order ``build_model`` lists them
(``train/models/models.py:79-101``; prepared in that order at ``train/train.py:189-193`` / ``train/cli.py:300-303``).
With ``safe_serialization=True`` the files are ``model[_<i>].safetensors`` instead.  Only the five modules of the
inference composition are read; pickles are opened with ``weights_only=True`` (tensors only, no code execution).

``save_packed`` / ``load_packed`` keep those five state dicts in ONE safetensors file with ``<module>.<key>`` names —
the same names ``stts_load_weight`` takes (include/stylish_hip.h) — so a serving process needs neither pickle nor the
training tree.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Mapping

import torch

# position of every model in the reference's build_model() Munch (models/models.py:79-101)
MODEL_ORDER = (
    "text_aligner", "duration_predictor", "pitch_energy_predictor", "speech_predictor", "mrd", "mpd", "pe_text_encoder",
    "pe_text_style_encoder", "pe_mel_style_encoder", "hubert_encoder", "cfm_mel_decoder", "cfm_pitch_predictor",
    "hubert_speech_predictor", "hubert_pitch_energy_predictor",
)
INFERENCE_MODULES = ("speech_predictor", "duration_predictor", "pitch_energy_predictor", "pe_text_encoder", "pe_text_style_encoder")


def _indexed(stem: str, ext: str, i: int) -> str:
    return f"{stem}{ext}" if i == 0 else f"{stem}_{i}{ext}"


def checkpoint_files(checkpoint_dir: str, modules: Iterable[str] = INFERENCE_MODULES) -> Dict[str, str]:
    """module name -> weight file inside an accelerate save directory."""
    out = {}
    for name in modules:
        i = MODEL_ORDER.index(name)
        for stem, ext in (("pytorch_model", ".bin"), ("model", ".safetensors")):
            p = os.path.join(checkpoint_dir, _indexed(stem, ext, i))
            if os.path.exists(p):
                out[name] = p
                break
        else:
            raise FileNotFoundError(f"{checkpoint_dir}: no weight file for '{name}' (model index {i})")
    return out


def _strip(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    # DistributedDataParallel-wrapped models save their keys under "module."
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_accelerate_checkpoint(checkpoint_dir: str, modules: Iterable[str] = INFERENCE_MODULES) -> Dict[str, Dict[str, torch.Tensor]]:
    """{module: state_dict} for the inference modules of a reference checkpoint directory."""
    out = {}
    for name, path in checkpoint_files(checkpoint_dir, modules).items():
        if path.endswith(".safetensors"):
            from safetensors.torch import load_file

            sd = load_file(path, device="cpu")
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        out[name] = _strip(sd)
    return out


def save_packed(path: str, state_dicts: Mapping[str, Mapping[str, torch.Tensor]]) -> None:
    from safetensors.torch import save_file

    flat = {}
    for mod, sd in state_dicts.items():
        for k, v in sd.items():
            flat[f"{mod}.{k}"] = v.detach().to(torch.float32).cpu().contiguous()
    save_file(flat, path, metadata={"format": "stylish_tts_amd.packed.v1"})


def load_packed(path: str) -> Dict[str, Dict[str, torch.Tensor]]:
    from safetensors.torch import load_file

    out: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in load_file(path, device="cpu").items():
        mod, key = k.split(".", 1)
        out.setdefault(mod, {})[key] = v
    return out


def load_into(modules: Mapping[str, torch.nn.Module], state_dicts: Mapping[str, Mapping[str, torch.Tensor]], strict: bool = True) -> None:
    """``load_state_dict`` every module of ``build_inference_modules`` from a checkpoint / packed file."""
    for name, mod in modules.items():
        if name not in state_dicts:
            raise KeyError(f"checkpoint has no module '{name}'")
        mod.load_state_dict(state_dicts[name], strict=strict)
