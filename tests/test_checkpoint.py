"""Checkpoint reader: an accelerate save directory written by accelerate's own writer -> the five inference state dicts."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from stylish_tts_amd import checkpoint, modules, params
from stylish_tts_amd.config import load_model_config


def _state_dicts(cfg, seed=3):
    return {m: {k: torch.from_numpy(v) for k, v in params.synth_state_dict(params.module_spec(m, cfg), seed).items()} for m in checkpoint.INFERENCE_MODULES}


@pytest.mark.parametrize("safe", [False, True])
def test_reads_accelerate_save_state_layout(tmp_path, safe):
    accelerate = pytest.importorskip("accelerate")
    from accelerate.checkpointing import save_accelerator_state

    cfg = load_model_config()
    sds = _state_dicts(cfg)
    # model_states in build_model() order (models/models.py:79-101); the training-only models get a stand-in tensor,
    # and one inference module is saved as DistributedDataParallel would name its keys
    states = []
    for name in checkpoint.MODEL_ORDER[:9]:
        if name in sds:
            sd = sds[name]
            states.append({"module." + k: v for k, v in sd.items()} if name == "pe_text_encoder" else dict(sd))
        else:
            states.append({"unused.weight": torch.zeros(2)})
    save_accelerator_state(str(tmp_path), states, [], [], [], 0, 0, safe_serialization=safe)
    files = checkpoint.checkpoint_files(str(tmp_path))
    assert os.path.basename(files["speech_predictor"]) == ("model_3.safetensors" if safe else "pytorch_model_3.bin")
    assert os.path.basename(files["duration_predictor"]) == ("model_1.safetensors" if safe else "pytorch_model_1.bin")
    got = checkpoint.load_accelerate_checkpoint(str(tmp_path))
    assert set(got) == set(checkpoint.INFERENCE_MODULES)
    for m in sds:
        assert set(got[m]) == set(sds[m])  # (safetensors stores keys sorted)
        for k in sds[m]:
            assert torch.equal(got[m][k], sds[m][k])
    mods = modules.build_inference_modules(cfg)
    checkpoint.load_into(mods, got)
    k = "decoder.encode.conv1.parametrizations.weight.original1"
    assert torch.equal(mods["speech_predictor"].state_dict()[k], sds["speech_predictor"][k])


def test_missing_module_is_reported(tmp_path):
    torch.save({"a": torch.zeros(1)}, tmp_path / "pytorch_model.bin")
    with pytest.raises(FileNotFoundError, match="no weight file for .speech_predictor."):
        checkpoint.load_accelerate_checkpoint(str(tmp_path))


def test_packed_file_roundtrip(tmp_path):
    cfg = load_model_config()
    sds = _state_dicts(cfg, seed=5)
    p = str(tmp_path / "voice.safetensors")
    checkpoint.save_packed(p, sds)
    back = checkpoint.load_packed(p)
    assert set(back) == set(sds)
    for m in sds:
        assert set(back[m]) == set(sds[m])
        for k in sds[m]:
            assert np.array_equal(back[m][k].numpy(), sds[m][k].numpy())
