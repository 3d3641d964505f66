"""Phoneme string → token ids, and the int16 wav writer: the host-side ends of the inference entry point.

``TextCleaner`` mirrors ``lib/text_utils.py:8-41``: the symbol table is ``pad + punctuation + letters + letters_ipa``
from the model config (``model.yml:81-85``), a symbol listed twice keeps its LAST position, and characters outside the
table are dropped with an error log.  ``frame_tokens`` adds the pad token either side as the reference's ONNX driver
does (``train/test_onnx.py:49-53``); ``to_int16`` / ``write_wav`` are its output stage (``train/test_onnx.py:79-90``).
"""
from __future__ import annotations

import logging
import struct
from typing import Iterable, List, Sequence

import numpy as np

logger = logging.getLogger(__name__)


class TextCleaner:
    def __init__(self, symbols):
        table = [symbols["pad"]] + list(symbols["punctuation"]) + list(symbols["letters"]) + list(symbols["letters_ipa"])
        self.word_index_dictionary = {ch: i for i, ch in enumerate(table)}
        self.size = len(table)

    def __call__(self, text: str) -> List[int]:
        ids = []
        for ch in text:
            i = self.word_index_dictionary.get(ch)
            if i is None:
                logger.error("Meld " + ch + ": " + text)
            else:
                ids.append(i)
        return ids


def frame_tokens(ids: Sequence[int], pad: int = 0) -> List[int]:
    """[pad] + ids + [pad]: the token row the reference feeds to both predictors."""
    return [pad] + [int(v) for v in ids] + [pad]


def to_int16(wave) -> np.ndarray:
    """``np.multiply(x, 32768).astype(np.int16)`` of the reference; |x| < 1 after the tanh, so no sample wraps."""
    a = wave.detach().cpu().numpy() if hasattr(wave, "detach") else np.asarray(wave)
    return np.multiply(a, 32768).astype(np.int16)


def write_wav(path: str, samples: Iterable, sample_rate: int = 24000) -> None:
    """Mono 16-bit PCM RIFF file, the format ``scipy.io.wavfile.write`` produces for an int16 vector."""
    pcm = np.ascontiguousarray(samples)
    if pcm.dtype != np.int16:
        pcm = to_int16(pcm)
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)
