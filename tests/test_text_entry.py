"""Tokeniser and wav writer against vectors produced by the reference (tests/golden/gen_golden.py:text_golden)."""
import json
import os

import numpy as np

from stylish_tts_amd.config import load_model_config
from stylish_tts_amd.text import TextCleaner, frame_tokens, to_int16, write_wav

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_tokens.json"), encoding="utf-8"))


def test_text_cleaner_matches_reference_ids():
    cfg = load_model_config()
    tc = TextCleaner(cfg.symbol)
    assert len(tc.word_index_dictionary) == G["table_size"]
    assert tc.size == cfg.text_encoder.tokens  # 178 symbols incl. the duplicated apostrophe (model.yml:74,81-85)
    for text, ids in zip(G["texts"], G["ids"]):
        assert tc(text) == ids


def test_frame_tokens_pads_both_ends():
    assert frame_tokens([5, 6]) == [0, 5, 6, 0]
    assert frame_tokens([]) == [0, 0]


def test_int16_and_wav_bytes_match_scipy(tmp_path):
    pcm = to_int16(np.asarray(G["wave"], dtype=np.float32))
    assert pcm.dtype == np.int16 and pcm.tolist() == G["pcm"]
    p = tmp_path / "a.wav"
    write_wav(str(p), pcm, 24000)
    assert p.read_bytes().hex() == G["wav_hex"]
