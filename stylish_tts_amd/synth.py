"""Seeded synthetic inputs (hash RNG shared by fixtures, tests and bench).

Shapes follow SURVEY.md §8(d): LJSpeech-shaped utterances at 80 mel frames/s
(hop 300 @ 24 kHz, ``train/config/model.yml``), token ids in [1, tokens) with the
pad id 0 at both ends (``train/dataloader.py:178-180``), pitch as piecewise-smooth
80–300 Hz contours with unvoiced (0 Hz) spans.
"""
from __future__ import annotations

import numpy as np

from .params import hash_normal, hash_uniform


def uniform(name: str, shape, seed: int = 0) -> np.ndarray:
    """U[0,1) fp32."""
    n = int(np.prod(shape))
    return ((hash_uniform(name, n, seed).astype(np.float64) + 1.0) * 0.5).astype(np.float32).reshape(shape)


def normal(name: str, shape, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return hash_normal(name, n, seed).reshape(shape)


def tokens(name: str, batch: int, length: int, vocab: int, seed: int = 0) -> np.ndarray:
    u = uniform(name, (batch, length), seed)
    ids = 1 + np.floor(u * (vocab - 1)).astype(np.int64)
    ids = np.clip(ids, 1, vocab - 1)
    ids[:, 0] = 0
    ids[:, -1] = 0
    return ids


def pitch_curve(name: str, batch: int, frames: int, seed: int = 0, unvoiced: float = 0.3) -> np.ndarray:
    """[batch, frames] fp32 Hz; 0 on unvoiced spans."""
    out = np.zeros((batch, frames), np.float32)
    for b in range(batch):
        u = uniform(f"{name}.{b}", (frames * 3 + 8,), seed)
        pos, j = 0, 0
        while pos < frames:
            seg = 4 + int(u[j] * 9)
            voiced = u[j + 1] >= unvoiced
            base = 80.0 + 220.0 * u[j + 2]
            j += 3
            end = min(frames, pos + seg)
            if voiced:
                tt = np.arange(end - pos, dtype=np.float32)
                out[b, pos:end] = base * (1.0 + 0.08 * np.sin(0.35 * tt + 6.0 * u[j]))
            pos = end
    return out


def durations_for(name: str, n_tokens: int, total_frames: int, seed: int = 0) -> np.ndarray:
    """Split total_frames over n_tokens, every duration ≥ 1 (deterministic)."""
    assert total_frames >= n_tokens
    w = uniform(name, (n_tokens,), seed).astype(np.float64) + 0.05
    extra = total_frames - n_tokens
    raw = w / w.sum() * extra
    d = np.floor(raw).astype(np.int64)
    rem = extra - int(d.sum())
    order = np.argsort(-(raw - d), kind="stable")
    d[order[:rem]] += 1
    return d + 1


def alignment_from_durations(d: np.ndarray) -> np.ndarray:
    """0/1 [P, T] matrix (same object the reference's DurationProcessor.duration_to_alignment builds,
    train/utils.py:476-489)."""
    P, T = len(d), int(d.sum())
    a = np.zeros((P, T), np.float32)
    idx = np.repeat(np.arange(P), d)
    a[idx, np.arange(T)] = 1.0
    return a


def path_noise(case: str, batch: int, t4: int, flow_dim: int = 128, hop4: int = 75, seed: int = 0):
    """The three explicit noise inputs of the frame-rate path for fixture/test `case`:
    prior_noise [B, flow_dim, T4] ~ N(0,1)  (replaces randn_like, models/flow.py:314),
    src_noise   [B, 1, hop4*T4]  ~ N(0,1)   (replaces randn before the 0.01 scale, models/generator.py:272),
    init_phase  [1, 1]           ~ U[0,1)   (replaces rand, models/generator.py:306; one scalar per call)."""
    return dict(
        prior_noise=normal(f"noise.{case}.prior", (batch, flow_dim, t4), seed),
        src_noise=normal(f"noise.{case}.src", (batch, 1, hop4 * t4), seed),
        init_phase=uniform(f"noise.{case}.phase", (1, 1), seed),
    )
