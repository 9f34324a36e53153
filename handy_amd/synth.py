"""Seeded synthetic 16 kHz mono f32 audio: three sinusoids (80-4000 Hz, amplitude 0.05 each) plus
white noise (sigma 0.02) under a slow 2-5 Hz envelope, clipped to [-1, 1]. The envelope gives
the log-mel real dynamic range, so the max-8 dB clamp of the frontend is exercised."""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000


def synth_pcm(seed: int, n_samples: int = 30 * SAMPLE_RATE) -> np.ndarray:
    rng = np.random.default_rng(1000003 * (seed + 1))
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    x = np.zeros(n_samples, dtype=np.float64)
    for _ in range(3):
        f = rng.uniform(80.0, 4000.0)
        ph = rng.uniform(0, 2 * np.pi)
        x += 0.05 * np.sin(2 * np.pi * f * t + ph)
    x += 0.02 * rng.standard_normal(n_samples)
    fe = rng.uniform(2.0, 5.0)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * fe * t + rng.uniform(0, 2 * np.pi))
    return np.clip(x * env, -1.0, 1.0).astype(np.float32)


def synth_batch(first_seed: int, count: int, n_samples: int = 30 * SAMPLE_RATE) -> np.ndarray:
    return np.stack([synth_pcm(first_seed + i, n_samples) for i in range(count)])
