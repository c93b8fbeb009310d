"""Deterministic synthetic audio for tests (int16-scale mono, as Kaldi waveforms are)."""
import numpy as np


def utterance(n, seed=0, sr=16000.0, scale=1.0, dtype=np.float32, flat=False):
    """Near-Gaussian noise at about -30 dBFS plus three drifting tones; values within +-32767*scale.
    flat=True: noise-dominated (tones below the noise), so no spectral bin sits at the float32 FFT noise floor."""
    rng = np.random.default_rng(0x5EED2026 ^ seed)
    t = np.arange(n, dtype=np.float64) / sr
    x = (3000.0 if flat else 1000.0) * rng.standard_normal(n)
    for k in range(3):
        f0 = rng.uniform(80.0, 0.45 * sr)
        a = rng.uniform(300.0, 1500.0) if flat else rng.uniform(500.0, 6000.0)
        x += a * np.sin(2 * np.pi * (f0 * t + 20.0 * k * t * t) + rng.uniform(0, 2 * np.pi))
    x = np.clip(np.round(x), -32767, 32767) * scale
    return x.astype(dtype)
