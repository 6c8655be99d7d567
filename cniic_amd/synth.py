"""Deterministic synthetic images, numpy restatement of the generator in
csrc/k_misc.hip (k_synth).  Integer-only, so CPU and GPU produce identical bytes.

  U  "uniform":    byte k = byte (k % 8) of splitmix64 output number k // 8 (seeded stream)
  P  "photo-like": per channel, integer bilinear interpolation of a hashed 64-px lattice, +-8 noise
"""
import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
SEED0 = 0x636E696963


def _mix(z):
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(w, h, seed=SEED0):
    with np.errstate(over="ignore"):
        k = np.arange(w * h * 3, dtype=np.uint64)
        word = _mix(np.uint64(seed) + GAMMA * (k // np.uint64(8) + np.uint64(1)))
        b = (word >> (np.uint64(8) * (k % np.uint64(8)))) & np.uint64(255)
    return b.astype(np.uint8).reshape(h, w, 3)


def _lattice(seed, i, j, ch):
    with np.errstate(over="ignore"):
        ident = (j.astype(np.uint64) << np.uint64(32)) | (i.astype(np.uint64) << np.uint64(2)) | np.uint64(ch)
        return (_mix(np.uint64(seed) + GAMMA * (ident + np.uint64(1))) & np.uint64(255)).astype(np.int64)


def photo(w, h, seed=SEED0):
    y, x = np.mgrid[0:h, 0:w]
    cx, fx, cy, fy = x >> 6, x & 63, y >> 6, y & 63
    idx = (y.astype(np.uint64) * np.uint64(w) + x.astype(np.uint64))
    seed2 = np.uint64(seed) ^ np.uint64(0xD1B54A32D192ED03)
    out = np.empty((h, w, 3), np.uint8)
    for ch in range(3):
        a, b = _lattice(seed, cx, cy, ch), _lattice(seed, cx + 1, cy, ch)
        c, d = _lattice(seed, cx, cy + 1, ch), _lattice(seed, cx + 1, cy + 1, ch)
        v = ((a * (64 - fx) + b * fx) * (64 - fy) + (c * (64 - fx) + d * fx) * fy) >> 12
        with np.errstate(over="ignore"):
            noise = (_mix(seed2 + GAMMA * (idx * np.uint64(4) + np.uint64(ch + 1))) & np.uint64(15)).astype(np.int64) - 8
        out[..., ch] = np.clip(v + noise, 0, 255).astype(np.uint8)
    return out
