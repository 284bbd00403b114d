"""Synthetic inputs of the benchmark configurations: splitmix64 with state0 = seed;
pixel i takes the (i+1)-th output r: R = r & 255, G = (r >> 8) & 255, B = (r >> 16) & 255, A = 255.
The generator is random access (state_i = seed + (i+1) * gamma), so a row band can be produced
directly on the GPU that owns it."""
import numpy as np

GAMMA = 0x9E3779B97F4A7C15
M1 = 0xBF58476D1CE4E5B9
M2 = 0x94D049BB133111EB

SEED_CFG2 = 0x5EED0002
SEED_CFG3 = 0x5EED0003
SEED_CFG4 = 0x5EED0400
SEED_CFG5 = 0x5EED0005


def uniform_rgba_numpy(seed, n, first=0):
    """(n, 4) uint8 pixels first .. first+n-1 of stream `seed` (host)."""
    with np.errstate(over="ignore"):
        i = np.arange(first + 1, first + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(M2)
        z = z ^ (z >> np.uint64(31))
    out = np.empty((n, 4), np.uint8)
    out[:, 0] = (z & np.uint64(255)).astype(np.uint8)
    out[:, 1] = ((z >> np.uint64(8)) & np.uint64(255)).astype(np.uint8)
    out[:, 2] = ((z >> np.uint64(16)) & np.uint64(255)).astype(np.uint8)
    out[:, 3] = 255
    return out


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def uniform_rgba_torch(seed, n, first=0, device="cuda", chunk=1 << 24):
    """same stream generated on `device` (two's-complement int64 arithmetic, logical shifts via masks)"""
    import torch
    out = torch.empty((n, 4), dtype=torch.uint8, device=device)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        i = torch.arange(first + c0 + 1, first + c0 + m + 1, dtype=torch.int64, device=device)
        z = i * _s64(GAMMA) + _s64(seed)
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _s64(M1)
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _s64(M2)
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        o = out[c0:c0 + m]
        o[:, 0] = (z & 255).to(torch.uint8)
        o[:, 1] = ((z >> 8) & 255).to(torch.uint8)
        o[:, 2] = ((z >> 16) & 255).to(torch.uint8)
        o[:, 3] = 255
    return out


def uniform_rgba_at(seed, indices):
    """(len(indices), 4) uint8: the pixels at the given linear indices of stream `seed` (random access)"""
    with np.errstate(over="ignore"):
        i = np.asarray(indices, dtype=np.uint64) + np.uint64(1)
        z = np.uint64(seed) + i * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(M2)
        z = z ^ (z >> np.uint64(31))
    out = np.empty((len(i), 4), np.uint8)
    out[:, 0] = (z & np.uint64(255)).astype(np.uint8)
    out[:, 1] = ((z >> np.uint64(8)) & np.uint64(255)).astype(np.uint8)
    out[:, 2] = ((z >> np.uint64(16)) & np.uint64(255)).astype(np.uint8)
    out[:, 3] = 255
    return out
