"""Synthetic inputs for benchmarks / parity tests.

No 7-Scenes data or trained weights are reachable from this environment, so sequences
are a fixed random texture rolled by (t*dx, t*dy) pixels plus +-4 noise -- the rolled
texture gives OFlowNet's cost volume a non-trivial optimum so the soft-argmax flow is
not uniform.
"""
import numpy as np


def synthetic_sequence(T, H=480, W=640, seed=1, dx=3, dy=2, noise=4, start=0):
    """Frames [start, start+T) of the seeded sequence (frame t depends only on (seed, t),
    so ranks can generate their own chunk)."""
    rng = np.random.default_rng(seed)
    # low-frequency texture: upsampled coarse noise + fine noise, 0..255
    coarse = rng.integers(0, 256, size=(H // 8 + 2, W // 8 + 2, 3)).astype(np.float32)
    tex = np.kron(coarse, np.ones((8, 8, 1), dtype=np.float32))[:H, :W]
    tex = 0.75 * tex + 0.25 * rng.integers(0, 256, size=(H, W, 3)).astype(np.float32)
    frames = np.empty((T, H, W, 3), dtype=np.uint8)
    for k in range(T):
        t = start + k
        f = np.roll(tex, shift=(t * dy, t * dx), axis=(0, 1))
        f = f + np.random.default_rng([seed, t]).integers(-noise, noise + 1, size=f.shape)
        frames[k] = np.clip(f, 0, 255).astype(np.uint8)
    return frames


def synthetic_transform(seed=7):
    """Fixed seeded rigid 4x4 (what transform.txt would hold, KFNet/train.py:49-58)."""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(3, 3))
    Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    M = np.eye(4, dtype=np.float32)
    M[:3, :3] = Q.astype(np.float32)
    M[:3, 3] = rng.uniform(-1, 1, size=3).astype(np.float32)
    return M
