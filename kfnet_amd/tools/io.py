"""Host-side file helpers with the semantics of the reference's tools/io.py."""
import glob
import os
import re


def read_lines(filepath):
    """tools/io.py:208-212: one stripped line per entry."""
    with open(filepath) as fin:
        lines = fin.readlines()
    return [line.strip() for line in lines]


def get_snapshot(folder):
    """tools/io.py:185-196 picks the newest `model.ckpt-N` (and chdir()s -- not reproduced).
    Here a model folder holds `kfnet_weights.npz` or `kfnet_weights-<step>.npz` keyed by TF
    variable names (kfnet_amd/weights.py); the highest step wins.  Returns (path, step)."""
    cands = glob.glob(os.path.join(folder, 'kfnet_weights*.npz'))
    if not cands:
        return None, 0

    def step(p):
        nums = re.findall(r'\d+', os.path.basename(p))
        return int(nums[-1]) if nums else 0
    best = max(cands, key=step)
    return best, step(best)
