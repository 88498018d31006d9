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


def confident_points(npy_file, thres=20.0):
    """What the downstream consumers do with a `coord_<i>.npy` record
    (vis/vis_scene_coordinate_map.py:10-26, and the PnP step of README.md:132-138): load the
    float32 [h,w,4] map, split scene coordinates (channels 0-2) from the confidence
    (channel 3 = 1/sigma) and keep the points whose confidence exceeds `thres`.
    Returns (points [n,3], pixel indices [n,2] as (row, col))."""
    import numpy as np
    rec = np.load(npy_file)
    if rec.ndim != 3 or rec.shape[2] != 4:
        raise ValueError('%s: expected a [h,w,4] scene-coordinate map, got %s' % (npy_file, rec.shape))
    keep = rec[:, :, 3] > thres
    return rec[:, :, 0:3][keep], np.argwhere(keep)
