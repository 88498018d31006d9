"""CPU tests of the eval driver's host helpers (file formats either side of the path)."""
import os

import numpy as np
import pytest

from kfnet_amd.KFNet import eval as kf_eval
from kfnet_amd.tools.io import get_snapshot, read_lines
from kfnet_amd.weights import from_network_load_dict, load_npz, save_npz, synthetic_weights


def test_get_transform_inverts_like_train_py(tmp_path):
    M = np.eye(4, dtype=np.float32)
    M[:3, 3] = [1, 2, 3]
    M[0, 0] = 2
    p = tmp_path / 'transform.txt'
    np.savetxt(p, M)
    T = kf_eval.get_transform(str(p))            # KFNet/train.py:49-58: inv(loadtxt)
    assert np.allclose(T @ M, np.eye(4), atol=1e-6)
    assert np.array_equal(kf_eval.get_transform(None), np.eye(4, dtype=np.float32))


def test_read_lines_strips(tmp_path):
    p = tmp_path / 'l.txt'
    p.write_text(' a/b.png \nc.png\n\n')
    assert read_lines(str(p)) == ['a/b.png', 'c.png', '']      # tools/io.py:208-212 semantics


def test_snapshot_picks_highest_step(tmp_path):
    assert get_snapshot(str(tmp_path)) == (None, 0)
    for n in ('kfnet_weights-100.npz', 'kfnet_weights-2500.npz', 'kfnet_weights-900.npz'):
        (tmp_path / n).write_bytes(b'')
    path, step = get_snapshot(str(tmp_path))
    assert step == 2500 and path.endswith('kfnet_weights-2500.npz')


def test_weight_container_roundtrip(tmp_path):
    W = {k: v for k, v in synthetic_weights(5).items() if k.startswith('Temporal/fc')}
    p = str(tmp_path / 'w.npz')
    save_npz(p, W)
    W2 = load_npz(p)
    assert set(W2) == set(W) and all(np.array_equal(W[k], W2[k]) for k in W)
    flat = from_network_load_dict({'fc1': {'kernel': W['Temporal/fc1/kernel']}}, 'Temporal')
    assert np.array_equal(flat['Temporal/fc1/kernel'], W['Temporal/fc1/kernel'])


def test_load_images_checks_size(tmp_path):
    from PIL import Image
    a = np.random.default_rng(0).integers(0, 256, size=(6, 8, 3), dtype=np.uint8)
    p = str(tmp_path / 'a.png')
    Image.fromarray(a).save(p)
    assert np.array_equal(kf_eval.load_images([p], (6, 8))[0], a)
    with pytest.raises(ValueError):
        kf_eval.load_images([p], (480, 640))


def test_invalid_scene_is_rejected():
    assert kf_eval.main(['--scene', 'livingroom']) == 1      # KFNet/train.py:142-144
