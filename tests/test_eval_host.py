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


# ---- kfnet_amd/pipeline.py: the decode/staging side of the image stream (no GPU needed) ----
def _png_sequence(tmp_path, T, size=(6, 8)):
    from PIL import Image
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(T,) + size + (3,), dtype=np.uint8)
    paths = []
    for i in range(T):
        p = str(tmp_path / ('f%03d.png' % i))
        Image.fromarray(frames[i]).save(p)
        paths.append(p)
    return frames, paths


@pytest.mark.parametrize('workers', [1, 4])
def test_chunk_loader_orders_and_ragged_tail(tmp_path, workers):
    from kfnet_amd.pipeline import ChunkLoader
    frames, paths = _png_sequence(tmp_path, 11)
    for source in (paths, frames):
        loader = ChunkLoader(source, (6, 8), chunk=4, workers=workers, pinned=False)
        assert len(loader) == 3
        got = [(lo, host.numpy().copy()) for lo, host in loader]
        assert [lo for lo, _ in got] == [0, 4, 8]
        assert [a.shape[0] for _, a in got] == [4, 4, 3]
        assert np.array_equal(np.concatenate([a for _, a in got]), frames)


def test_chunk_loader_buffer_lifetime(tmp_path):
    """A chunk stays intact while depth-1 further chunks are requested (the upload of chunk k
    may still be in flight when chunk k+1 is being filled)."""
    from kfnet_amd.pipeline import ChunkLoader
    frames, _ = _png_sequence(tmp_path, 12)
    held = []
    for lo, host in ChunkLoader(frames, (6, 8), chunk=2, depth=3, pinned=False):
        held.append((lo, host))
        for plo, ph in held[-2:]:
            assert np.array_equal(ph.numpy(), frames[plo:plo + 2])


def test_chunk_loader_propagates_decode_errors_and_early_exit(tmp_path):
    from kfnet_amd.pipeline import ChunkLoader
    frames, paths = _png_sequence(tmp_path, 6)
    with pytest.raises(ValueError):
        list(ChunkLoader(paths, (480, 640), chunk=4, pinned=False))      # wrong size, raised in the consumer
    with pytest.raises(ValueError):
        ChunkLoader(frames.astype(np.float32), (6, 8), chunk=4, pinned=False)
    loader = ChunkLoader(paths, (6, 8), chunk=1, pinned=False)
    for lo, host in loader:          # consumer walks away: the producer thread must not hang
        break
    loader.thread.join(timeout=10)
    assert not loader.thread.is_alive()
    assert list(ChunkLoader([], (6, 8), chunk=4, pinned=False)) == []     # empty sequence


def test_confident_points_reads_records_like_the_vis_tools(tmp_path):
    from kfnet_amd.tools.io import confident_points
    rec = np.zeros((3, 4, 4), np.float32)
    rec[..., :3] = np.arange(36, dtype=np.float32).reshape(3, 4, 3)
    rec[..., 3] = 1.0
    rec[1, 2, 3] = 25.0
    rec[2, 0, 3] = 20.0            # not strictly above the threshold
    p = str(tmp_path / 'coord_0.npy')
    np.save(p, rec)
    pts, idx = confident_points(p, 20.0)
    assert pts.shape == (1, 3) and np.array_equal(pts[0], rec[1, 2, :3]) and idx.tolist() == [[1, 2]]
    np.save(p, rec[..., :3])
    with pytest.raises(ValueError):
        confident_points(p)


def test_chunk_loader_short_first_chunk(tmp_path):
    """first_chunk: only the first chunk is short (its decode is exposed: get the GPU going early), the rest are `chunk`."""
    from kfnet_amd.pipeline import ChunkLoader
    frames, paths = _png_sequence(tmp_path, 11)
    loader = ChunkLoader(paths, (6, 8), chunk=4, workers=2, pinned=False, first_chunk=2)
    assert loader.bounds() == [(0, 2), (2, 6), (6, 10), (10, 11)] and len(loader) == 4
    got = [(lo, host.numpy().copy()) for lo, host in loader]
    assert [g[0] for g in got] == [0, 2, 6, 10] and [g[1].shape[0] for g in got] == [2, 4, 4, 1]
    assert np.array_equal(np.concatenate([g[1] for g in got]), frames)
    assert ChunkLoader(frames, (6, 8), chunk=4, pinned=False, first_chunk=9).bounds()[0] == (0, 4)     # never longer than a chunk


def test_chunk_loader_ramp(tmp_path):
    """first_chunk as a sequence = a ramp of short chunks (own buffers, outside the rotation) in front of the full ones; every
    chunk handed out stays intact while later chunks are produced; a sequence shorter than the ramp just ends inside it."""
    from kfnet_amd.pipeline import ChunkLoader
    frames, paths = _png_sequence(tmp_path, 23)
    loader = ChunkLoader(paths, (6, 8), chunk=8, workers=3, pinned=False, first_chunk=(2, 4, 8, 9), depth=2)
    assert loader.ramp == [2, 4]                       # entries >= chunk are not a ramp
    assert loader.bounds() == [(0, 2), (2, 6), (6, 14), (14, 22), (22, 23)]
    views = [(lo, host) for lo, host in loader]        # (no copies: the ramp chunks must still hold their frames at the end)
    assert np.array_equal(views[0][1].numpy(), frames[0:2]) and np.array_equal(views[1][1].numpy(), frames[2:6])
    assert np.array_equal(views[-1][1].numpy(), frames[22:23])
    got = [(lo, host.numpy().copy()) for lo, host in ChunkLoader(frames, (6, 8), chunk=8, pinned=False, first_chunk=[3, 5])]
    assert [g[0] for g in got] == [0, 3, 8, 16] and np.array_equal(np.concatenate([g[1] for g in got]), frames)
    short = ChunkLoader(frames[:4], (6, 8), chunk=8, pinned=False, first_chunk=(3, 5))
    assert short.bounds() == [(0, 3), (3, 4)]
    assert np.array_equal(np.concatenate([h.numpy().copy() for _, h in short]), frames[:4])
