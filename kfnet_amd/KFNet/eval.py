"""Per-sequence prediction entry point with the reference's `KFNet/eval.py` command line
and output contract (README.md:126-130, KFNet/eval.py:31-177):

    python -m kfnet_amd.KFNet.eval --input_folder I --output_folder O --model_folder M --scene S [--NIS]

I holds image_list.txt (+ optional label_list.txt) and transform.txt; for every image one
`coord_<index>.npy` float32 [h,w,4] = concat(T.x_KF, 1/sigma_KF) is written to O.
`--synthetic T` replaces the image list by a seeded synthetic sequence (no dataset is
reachable from the build environment); `--random_weights` replaces the checkpoint.

Host loop semantics kept from eval.py: sequence_length = 500 irrespective of --scene
(KFNetDataSpec() is built with the default scene), reset at i % 500 == 0,
raw (untransformed, ungated) KF state fed back, NIS gate on the output only.
When label_list.txt is present the reference's per-frame log line (losses, accuracies, median
distance errors in cm, NIS-in-band fraction) and the final summary are printed
(kfnet_amd/KFNet/metrics.py).  Not reproduced: --show plotting.
"""
import argparse
import os
import sys

import numpy as np

from ..tools.io import get_snapshot, read_lines
from ..weights import load_npz, synthetic_weights

SCENES = ('chess', 'fire', 'heads', 'office', 'pumpkin', 'redkitchen', 'stairs')


def get_transform(transform_file=None):
    """KFNet/train.py:49-58."""
    if transform_file:
        transform = np.loadtxt(transform_file, dtype=np.float32)
        return np.linalg.inv(transform)
    return np.eye(4, dtype=np.float32)


def load_images(paths, image_size):
    """tf.image.decode_png(channels=3) replacement (KFNet/train.py:213-217)."""
    from ..pipeline import decode_image
    H, W = image_size
    out = np.empty((len(paths), H, W, 3), dtype=np.uint8)
    for i, p in enumerate(paths):
        out[i] = decode_image(p, image_size)
    return out


def eval(image_paths, transform, weights, output_folder, nis=False, image_size=(480, 640), batch=4,
         frames=None, sequence_length=500, chunk=256, verbose=True, label_paths=None, labels=None,
         decode_workers=8):
    """Runs the sequence and writes coord_<i>.npy files; returns the [T,h,w,4] records.
    With label maps (label_list.txt, or `labels` [T,H,W,4] in memory) the reference's per-frame
    log line and final median/mean/std summary are printed (KFNet/eval.py:113-118,162-164);
    returns (records, metrics) in that case."""
    from ..engine import KFNetEngine
    from . import metrics as M
    from ..pipeline import ChunkLoader, StreamedSequence
    T = len(image_paths) if frames is None else frames.shape[0]
    want_metrics = label_paths is not None or labels is not None
    eng = KFNetEngine(weights, image_size=image_size, batch=batch, transform=transform,
                      reset_period=sequence_length, nis_gate=7.815 if nis else 0.0, max_chunk=chunk,
                      emit_debug=want_metrics)
    records, all_metrics = [], []

    def label(i):
        return labels[i] if labels is not None else M.read_label(label_paths[i], image_size)

    def emit(lo, rec):
        records.append(rec)
        if output_folder and os.path.isdir(output_folder):
            for k in range(rec.shape[0]):
                np.save(os.path.join(output_folder, 'coord_%d.npy' % (lo + k)), rec[k].astype(np.float32))

    # decode thread + pinned staging (the reference's queue runners, KFNet/train.py:195-239)
    loader = ChunkLoader(frames if frames is not None else list(image_paths), image_size, chunk,
                         workers=decode_workers)
    if not want_metrics:
        # uploads / compute / downloads overlapped on three streams
        for lo, rec in StreamedSequence(eng, chunk).run(loader):
            emit(lo, rec.copy())
            if verbose:
                print('frames %d~%d done' % (lo, lo + rec.shape[0] - 1))
        return np.concatenate(records) if records else np.zeros((0, eng.h, eng.w, 4), np.float32)
    # with labels: per-frame metrics need the intermediate maps of every chunk on the host
    for lo, host in loader:
        hi = lo + int(host.shape[0])
        dev = eng.upload_frames(host.numpy())
        rec = eng.process(dev, t0=lo).cpu().numpy()   # state and feature ring carry over
        emit(lo, rec)
        dbg = eng.debug(hi - lo)
        for k in range(hi - lo):
            i = lo + k
            # pair schedule of KFNet/train.py:67-71: step 0 = (1, 0), step i = (i-1, i)
            pair = (1, 0) if i == 0 else (i - 1, i)
            reset = sequence_length > 0 and i % sequence_length == 0
            m = M.frame_metrics(i, pair, dbg['meas'][k], dbg['temp'][k], rec[k], dbg['nis'][k],
                                (label(min(pair[0], T - 1)), label(pair[1])), transform, reset, (eng.h, eng.w))
            all_metrics.append(m)
            if verbose:
                print(M.format_line(m))
    records = np.concatenate(records)
    if verbose and all_metrics:
        for name, fn in (('Median dist error: ', np.median), ('Mean dist error: ', np.mean), ('stddev error: ', np.std)):
            print(name, fn([m['d_m'] for m in all_metrics]), fn([m['d_t'] for m in all_metrics]),
                  fn([m['d_kf'] for m in all_metrics]))
    return records, all_metrics


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--input_folder', default='')
    ap.add_argument('--output_folder', default='')
    ap.add_argument('--model_folder', default='')
    ap.add_argument('--scene', default='')
    ap.add_argument('--NIS', action='store_true')
    ap.add_argument('--gpu', type=int, default=0)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--synthetic', type=int, default=0, help='use a seeded synthetic sequence of this many frames')
    ap.add_argument('--random_weights', action='store_true')
    a = ap.parse_args(argv)
    if a.scene not in SCENES:
        print('Invalid scene:', a.scene)   # KFNet/train.py:142-144
        return 1
    if a.random_weights:
        W = synthetic_weights(1234)
    else:
        snapshot, step = get_snapshot(a.model_folder)
        if snapshot is None:
            print('no kfnet_weights*.npz in', a.model_folder)
            return 1
        W = load_npz(snapshot)
    import torch
    torch.cuda.set_device(a.gpu)
    if a.synthetic > 0:
        from ..synth import synthetic_sequence, synthetic_transform
        frames = synthetic_sequence(a.synthetic)
        transform = np.linalg.inv(synthetic_transform())
        eval(None, transform, W, a.output_folder, a.NIS, frames=frames, batch=a.batch)
        return 0
    image_list = os.path.join(a.input_folder, 'image_list.txt')
    transform_file = os.path.join(a.input_folder, 'transform.txt')
    image_paths = read_lines(image_list)
    print('----------------------------------')
    print('scene: ', a.scene)
    print('image number: ', len(image_paths))
    print('----------------------------------')
    label_list = os.path.join(a.input_folder, 'label_list.txt')
    label_paths = read_lines(label_list) if os.path.exists(label_list) else None
    if label_paths is not None:
        assert len(image_paths) == len(label_paths)   # KFNet/eval.py:37
    eval(image_paths, get_transform(transform_file), W, a.output_folder, a.NIS, batch=a.batch, label_paths=label_paths)
    return 0


if __name__ == '__main__':
    sys.exit(main())
