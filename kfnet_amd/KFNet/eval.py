"""Per-sequence prediction entry point with the reference's `KFNet/eval.py` command line
and output contract (README.md:126-130, KFNet/eval.py:31-177):

    python -m kfnet_amd.KFNet.eval --input_folder I --output_folder O --model_folder M --scene S [--NIS]

I holds image_list.txt (+ optional label_list.txt) and transform.txt; for every image one
`coord_<index>.npy` float32 [h,w,4] = concat(T.x_KF, 1/sigma_KF) is written to O.
`--synthetic T` replaces the image list by a seeded synthetic sequence (no dataset is
reachable from the build environment); `--random_weights` replaces the checkpoint.

Launched under `python -m torch.distributed.run --nproc-per-node N -m kfnet_amd.KFNet.eval ...`
the sequence is frame-sharded (BASELINE config 4): rank r processes its contiguous chunk on
GPU LOCAL_RANK (falling back to sharing GPUs over gloo when there are fewer GPUs than ranks),
receives the Kalman state from rank r-1 just before its scan and writes its own
coord_<i>.npy files -- bit-identical to a single-process run (kfnet_amd/dist.py).

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
# first frames of the test sequences per scene (get_indexes(False), KFNet/train.py:82-141)
M_TEST_SEQUENCE_LENGTH = {'chess': 1000, 'fire': 1000, 'heads': 1000, 'office': 1000, 'pumpkin': 1000,
                          'redkitchen': 1000, 'stairs': 500}


def get_transform(transform_file=None):
    """KFNet/train.py:49-58."""
    if transform_file:
        transform = np.loadtxt(transform_file, dtype=np.float32)
        return np.linalg.inv(transform)
    return np.eye(4, dtype=np.float32)


def load_images(paths, image_size, workers=None):
    """tf.image.decode_png(channels=3) replacement (KFNet/train.py:213-217): the library's threaded decoder
    (kfn_decode_png_rgb8 through pipeline.decode_png_batch), PIL where the library is not built."""
    from ..pipeline import _native_png_available, decode_image, decode_png_batch
    H, W = image_size
    out = np.empty((len(paths), H, W, 3), dtype=np.uint8)
    if _native_png_available():
        decode_png_batch(list(paths), out, image_size, workers or max(4, min(32, (os.cpu_count() or 8) // 2)))
        return out
    for i, p in enumerate(paths):
        out[i] = decode_image(p, image_size)
    return out


# eval.py's host loop resets the filter every `spec.sequence_length` frames, and KFNetDataSpec() is built with
# the default scene there: 500 whatever --scene says.  ONE constant for every caller.
RESET_PERIOD = 500


def eval_sharded(image_paths, transform, weights, output_folder, rank, world, link, nis=False,
                 image_size=(480, 640), batch=4, frames=None, sequence_length=RESET_PERIOD, verbose=True,
                 device=None, decode_workers=None):
    """Frame-sharded prediction (BASELINE config 4): this rank owns the contiguous chunk
    `chunk_bounds(T, world, rank)`, runs the state-independent heavy phase for it at once,
    receives the [h,w,4] Kalman state from rank-1 (unless its chunk starts on a reset
    frame), scans, sends the state on and writes coord_<i>.npy for its own frames.
    Returns (first_frame, records [n,h,w,4])."""
    import torch
    from ..dist import chunk_bounds, needs_state, run_chunk
    from ..engine import KFNetEngine
    T = len(image_paths) if frames is None else frames.shape[0]
    lo, hi = chunk_bounds(T, world, rank)
    dev = device if device is not None else 'cuda:%d' % torch.cuda.current_device()
    eng = KFNetEngine(weights, image_size=image_size, batch=batch, transform=transform,
                      reset_period=sequence_length, nis_gate=7.815 if nis else 0.0, max_chunk=max(hi - lo, 1),
                      device=dev)
    need_prev = 1 if (hi > lo and needs_state(lo, sequence_length)) else 0
    if frames is not None:
        host = np.ascontiguousarray(frames[lo - need_prev:hi])
    else:
        host = load_images(image_paths[lo - need_prev:hi], image_size, decode_workers)
    dev_all = eng.upload_frames(host) if host.shape[0] else torch.empty((0, eng.H, eng.W, 3), dtype=torch.uint8, device=dev)
    rec = run_chunk(eng, dev_all[need_prev:], lo, rank, world, link, dev_all[0] if need_prev else None)
    rec = rec.cpu().numpy().copy()
    if output_folder and os.path.isdir(output_folder):
        for k in range(rec.shape[0]):
            np.save(os.path.join(output_folder, 'coord_%d.npy' % (lo + k)), rec[k].astype(np.float32))
    if verbose:
        print('rank %d/%d: frames %d~%d done' % (rank, world, lo, hi - 1))
    return lo, rec


def eval_sharded_cyclic(image_paths, transform, weights, output_folder, rank, world, link, block, nis=False,
                        image_size=(480, 640), batch=4, frames=None, sequence_length=RESET_PERIOD, verbose=True, device=None):
    """Block-cyclic frame sharding (kfnet_amd.dist.run_cyclic): blocks of `block` frames dealt round-robin, block j on rank
    j % world; the Kalman state hops rank -> rank+1 once per block, so a rank scans its block while the others are still in
    the heavy phase of theirs.  Same records, bit for bit, as eval_sharded and as a single process.
    Returns [(first_frame, records [n,h,w,4])] of this rank's blocks."""
    import torch
    from ..dist import run_cyclic
    from ..engine import KFNetEngine
    T = len(image_paths) if frames is None else frames.shape[0]
    dev = device if device is not None else 'cuda:%d' % torch.cuda.current_device()
    eng = KFNetEngine(weights, image_size=image_size, batch=batch, transform=transform, reset_period=sequence_length,
                      nis_gate=7.815 if nis else 0.0, max_chunk=max(int(block), 1), device=dev)

    def frames_of(lo, hi):       # only this rank's blocks (+ the frame in front of each) are ever decoded / uploaded
        host = np.ascontiguousarray(frames[lo:hi]) if frames is not None else load_images(image_paths[lo:hi], image_size)
        return eng.upload_frames(host)

    out = []

    def on_block(lo, rec):
        r = rec.cpu().numpy().copy()
        if output_folder and os.path.isdir(output_folder):
            for k in range(r.shape[0]):
                np.save(os.path.join(output_folder, 'coord_%d.npy' % (lo + k)), r[k].astype(np.float32))
        out.append((lo, r))

    run_cyclic(eng, frames_of, T, int(block), rank, world, link, on_block=on_block)
    if verbose:
        print('rank %d/%d: %d blocks of %d frames done (block-cyclic)' % (rank, world, len(out), block))
    return out


def eval(image_paths, transform, weights, output_folder, nis=False, image_size=(480, 640), batch=4,
         frames=None, sequence_length=RESET_PERIOD, chunk=256, verbose=True, label_paths=None, labels=None,
         decode_workers=None, device=None, metrics_sequence_length=1000, engine=None, save_workers=2, stats=None, ramp=None):
    """Runs the sequence and writes coord_<i>.npy files; returns the [T,h,w,4] records.
    With label maps (label_list.txt, or `labels` [T,H,W,4] in memory) the reference's per-frame
    log line and final median/mean/std summary are printed (KFNet/eval.py:113-118,162-164);
    returns (records, metrics) in that case.  `engine`: a ready KFNetEngine to run on (its batch / max_chunk / transform
    are used; bench.py times the path without the one-off graph build); `save_workers`: threads that write the
    coord_<i>.npy files behind the consumer loop (np.save releases the GIL in the file write); `stats`: a dict that receives
    where the consumer thread's wall time went (StreamedSequence.stats + `emit` and `saves_wait` seconds); `ramp`: lengths of
    the first chunks (default (8, 16) in front of chunks of >= 32 frames: the first decode is exposed, keep it short, and let
    each ramp chunk's compute cover the next one's decode -- pipeline.ChunkLoader)."""
    from ..engine import KFNetEngine
    from . import metrics as M
    from ..pipeline import ChunkLoader, StreamedSequence
    T = len(image_paths) if frames is None else frames.shape[0]
    want_metrics = label_paths is not None or labels is not None
    if device is None:   # --gpu N: everything (buffers, streams, launches) lives on the CURRENT device
        import torch
        device = 'cuda:%d' % torch.cuda.current_device()
    eng = engine if engine is not None else KFNetEngine(
        weights, image_size=image_size, batch=batch, transform=transform, reset_period=sequence_length,
        nis_gate=7.815 if nis else 0.0, max_chunk=chunk, emit_metrics=want_metrics, device=device)
    if engine is not None:
        # a supplied engine carries its own settings; an argument that DISAGREES with them must not be dropped silently
        # (eval(..., nis=True, engine=eng_without_gate) used to return ungated records)
        chunk = min(chunk, eng.max_chunk)
        if want_metrics and not eng.emit_metrics:
            raise ValueError('labels given, but the engine was built without emit_metrics')
        if bool(nis) != (eng.nis_gate > 0.0):
            raise ValueError('eval(nis=%r) disagrees with the engine (nis_gate=%g)' % (nis, eng.nis_gate))
        if int(sequence_length) != int(eng.reset_period):
            raise ValueError('eval(sequence_length=%d) disagrees with the engine (reset_period=%d)'
                             % (sequence_length, eng.reset_period))
        if tuple(image_size) != (eng.H, eng.W):
            raise ValueError('eval(image_size=%r) disagrees with the engine (%dx%d)' % (tuple(image_size), eng.H, eng.W))
        t_arg = None if transform is None else np.asarray(transform, dtype=np.float32)
        if (t_arg is None) != (eng.transform is None) or (t_arg is not None and not np.array_equal(t_arg, eng.transform)):
            raise ValueError('eval(transform=...) disagrees with the transform the engine was built with')
    records, all_metrics = [], []
    # coord_<i>.npy (KFNet/eval.py:121-126) written off the consumer thread: the next chunk's records can be
    # fetched while the previous chunk's 76.8 KB files are still going to disk
    from concurrent.futures import ThreadPoolExecutor
    saver = ThreadPoolExecutor(max(1, int(save_workers))) if (output_folder and os.path.isdir(output_folder)) else None
    pending_saves = []

    def save_chunk(lo, rec):
        for k in range(rec.shape[0]):
            np.save(os.path.join(output_folder, 'coord_%d.npy' % (lo + k)), rec[k])

    def emit(lo, rec):
        rec = np.ascontiguousarray(rec, dtype=np.float32)
        records.append(rec)
        if saver is not None:
            pending_saves.append(saver.submit(save_chunk, lo, rec))

    # decode thread + pinned staging (the reference's queue runners, KFNet/train.py:195-239);
    # uploads / compute / downloads overlapped on three streams -- with or without labels
    # chunks in flight on the GPU: 3 (a second chunk queued behind the running one); the metrics reduction double-buffers
    # its results, so it keeps 2.  The loader rotates one host buffer more than that: a buffer is recycled only after the
    # records of the chunk that was uploaded from it have been handed out, i.e. its upload is complete.
    in_flight = 2 if want_metrics else 3
    if not decode_workers:      # decode threads of the library (kfn_decode_png_rgb8): a chunk's files side by side
        decode_workers = max(4, min(32, (os.cpu_count() or 8) // 2))
    loader = ChunkLoader(frames if frames is not None else list(image_paths), image_size, chunk,
                         workers=decode_workers, first_chunk=[r for r in ((8, 16) if ramp is None else ramp) if r < chunk],
                         depth=in_flight + 1)
    dm = M.DeviceMetrics(eng) if want_metrics else None
    plan = {}     # chunk index -> (first, n, global pairs)

    def label_grid(i):
        if labels is not None:
            return M.resize_nearest(labels[i], (eng.h, eng.w))
        return M.read_label_grid(label_paths[i], image_size, (eng.h, eng.w))

    def after_process(k, lo, n):
        """Right behind the scan of chunk k: label grids of the frames its pairs refer to -> device,
        kfn_eval_metrics, results -> pinned host slot k & 1."""
        pairs = M.pair_schedule(lo, n, T, metrics_sequence_length)
        base, top = max(min(int(pairs.min()), lo), 0), min(max(int(pairs.max()), lo + n - 1) + 1, T)
        rows = np.stack([label_grid(i) for i in range(base, top)])
        dm.launch(k & 1, lo, n, rows, np.clip(pairs, base, top - 1) - base)
        plan[k] = (lo, n, pairs)

    k = 0
    import time
    t_emit = t_saves = 0.0
    seq = StreamedSequence(eng, chunk, depth=in_flight)
    try:
        for lo, rec in seq.run(loader, after_process=after_process if want_metrics else None):
            t_e = time.perf_counter()
            emit(lo, rec.copy())
            t_emit += time.perf_counter() - t_e
            if want_metrics:
                first, n, pairs = plan.pop(k)
                for m in dm.collect(k & 1, first, n, pairs):
                    all_metrics.append(m)
                    if verbose:
                        print(M.format_line(m))
            elif verbose:
                print('frames %d~%d done' % (lo, lo + rec.shape[0] - 1))
            k += 1
        t_e = time.perf_counter()
        for f in pending_saves:
            f.result()          # re-raise write errors; every file is on disk when eval() returns
        t_saves = time.perf_counter() - t_e
        if stats is not None:
            stats.update(getattr(seq, 'stats', {}), emit=t_emit, saves_wait=t_saves,
                         producer={k: round(v, 4) for k, v in getattr(loader, 'stats', {}).items()})
    finally:
        if saver is not None:   # also when the loop raised: no writer thread outlives the call
            saver.shutdown()
    records = np.concatenate(records) if records else np.zeros((0, eng.h, eng.w, 4), np.float32)
    if not want_metrics:
        return records
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
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--sharding', choices=['contiguous', 'cyclic'], default='contiguous',
                    help='multi-process runs: contiguous chunks per rank, or blocks of --block frames dealt round-robin')
    ap.add_argument('--block', type=int, default=32, help='--sharding cyclic: frames per block')
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
    size = (a.height, a.width)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        return _main_sharded(a, W, size, rank, world)
    torch.cuda.set_device(a.gpu)
    device = 'cuda:%d' % a.gpu
    if a.synthetic > 0:
        from ..synth import synthetic_sequence, synthetic_transform
        frames = synthetic_sequence(a.synthetic, a.height, a.width)
        transform = np.linalg.inv(synthetic_transform())
        eval(None, transform, W, a.output_folder, a.NIS, image_size=size, frames=frames, batch=a.batch, device=device)
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
    eval(image_paths, get_transform(transform_file), W, a.output_folder, a.NIS, image_size=size, batch=a.batch,
         label_paths=label_paths, device=device,
         metrics_sequence_length=M_TEST_SEQUENCE_LENGTH.get(a.scene, 1000))
    return 0


def _main_sharded(a, W, size, rank, world):
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE)."""
    import torch
    import torch.distributed as dist
    from ..dist import make_link
    ndev = torch.cuda.device_count()
    dev_index = int(os.environ.get('LOCAL_RANK', '0')) % max(ndev, 1)
    if a.gpu != 0 and rank == 0:
        print('WARNING: --gpu %d is ignored under torch.distributed.run: rank r uses device LOCAL_RANK' % a.gpu,
              file=sys.stderr)
    torch.cuda.set_device(dev_index)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = os.environ.get('KFN_DIST_BACKEND', 'nccl' if ndev >= world else 'gloo')
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_index))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    link = make_link(dist, rank, world, dev_index, prefer=os.environ.get('KFN_STATE_LINK', 'auto'))
    try:
        if a.synthetic > 0:
            from ..synth import synthetic_sequence, synthetic_transform
            from ..dist import chunk_bounds, needs_state
            transform = np.linalg.inv(synthetic_transform())
            if a.sharding == 'cyclic':
                # (the generator is a function of (seed, frame index): every rank synthesises only what its blocks need)
                gen = _SyntheticFrames(a.synthetic, a.height, a.width)
                eval_sharded_cyclic(None, transform, W, a.output_folder, rank, world, link, a.block, a.NIS, image_size=size,
                                    batch=a.batch, frames=gen, sequence_length=RESET_PERIOD)
                torch.cuda.synchronize()
                dist.barrier()
                return 0
            lo, hi = chunk_bounds(a.synthetic, world, rank)
            # every rank generates only the frames it needs (frame t depends on (seed, t) alone)
            first = lo - (1 if (hi > lo and needs_state(lo, RESET_PERIOD)) else 0)
            part = synthetic_sequence(hi - first, a.height, a.width, start=first)
            frames = _ShiftedFrames(part, first, a.synthetic)
            eval_sharded(None, transform, W, a.output_folder, rank, world, link, a.NIS, image_size=size,
                         batch=a.batch, frames=frames, sequence_length=RESET_PERIOD)
        else:
            if os.path.exists(os.path.join(a.input_folder, 'label_list.txt')) and rank == 0:
                # the single-process run prints eval.py's per-frame l_/a_/d_/nis line and the median summary from
                # these labels; the sharded run writes the same coord_<i>.npy files but computes no metrics
                print('WARNING: label_list.txt found, but the sharded run (WORLD_SIZE=%d) does not evaluate labels: '
                      'no per-frame log line and no median summary will be printed.  Run single-process '
                      '(python -m kfnet_amd.KFNet.eval --gpu N ...) for the metrics.' % world, file=sys.stderr)
            image_paths = read_lines(os.path.join(a.input_folder, 'image_list.txt'))
            if a.sharding == 'cyclic':
                eval_sharded_cyclic(image_paths, get_transform(os.path.join(a.input_folder, 'transform.txt')), W,
                                    a.output_folder, rank, world, link, a.block, a.NIS, image_size=size, batch=a.batch)
            else:
                eval_sharded(image_paths, get_transform(os.path.join(a.input_folder, 'transform.txt')), W,
                             a.output_folder, rank, world, link, a.NIS, image_size=size, batch=a.batch)
        torch.cuda.synchronize()
        dist.barrier()
    finally:
        if link is not None:
            link.close()
        dist.destroy_process_group()
    return 0


class _SyntheticFrames(object):
    """The seeded synthetic sequence, generated slice by slice (frame t depends on (seed, t) alone)."""

    def __init__(self, total, height, width):
        self.shape = (total, height, width, 3)
        self.h, self.w = height, width

    def __getitem__(self, sl):
        from ..synth import synthetic_sequence
        return synthetic_sequence(sl.stop - sl.start, self.h, self.w, start=sl.start)


class _ShiftedFrames(object):
    """A window [first, first+n) of a T-frame sequence that indexes like the whole sequence
    (so a rank holds only its own frames)."""

    def __init__(self, part, first, total):
        self.part, self.first, self.shape = part, first, (total,) + part.shape[1:]

    def __getitem__(self, sl):
        return self.part[sl.start - self.first:sl.stop - self.first]


if __name__ == '__main__':
    sys.exit(main())
