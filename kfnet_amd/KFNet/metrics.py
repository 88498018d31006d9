"""Per-frame evaluation numbers of KFNet/eval.py, reduced on the device.

eval.py prints for every step `l_m, l_t, l_kf, a_m, a_t, a_kf` (three CoordLossWithUncertainty results,
KFNet/KFNet.py:192-232 via KFNet/train.py:252-257), `d_m, d_t, d_kf` (median distance errors in cm,
KFNet/eval.py:17-29) and `nis` (share of NIS values inside (0.0157, 2.706), eval.py:10-15).  Everything that
is a sum or a count over the label grid is reduced by kfn_eval_metrics (one workgroup per frame) from the
buffers the Kalman scan wrote -- nothing but [T,16] numbers and the [T,3,h,w] distance maps crosses PCIe;
the host takes the medians (as the reference does, np.median over the positive entries) and formats the
line.  The host side here reads labels and builds the step schedule:

  * labels are raw float32 [H,W,4] dumps (3 scene coordinates + validity mask, KFNet/train.py:219-222),
    nearest-down-sampled like tf.image.resize_nearest_neighbor: source pixel (8y, 8x) --
    done while READING (a strided view of the file), so 76.8 KB per frame are uploaded, not 4.9 MB;
  * the frame pair of step i is (i-1, i), and (s+1, s) at the first frame s of every test sequence of the
    scene (KFNet/train.py:67-71, :82-141): the losses compare the prediction with BOTH labels of the pair
    (the [2,h,w,.] ground-truth batch broadcasts against the [1,h,w,.] prediction), the distance errors
    use the second one.

Known deviation: at a later sequence start the reference also feeds the IMAGE pair reversed, which changes
the (discarded, because the step is a reset) prediction its l_t / l_kf / nis are computed from; here the
prediction of that step comes from the forward pair.  The coord_<i>.npy outputs are unaffected.
"""
import ctypes as C

import numpy as np

from .. import _lib

FORMAT = ("%d, frame %d~%d, l_m = %.3f, l_t = %.3f, l_kf = %.3f, a_m = %.3f, a_t = %.3f, a_kf = %.3f, "
          "d_m = %.3f, d_t = %.3f, d_kf = %.3f, nis = %.3f")   # KFNet/eval.py:113-118

# first frames of the test sequences per scene (get_indexes(is_training=False), KFNet/train.py:82-141)
TEST_SEQUENCE_LENGTH = {'chess': 1000, 'fire': 1000, 'heads': 1000, 'office': 1000, 'pumpkin': 1000,
                        'redkitchen': 1000, 'stairs': 500}


def read_label(path, image_size=(480, 640)):
    """tf.decode_raw(float32) + reshape [H,W,4] (KFNet/train.py:219-222)."""
    H, W = image_size
    a = np.fromfile(path, dtype=np.float32)
    if a.size != H * W * 4:
        raise ValueError('%s holds %d floats, expected %d' % (path, a.size, H * W * 4))
    return a.reshape(H, W, 4)


def nearest_rows_cols(image_size, grid_hw):
    """Source rows / columns of tf.image.resize_nearest_neighbor(align_corners=False):
    src = min(floor(dst * in / out), in - 1)."""
    H, W = image_size
    h, w = grid_hw
    ys = np.minimum((np.arange(h) * (H / float(h))).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / float(w))).astype(np.int64), W - 1)
    return ys, xs


def resize_nearest(x, out_hw):
    ys, xs = nearest_rows_cols(x.shape[-3:-1], out_hw)
    return x[..., ys[:, None], xs[None, :], :]


def read_label_grid(path, image_size, grid_hw):
    """The [h,w,4] nearest-down-sampled label of one frame, read through a memory map so that only
    the pixels that are kept are touched."""
    H, W = image_size
    m = np.memmap(path, dtype=np.float32, mode='r')
    if m.size != H * W * 4:
        raise ValueError('%s holds %d floats, expected %d' % (path, m.size, H * W * 4))
    ys, xs = nearest_rows_cols(image_size, grid_hw)
    return np.ascontiguousarray(m.reshape(H, W, 4)[ys[:, None], xs[None, :], :])


def pair_schedule(first, count, total, sequence_length):
    """Frame pairs (a, b) of steps first .. first+count-1 (KFNet/train.py:67-71): (s+1, s) at a sequence
    start s, (i-1, i) otherwise; indices are clamped to the list like tf.gather would fail to be."""
    out = np.empty((count, 2), dtype=np.int64)
    for k in range(count):
        i = first + k
        if sequence_length > 0 and i % sequence_length == 0:
            out[k] = (min(i + 1, total - 1), i)
        else:
            out[k] = (i - 1, i)
    return out


def dist_median(dmap):
    """np.median over the positive entries of a distance map in cm (KFNet/eval.py:27-29)."""
    pos = dmap[dmap > 0]
    return float(np.median(pos)) if pos.size else float('nan')


class DeviceMetrics(object):
    """kfn_eval_metrics on the buffers of a KFNetEngine built with emit_metrics=True."""

    def __init__(self, eng, dist_threshold=0.05):
        if not getattr(eng, 'emit_metrics', False):
            raise ValueError('the engine must be built with emit_metrics=True')
        torch = eng.torch
        self.eng, self.torch = eng, torch
        self.dist_threshold = float(dist_threshold)
        T = eng.max_chunk
        dev = eng.device
        # two result slots: the reduction of chunk k+1 is enqueued before chunk k's numbers are consumed
        self.stats = [torch.zeros((T, 16), dtype=torch.float32, device=dev) for _ in range(2)]
        self.dist = [torch.zeros((T, 3, eng.h, eng.w), dtype=torch.float32, device=dev) for _ in range(2)]
        self.h_stats = [torch.zeros((T, 16), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.h_dist = [torch.zeros((T, 3, eng.h, eng.w), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.done = [None, None]
        # per-slot inputs too (labels, pair table, reset flags), staged through pinned host buffers and copied
        # stream-ordered on the compute stream: chunk k+1's labels can never overwrite chunk k's while its
        # reduction is still reading them, and no upload blocks the host (upload / compute / download overlap)
        self.labels = [torch.zeros((T + 2, eng.h, eng.w, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        self.pairs = [torch.zeros((T, 2), dtype=torch.int32, device=dev) for _ in range(2)]
        self.resets = [torch.zeros((T,), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.h_labels = [torch.zeros((T + 2, eng.h, eng.w, 4), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.h_pairs = [torch.zeros((T, 2), dtype=torch.int32).pin_memory() for _ in range(2)]
        self.h_resets = [torch.zeros((T,), dtype=torch.uint8).pin_memory() for _ in range(2)]
        tr = eng.transform
        self.t12 = None if tr is None else (C.c_float * 12)(*[float(v) for v in np.asarray(tr, np.float32)[:3, :4].reshape(-1)])

    def launch(self, slot, first, count, label_rows, pairs):
        """Enqueue (on the engine's stream, right behind the scan) the reduction for the `count` frames just
        scanned (global indices first..) and the download of its results into pinned host slot `slot`.
        label_rows [L,h,w,4] host float32 = the label grids this chunk refers to, pairs [count,2] = rows of it."""
        eng, torch = self.eng, self.torch
        L = int(label_rows.shape[0])
        if L > self.labels[0].shape[0] or count > self.stats[0].shape[0]:
            raise ValueError('chunk larger than the metric buffers')
        if self.done[slot] is not None:
            self.done[slot].synchronize()      # the pinned staging of this slot is free once its last use finished
        rp = eng.reset_period
        flags = np.array([1 if (rp > 0 and (first + k) % rp == 0) else 0 for k in range(count)], dtype=np.uint8)
        self.h_labels[slot][:L].copy_(torch.from_numpy(np.ascontiguousarray(label_rows, dtype=np.float32)))
        self.h_pairs[slot][:count].copy_(torch.from_numpy(np.ascontiguousarray(pairs, dtype=np.int32)))
        self.h_resets[slot][:count].copy_(torch.from_numpy(flags))
        with torch.cuda.stream(torch.cuda.current_stream(eng.device)):
            self.labels[slot][:L].copy_(self.h_labels[slot][:L], non_blocking=True)
            self.pairs[slot][:count].copy_(self.h_pairs[slot][:count], non_blocking=True)
            self.resets[slot][:count].copy_(self.h_resets[slot][:count], non_blocking=True)
        rc = eng.lib.kfn_eval_metrics(eng.c_meas.ptr, eng.c_temp.ptr, eng.c_kf.ptr, eng.c_rec.ptr, eng.c_nis.ptr,
                                      self.labels[slot].data_ptr(), self.pairs[slot].data_ptr(),
                                      self.resets[slot].data_ptr(), self.t12,
                                      int(count), int(eng.hw), self.dist_threshold, float(eng.net.min_uncertainty),
                                      self.stats[slot].data_ptr(), self.dist[slot].data_ptr(), eng._stream())
        _lib.check(rc, 'kfn_eval_metrics')
        self.h_stats[slot][:count].copy_(self.stats[slot][:count], non_blocking=True)
        self.h_dist[slot][:count].copy_(self.dist[slot][:count], non_blocking=True)
        self.done[slot] = torch.cuda.Event()
        self.done[slot].record(torch.cuda.current_stream(eng.device))

    def collect(self, slot, first, count, pairs_global):
        """Wait for slot `slot` and finish: one dict per frame with the fields of eval.py's log line."""
        self.done[slot].synchronize()
        st = self.h_stats[slot][:count].numpy()
        dm = self.h_dist[slot][:count].numpy()
        out = []
        for k in range(count):
            s = st[k]
            valid = float(s[6])
            m = dict(i=first + k, pair=(int(pairs_global[k][0]), int(pairs_global[k][1])),
                     l_m=float(s[0]) / valid, l_t=float(s[1]) / valid, l_kf=float(s[2]) / valid,
                     a_m=(valid - float(s[3])) / valid, a_t=(valid - float(s[4])) / valid, a_kf=(valid - float(s[5])) / valid,
                     d_m=dist_median(dm[k, 0]), d_t=dist_median(dm[k, 1]), d_kf=dist_median(dm[k, 2]),
                     nis=(float(s[8]) / float(s[7])) if s[7] > 0 else 0.0)
            out.append(m)
        return out


def format_line(m):
    return FORMAT % (m['i'], m['pair'][0], m['pair'][1], m['l_m'], m['l_t'], m['l_kf'], m['a_m'], m['a_t'],
                     m['a_kf'], m['d_m'], m['d_t'], m['d_kf'], m['nis'])
