"""ORACLE (test infrastructure -- never imported by kfnet_amd/): numpy restatement of the per-frame
evaluation numbers of KFNet/eval.py, the checker of the device reduction
kfn_eval_metrics (kfnet_amd/csrc/kfn_metrics.hip, kfnet_amd/KFNet/metrics.py).

Follows KFNet.CoordLossWithUncertainty (KFNet/KFNet.py:192-232), get_NIS_measurement and dist_error
(KFNet/eval.py:10-29) and the step body of eval.py:77-118.  Parity unpinned with respect to TensorFlow
(no reference run can be captured here); pinned by the hand-computed values of tests/test_metrics.py.

Quirks reproduced on purpose:
  * `loss_map = tf.minimum(loss_map, -2.0)` (KFNet/KFNet.py:216): the per-pixel NLL is
    capped from ABOVE at -2;
  * in eval.py the ground truth fed to the losses is the PAIR batch [2,h,w,3] while the
    prediction is [1,h,w,3], so both frames of the pair are compared with the same
    prediction (broadcast) and `valid_pixel` sums both masks (+1);
  * `dist_error` and the log line use the second frame of the pair only.
"""
import numpy as np


def read_label(path, image_size=(480, 640)):
    """tf.decode_raw(float32) + reshape [H,W,4] (KFNet/train.py:219-222): 3 scene
    coordinates + 1 validity mask per full-resolution pixel."""
    H, W = image_size
    a = np.fromfile(path, dtype=np.float32)
    if a.size != H * W * 4:
        raise ValueError('%s holds %d floats, expected %d' % (path, a.size, H * W * 4))
    return a.reshape(H, W, 4)


def resize_nearest(x, out_hw):
    """tf.image.resize_nearest_neighbor(align_corners=False): src = floor(dst*in/out)
     -> picks pixel (8y, 8x) for 480x640 -> 60x80."""
    H, W = x.shape[-3], x.shape[-2]
    h, w = out_hw
    ys = np.minimum((np.arange(h) * (H / float(h))).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / float(w))).astype(np.int64), W - 1)
    return x[..., ys[:, None], xs[None, :], :]


def coord_loss_with_uncertainty(pred_coord, uncertainty, gt_coords, mask, dist_threshold=0.05,
                                min_uncertainty=1e-5):
    """KFNet.CoordLossWithUncertainty (KFNet/KFNet.py:192-232) on already transformed,
    already down-sampled inputs.  pred_coord [1,h,w,3], uncertainty [1,h,w,1], gt_coords
    [B,h,w,3], mask [B,h,w,1] (B = 2 in eval.py).  Returns (loss, accuracy)."""
    pred = np.asarray(pred_coord, np.float32)
    unc = np.maximum(np.asarray(uncertainty, np.float32), np.float32(min_uncertainty))
    gt = np.asarray(gt_coords, np.float32)
    m = (np.asarray(mask, np.float32) == 1.0).astype(np.float32)
    diff = np.sum(np.square(pred - gt), axis=-1, keepdims=True)
    loss_map = 3.0 * np.log(unc) + diff / (2.0 * np.square(unc))
    loss_map = np.minimum(loss_map, np.float32(-2.0))
    valid_pixel = m.sum() + 1.0
    diff = m * diff
    loss_map = m * loss_map
    loss = loss_map.sum() / valid_pixel
    thres = np.maximum(diff - dist_threshold * dist_threshold, 0)
    num_accurate = valid_pixel - np.count_nonzero(thres)
    return float(loss), float(num_accurate / valid_pixel)


def get_NIS_measurement(out_NIS):
    """KFNet/eval.py:10-15: fraction of positive NIS values inside (0.0157, 2.706)."""
    a = np.asarray(out_NIS).reshape(-1)
    a = a[a > 0.0]
    if a.size == 0:
        return 0.0
    return float(np.count_nonzero((a > 0.0157) & (a < 2.706))) / float(a.size)


def dist_error(coords, gt_coords, mask):
    """KFNet/eval.py:17-29: median Euclidean error in cm over valid pixels, and the map."""
    d = np.sqrt(np.sum(np.square(coords - gt_coords), axis=-1)) * mask[:, :, 0]
    pos = d[d > 0]
    med = float(np.median(pos)) * 100.0 if pos.size else float('nan')
    return med, d * 100


def apply_transform(coords, T):
    """KFNet/util.py:12-40 on host: x' = (T [x;1])[0:3]."""
    T = np.asarray(T, np.float32)
    return coords @ T[:3, :3].T + T[:3, 3]


FORMAT = ("%d, frame %d~%d, l_m = %.3f, l_t = %.3f, l_kf = %.3f, a_m = %.3f, a_t = %.3f, a_kf = %.3f, "
          "d_m = %.3f, d_t = %.3f, d_kf = %.3f, nis = %.3f")   # KFNet/eval.py:113-118


def frame_metrics(i, pair, meas, temp, kf_raw, rec, nis, labels_pair, transform, reset, grid_hw):
    """All fields of one eval.py log line (KFNet/eval.py:77-118).  meas / temp / kf_raw [h,w,4] = the
    raw (coord, sigma) maps the GRAPH computes on this step (measurement, prediction, KF estimate before
    the NIS gate and the reset override); rec [h,w,4] = the record as emitted (T.x_KF or, gated / on a
    reset, T.z; 1/sigma); nis [h,w,3] as the graph computes it; labels_pair = (label[a], label[b])
    full-resolution [H,W,4]; reset = the host re-initialised the filter on this step (eval.py:94-101)."""
    T = transform
    gt = np.stack([resize_nearest(l, grid_hw) for l in labels_pair])        # [2,h,w,4]
    gt_c, gt_m = gt[..., :3], gt[..., 3:4]
    t_meas = apply_transform(meas[None, ..., :3], T)
    t_temp = apply_transform(temp[None, ..., :3], T)
    t_kf = apply_transform(kf_raw[None, ..., :3], T)
    # in-graph losses (KFNet/train.py:252-257): raw graph outputs, transform applied inside the loss
    l_m, a_m = coord_loss_with_uncertainty(t_meas, meas[None, ..., 3:4], gt_c, gt_m)
    l_t, a_t = coord_loss_with_uncertainty(t_temp, temp[None, ..., 3:4], gt_c, gt_m)
    l_kf, a_kf = coord_loss_with_uncertainty(t_kf, kf_raw[None, ..., 3:4], gt_c, gt_m)
    if reset:   # eval.py:98-101 overrides the temp/KF OUTPUTS by the measurement for dist_error
        t_temp = t_meas
    d_m, _ = dist_error(t_meas[0], gt_c[-1], gt_m[-1])
    d_t, _ = dist_error(t_temp[0], gt_c[-1], gt_m[-1])
    d_kf, _ = dist_error(rec[..., :3], gt_c[-1], gt_m[-1])   # emitted coords: gated / reset-overridden
    return dict(i=i, pair=pair, l_m=l_m, l_t=l_t, l_kf=l_kf, a_m=a_m, a_t=a_t, a_kf=a_kf, d_m=d_m, d_t=d_t,
                d_kf=d_kf, nis=get_NIS_measurement(nis))


def format_line(m):
    return FORMAT % (m['i'], m['pair'][0], m['pair'][1], m['l_m'], m['l_t'], m['l_kf'], m['a_m'], m['a_t'],
                     m['a_kf'], m['d_m'], m['d_t'], m['d_kf'], m['nis'])
