"""CPU tests of the eval.py metrics: the numpy oracle against hand-computed values
(what pins it), and the host side of kfnet_amd.KFNet.metrics (label reading, step schedule, line format).
The device reduction itself is checked against the oracle in tests/test_gpu_e2e.py."""
import numpy as np

from oracle import kfnet_metrics_oracle as M
from kfnet_amd.KFNet import metrics as HM


def test_resize_nearest_picks_8y_8x():
    x = np.arange(480 * 640, dtype=np.float32).reshape(480, 640, 1)
    y = M.resize_nearest(x, (60, 80))
    assert y.shape == (60, 80, 1) and y[3, 5, 0] == x[24, 40, 0] and y[59, 79, 0] == x[472, 632, 0]


def test_label_roundtrip(tmp_path):
    a = np.random.default_rng(0).normal(size=(480, 640, 4)).astype(np.float32)
    p = tmp_path / 'l.bin'
    a.tofile(p)                       # README.md:74: labels are numpy .tofile() dumps
    assert np.array_equal(M.read_label(str(p)), a)


def test_coord_loss_known_values():
    pred = np.zeros((1, 2, 2, 3), np.float32)
    unc = np.full((1, 2, 2, 1), np.exp(-1.0), np.float32)       # 3*log(unc) = -3
    gt = np.zeros((2, 2, 2, 3), np.float32)
    gt[1, 0, 0, 0] = 0.1                                         # one pixel 10 cm off in frame 2
    mask = np.ones((2, 2, 2, 1), np.float32)
    mask[0, 1, 1, 0] = 0.0
    loss, acc = M.coord_loss_with_uncertainty(pred, unc, gt, mask)
    # per-pixel NLL: -3 everywhere (capped at <= -2), except the off pixel: -3 + 0.01/(2 e^-2) = -2.963
    off = -3.0 + 0.01 / (2 * np.exp(-2.0))
    valid = 7 + 1.0
    assert np.isclose(loss, (6 * -3.0 + off) / valid, atol=1e-5)
    assert np.isclose(acc, (valid - 1) / valid)                  # the 10 cm pixel exceeds 5 cm
    # cap from above at -2 (KFNet.py:216)
    loss2, _ = M.coord_loss_with_uncertainty(pred, np.ones_like(unc), gt, mask)
    assert np.isclose(loss2, 7 * -2.0 / valid)


def test_nis_band_and_dist_error():
    nis = np.array([0.0, 0.01, 0.5, 1.0, 3.0, -1.0])
    assert np.isclose(M.get_NIS_measurement(nis), 2 / 4.0)       # positives: .01 .5 1 3 -> .5 and 1 in band
    c = np.zeros((2, 2, 3)); g = np.zeros((2, 2, 3)); g[0, 0] = [0.03, 0.04, 0.0]; g[1, 1] = [0.3, 0.4, 0.0]
    m = np.ones((2, 2, 1)); m[1, 1] = 0
    med, dmap = M.dist_error(c, g, m)
    assert np.isclose(med, 5.0) and np.isclose(dmap[0, 0], 5.0) and dmap[1, 1] == 0


def test_log_line_format():
    m = dict(i=3, pair=(2, 3), l_m=-2.5, l_t=-2.4, l_kf=-2.6, a_m=0.5, a_t=0.4, a_kf=0.6, d_m=3.0, d_t=4.0,
             d_kf=2.0, nis=0.7)
    assert M.format_line(m).startswith('3, frame 2~3, l_m = -2.500, l_t = -2.400, l_kf = -2.600, a_m = 0.500')


def test_host_label_grid_equals_tf_nearest_resize(tmp_path):
    a = np.random.default_rng(1).normal(size=(480, 640, 4)).astype(np.float32)
    p = tmp_path / 'l.bin'
    a.tofile(p)
    g = HM.read_label_grid(str(p), (480, 640), (60, 80))
    assert g.shape == (60, 80, 4) and np.array_equal(g, M.resize_nearest(a, (60, 80)))
    assert np.array_equal(HM.resize_nearest(a, (60, 80)), M.resize_nearest(a, (60, 80)))
    assert HM.format_line(dict(i=3, pair=(2, 3), l_m=-2.5, l_t=-2.4, l_kf=-2.6, a_m=0.5, a_t=0.4, a_kf=0.6, d_m=3.0,
                               d_t=4.0, d_kf=2.0, nis=0.7)) == M.format_line(dict(i=3, pair=(2, 3), l_m=-2.5, l_t=-2.4,
                               l_kf=-2.6, a_m=0.5, a_t=0.4, a_kf=0.6, d_m=3.0, d_t=4.0, d_kf=2.0, nis=0.7))


def test_pair_schedule_follows_get_indexes():
    """KFNet/train.py:67-71: [[s+1, s], [s, s+1], ...] per test sequence; 'stairs' sequences are 500 long."""
    ps = HM.pair_schedule(0, 4, 2000, 1000)
    assert ps.tolist() == [[1, 0], [0, 1], [1, 2], [2, 3]]
    assert HM.pair_schedule(998, 4, 2000, 1000).tolist() == [[997, 998], [998, 999], [1001, 1000], [1000, 1001]]
    assert HM.pair_schedule(499, 3, 1000, 500).tolist() == [[498, 499], [501, 500], [500, 501]]
    assert HM.pair_schedule(0, 1, 1, 1000).tolist() == [[0, 0]]
    assert HM.TEST_SEQUENCE_LENGTH['stairs'] == 500 and HM.TEST_SEQUENCE_LENGTH['heads'] == 1000
