"""-m gpu end-to-end parity: the full KFNet sequence engine (HIP) against the CPU oracle.

Tolerance (BASELINE.json north_star): max-abs <= 1e-4 on the scene-coordinate channels;
the confidence channel 1/sigma is O(1..1e3), so its tolerance is RELATIVE 1e-4
(SURVEY.md §7 'Tolerance on channel 3')."""
import numpy as np
import pytest

from oracle import kfnet_oracle as O
from oracle import kfnet_oracle_torch as OT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


COORD_TOL = 1e-4
CONF_RTOL = 1e-4


# config 5's tolerance statement (test_config5_tolerance_at_bench_scale; numbers in profiles/r03_bench_c5.json)
C5_DELTA_PX = 0.05                 # flows agree / crossings lie within this distance of a sampler step
C5_MASKED_FRACTION_BOUND = 1e-2    # descendants of step crossings (measured 2.5e-3 over 4 x 16 frames)
C5_OUTSIDE_FRACTION_BOUND = 2e-4   # pixel-frames actually outside the tolerance (measured 8.6e-5)


def _check(rec, ref):
    assert rec.shape == ref.shape
    dc = np.abs(rec[..., 0:3] - ref[..., 0:3]).max()
    dr = (np.abs(rec[..., 3] - ref[..., 3]) / np.abs(ref[..., 3])).max()
    assert dc <= COORD_TOL, 'coord max-abs %g' % dc
    assert dr <= CONF_RTOL, 'confidence max-rel %g' % dr
    return dc, dr


@pytest.mark.parametrize('batch', [1, 2, 4])
def test_small_sequence_vs_fp64_oracle(batch):
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    imgs = synthetic_sequence(7, 64, 96, seed=1)
    T4 = O.get_transform(synthetic_transform())
    ref, dbg = O.eval_sequence(imgs, W, T4, reset_period=5, dtype=np.float64, return_debug=True)
    eng = KFNetEngine(W, image_size=(64, 96), batch=batch, transform=T4, reset_period=5, max_chunk=16,
                      emit_debug=True)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    d = eng.debug(7)
    # stage-level diagnostics first (clearer failures than the final record)
    z = np.concatenate([np.concatenate([x['z'][0], x['sz'][0]], -1)[None] for x in dbg])
    assert np.abs(d['meas'] - z).max() < 5e-5
    for t in (1, 2, 3, 4, 6):
        assert np.abs(d['flow'][t] - dbg[t]['flow'][0]).max() < 5e-5
        assert np.abs(d['sigma_trans'][t].reshape(-1) - dbg[t]['sigma_trans'].reshape(-1)).max() < 1e-6
    _check(rec, ref)


@pytest.mark.parametrize('options', [dict(factor_cost_volume=False), dict(fuse_oflow_window=False),
                                     dict(fuse_cost_volume=False), dict(fuse_oflow_window=False, fuse_oflow_tail=False)])
def test_unfused_oflow_routes_vs_fp64_oracle(options):
    """The routes the default graph does not take (conv0 had become a Winograd layer whose re-packed kernel the
    loader-generated cost-volume route read as a direct-conv matrix -- silently wrong flow, no test): the cost volume
    generated in conv0's loader (factor_cost_volume=False), the round-2 launches instead of the window-resident ends
    (fuse_oflow_window=False), the materialised volume (fuse_cost_volume=False)."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    imgs = synthetic_sequence(4, 64, 96, seed=1)
    T4 = O.get_transform(synthetic_transform())
    ref, dbg = O.eval_sequence(imgs, W, T4, reset_period=500, dtype=np.float64, return_debug=True)
    eng = KFNetEngine(W, image_size=(64, 96), batch=2, transform=T4, reset_period=500, max_chunk=8, emit_debug=True,
                      graph_options=options)
    names = [type(op).__name__ for op in eng.net.pair_ops]
    if options.get('factor_cost_volume') is False:
        assert 'CostVolumeConvOp' in names and 'OFlowHeadOp' not in names
    if options.get('fuse_oflow_window') is False:
        assert 'OFlowHeadOp' not in names and 'OFlowTail2Op' not in names
    if options.get('fuse_cost_volume') is False:
        assert 'CostVolumeOp' in names
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    d = eng.debug(4)
    for t in (1, 2, 3):
        assert np.abs(d['flow'][t] - dbg[t]['flow'][0]).max() < 5e-5, (options, t)
    _check(rec, ref)


def test_nis_gate_and_odd_grid():
    """540x960-style odd intermediate sizes (SAME pad (1,1) on odd rows) at reduced scale:
    68x120 image -> 34x60 -> 17x30 -> 9x15 grid; plus the --NIS output gate."""
    from kfnet_amd.engine import KFNetEngine, grid_size
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    assert grid_size((540, 960)) == (68, 120) and grid_size((68, 120)) == (9, 15)
    W = synthetic_weights(99)
    imgs = synthetic_sequence(4, 68, 120, seed=5)
    T4 = np.eye(4, dtype=np.float32)
    ref = O.eval_sequence(imgs, W, T4, reset_period=500, nis_gate=True, dtype=np.float64)
    eng = KFNetEngine(W, image_size=(68, 120), batch=2, transform=T4, reset_period=500, nis_gate=7.815,
                      max_chunk=8)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    _check(rec, ref)


# (full-size 480x640 parity: tests/test_golden.py::test_hip_engine_reproduces_full_size_golden compares
#  against the committed fp64-oracle records of tests/golden/kfnet_full.npz instead of recomputing them)


def test_channel_blocked_activations_change_nothing_but_addresses():
    """Graph.activation_layout_c16 (round 6): at batch 8 every tensor from conv1b's output to conv5's lives channel-blocked
    (KFN_LAYOUT_C16) between the Winograd launches -- the records of an 8-frame 480x640 sequence are BIT-IDENTICAL to the all-NHWC
    graph's (same products in the same order; only addresses move), an intermediate tensor read back through Tensor.numpy() is the
    same NHWC array in both, and a small graph (64x96, where neither Winograd form is routed) blocks nothing."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(11)
    imgs = synthetic_sequence(8, 480, 640, seed=4)
    outs, mids = [], []
    for c16 in (True, False):
        eng = KFNetEngine(W, image_size=(480, 640), batch=8, reset_period=500, max_chunk=8, graph_options=dict(activation_layout_c16=c16))
        sc = eng.net.scoordnet
        blocked = sorted(n for n in ('conv1a', 'conv1b', 'conv2a', 'conv2b', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5', 'conv6')
                         if sc.get_output_by_name(n).layout == 'c16')
        assert blocked == (['conv1b', 'conv2a', 'conv2b', 'conv3a', 'conv3b', 'conv4a', 'conv4b', 'conv5'] if c16 else []), blocked
        outs.append(eng.process(eng.upload_frames(imgs)).cpu().numpy().copy())
        mids.append(sc.get_output_by_name('conv3a').numpy().copy())
        del eng
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(mids[0], mids[1]) and np.abs(mids[0]).max() > 0
    small = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=500, max_chunk=4)
    assert all(t.layout == 'nhwc' for t in small.net.scoordnet.layers.values() if hasattr(t, 'layout'))


def test_chunked_equals_single_pass():
    """Processing a sequence as two chunks with state/feature hand-over (what two ranks do)
    is bit-identical to one pass."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(7)
    imgs = synthetic_sequence(8, 64, 96, seed=2)
    eng = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=500, max_chunk=8)
    dev = eng.upload_frames(imgs)
    one = eng.process(dev).cpu().numpy().copy()
    eng2 = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=500, max_chunk=8)
    a = eng2.process(dev[:3], t0=0).cpu().numpy().copy()
    state = eng2.get_state().clone()
    eng3 = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=500, max_chunk=8)
    eng3.prime(dev[2])
    eng3.get_state().copy_(state)
    b = eng3.process(dev[3:], t0=3).cpu().numpy().copy()
    assert np.array_equal(np.concatenate([a, b]), one)


def test_two_rank_bench_on_one_gpu(tmp_path):
    """bench.py's multi-rank path (frame sharding, priming, state hand-off, barriers, max-over-
    ranks timing) with 2 ranks sharing this GPU over gloo; the RCCL transport itself needs
    one GPU per rank and is exercised by the driver's multi-GPU run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KFN_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(root, 'bench.py'),
           '--gpus', '2', '--steps', '6', '--warmup', '2', '--batch', '2', '--height', '64', '--width', '96',
           '--no-kalman-roofline']
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['frames_total'] == 12 and out['value'] > 0


def test_bench_gpus_2_launches_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` -- the plain form, no torch.distributed.run around it -- must start
    two ranks itself.  On a one-GPU box the ranks share the GPU over gloo and the
    line says so; on a multi-GPU node the same command hands the state over RCCL (rccl_ranks)."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'KFN_DIST_BACKEND')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
           '--batch', '2', '--height', '64', '--width', '96', '--no-kalman-roofline', '--min-seconds', '0.2', '--block', '2',
           '--detail', str(tmp_path / 'detail.json')]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0)'
    assert lines[0] == r.stdout.strip().splitlines()[-1] and len(lines[0]) < 4096      # the driver's line: last, compact
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['handoff']['scan_chain_ms'] > 0 and line['sharding_cyclic']['block'] == 2
    out = json.load(open(tmp_path / 'detail.json'))                 # the full result (sidecar file)
    assert out['value'] == line['value']
    assert out['n_gpus'] == 2 and out['self_launched'] is True
    assert out['config']['frames_total'] == 12 and out['value'] > 0
    assert len(out['rccl_ranks']) == 2 and len(out['rank_devices']) == 2
    # the serial chain of the sharded configuration, measured: two scans and one hand-off (Next #4c)
    h = out['handoff']
    assert len(h['scan_ms_per_rank']) == 2 and h['scan_chain_ms'] >= max(h['scan_ms_per_rank'])
    assert h['handoff_us'][0]['send_us'] is not None and h['handoff_us'][1]['recv_wait_us'] is not None
    assert h['handoff_us'][0]['recv_wait_us'] is None and h['handoff_us'][1]['send_us'] is None
    # ... and the block-cyclic sharding of the same 12-frame job measured beside it (Next #9): --block 2 -> six
    # blocks over two ranks, the state hopping five times
    assert out['sharding'] == 'contiguous' and h['sharding'] == 'contiguous' and h['tail_ms'] > 0
    cy = out['sharding_cyclic']
    assert cy['sharding'] == 'block-cyclic' and cy['block'] == 2 and cy['blocks_total'] == 6 and cy['value'] > 0
    assert len(cy['recv_wait_ms_per_rank']) == 2 and cy['tail_ms'] is not None
    if torch.cuda.device_count() >= 2:
        assert out['dist_backend'] == 'nccl' and out['rank_devices'] == [0, 1]
        assert out['rccl_ranks'] == [[0, 2], [1, 2]], out['rccl_ranks']   # kfn_comm_rank on every rank
        assert 'RCCL' in out['state_link']
    else:
        assert out['dist_backend'] == 'gloo' and out['rccl_ranks'] == [None, None]
        assert 'FUNCTIONAL FALLBACK' in out['config']['parallelism']


def test_bench_gpus_8_carries_the_config4_block(tmp_path):
    """The driver's 8-rank command with fewer than 256 steps also runs BASELINE configs[3] -- 8 x 256 = 2048 frames, state
    handed rank -> rank -- and reports it with the measured chain (Next #4b).  Here the 8 ranks share the
    box's GPU(s) (gloo fallback when there are fewer than 8) on small frames; the arithmetic path is the driver's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'KFN_DIST_BACKEND')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '2',
           '--batch', '2', '--height', '64', '--width', '96', '--no-kalman-roofline', '--min-seconds', '0.2',
           '--detail', str(tmp_path / 'detail.json')]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    out = json.load(open(tmp_path / 'detail.json'))
    assert out['n_gpus'] == 8 and out['config']['frames_total'] == 32
    c4 = out['config4_2048_frames']
    assert c4['frames_total'] == 2048 and c4['frames_per_rank'] == 256 and c4['value'] > 0
    assert c4['is_literal_config4'] is False and c4['image'] == '64x96'      # (480x640 on the driver's node)
    h = c4['handoff']
    assert len(h['scan_ms_per_rank']) == 8 and h['scan_chain_ms'] >= sum(h['scan_ms_per_rank']) * 0.5
    # chunks [256 r, 256 r + 256): frames 500, 1000, 1500, 2000 are inside chunks 1, 3, 5, 7 -- every boundary hands over
    assert all(x['recv_wait_us'] is not None for x in h['handoff_us'][1:]) and h['handoff_us'][0]['recv_wait_us'] is None
    assert line['config4_2048_frames']['value'] == c4['value'] and line['n_gpus'] == 8


def test_config5_shape_batch_of_sequences():
    """BASELINE config 5 geometry in fp32: 540x960 frames -> 68x120 grid (odd 135-row level,
    SAME pad (1,1) there; state sized by ceil), two independent sequences advanced
    by one batched scan launch; against the torch-CPU oracle."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    seqs = np.stack([synthetic_sequence(2, 540, 960, seed=3), synthetic_sequence(2, 540, 960, seed=4)])
    T4 = np.eye(4, dtype=np.float32)
    eng = KFNetEngine(W, image_size=(540, 960), batch=2, transform=T4, reset_period=500, max_chunk=4)
    import torch
    rec = eng.process_sequences(torch.from_numpy(seqs).cuda()).cpu().numpy()
    assert rec.shape == (2, 2, 68, 120, 4)
    for s in range(2):
        ref = OT.eval_sequence(seqs[s], W, T4, reset_period=500)
        _check(rec[s], ref)


def test_eval_cli_writes_npy(tmp_path):
    """kfnet_amd.KFNet.eval keeps the reference's flags and coord_<i>.npy output contract
    (KFNet/eval.py:121-126): float32 [60,80,4] = (T.x, 1/sigma)."""
    from kfnet_amd.KFNet import eval as kf_eval
    out = tmp_path / 'out'
    out.mkdir()
    rc = kf_eval.main(['--scene', 'heads', '--output_folder', str(out), '--synthetic', '3', '--random_weights',
                       '--batch', '2'])
    assert rc == 0
    files = sorted(p.name for p in out.iterdir())
    assert files == ['coord_0.npy', 'coord_1.npy', 'coord_2.npy']
    a = np.load(out / 'coord_1.npy')
    assert a.shape == (60, 80, 4) and a.dtype == np.float32 and np.all(np.isfinite(a)) and np.all(a[..., 3] > 0)
    assert kf_eval.main(['--scene', 'nowhere']) == 1   # KFNet/train.py:142-144: invalid scene


def test_long_recursion_error_does_not_grow():
    """48 frames without a reset: the recurrent state feeds back 47 times, so any systematic
    fp32 drift of the HIP warp/fuse chain against the fp64 oracle would accumulate here."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(4321)
    imgs = synthetic_sequence(48, 64, 96, seed=9)
    T4 = O.get_transform(synthetic_transform())
    ref = O.eval_sequence(imgs, W, T4, reset_period=500, dtype=np.float64)
    eng = KFNetEngine(W, image_size=(64, 96), batch=5, transform=T4, reset_period=500, max_chunk=48)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    _check(rec, ref)
    d_first = np.abs(rec[1:9, ..., :3] - ref[1:9, ..., :3]).max()
    d_last = np.abs(rec[-8:, ..., :3] - ref[-8:, ..., :3]).max()
    assert d_last < 10 * max(d_first, 1e-7)


def test_eval_with_labels_prints_reference_metrics(capsys):
    """eval() with label maps: the log-line fields come from kfn_eval_metrics on the device (sums / counts)
    + host medians, through the STREAMED pipeline; every field against the numpy oracle evaluated on the
    scan's own debug buffers -- including a reset step inside the run (graph-view losses, host-view
    distances), the NIS gate, and the (s+1, s) pair of a sequence start."""
    from kfnet_amd.KFNet import eval as kf_eval
    from kfnet_amd.KFNet import metrics as HM
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    from oracle import kfnet_metrics_oracle as MO
    rng = np.random.default_rng(5)
    n, size, grid = 7, (64, 96), (8, 12)
    frames = synthetic_sequence(n, 64, 96, seed=6)
    labels = (rng.normal(size=(n, 64, 96, 4)) * 0.05).astype(np.float32)
    labels[..., 3] = (rng.random(size=(n, 64, 96)) > 0.3).astype(np.float32)
    T4 = np.linalg.inv(synthetic_transform())
    W = synthetic_weights(3)
    for nis in (False, True):
        rec, mets = kf_eval.eval(None, T4, W, None, nis=nis, image_size=size, batch=2, frames=frames, labels=labels,
                                 chunk=3, sequence_length=4, metrics_sequence_length=4)
        assert len(mets) == n and mets[0]['pair'] == (1, 0) and mets[2]['pair'] == (1, 2) and mets[4]['pair'] == (5, 4)
        out = capsys.readouterr().out
        assert '2, frame 1~2, l_m = ' in out and 'Median dist error:' in out
        # oracle on the same intermediate maps (one resident pass with the debug outputs)
        eng = KFNetEngine(W, image_size=size, batch=2, transform=T4, reset_period=4, nis_gate=7.815 if nis else 0.0,
                          max_chunk=n, emit_metrics=True)
        rec2 = eng.process(eng.upload_frames(frames)).cpu().numpy()
        assert np.array_equal(rec2, rec)          # chunked + streamed == resident
        d = eng.debug(n)
        kf_raw = eng.c_kf.root_storage.buf[:n * eng.hw * 4].view(n, 8, 12, 4).cpu().numpy()
        for i in range(n):
            a, b = mets[i]['pair']
            ref = MO.frame_metrics(i, (a, b), d['meas'][i], d['temp'][i], kf_raw[i], rec[i], d['nis'][i],
                                   (labels[a], labels[b]), T4, i % 4 == 0, grid)
            for key in ('l_m', 'l_t', 'l_kf', 'a_m', 'a_t', 'a_kf', 'nis'):
                assert np.isclose(mets[i][key], ref[key], rtol=2e-5, atol=1e-6), (i, key, mets[i][key], ref[key])
            for key in ('d_m', 'd_t', 'd_kf'):
                assert np.isclose(mets[i][key], ref[key], rtol=1e-4), (i, key, mets[i][key], ref[key])
        assert mets[4]['d_t'] == mets[4]['d_m']      # reset step: temp output := measurement (eval.py:98)
        assert mets[4]['l_t'] != mets[4]['l_m']      # ... but the losses see the graph's prediction
    assert HM.format_line(mets[1]).startswith('1, frame 0~1, l_m = ')


def test_config5_fp16_convs_fp32_kalman():
    """BASELINE config 5: fp16-operand convolutions (fp32 accumulate) + fp32 Kalman, at the
    540x960 geometry.  Own tolerance (stated here): coord max-abs <= 2e-2, confidence
    max-rel <= 5e-2 against the fp32 oracle -- this is NOT the headline parity path."""
    import torch
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.graph import ConvOp
    from kfnet_amd import _lib
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    seq = synthetic_sequence(3, 540, 960, seed=3)
    T4 = np.eye(4, dtype=np.float32)
    eng = KFNetEngine(W, image_size=(540, 960), batch=3, transform=T4, reset_period=500, max_chunk=3,
                      conv_operands='f16')
    n16 = sum(1 for op in eng.heavy_ops if isinstance(op, ConvOp) and op.operand_dtype == _lib.OPERAND_F16)
    assert n16 >= 25
    rec = eng.process(eng.upload_frames(seq)).cpu().numpy()
    ref = OT.eval_sequence(seq, W, T4, reset_period=500)
    dc = np.abs(rec[..., :3] - ref[..., :3]).max()
    dr = (np.abs(rec[..., 3] - ref[..., 3]) / np.abs(ref[..., 3])).max()
    print('fp16-operand convs: coord max-abs %.3g, confidence max-rel %.3g' % (dc, dr))
    assert dc <= 2e-2 and dr <= 5e-2
    assert dc > 1e-6   # and it is measurably not the fp32 path


def test_config5_tolerance_at_bench_scale():
    """Config 5's tolerance as it is actually true, at the scale bench.py --config c5
    reports: 4 sequences x 16 frames of 540x960, fp16 path (fp16 MFMA operands, fp16 activations in
    SCoordNet) against the fp32 HIP path (which the tests above hold to the oracle at 1e-4).

    The reference sampler (tools/util.py:36-93) is a step function of the flow at x in {0, W-1} /
    y in {0, H-1}: a sample just outside evaluates to 0, just inside to the border value.  The two
    paths' flows agree to ~3e-3 px, so now and then (3 times in 522 240 pixel-frames here) a sample
    within 1e-3 px of a step lands on different sides in the two paths; that pixel then differs by
    the whole state value, the recurrent state keeps the difference, and pixels that later sample
    it inherit a share.  Statement, all four parts asserted:
      1. the flows agree to better than delta = 0.05 px and every crossing lies within delta of a step;
      2. every pixel that is NOT a descendant of a crossing (kfnet_amd/tools/parity.py propagates
         the taint exactly as the state propagates) meets coord max-abs <= 2e-2, confidence
         max-rel <= 5e-2 -- no deviation is left unexplained by a crossing;
      3. the descendants are <= 1 % of the pixel-frames (measured 0.25 %); a sound mask cannot be
         smaller, the bilinear sampler spreads a state over up to 4 pixels per frame until the reset;
      4. the pixel-frames actually outside the tolerance are <= 2e-4 of all (measured 8.6e-5)."""
    import torch
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.tools.parity import masked_parity, merge_parity
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    S, T = 4, 16
    seqs = np.stack([synthetic_sequence(T, 540, 960, seed=3 + s) for s in range(S)])
    dev = torch.from_numpy(seqs).cuda()
    T4 = np.eye(4, dtype=np.float32)
    recs, flows = {}, {}
    for mode in ('f32', 'f16'):
        eng = KFNetEngine(W, image_size=(540, 960), batch=8, transform=T4, reset_period=500, max_chunk=S * T,
                          conv_operands=mode)
        recs[mode] = eng.process_sequences(dev).cpu().numpy().copy()
        flows[mode] = eng.debug(S * T)['flow'].reshape(S, T, eng.h, eng.w, 2).copy()
        if mode == 'f16':
            assert any(getattr(op, 'y', None) is not None and op.y.dtype == 'f16' for op in eng.heavy_ops), \
                'config 5 must keep SCoordNet activations in fp16'
        del eng
        torch.cuda.empty_cache()
    mp = merge_parity([masked_parity(recs['f16'][s], recs['f32'][s], flows['f32'][s], coord_tol=2e-2,
                                     conf_rel_tol=5e-2, delta=C5_DELTA_PX, reset_period=500, test_flow=flows['f16'][s])
                       for s in range(S)])
    print('config 5 at bench scale:', mp)
    assert mp['pixels'] == S * T * 68 * 120
    assert mp['flow_max_abs_diff_px'] < C5_DELTA_PX and mp['crossing_max_step_distance_px'] < C5_DELTA_PX, mp
    assert mp['unmasked_outside_tolerance'] == 0, mp
    assert mp['masked_fraction'] <= C5_MASKED_FRACTION_BOUND, mp
    assert mp['outside_tolerance_fraction'] <= C5_OUTSIDE_FRACTION_BOUND, mp
    assert mp['all_pixels_coord_max_abs'] > 1e-6     # and it is measurably not the fp32 path


def test_get_kf_coord2_and_temporal_coord2_api():
    """The reference's optional API (KFNet/KFNet.py:476-502) at the graph level, one frame per run as the reference
    evaluates it: GetTemporalCoord2 = the process-model prediction of the step, GetKFCoord2 = its fusion with the
    measurement in the symmetric-variance form -- bit exact vs the fp32 oracle applied to the graph's own prediction /
    measurement buffers, and the recursive state itself still follows BuildKFCoord."""
    from kfnet_amd.graph import Graph
    from kfnet_amd.KFNet.KFNet import KFNet, KFNetDataSpec
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    imgs = synthetic_sequence(3, 64, 96, seed=2)
    g = Graph()
    images = g.placeholder((1, 64, 96, 3), 'u8', name='images')
    state = g.placeholder((1, 8, 12, 4), name='last_state')
    net = KFNet(images, KFNetDataSpec(batch_size=1, image_size=(64, 96)))
    net.GetKFCoordRecursive(state.channels(0, 3), state.channels(3, 1), reset_period=500, emit_temp=True)
    t_coord, t_unc = net.GetTemporalCoord2()
    kf2_coord, kf2_unc = net.GetKFCoord2()
    g.finalize('cuda:0')
    g.load_weights(W)
    for t in range(3):
        images.upload(imgs[t:t + 1])
        net._kalman.t0 = t
        g.run()
    temp, meas, st = net.temp.numpy(), net.GetMeasureCoord()[0].base.numpy(), state.numpy()
    got = kf2_coord.base.numpy()
    assert temp[..., 3].min() > 0 and np.isfinite(got).all()
    rc, ru = O.get_kf_coord2(temp[..., :3], temp[..., 3:4], meas[..., :3], meas[..., 3:4])
    assert np.array_equal(got[..., :3], rc) and np.array_equal(got[..., 3:4], ru)
    r1c, r1u = O.build_kf_coord(temp[..., :3], temp[..., 3:4], meas[..., :3], meas[..., 3:4])
    assert np.allclose(st[..., :3], r1c, atol=1e-6) and np.allclose(st[..., 3:4], r1u, rtol=1e-5)
    assert t_coord.shape == (1, 8, 12, 3) and t_unc.shape == (1, 8, 12, 1) and kf2_unc.shape == (1, 8, 12, 1)


def test_config5_batch_independence_at_full_size():
    """Size-independent property at config 5's full geometry (540x960, fp16 path): no layer couples batch elements, so
    a sequence's records do not depend on which other sequences share the launch, nor on the tower batch size --
    bit-identical (the fp16-activation kernels round per element; tile membership must not matter)."""
    import torch
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    seqs = np.stack([synthetic_sequence(5, 540, 960, seed=11 + s) for s in range(2)])
    dev = torch.from_numpy(seqs).cuda()
    T4 = np.eye(4, dtype=np.float32)
    eng = KFNetEngine(W, image_size=(540, 960), batch=5, transform=T4, reset_period=500, max_chunk=10, conv_operands='f16')
    both = eng.process_sequences(dev).cpu().numpy().copy()
    del eng
    torch.cuda.empty_cache()
    eng = KFNetEngine(W, image_size=(540, 960), batch=2, transform=T4, reset_period=500, max_chunk=5, conv_operands='f16')
    for s in range(2):
        alone = eng.process_sequences(dev[s:s + 1]).cpu().numpy()
        assert np.array_equal(alone[0], both[s]), 'sequence %d depends on its batch neighbours' % s


@pytest.mark.parametrize('cfg', [('f32', (480, 640), 8), ('f16', (540, 960), 4)])
def test_records_are_reproducible_under_memory_load(cfg):
    """The whole path, fp32 (config 3's geometry) and fp16 (config 5's), run again and again while another stream keeps the
    memory system busy: every pass bit-identical to an unloaded one.  (What round 5's store hazard broke for config 5 --
    DESIGN 3.5 -- checked end to end: two towers on two streams plus the copy stream, all kernels of the step.)"""
    import torch
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    operands, size, frames = cfg
    W = synthetic_weights(1234)
    dev = torch.from_numpy(synthetic_sequence(frames, size[0], size[1], seed=5)).cuda()
    eng = KFNetEngine(W, image_size=size, batch=min(frames, 4), transform=np.eye(4, dtype=np.float32), reset_period=500,
                      max_chunk=frames, conv_operands=operands)
    side = torch.cuda.Stream()
    big_a = torch.randn(64 << 20, device='cuda')
    big_b = torch.empty_like(big_a)

    def run(load):
        torch.cuda.synchronize()      # (t0 = 0 is a reset frame: a pass does not depend on the state the previous one left)
        if load:
            with torch.cuda.stream(side):
                for _ in range(8):
                    big_b.copy_(big_a)
        rec = eng.process(dev, t0=0).clone()
        torch.cuda.synchronize()
        return rec

    ref = run(False)
    bad = [r for r in range(12) if not torch.equal(run(True), ref)]
    assert not bad, 'passes %s differ from the unloaded one' % bad


@pytest.mark.parametrize('size,batch', [((64, 96), 2), ((480, 640), 3)])
def test_f16x3_split_mode_meets_fp32_tolerance(size, batch):
    """conv_operands='f16x3' (operands split into hi+lo halfs, 3 fp16 MFMA products, fp32
    accumulate) must meet the SAME tolerance as the fp32 path: coord max-abs <= 1e-4,
    confidence max-rel <= 1e-4."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.graph import ConvOp
    from kfnet_amd import _lib
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    imgs = synthetic_sequence(4, size[0], size[1], seed=1)
    T4 = O.get_transform(synthetic_transform())
    if size[0] <= 64:
        ref = O.eval_sequence(imgs, W, T4, reset_period=500, dtype=np.float64)
    else:
        ref = OT.eval_sequence(imgs, W, T4, reset_period=500)
    eng = KFNetEngine(W, image_size=size, batch=batch, transform=T4, reset_period=500, max_chunk=4,
                      conv_operands='f16x3')
    assert sum(1 for op in eng.heavy_ops if isinstance(op, ConvOp) and op.operand_dtype == _lib.OPERAND_F16X3) >= 12
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    dc, dr = _check(rec, ref)
    print('f16x3 %s: coord max-abs %.3g, confidence max-rel %.3g' % (size, dc, dr))


def test_edge_cases_empty_single_and_ragged_chunks():
    """Empty chunk, single frame, T < batch, and ragged chunking all agree with one pass."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(11)
    imgs = synthetic_sequence(7, 64, 96, seed=12)
    eng = KFNetEngine(W, image_size=(64, 96), batch=4, reset_period=500, max_chunk=8)
    dev = eng.upload_frames(imgs)
    one = eng.process(dev).cpu().numpy().copy()
    eng2 = KFNetEngine(W, image_size=(64, 96), batch=4, reset_period=500, max_chunk=8)
    assert tuple(eng2.process(dev[:0]).shape) == (0, 8, 12, 4)
    parts = []
    for lo, hi in ((0, 1), (1, 1), (1, 3), (3, 7)):       # single frame, empty, T < batch, rest
        parts.append(eng2.process(dev[lo:hi], t0=lo).cpu().numpy().copy())
    assert np.array_equal(np.concatenate(parts), one)
    with pytest.raises(ValueError):
        eng2.process(eng2.upload_frames(synthetic_sequence(9, 64, 96)))   # exceeds max_chunk


def test_hipgraph_replay_of_heavy_phase_is_bit_identical():
    """The C ABI claims every launch is hipGraph-capturable: capture the two-stream heavy phase
    of a batch once, replay it for every batch, and get bit-identical records."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(21)
    imgs = synthetic_sequence(9, 64, 96, seed=22)
    a = KFNetEngine(W, image_size=(64, 96), batch=3, reset_period=500, max_chunk=9)
    ref = a.process(a.upload_frames(imgs)).cpu().numpy().copy()
    b = KFNetEngine(W, image_size=(64, 96), batch=3, reset_period=500, max_chunk=9, use_graph=True)
    got = b.process(b.upload_frames(imgs)).cpu().numpy().copy()
    assert b._graph is not None
    assert np.array_equal(got, ref)


def test_streamed_pipeline_is_bit_identical_to_resident(tmp_path):
    """PNG list -> ChunkLoader (decode thread, pinned staging) -> StreamedSequence (upload /
    compute / download on three streams, ragged last chunk) gives exactly the records of one
    resident pass; eval() over image paths writes them as coord_<i>.npy."""
    from PIL import Image
    from kfnet_amd.KFNet import eval as kf_eval
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.pipeline import ChunkLoader, StreamedSequence
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(7)
    imgs = synthetic_sequence(13, 64, 96, seed=4)
    paths = []
    for i in range(13):
        paths.append(str(tmp_path / ('im%02d.png' % i)))
        Image.fromarray(imgs[i]).save(paths[-1])
    eng = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=5, max_chunk=13)
    ref = eng.process(eng.upload_frames(imgs)).cpu().numpy().copy()
    for source, depth in ((paths, 3), (imgs, 2), (imgs, 4)):      # chunks in flight on the GPU (round 5: 3 by default)
        eng2 = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=5, max_chunk=4)
        got = [(lo, rec.copy()) for lo, rec in StreamedSequence(eng2, 4, depth=depth).run(
            ChunkLoader(source, (64, 96), 4, workers=3, depth=depth + 1))]
        assert [lo for lo, _ in got] == [0, 4, 8, 12]
        assert np.array_equal(np.concatenate([r for _, r in got]), ref)
    # a short first chunk (what eval() asks for) and a ragged tail
    eng3 = KFNetEngine(W, image_size=(64, 96), batch=2, reset_period=5, max_chunk=4)
    got = [(lo, rec.copy()) for lo, rec in StreamedSequence(eng3, 4).run(ChunkLoader(paths, (64, 96), 4, workers=2, first_chunk=2))]
    assert [lo for lo, _ in got] == [0, 2, 6, 10] and np.array_equal(np.concatenate([r for _, r in got]), ref)
    out = tmp_path / 'out'
    out.mkdir()
    rec = kf_eval.eval(paths, None, W, str(out), image_size=(64, 96), batch=2, sequence_length=5, chunk=4,
                       verbose=False)
    assert np.array_equal(rec, ref)
    assert np.array_equal(np.load(out / 'coord_12.npy'), ref[12])


def test_large_grid_uses_global_memory_scan():
    """808x968 frames give a 101x121 grid (12 221 px): the Kalman state no longer fits the
    LDS, so the scan runs frame by frame with the state in global memory -- same results."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(1234)
    imgs = synthetic_sequence(3, 808, 968, seed=6)
    T4 = O.get_transform(synthetic_transform())
    ref = OT.eval_sequence(imgs, W, T4, reset_period=500)
    eng = KFNetEngine(W, image_size=(808, 968), batch=1, transform=T4, reset_period=500, max_chunk=3)
    rec = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    assert rec.shape == (3, 101, 121, 4)
    _check(rec, ref)
