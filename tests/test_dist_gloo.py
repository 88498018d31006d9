"""World-size-2 `gloo` test (CPU) of the frame-sharded scan hand-off protocol used on the
GPUs (kfnet_amd/dist.py): rank r receives the Kalman state, scans its chunk, sends it on.
The chunk scan itself is the fp32 numpy oracle here (test infrastructure); what is under
test is chunking, reset-boundary independence and the send/recv pairing -- the sharded
result must be bit-identical to the serial scan."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kfnet_amd.dist import chunk_bounds, needs_state, scan_sharded_host


def _inputs(T, H, W, seed=0):
    rng = np.random.default_rng(seed)
    flow = (rng.normal(size=(T, H, W, 2)) * 1.5).astype(np.float32)
    sig = np.abs(rng.normal(size=(T, H, W, 1)) * 0.05).astype(np.float32)
    meas = rng.normal(size=(T, H, W, 4)).astype(np.float32)
    meas[..., 3] = np.abs(meas[..., 3]) * 0.3 + 0.05
    return flow, sig, meas


def _scan(flow, sig, meas, state, t0, reset_period):
    """Serial fp32 oracle scan of a chunk; returns records, final state."""
    from oracle import kfnet_oracle as O
    T, H, W, _ = flow.shape
    recs = []
    sx, ss = state[None, ..., 0:3].copy(), state[None, ..., 3:4].copy()
    for t in range(T):
        z, sz = meas[t][None, ..., 0:3], meas[t][None, ..., 3:4]
        if reset_period > 0 and (t0 + t) % reset_period == 0:
            sx, ss = z, sz
        else:
            pm = O.get_pixel_map(H, W, np.float32) + flow[t][None]
            tx = O.bilinear_sampler(sx, pm)
            lu = O.bilinear_sampler(ss, pm)
            e2 = np.float32(1e-5) ** 2
            ts = np.sqrt(np.maximum(sig[t][None] ** 2, e2) + np.maximum(lu * lu, e2))
            sx, ss = O.build_kf_coord(tx, ts, z, sz)
        recs.append(np.concatenate([sx[0], 1.0 / ss[0]], -1))
    return np.stack(recs), np.concatenate([sx[0], ss[0]], -1)


def _worker(rank, world, port, T, H, W, reset_period, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flow, sig, meas = _inputs(T, H, W)
    lo, hi = chunk_bounds(T, world, rank)
    state = torch.zeros(H, W, 4)
    recs = {}

    def chunk_fn(buf):
        r, s = _scan(flow[lo:hi], sig[lo:hi], meas[lo:hi], buf.numpy().copy(), lo, reset_period)
        buf.copy_(torch.from_numpy(s))
        recs['r'] = r

    scan_sharded_host(None, chunk_fn, rank, world, dist, state, lo, reset_period)
    np.save(os.path.join(out_dir, 'rec_%d.npy' % rank), recs['r'])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('T,reset_period', [(11, 500), (12, 6), (9, 4)])
def test_sharded_scan_equals_serial(tmp_path, T, reset_period):
    H, W, world = 6, 9, 2
    mp.spawn(_worker, args=(world, _free_port(), T, H, W, reset_period, str(tmp_path)), nprocs=world, join=True)
    flow, sig, meas = _inputs(T, H, W)
    ref, _ = _scan(flow, sig, meas, np.zeros((H, W, 4), np.float32), 0, reset_period)
    got = np.concatenate([np.load(tmp_path / ('rec_%d.npy' % r)) for r in range(world)])
    assert np.array_equal(got, ref)   # deterministic, order-fixed scan: bit exact


def test_chunking_rules():
    assert [chunk_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert chunk_bounds(2048, 8, 3) == (768, 1024)
    assert not needs_state(0, 500) and not needs_state(1000, 500) and needs_state(256, 500)
    assert needs_state(256, 0)
