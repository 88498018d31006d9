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

from kfnet_amd.dist import chunk_bounds, handoff_plan, needs_state, scan_sharded_host


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

    scan_sharded_host(None, chunk_fn, rank, world, dist, state, lo, reset_period, n_frames=hi - lo)
    np.save(os.path.join(out_dir, 'rec_%d.npy' % rank), recs.get('r', np.zeros((0, H, W, 4), np.float32)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('T,reset_period,world', [(11, 500, 2), (12, 6, 2), (9, 4, 2), (3, 500, 4),
                                                  (2048, 500, 8)])   # last = BASELINE config 4's chunking
def test_sharded_scan_equals_serial(tmp_path, T, reset_period, world):
    """(2048, 500, 8): eight contiguous 256-frame chunks, resets at 500/1000/1500/2000 fall INSIDE
    chunks 1, 3, 5 and 7; chunk 4 starts at 1024 (needs the state), no chunk starts on a reset.
    (3, 500, 4): an empty trailing chunk."""
    H, W = 6, 9
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


def test_handoff_pairing_is_symmetric():
    """Every send has exactly one matching recv, for any chunking and reset period; a chunk
    that starts on a reset frame neither receives nor makes its predecessor send."""
    for total, world, period in [(2048, 8, 500), (2000, 4, 500), (12, 2, 6), (3, 4, 500), (1000, 3, 0), (7, 7, 2)]:
        plans = []
        for r in range(world):
            lo, hi = chunk_bounds(total, world, r)
            plans.append((lo, hi, handoff_plan(lo, hi - lo, r, world, period)))
        assert plans[0][2][0] is False and plans[-1][2][1] is False
        for r in range(world - 1):
            assert plans[r][2][1] == plans[r + 1][2][0], (total, world, period, r)
            assert plans[r][1] == plans[r + 1][0]
    # 2000 frames over 4 ranks with period 500: every chunk starts on a reset -> no message at all
    assert all(handoff_plan(500 * r, 500, r, 4, 500) == (False, False) for r in range(4))
    # config 4: all 7 messages are needed
    assert [handoff_plan(256 * r, 256, r, 8, 500) for r in range(8)] == \
        [(False, True)] + [(True, True)] * 6 + [(True, False)]


class _NcclLookalike(object):
    """torch.distributed over gloo that claims to be 'nccl', so that make_link's RCCL branch (and its
    collective fallback) can be driven without GPUs."""

    def __init__(self, d):
        self._d = d
        self.ReduceOp = d.ReduceOp

    def get_backend(self):
        return 'nccl'

    def __getattr__(self, name):
        return getattr(self._d, name)


def _link_worker(rank, world, port, fail_rank, prefer, out_dir, bad_index=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from kfnet_amd import _lib, dist as kd
    if rank == fail_rank and not bad_index:
        def broken():
            raise _lib.KfnError('simulated: librccl cannot be bound on this rank')
        _lib.load = broken
    try:
        # (bad_index: the failing rank's device index is not even an integer -- ADVICE r4: that must fail like any
        #  other local precondition, collectively, not raise TypeError past make_link's except clause)
        link = kd.make_link(_NcclLookalike(dist), rank, world, None if (bad_index and rank == fail_rank) else 0,
                            prefer=prefer)
        name = type(link).__name__
        # the fallback link must be usable at once: one hand-off 0 -> 1
        buf = torch.full((2, 3, 4), float(rank + 1))
        link.via_host = True
        if rank == 0:
            link.send(buf, 1)
        elif rank == 1:
            link.recv(buf, 0)
            assert float(buf[0, 0, 0]) == 1.0
    except _lib.KfnError:
        name = 'raised'
    with open(os.path.join(out_dir, 'link_%d.txt' % rank), 'w') as f:
        f.write(name)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize('fail_rank', [0, 1])
@pytest.mark.parametrize('prefer', ['auto', 'cabi'])
def test_make_link_fallback_is_collective(tmp_path, fail_rank, prefer):
    """ADVICE r2 (dist.py:139): one rank failing to bind RCCL must not leave the others in the id
    broadcast or on a different transport.  Whichever rank fails, BOTH ranks take TorchLink
    ('auto') or BOTH raise ('cabi'), and nobody hangs."""
    mp.spawn(_link_worker, args=(2, _free_port(), fail_rank, prefer, str(tmp_path)), nprocs=2, join=True)
    names = [open(tmp_path / ('link_%d.txt' % r)).read() for r in range(2)]
    assert names == (['TorchLink'] * 2 if prefer == 'auto' else ['raised'] * 2), names


@pytest.mark.timeout(120)
@pytest.mark.parametrize('prefer', ['auto', 'cabi'])
def test_make_link_bad_device_index_fails_collectively(tmp_path, prefer):
    """ADVICE r4 (dist.py:179): a device index that is None / out of range on ONE rank is a local precondition
    failure: the flag reduction still runs on that rank (it falls back to the current device / the CPU), so both
    ranks end on TorchLink ('auto') or both raise KfnError ('cabi') -- nobody hangs, nothing escapes as TypeError."""
    mp.spawn(_link_worker, args=(2, _free_port(), 1, prefer, str(tmp_path), True), nprocs=2, join=True)
    names = [open(tmp_path / ('link_%d.txt' % r)).read() for r in range(2)]
    assert names == (['TorchLink'] * 2 if prefer == 'auto' else ['raised'] * 2), names


# ---- block-cyclic sharding (kfnet_amd/dist.py: cyclic_blocks / cyclic_handoff_plan / scan_cyclic_host) ---------------------------
def _cyclic_worker(rank, world, port, T, H, W, reset_period, block, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from kfnet_amd.dist import scan_cyclic_host
    flow, sig, meas = _inputs(T, H, W)
    state = torch.zeros(H, W, 4)

    def block_fn(buf, lo, hi):
        r, s = _scan(flow[lo:hi], sig[lo:hi], meas[lo:hi], buf.numpy().copy(), lo, reset_period)
        buf.copy_(torch.from_numpy(s))
        return r

    for lo, r in scan_cyclic_host(block_fn, rank, world, dist, state, T, block, reset_period):
        np.save(os.path.join(out_dir, 'blk_%06d.npy' % lo), r)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('T,reset_period,world,block', [(23, 500, 2, 4), (24, 6, 2, 3), (24, 6, 4, 6), (5, 500, 4, 2), (7, 3, 3, 1),
                                                        (2048, 500, 8, 64)])
def test_cyclic_sharded_scan_equals_serial(tmp_path, T, reset_period, world, block):
    """Blocks of `block` frames dealt round-robin over `world` gloo ranks, the state hopping rank -> rank+1 (wrapping from the
    last rank to rank 0) once per block: bit-identical to the serial scan.  (24, 6, 4, 6): every block starts on a reset -> no
    message at all; (24, 6, 2, 3): every second one; (5, 500, 4, 2): fewer blocks than ranks; (2048, 500, 8, 64) = BASELINE
    config 4's sequence in 32 blocks, four revolutions of the ring, resets inside blocks 7, 15, 23 and 31."""
    H, W = 6, 9
    mp.spawn(_cyclic_worker, args=(world, _free_port(), T, H, W, reset_period, block, str(tmp_path)), nprocs=world, join=True)
    flow, sig, meas = _inputs(T, H, W)
    ref, _ = _scan(flow, sig, meas, np.zeros((H, W, 4), np.float32), 0, reset_period)
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith('blk_'))
    assert len(files) == -(-T // block)
    got = np.concatenate([np.load(tmp_path / f) for f in files])
    assert np.array_equal(got, ref)


def test_cyclic_plan_pairs_every_send_with_one_recv():
    from kfnet_amd.dist import cyclic_blocks, cyclic_handoff_plan
    for total, world, block, period in [(2048, 8, 32, 500), (2048, 8, 64, 500), (100, 3, 7, 10), (9, 4, 2, 0), (50, 2, 25, 25), (5, 8, 1, 500)]:
        owned = {}
        for r in range(world):
            for j, lo, hi in cyclic_blocks(total, block, r, world):
                assert j % world == r and j not in owned
                owned[j] = (r, lo, hi)
        assert sorted(owned) == list(range(-(-total // block)))
        assert [owned[j][1] for j in sorted(owned)] == [j * block for j in sorted(owned)] and owned[max(owned)][2] == total
        for j in sorted(owned):
            r, lo, hi = owned[j]
            src, dst = cyclic_handoff_plan(lo, hi, total, r, world, period)
            if j == 0:
                assert src is None
            if dst is not None:                      # the next block exists, is owned by dst, and expects the state from r
                r2, lo2, hi2 = owned[j + 1]
                assert r2 == dst and lo2 == hi and cyclic_handoff_plan(lo2, hi2, total, r2, world, period)[0] == r
            elif j + 1 in owned:                     # no send: the next block must not wait for one
                r2, lo2, hi2 = owned[j + 1]
                assert cyclic_handoff_plan(lo2, hi2, total, r2, world, period)[0] is None
            if world == 1:
                assert src is None and dst is None
    # 50 frames, 2 ranks, blocks of 25, resets every 25: both blocks start on a reset -> nothing travels
    assert cyclic_handoff_plan(0, 25, 50, 0, 2, 25) == (None, None) and cyclic_handoff_plan(25, 50, 50, 1, 2, 25) == (None, None)
