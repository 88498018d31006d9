"""-m gpu: BASELINE config 4 ("8-GPU frame-sharded 2048-frame seq, Kalman state passed
rank->rank") exercised functionally on ONE GPU.

The path under test is the product's: kfnet_amd.KFNet.eval under torch.distributed.run
(8 processes, contiguous 256-frame chunks, resets at 500/1000/1500/2000 INSIDE chunks 1/3/5/7,
state hand-off r -> r+1 before every scan) and kfnet_amd.dist.run_chunk.  A sharded run must
be BIT-IDENTICAL to a single pass: every per-pixel operation is deterministic and the scan
order is fixed.  On a one-GPU box the ranks share the device and the 76.8 KB message goes
through gloo; the RCCL transports (kfn_send_state / kfn_recv_state, torch 'nccl') need one
GPU per rank and are covered by tests/test_gpu_comm.py where that is available."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_eight_ranks_2048_frames_bit_identical_to_single_pass(tmp_path):
    """2048 frames (reduced 64x96 images -> 8x12 grid), 8 gloo ranks sharing this GPU."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    T, world = 2048, 8
    out = tmp_path / 'sharded'
    out.mkdir()
    env = dict(os.environ, KFN_DIST_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), '-m', 'kfnet_amd.KFNet.eval',
           '--scene', 'heads', '--synthetic', str(T), '--random_weights', '--batch', '8',
           '--height', '64', '--width', '96', '--output_folder', str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    for k in range(world):
        assert 'rank %d/%d: frames %d~%d done' % (k, world, 256 * k, 256 * k + 255) in r.stdout
    # single pass in this process, same inputs (eval's --synthetic/--random_weights seeds)
    imgs = synthetic_sequence(T, 64, 96)
    T4 = np.linalg.inv(synthetic_transform())
    eng = KFNetEngine(synthetic_weights(1234), image_size=(64, 96), batch=8, transform=T4, reset_period=500,
                      max_chunk=T)
    one = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    got = np.stack([np.load(out / ('coord_%d.npy' % i)) for i in range(T)])
    assert got.shape == one.shape == (T, 8, 12, 4)
    assert np.array_equal(got, one)
    # the resets really fall inside chunks and matter: frame 500's record is the pure measurement
    assert not np.array_equal(one[499], one[500])


def _run_sharded_in_process(W, dev_frames, world, T4, batch, size, reset_period=500):
    """The 8 ranks' run_chunk calls one after the other in THIS process (LoopbackLink): same
    code path and pairing rule as the real transports, separate engines per rank."""
    from kfnet_amd.dist import LoopbackLink, chunk_bounds, needs_state, run_chunk
    from kfnet_amd.engine import KFNetEngine
    T = int(dev_frames.shape[0])
    mailbox, parts = {}, []
    for r in range(world):
        lo, hi = chunk_bounds(T, world, r)
        eng = KFNetEngine(W, image_size=size, batch=batch, transform=T4, reset_period=reset_period,
                          max_chunk=max(hi - lo, 1))
        prev = dev_frames[lo - 1] if (hi > lo and needs_state(lo, reset_period)) else None
        rec = run_chunk(eng, dev_frames[lo:hi], lo, r, world, LoopbackLink(mailbox, r), prev)
        parts.append(rec.cpu().numpy().copy())
        del eng
    assert not mailbox, 'unconsumed state messages: %s' % list(mailbox)
    return np.concatenate(parts)


def test_config4_full_size_2048_frames_chunked_8x256_equals_single_pass():
    """480x640, 2048 frames: one engine pass vs 8 contiguous 256-frame chunks with the state
    handed chunk to chunk (resets at 500/1000/1500/2000 inside chunks) -- bit-identical."""
    import torch
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    T, world, size = 2048, 8, (480, 640)
    W = synthetic_weights(1234)
    T4 = np.linalg.inv(synthetic_transform())
    # 2048 distinct frames would take a minute of host time to synthesise; a 64-frame seeded
    # sequence repeated 32x exercises the same code (the test compares two GPU runs)
    base = synthetic_sequence(64, size[0], size[1], seed=2)
    eng = KFNetEngine(W, image_size=size, batch=17, transform=T4, reset_period=500, max_chunk=T)
    dev = eng.upload_frames(base).repeat(T // 64, 1, 1, 1)
    one = eng.process(dev).cpu().numpy().copy()
    del eng
    torch.cuda.empty_cache()
    got = _run_sharded_in_process(W, dev, world, T4, 17, size)
    assert got.shape == (T, 60, 80, 4) and np.all(np.isfinite(got))
    assert np.array_equal(got, one)


@pytest.mark.parametrize('T,world,period', [(40, 4, 10), (37, 5, 7), (3, 4, 500), (24, 3, 8)])
def test_sharded_edge_cases_in_process(T, world, period):
    """Chunks that START on a reset frame (no message: (40,4,10), (24,3,8)), ragged chunks with
    resets inside (37,5,7) and more ranks than frames (empty chunks forward the state)."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(5)
    imgs = synthetic_sequence(T, 64, 96, seed=9)
    eng = KFNetEngine(W, image_size=(64, 96), batch=3, reset_period=period, max_chunk=T)
    dev = eng.upload_frames(imgs)
    one = eng.process(dev).cpu().numpy().copy()
    got = _run_sharded_in_process(W, dev, world, None, 3, (64, 96), reset_period=period)
    assert np.array_equal(got, one)


# ---- block-cyclic sharding (kfnet_amd.dist.run_cyclic / iter_cyclic; round 5) -----------------------------------------------------
@pytest.mark.parametrize('T,world,block,period', [(300, 8, 16, 100), (70, 3, 8, 24), (50, 4, 7, 500), (9, 8, 2, 4)])
def test_cyclic_sharding_in_process_equals_single_pass(T, world, block, period):
    """N "ranks" = N engines in this process, their block generators advanced in GLOBAL block order (what the real transports
    do in time), the state travelling through a LoopbackLink mailbox (rank -> rank+1, wrapping from the last rank to 0): the
    records equal a single pass bit for bit.  (300, 8, 16, 100): 19 blocks over 8 ranks, resets inside blocks and -- at 96 =
    6 x 16 -- none on a block boundary; (70, 3, 8, 24): resets at 24 / 48 ON block boundaries (no message there), ragged last
    block; (9, 8, 2, 4): fewer blocks than ranks."""
    from kfnet_amd.dist import LoopbackLink, cyclic_blocks, iter_cyclic
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    W = synthetic_weights(5)
    T4 = np.linalg.inv(synthetic_transform())
    imgs = synthetic_sequence(T, 64, 96, seed=4)
    one_eng = KFNetEngine(W, image_size=(64, 96), batch=3, transform=T4, reset_period=period, max_chunk=T)
    dev = one_eng.upload_frames(imgs)
    one = one_eng.process(dev).cpu().numpy().copy()
    mailbox = {}
    gens, asked = [], [[] for _ in range(world)]
    for r in range(world):
        eng = KFNetEngine(W, image_size=(64, 96), batch=3, transform=T4, reset_period=period, max_chunk=block)

        def frames_of(lo, hi, r=r):
            asked[r].append((lo, hi))
            return dev[lo:hi]
        gens.append(iter_cyclic(eng, frames_of, T, block, r, world, LoopbackLink(mailbox, r)))
    got = np.zeros_like(one)
    nblocks = -(-T // block)
    for j in range(nblocks):
        lo, rec = next(gens[j % world])
        assert lo == j * block
        got[lo:lo + rec.shape[0]] = rec.cpu().numpy()
    for g in gens:
        with pytest.raises(StopIteration):
            next(g)
    assert not mailbox, 'unconsumed state messages: %s' % list(mailbox)
    assert np.array_equal(got, one)
    # a rank touches only its own blocks and the frame in front of each
    for r in range(world):
        own = [(lo, hi) for _, lo, hi in cyclic_blocks(T, block, r, world)]
        assert [(hi) for _, hi in asked[r]] == [hi for _, hi in own]
        assert all(a_lo in (lo, lo - 1) for (a_lo, _), (lo, _) in zip(asked[r], own))


def test_cyclic_eval_cli_four_gloo_ranks_bit_identical_to_single_pass(tmp_path):
    """python -m torch.distributed.run ... -m kfnet_amd.KFNet.eval --sharding cyclic --block 16: four processes sharing this
    GPU over gloo, 13 blocks (three revolutions of the ring + one), every rank writes the coord_<i>.npy of its own blocks."""
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    T, world, block = 200, 4, 16
    out = tmp_path / 'cyclic'
    out.mkdir()
    env = dict(os.environ, KFN_DIST_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), '-m', 'kfnet_amd.KFNet.eval',
           '--scene', 'heads', '--synthetic', str(T), '--random_weights', '--batch', '8', '--sharding', 'cyclic', '--block', str(block),
           '--height', '64', '--width', '96', '--output_folder', str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'rank 0/4: 4 blocks of 16 frames done (block-cyclic)' in r.stdout and 'rank 3/4: 3 blocks' in r.stdout
    imgs = synthetic_sequence(T, 64, 96)
    T4 = np.linalg.inv(synthetic_transform())
    eng = KFNetEngine(synthetic_weights(1234), image_size=(64, 96), batch=8, transform=T4, reset_period=500, max_chunk=T)
    one = eng.process(eng.upload_frames(imgs)).cpu().numpy()
    got = np.stack([np.load(out / ('coord_%d.npy' % i)) for i in range(T)])
    assert np.array_equal(got, one)
