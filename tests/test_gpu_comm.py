"""-m gpu: the RCCL entry points of the C ABI (kfn_comm_*, kfn_send_state, kfn_recv_state).

World-1 init/destroy runs on any GPU box; the two-rank transfer needs two GPUs and is skipped
otherwise (the driver's multi-GPU node is the first place it can run -- until then the RCCL
transport is UNTESTED on hardware)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from kfnet_amd import _lib

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_world1_init_rank_destroy():
    import torch
    lib = _lib.load()
    uid = C.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(lib.kfn_comm_unique_id(uid, _lib.COMM_ID_BYTES), 'kfn_comm_unique_id')
    assert any(b != 0 for b in uid.raw)
    comm = C.c_void_p()
    _lib.check(lib.kfn_comm_init(C.byref(comm), 0, 1, uid, torch.cuda.current_device()), 'kfn_comm_init')
    r, n = C.c_int(-1), C.c_int(-1)
    _lib.check(lib.kfn_comm_rank(comm, C.byref(r), C.byref(n)), 'kfn_comm_rank')
    assert (r.value, n.value) == (0, 1)
    # a world of one has no valid peer: both calls must be rejected, not hang
    buf = torch.zeros(8 * 12 * 4, device='cuda')
    assert lib.kfn_send_state(comm, 0, buf.data_ptr(), 8, 12, None) == -1
    assert lib.kfn_recv_state(comm, 1, buf.data_ptr(), 8, 12, None) == -1
    assert b'bad peer' in lib.kfn_last_error()
    _lib.check(lib.kfn_comm_destroy(comm), 'kfn_comm_destroy')


def test_rccl_link_world1_via_python_wrapper():
    import torch
    from kfnet_amd.dist import RcclLink
    link = RcclLink(0, 1, torch.cuda.current_device())
    link.close()
    link.close()   # idempotent


_TWO_RANK = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from kfnet_amd.dist import RcclLink, TorchLink
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(rank)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
ok = True
for mk in (lambda: RcclLink(rank, world, rank, dist=dist, grid_hw=(60, 80)), lambda: TorchLink(dist)):
    link = mk()
    state = torch.full((60 * 80 * 4,), float(rank + 1), device='cuda')
    if rank == 0:
        state += torch.arange(60 * 80 * 4, device='cuda') * 1e-3
        link.send(state, 1)
    else:
        link.recv(state, 0)
        torch.cuda.synchronize()
        want = 1.0 + torch.arange(60 * 80 * 4, device='cuda') * 1e-3
        ok = ok and bool(torch.equal(state, want))
    torch.cuda.synchronize()
    link.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
'''


def test_two_rank_state_transfer_over_rccl(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL send/recv between ranks)')
    script = tmp_path / 'two_rank.py'
    script.write_text(_TWO_RANK % ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # ... and the benchmark's own two-rank form on the same two GPUs: the plain command the driver would run, which
    # must end on the production transport and report RCCL's OWN view of the communicator (Next #4d)
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    b = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '8', '--warmup', '2',
                        '--min-seconds', '0.5', '--no-kalman-roofline', '--detail', str(tmp_path / 'detail.json')],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-3000:]
    compact = json.loads(b.stdout.strip().splitlines()[-1])
    assert compact['n_gpus'] == 2 and compact['rccl_ranks'] == [[0, 2], [1, 2]]
    line = json.load(open(tmp_path / 'detail.json'))
    assert line['n_gpus'] == 2 and line['dist_backend'] == 'nccl'
    assert line['rccl_ranks'] == [[0, 2], [1, 2]]                      # ncclCommUserRank / ncclCommCount of every rank
    assert line['state_link'].startswith('C-ABI kfn_send_state')
    assert line['rank_devices'] == [0, 1]
    h = line['handoff']
    assert h['scan_chain_ms'] > 0 and len(h['handoff_us']) == 2
    assert h['handoff_us'][1]['recv_wait_us'] is not None and h['handoff_us'][0]['send_us'] is not None


def test_eval_gpu_flag_selects_device(tmp_path):
    """--gpu N (KFNet/train.py:19): buffers, streams and launches all on device N."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    from kfnet_amd.KFNet import eval as kf_eval
    out0, out1 = tmp_path / 'g0', tmp_path / 'g1'
    out0.mkdir(); out1.mkdir()
    common = ['--scene', 'heads', '--synthetic', '3', '--random_weights', '--batch', '2', '--height', '64', '--width', '96']
    assert kf_eval.main(common + ['--gpu', '0', '--output_folder', str(out0)]) == 0
    assert kf_eval.main(common + ['--gpu', '1', '--output_folder', str(out1)]) == 0
    torch.cuda.set_device(0)
    for i in range(3):
        assert np.array_equal(np.load(out0 / ('coord_%d.npy' % i)), np.load(out1 / ('coord_%d.npy' % i)))
