#!/usr/bin/env python
"""bench.py -- frames/sec of the MI355X-native KFNet prediction path (the headline pass + the driver's line).

A "step" is ONE 480x640 frame through the whole hot path (SCoordNet + flow-feature tower + cost volume + OFlowNet +
flow head + warp / Kalman fuse / emit).  Workload = BASELINE.json configs[2]: full KFNet on a synthetic 480x640
sequence, random weights, fp32.  The timed region starts with the uint8 frames already resident in HBM and ends when
the [K,60,80,4] records are in HBM.

N > 1 (one rank per GPU under torch.distributed.run; `python bench.py --gpus N` starts the ranks itself): the N*K-frame
sequence is sharded into contiguous K-frame chunks (weak scaling); every rank runs the state-independent heavy phase
of its chunk at once, the recurrent Kalman state (76.8 KB) goes rank r -> r+1 by RCCL send/recv before the scan.

Timing: W untimed warm-up steps, then the K-step pass is repeated until --min-seconds are timed; every repetition is
exactly K steps bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks; the line reports the
MEDIAN repetition.

OUTPUT.  The LAST stdout line is ONE compact JSON object (< 4 KB, no prose; `compact_line`): the contract fields,
`roofline` of the step's dominant kernel (executed MFMA FLOPs / HIP-event time of its launches; `traffic` = PMC bytes
per launch quoted from profiles/), `cpu_baseline` (the torch-CPU restatement on config 1's 16 frames), and one number
each for the side measurements (value_streamed, parity, Kalman rooflines, config 5, config 2, multi-rank extras).
Everything else -- per-kernel tables, telemetry, notes -- goes to the sidecar file gpurun_out/bench_detail.json
(`--detail PATH`; bench_extra.py holds the code of those blocks).  `--config c2` / `--config c5` print those
configurations alone.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from bench_extra import (PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, Telemetry, auto_batch, bench_c2, bench_c5,  # noqa: E402,F401
                         chain_timing, config3_literal, config4_literal, cyclic_sharding_block, host_streamed,
                         kalman_fuse_roofline, kalman_roofline, latest_pmc_traffic, measure_c2, measure_c5,
                         per_kernel_profile, pipeline_io_bytes, timed_repetitions)

COMPACT_LIMIT = 4096     # bytes; the driver reads a bounded tail of stdout (BENCH_r05: a 21 KB line was not parsed)


def _pick(d, *ks):
    return None if not isinstance(d, dict) else {k: d.get(k) for k in ks if d.get(k) is not None}


def _short(s, n=160):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + '...'


def compact_line(out):
    """The driver's line from the full result dict: contract fields + roofline + cpu_baseline + one number per side
    measurement.  Strict JSON, under COMPACT_LIMIT bytes whatever the detail holds (tests/test_host_logic.py)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data')
    line = {k: out.get(k) for k in keep}
    line['dtype'] = _short(line['dtype'], 48)
    line['data'] = _short(line['data'], 96)
    cfg = out.get('config') or {}
    line['config'] = {k: (_short(v) if isinstance(v, str) else v) for k, v in cfg.items()}
    for k in ('repetitions', 'timed_seconds', 'state_link', 'dist_backend', 'rccl_ranks', 'sharding', 'watchdog_fired'):
        if out.get(k) is not None:
            line[k] = out[k]
    rf = out.get('roofline')
    if rf is not None:
        r = _pick(rf, 'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'launches_per_batch',
                  'executed_gflop_per_launch_avg', 'share_of_step_time')
        t = rf.get('traffic')
        r['traffic'] = None if not t else {'fetch': t.get('fetch_bytes_per_launch'), 'write': t.get('write_bytes_per_launch'),
                                           'hbm': t.get('hbm_bytes_per_launch'), 'algorithmic': t.get('algorithmic_bytes_per_launch'),
                                           'from': t.get('quoted_from'), 'tower_batch': (t.get('sampled') or {}).get('tower_batch')}
        line['roofline'] = r
    cb = out.get('cpu_baseline')
    if cb is not None:
        c = _pick(cb, 'value', 'unit', 'cores', 'kind', 'deduplicated_value')
        c['sample'] = _short(cb.get('sample', ''), 140)
        line['cpu_baseline'] = c
        line['speedup_vs_cpu_baseline'] = out.get('speedup_vs_cpu_baseline')
    hs = out.get('host_streamed')
    if hs is not None:
        line['value_streamed'] = hs['value']
    png = out.get('eval_png_end_to_end')
    if png is not None:
        line['value_png_to_npy'] = png.get('value')
    c3 = out.get('config3_256_frames')
    if c3 is not None:
        line['config3_256_frames'] = _pick(c3, 'value', 'ms_per_step')
    pc = out.get('parity_vs_cpu_restatement')
    if pc is not None:
        line['parity'] = _pick(pc, 'frames', 'coord_max_abs', 'conf_max_rel')
    if 'roofline_kalman' in out:
        line['roofline_kalman'] = {'bound': 'hbm', 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                   'T64': out['roofline_kalman'].get('frac'),
                                   'T256': (out.get('roofline_kalman_T256') or {}).get('frac'),
                                   'fuse': (out.get('roofline_kalman_fuse') or {}).get('frac')}
    c5 = out.get('config5_960x540')
    if c5 is not None:
        rf5 = c5.get('roofline') or {}
        p5 = c5.get('parity_vs_fp32_path') or {}
        line['config5'] = {'value': c5.get('value'), 'unit': c5.get('unit'), 'kernel': rf5.get('kernel'), 'frac': rf5.get('frac'),
                           'unmasked_outside_tolerance': p5.get('unmasked_outside_tolerance')}
    c2 = out.get('config2_single_frame')
    if c2 is not None:
        line['config2_ms'] = (c2.get('latency_ms') or {}).get('hipgraph_replay_median')
    for k, ks in (('handoff', ('scan_chain_ms', 'tail_ms', 'handoff_us_net', 'error')),
                  ('sharding_cyclic', ('value', 'block', 'tail_ms', 'error')),
                  ('config4_2048_frames', ('value', 'ms_per_step', 'error'))):
        if out.get(k) is not None:
            line[k] = {a: (_short(b) if isinstance(b, str) else b) for a, b in (_pick(out[k], *ks) or {}).items()}
    if out.get('multi_rank_extras'):
        line['multi_rank_extras'] = _short(out['multi_rank_extras'], 80)
    if out.get('detail_file'):
        line['detail_file'] = out['detail_file']
    s = json.dumps(line, allow_nan=False, separators=(',', ':'))
    if len(s) >= COMPACT_LIMIT:      # cannot happen with the fields above; if it ever does, drop the optional ones
        for k in ('config4_2048_frames', 'sharding_cyclic', 'handoff', 'config3_256_frames', 'rccl_ranks', 'multi_rank_extras'):
            line.pop(k, None)
        s = json.dumps(line, allow_nan=False, separators=(',', ':'))
    assert len(s) < COMPACT_LIMIT, len(s)
    return s


def _json_safe(o):
    """NaN / inf -> None, numpy scalars -> Python (the sidecar stays strict JSON too)."""
    if isinstance(o, dict):
        return {str(k): _json_safe(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_json_safe(v) for v in o]
    if isinstance(o, (np.floating, float)):
        f = float(o)
        return f if np.isfinite(f) else None
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, np.bool_):
        return bool(o)
    return o


def emit(out, args):
    """Full result -> sidecar file; compact line -> the LAST line of stdout."""
    out = _json_safe(out)
    path = args.detail
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(out, f, indent=1, allow_nan=False)
            f.write('\n')
        out['detail_file'] = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:
        sys.stderr.write('bench.py: could not write %s: %s\n' % (path, e))
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=256,
                    help='frames per GPU in the timed region (BASELINE config 3: a 256-frame sequence)')
    ap.add_argument('--warmup', type=int, default=17)
    ap.add_argument('--config', choices=['c3', 'c2', 'c5'], default='c3',
                    help='BASELINE.json config: c3 = full KFNet 480x640 sequence (headline), c2 = SCoordNet-only '
                         'single-frame latency, c5 = 960x540 batch-of-sequences, fp16 convs + fp32 Kalman')
    ap.add_argument('--batch', type=int, default=0,
                    help='frames per tower launch; 0 = auto: the size in 15..32 that splits --steps with the least '
                         'ragged tail (the rate is flat over that range; 17 when --steps is a multiple of 17)')
    ap.add_argument('--min-seconds', type=float, default=2.0,
                    help='repeat the K-step pass until this much time has been measured (median reported)')
    ap.add_argument('--sequences', type=int, default=4, help='c5: independent sequences per scan launch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-streamed', action='store_true',
                    help='skip the PCIe-inclusive run (frames from pinned host memory)')
    ap.add_argument('--no-alt-modes', action='store_true',
                    help='skip the extra (non-headline) measurement of the f16x3 split-operand mode')
    ap.add_argument('--one-stream', action='store_true', help='serialise the two towers on one stream')
    ap.add_argument('--graph', action='store_true', help='replay the heavy phase of full batches from a captured hipGraph')
    ap.add_argument('--autotune', action='store_true',
                    help='time every tile config per layer at start-up (the heuristic is within ~2%% of it)')
    ap.add_argument('--cpu-steps', type=int, default=16,
                    help='frames of the CPU baseline (BASELINE config 1 is a 16-frame 480x640 sequence)')
    ap.add_argument('--no-kalman-roofline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='c3 on one GPU: skip the config5_960x540 / config2_single_frame blocks')
    ap.add_argument('--no-eval-png', action='store_true', help='skip the PNG -> coord_<i>.npy end-to-end block')
    ap.add_argument('--block', type=int, default=32, help='multi-rank runs: frames per block of the block-cyclic sharding measured beside the contiguous one')
    ap.add_argument('--decode-workers', type=int, default=0, help='PNG decode threads of the end-to-end block (0 = min(32, cores / 2))')
    ap.add_argument('--eval-chunk', type=int, default=32, help='frames per host chunk of the PNG end-to-end block')
    ap.add_argument('--eval-ramp', default='', help='lengths of the first chunks of the PNG end-to-end block, e.g. "8,16" (default: eval()\'s own: 8, 16); "0" = none')
    ap.add_argument('--no-config3', action='store_true',
                    help='when --steps < 256: skip the additional literal 256-frame / batch-32 pass of BASELINE configs[2]')
    ap.add_argument('--conv-operands', choices=['f32', 'f16', 'f16x3'], default='f32',
                    help="f16 = BASELINE config 5's fp16-operand convs (fp32 accumulate, fp32 Kalman); NOT the headline")
    ap.add_argument('--graph-option', action='append', default=[], metavar='NAME=VALUE',
                    help='a routing switch of kfnet_amd.graph.Graph for an A/B run of the main engine, e.g. '
                         'winograd_f43_eight_wave=0 (recorded in the line as graph_options)')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--detail', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='sidecar file for the full result (per-kernel tables, telemetry, notes); stdout gets the compact line')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: start the N ranks
    ourselves (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 at a free
    port) and pass rank 0's JSON line through.  The driver's own form -- `python -m
    torch.distributed.run ... bench.py --gpus N` -- sets WORLD_SIZE and never comes here."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL's P2P buffers need it on this driver
    env['KFN_BENCH_SELF_LAUNCHED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(frames, W, T4, steps):
    """Reference-faithful CPU restatement (oracle/kfnet_oracle_torch.py): both towers on a
    2-frame batch per step, 64 materialised shifts, unfused ops (KFNet/eval.py:41,77-104)."""
    import torch
    from oracle import kfnet_oracle_torch as OT
    cores = torch.get_num_threads()
    sx = ss = None
    recs = []
    # one untimed warm-up step (oneDNN primitive creation)
    OT.eval_step_reference_style(np.stack([frames[1], frames[0]]), W, None, None, T4, True)
    t0 = time.time()
    for i in range(steps):
        pair = np.stack([frames[1], frames[0]]) if i == 0 else np.stack([frames[i - 1], frames[i]])
        rec, sx, ss = OT.eval_step_reference_style(pair, W, sx, ss, T4, i % 500 == 0)
        recs.append(rec)
    dt = time.time() - t0
    # the same restatement with the reference's redundancy removed (towers once per frame,
    # SURVEY.md F9), so that the GPU/CPU ratio can be read without it
    nd = max(2, min(steps, 8))
    t1 = time.time()
    OT.eval_sequence(frames[:nd], W, T4, 500, dedup=True)
    dt_d = time.time() - t1
    return {'value': round(steps / dt, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d eval.py-style steps (2-frame tower batches, 64 shifts) of the same 480x640 '
                      'sequence, torch-CPU fp32, %.1f s' % (steps, dt),
            'deduplicated_value': round(nd / dt_d, 4),
            'deduplicated_sample': '%d frames, towers once per frame, %.1f s' % (nd, dt_d)}, np.stack(recs)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and args.config == 'c3':
        raise SystemExit(self_launch(args))
    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if args.config == 'c2':
        return bench_c2(args, device)
    if args.config == 'c5':
        return bench_c5(args, device)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL over xGMI when every rank has its own GPU (the production path); a functional
        # fallback over gloo lets the sharded path be exercised with ranks sharing a GPU
        backend = os.environ.get('KFN_DIST_BACKEND', 'nccl' if ndev >= world else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from kfnet_amd.KFNet.eval import get_transform  # noqa: F401
    from kfnet_amd.engine import KFNetEngine
    from kfnet_amd.synth import synthetic_sequence, synthetic_transform
    from kfnet_amd.weights import synthetic_weights
    from kfnet_amd.dist import make_link, needs_state, run_chunk
    from kfnet_amd._lib import check as _lib_check

    K, Wm = args.steps, args.warmup
    B = max(1, min(args.batch, K)) if args.batch > 0 else auto_batch(K)
    Wt = synthetic_weights(1234)
    T4 = np.linalg.inv(synthetic_transform())      # = get_transform(transform.txt) (KFNet/train.py:49-58)
    # the whole job is one N*K-frame sequence; rank r owns frames [r*K, (r+1)*K)
    # (synthetic frames are generated per rank from the global frame index)
    lo = rank * K
    need_prev = 1 if needs_state(lo, 500) else 0
    frames_all = synthetic_sequence(K + need_prev, args.height, args.width, seed=1, start=lo - need_prev)
    gopts = {}
    for kv in args.graph_option:
        name, _, val = kv.partition('=')
        gopts[name] = int(val) if val.lstrip('-').isdigit() else val
    eng = KFNetEngine(Wt, image_size=(args.height, args.width), batch=B, transform=T4, reset_period=500,
                      max_chunk=max(K, Wm, B), device=str(device), autotune=args.autotune,
                      conv_operands=args.conv_operands, use_graph=args.graph, graph_options=gopts or None)
    eng.two_streams = not args.one_stream
    dev_all = eng.upload_frames(frames_all)
    dev_prev = dev_all[0] if need_prev else None
    dev_frames = dev_all[need_prev:]
    link = make_link(dist, rank, world, dev_index, prefer=os.environ.get('KFN_STATE_LINK', 'auto'))

    # warm-up: W untimed steps (+ one untimed sharded pass that opens the p2p channels)
    if Wm > 0:
        eng.process(dev_frames[:min(Wm, K)], t0=lo)
    torch.cuda.synchronize()
    if dist is not None:
        # (the full chunk: the send/recv pairing rule is a function of the chunk boundaries)
        run_chunk(eng, dev_frames, lo, rank, world, link, dev_prev)
        torch.cuda.synchronize()
    tele = Telemetry(dev_index)
    with tele:
        times = timed_repetitions(lambda: run_chunk(eng, dev_frames, lo, rank, world, link, dev_prev),
                                  dist, device, backend, args.min_seconds)
    elapsed = float(np.median(times))
    # what every rank actually ran on: its device, and the (rank, nranks) its C-ABI RCCL communicator reports
    # (kfn_comm_rank; None when the hand-off goes through torch.distributed), all-gathered for the line
    me = {'rank': rank, 'device': dev_index, 'device_name': torch.cuda.get_device_name(dev_index)}
    if link is not None and hasattr(link, 'comm'):
        import ctypes as C
        r_, n_ = C.c_int(-1), C.c_int(-1)
        _lib_check(link.lib.kfn_comm_rank(link.comm, C.byref(r_), C.byref(n_)), 'kfn_comm_rank')
        me['rccl'] = [r_.value, n_.value]
    else:
        me['rccl'] = None
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    total_frames = K * world
    fps = total_frames / elapsed

    out = {
        'metric': 'frames/sec on 480x640 seq', 'value': round(fps, 3), 'unit': 'frames/s',
        'n_gpus': world, 'steps': K, 'warmup': Wm, 'ms_per_step': round(elapsed * 1e3 / K, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f16': 'f16 conv operands (f32 accumulate, f32 activations, f32 Kalman)',
                  'f16x3': 'f32 emulated by 3 fp16 MFMA products of hi/lo-split operands (f32 accumulate)'}[args.conv_operands],
        'data': 'synthetic (rolled random texture uint8 frames, seeded He-uniform random weights)',
        'repetitions': len(times), 'timed_seconds': round(float(np.sum(times)), 3),
        'ms_per_step_min_max': [round(min(times) * 1e3 / K, 4), round(max(times) * 1e3 / K, 4)],
        'timing': 'each repetition = exactly %d steps bracketed by barrier+synchronize, MAX over ranks; value and '
                  'ms_per_step are the MEDIAN repetition' % K,
        'config': {'workload': 'full KFNet (SCoordNet+OFlowNet+Kalman) %d-frame %dx%d seq per GPU, random weights'
                               % (K, args.height, args.width),
                   'frames_total': total_frames, 'tower_batch': B, 'reset_period': 500,
                   'parallelism': 'frame-sharded x%d, Kalman state rank->rank via %s%s'
                                  % (world, link.name if link is not None else 'nothing (single GPU)',
                                     '' if ndev >= world else
                                     ' -- FUNCTIONAL FALLBACK: only %d GPU(s) visible, ranks share them and the state '
                                     'goes through the host (gloo); not an xGMI number' % ndev)},
        'graph_options': gopts or None,
        'state_link': link.name if link is not None else None,
        'dist_backend': backend,
        'devices_visible': ndev,
        'rccl_ranks': [ri['rccl'] for ri in ranks_info],
        'rank_devices': [ri['device'] for ri in ranks_info],
        'self_launched': bool(os.environ.get('KFN_BENCH_SELF_LAUNCHED')),
        'gpu_telemetry_rank0': tele.summary(),
    }
    # ---- collective extras of a multi-rank run (every rank takes part; rank 0 reports) -------------------
    # The headline numbers are complete at this point.  The extras below are further collectives (event-timed chains, the
    # block-cyclic pass, config 4): should one of them hang on a fabric this build has never run on (no multi-GPU box was
    # available to it), a watchdog on every rank ends the process after KFN_BENCH_EXTRAS_DEADLINE seconds -- rank 0 prints the
    # line it has, marked -- instead of leaving the driver without any line.
    watchdog = None
    if dist is not None:
        import threading
        deadline = float(os.environ.get('KFN_BENCH_EXTRAS_DEADLINE', '300'))

        def bail():
            if rank == 0:
                out['multi_rank_extras'] = ('TIMED OUT after %.0f s: handoff / sharding_cyclic / config4_2048_frames and the '
                                            'rank-0 rooflines are missing from this line; the headline fields are complete' % deadline)
                out['watchdog_fired'] = True
                emit(out, args)
            os._exit(3)
        watchdog = threading.Timer(deadline, bail)
        watchdog.daemon = True
        watchdog.start()
    if dist is not None:
        extras_ok = [True]

        def extra(key, fn):
            """An extra is a sequence of collectives: a rank that failed inside one must not walk into the next while its
            peers are still in the previous (mispaired send/recv).  Every rank reports its outcome, all-reduced (MIN) after
            each extra; after the first failure anywhere, every rank skips the rest.  (A rank that dies INSIDE a collective
            leaves the others waiting: that case is the watchdog's.)"""
            if not extras_ok[0]:
                out[key] = {'error': 'skipped: an earlier multi-rank extra failed on some rank'}
                return
            ok = 1
            try:
                out[key] = fn()
            except Exception as e:      # noqa: BLE001
                out[key] = {'error': '%s: %s' % (type(e).__name__, e)}
                ok = 0
            flag = torch.tensor([ok], device=device if backend == 'nccl' else 'cpu', dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                extras_ok[0] = False
                if ok:
                    out[key] = dict(out[key] if isinstance(out[key], dict) else {}, error='failed on another rank')
        extra('handoff', lambda: chain_timing(lambda timer: run_chunk(eng, dev_frames, lo, rank, world, link, dev_prev, timer=timer),
                                              dist, device, world))
        out['sharding'] = 'contiguous'       # of the headline `value`; the block-cyclic alternative is measured beside it
        extra('sharding_cyclic', lambda: cyclic_sharding_block(args, eng, rank, world, K, link, dist, device, backend))
        if world == 8 and K < 256 and not args.no_config3 and args.conv_operands == 'f32':
            extra('config4_2048_frames', lambda: config4_literal(args, Wt, T4, rank, world, link, dist, device, backend, dev_index))
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        rows = per_kernel_profile(eng, dev_frames)
        heavy_ms = sum(r[3] for r in rows)
        by_kernel = {}
        for r in rows:
            k = by_kernel.setdefault(r[1], [0, 0.0, 0.0, 0.0])
            k[0] += 1; k[1] += r[2]; k[2] += r[3]; k[3] += r[4]
        # dominant kernel = the instantiation with the largest share of the step time
        dom = max(by_kernel, key=lambda k: by_kernel[k][2])
        n_dom, fl_dom, ms_dom, ex_dom = by_kernel[dom]
        tf = fl_dom / (ms_dom * 1e-3) / 1e12
        tf_exec = ex_dom / (ms_dom * 1e-3) / 1e12
        conv_ms = sum(v[2] for k, v in by_kernel.items() if k.startswith('conv_mfma_kernel'))
        conv_fl = sum(v[1] for k, v in by_kernel.items() if k.startswith('conv_mfma_kernel'))
        # algorithmic HBM bytes of one launch of the dominant kernel: its input and output tensors once, its weights once
        # (averaged over its launches of a batch, at the tower batch the PMC passes sampled so that it compares with them)
        def alg_bytes(batch):
            tot, n = 0.0, 0
            for op in eng.heavy_ops:
                if hasattr(op, 'kernel_name') and op.kernel_name(eng.lib) == dom and hasattr(op, 'x') and hasattr(op, 'y'):
                    esz = {'f32': 4, 'f16': 2, 'u8': 1}
                    act = sum(float(np.prod(t.shape[1:])) * esz[t.dtype] for t in (op.x, op.y))
                    tot += act * batch + float(np.prod(op.kernel.shape)) * 4
                    n += 1
            return int(tot / n) if n else None
        traffic = None
        tpath, tj = latest_pmc_traffic()
        traffic = tj.get(dom)
        if traffic is not None:
            traffic = dict(traffic, algorithmic_bytes_per_launch=alg_bytes((tj.get('__sampled__') or {}).get('tower_batch', 32)))
            # NOT measured by this run: PMC counters need their own rocprofv3 passes (tools/profile_round.sh)
            traffic = dict(traffic, quoted_from=os.path.relpath(tpath, ROOT), sampled=tj.get('__sampled__', {
                'command': 'bench.py --steps 64 --batch 32', 'tower_batch': 32}), this_run_tower_batch=B)
        out['roofline'] = {
            'kernel': dom, 'bound': 'mfma',
            # FLOPs the fp32 MFMA pipe actually executes in this kernel / its time.  For the direct implicit GEMM
            # that IS the algorithmic (nominal dense) FLOP count of SURVEY App. C; the Winograd kernels execute
            # 16/36 (F(2x2,3x3)) or 9/36 (F(4x4,3x3)) of it, so the hardware-utilisation number is reported here and
            # the algorithmic rate beside it.
            'achieved': round(tf_exec, 2), 'peak': PEAK_F32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(tf_exec / PEAK_F32_MFMA_TFLOPS, 4),
            'traffic': traffic,
            'algorithmic_tflops': round(tf, 2),      # a rate (nominal FLOPs / time), NOT a fraction of the peak
            'note': ('wino3_kernel / wino2_kernel = single-kernel Winograd F(2x2,3x3), wino4b_kernel / wino4_kernel = F(4x4,3x3) on eight / four waves '
                     '(kfn_conv2d_winograd_fused, fp32; the waves of a workgroup share one input transform through LDS): '
                     'achieved = FLOPs the MFMAs execute (16/36 resp. 9/36 of the nominal direct-convolution FLOPs + tile-block '
                     'padding) / time, algorithmic_* counts the nominal FLOPs of SURVEY App. C and may exceed the MFMA peak; '
                     'traffic = PMC HBM-side bytes per launch, averaged over the launches of a batch like avg_launch_ms. '
                     'conv_mfma_kernel<TM,TN,WM,WN,BK,MODE,PREC>: MODE 0 direct implicit GEMM, 1 transposed, 2 the 16 GEMMs of the '
                     'two-kernel Winograd form, 3 conv0 with the cost volume in the loader'),
            'launches_per_batch': n_dom,
            'algorithmic_gflop_per_launch_avg': round(fl_dom / n_dom / 1e9, 3),
            'executed_gflop_per_launch_avg': round(ex_dom / n_dom / 1e9, 3),
            'avg_launch_ms': round(ms_dom / n_dom, 4),
            'share_of_step_time': round(ms_dom / heavy_ms, 4),
            'all_conv_mfma_algorithmic_tflops': round(conv_fl / (conv_ms * 1e-3) / 1e12, 2) if conv_ms else None,
            'all_conv_mfma_share_of_step_time': round(conv_ms / heavy_ms, 4),
        }
        out['kernels_ms_per_batch'] = {k: {'launches': v[0], 'ms': round(v[2], 4),
                                           'tflops': round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[1] else None,
                                           'executed_tflops': round(v[3] / (v[2] * 1e-3) / 1e12, 2) if v[3] else None}
                                       for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][2])}
        out['per_kernel_ms_per_batch'] = {r[0] + ('#%d' % i): round(r[3], 4) for i, r in enumerate(rows)}
        top = sorted(rows, key=lambda r: -r[3])[:6]
        io_b = pipeline_io_bytes(eng)
        out['pipeline_hbm'] = {
            'algorithmic_bytes_per_frame': int(io_b / B),
            'achieved_GBs': round(io_b / B * fps / 1e9, 1), 'peak': PEAK_HBM_GBS,
            'frac_of_hbm_roofline': round(io_b / B * fps / 1e9 / PEAK_HBM_GBS, 4),
            'note': 'layer-by-layer fp32 activation + weight traffic / frame time: the path is MFMA-bound, not HBM-bound'}
        if eng.tuned:
            out['autotuned_tile_config'] = {k.split('@')[0]: v[0] for k, v in eng.tuned.items()}
        out['top_layers'] = [{'op': r[0], 'ms': round(r[3], 3),
                              'tflops': round(r[2] / (r[3] * 1e-3) / 1e12, 1) if r[2] else None} for r in top]
        if not args.no_kalman_roofline:
            out['roofline_kalman'] = kalman_roofline(device)                        # S = 256 x T = 64
            out['roofline_kalman_T256'] = kalman_roofline(device, S=256, T=256)     # SURVEY 8(d)'s default shape
            out['roofline_kalman_fuse'] = kalman_fuse_roofline(device)
        single_480 = world == 1 and (args.height, args.width) == (480, 640) and args.conv_operands == 'f32'
        if single_480 and not args.no_config3:
            # the literal 256-frame pass of BASELINE configs[2], and on the same engine / sequence the two
            # transfer-inclusive forms: frames from pinned host memory, and PNG files -> coord_<i>.npy files
            c3, extra = config3_literal(args, Wt, T4, synthetic_transform(), device, dev_index)
            if K < 256:
                out['config3_256_frames'] = c3
            out.update(extra)
        elif world == 1 and not args.no_host_streamed:
            out['host_streamed'] = host_streamed(eng, frames_all[need_prev:], dev_frames)
        if world == 1 and not args.no_cpu_baseline:
            host_frames = frames_all[need_prev:need_prev + max(args.cpu_steps, 2)]
            if host_frames.shape[0] < max(args.cpu_steps, 2):     # --steps below config 1's 16 frames
                host_frames = synthetic_sequence(max(args.cpu_steps, 2), args.height, args.width, seed=1)
            cb, cpu_recs = cpu_baseline(host_frames, Wt, T4, args.cpu_steps)
            cb['sample_is_config1'] = bool(args.cpu_steps == 16)
            out['cpu_baseline'] = cb
            gpu_recs = eng.process(eng.upload_frames(host_frames[:args.cpu_steps]), t0=0).cpu().numpy()
            out['parity_vs_cpu_restatement'] = {
                'frames': int(args.cpu_steps),
                'coord_max_abs': float(np.abs(gpu_recs[..., :3] - cpu_recs[..., :3]).max()),
                'conf_max_rel': float((np.abs(gpu_recs[..., 3] - cpu_recs[..., 3]) / np.abs(cpu_recs[..., 3])).max()),
                'tolerance': 'coord max-abs <= 1e-4, confidence max-rel <= 1e-4'}
            out['speedup_vs_cpu_baseline'] = round(fps / cb['value'], 1)
            if not args.no_alt_modes and args.conv_operands == 'f32':
                # NOT the headline: same workload with every wide forward conv evaluated as
                # hi*hi + hi*lo + lo*hi of fp16-split operands on the fp16 MFMA (fp32 accumulate)
                eng2 = KFNetEngine(Wt, image_size=(args.height, args.width), batch=B, transform=T4, reset_period=500,
                                   max_chunk=max(K, Wm, B, args.cpu_steps), device=str(device), conv_operands='f16x3')
                eng2.process(dev_frames[:min(Wm, K)], t0=0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng2.process(dev_frames, t0=0)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                g2 = eng2.process(eng2.upload_frames(host_frames[:args.cpu_steps]), t0=0).cpu().numpy()
                del eng2
                out['alt_mode_f16x3'] = {
                    'value': round(K / dt2, 3), 'unit': 'frames/s',
                    'dtype': 'f32 emulated: operands split into fp16 hi+lo, 3 fp16-MFMA products, f32 accumulate',
                    'coord_max_abs_vs_cpu': float(np.abs(g2[..., :3] - cpu_recs[..., :3]).max()),
                    'conf_max_rel_vs_cpu': float((np.abs(g2[..., 3] - cpu_recs[..., 3]) / np.abs(cpu_recs[..., 3])).max()),
                    'note': 'opt-in (KFNetEngine(conv_operands="f16x3")); reported beside, never as, the fp32 headline value'}
        if single_480 and not args.no_extra_configs:
            # BASELINE configs[4] and configs[1] inside the driver's one line (VERDICT r3 Next #1)
            del eng
            torch.cuda.empty_cache()
            c5 = measure_c5(args, device, T=64, min_seconds=2.0, with_parity=True)
            out['config5_960x540'] = {k: c5[k] for k in ('value', 'unit', 'ms_per_step', 'dtype', 'steps', 'repetitions', 'config',
                                                          'roofline', 'gpu_telemetry', 'kernels_ms_per_batch', 'tolerance',
                                                          'parity_vs_fp32_path') if k in c5}
            c2 = measure_c2(args, device, min_seconds=1.0, min_steps=50)
            out['config2_single_frame'] = {k: c2[k] for k in ('value', 'unit', 'ms_per_step', 'dtype', 'config', 'latency_ms',
                                                               'roofline', 'per_layer_batch1')}
        emit(out, args)
    if link is not None:
        link.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
