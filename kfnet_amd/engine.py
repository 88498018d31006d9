"""Sequence engine: drives the KFNet graph over a stream of frames.

This is the native replacement of the per-frame `sess.run` loop of KFNet/eval.py:77-126.
A chunk of T frames is processed in two phases:

  heavy (state-independent, frame-parallel): for every batch of B frames run both towers,
      the cost volume, OFlowNet and the flow head; the per-frame scan inputs
      (flow 8 B/px, sigma_trans 4 B/px, measurement 16 B/px) are appended to chunk buffers;
  scan  (the only sequential part): ONE kfn_kalman_scan launch walks the T frames with
      the recurrent state in LDS and emits the [T,h,w,4] records.

In a multi-GPU run every rank does `heavy` for its own contiguous chunk immediately and
only the scan waits for the 76.8 KB state of the previous rank (kfnet_amd/dist.py).
"""
import numpy as np

from . import _lib
from .graph import Graph, KalmanScanOp
from .KFNet.KFNet import KFNet, KFNetDataSpec


def grid_size(image_hw):
    """Label-grid size = three stride-2 SAME convs = ceil(./8) (not eval.py's `//8`,
    which is wrong for 540 rows -- SURVEY.md F11)."""
    h, w = image_hw
    for _ in range(3):
        h, w = -(-h // 2), -(-w // 2)
    return h, w


class KFNetEngine(object):
    def __init__(self, weights, image_size=(480, 640), batch=4, transform=None, reset_period=500,
                 nis_gate=0.0, max_chunk=256, device='cuda:0', emit_debug=False, autotune=False,
                 conv_operands='f32', use_graph=False, emit_metrics=False, graph_options=None):
        import torch
        self.torch = torch
        self.B = int(batch)
        self.H, self.W = image_size
        self.h, self.w = grid_size(image_size)
        self.reset_period = int(reset_period)
        self.nis_gate = float(nis_gate)
        self.transform = None if transform is None else np.asarray(transform, dtype=np.float32)
        self.max_chunk = int(max_chunk)
        emit_debug = bool(emit_debug or emit_metrics)
        self.emit_debug = emit_debug
        self.emit_metrics = bool(emit_metrics)

        g = self.graph = Graph()
        g.conv_operands = conv_operands
        if torch.cuda.is_available():
            import ctypes as C
            lds = C.c_int(0)
            dev_index = torch.device(device).index
            _lib.check(_lib.load().kfn_device_info(torch.cuda.current_device() if dev_index is None else dev_index,
                                                   None, C.byref(lds), None, 0), 'kfn_device_info')
            if lds.value > 0:
                g.lds_bytes_per_cu = int(lds.value)
        for key, val in (graph_options or {}).items():     # routing switches of kfnet_amd.graph.Graph (tests, A/B runs)
            if not hasattr(g, key):
                raise ValueError('unknown Graph option %r' % key)
            setattr(g, key, val)
        spec = KFNetDataSpec(batch_size=self.B, image_size=image_size)
        self.images = g.placeholder((self.B, self.H, self.W, 3), 'u8', name='images')
        self.state = g.placeholder((1, self.h, self.w, 4), name='last_state')
        last_coord = self.state.channels(0, 3, name='last_coord')
        last_unc = self.state.channels(3, 1, name='last_uncertainty')
        self.net = KFNet(self.images, spec)
        self.net.GetKFCoordRecursive(last_coord, last_unc, transform=self.transform,
                                     reset_period=self.reset_period, nis_gate=self.nis_gate,
                                     emit_temp=emit_debug, emit_nis=emit_debug)
        self.meas = self.net.GetMeasureCoord()[0].base          # [B,h,w,4]
        self.flow = self.net.prob.flow                           # [B*hw,1,1,2]
        self.sigma_t = self.net.oflownet.get_output_by_name('uncertainty')  # [B*hw,1,1,1]
        # chunk-level scan buffers
        hw = self.h * self.w
        T = self.max_chunk
        # The three per-frame scan inputs are WRITTEN IN PLACE: the launches that produce a batch's measurement / flow /
        # sigma_trans write straight into frame slots [d0, d0 + B) of the chunk buffers (Tensor.slide before every batch),
        # so nothing is copied between the heavy phase and the scan.  B frames of slack behind slot T: a partial batch
        # still addresses B slots, and the hipGraph-replay mode (pointers fixed at capture) parks its batch there.
        Ts = T + self.B
        self.c_flow = g.tensor((Ts, self.h, self.w, 2), name='chunk_flow')
        self.c_sigma = g.tensor((Ts, self.h, self.w, 1), name='chunk_sigma_trans')
        self.c_meas = g.tensor((Ts, self.h, self.w, 4), name='chunk_meas')
        for t, c in ((self.meas, self.c_meas), (self.flow, self.c_flow), (self.sigma_t, self.c_sigma)):
            if t.base is not None or t.ld != t.C or t.ch_off != 0:
                raise _lib.KfnError('scan input %r is not a dense root tensor: cannot be produced in place' % t.name)
            t.rebind(c.storage, 0, t.C)
        # ... and the flow-feature ring is LONG: [T + B + 1] maps instead of [B + 1].  Batch k's maps are written behind
        # batch k-1's, so "the previous frame's map" is simply the slot before -- the per-batch hand-over copy of the last
        # map to slot 0 happens once per chunk (at the start of heavy()), not once per batch.
        ring = self.net.temp_feat_maps
        self.ring_slots = T + self.B + 1
        from .graph import Storage
        ring_store = Storage(self.ring_slots * self.h * self.w * ring.C, ring.dtype)
        g.storages.append(ring_store)
        ring.rebind(ring_store, 0, ring.C)
        self._ring_pos = 0          # ring slot that holds the map of the frame before the next batch
        self.c_rec = g.tensor((T, self.h, self.w, 4), name='chunk_records')
        self.c_temp = g.tensor((T, self.h, self.w, 4), name='chunk_temp') if emit_debug else None
        self.c_nis = g.tensor((T, self.h, self.w, 3), name='chunk_nis') if emit_debug else None
        # eval.py's log line needs the raw KF estimate and the graph's (not the host's) view of reset steps
        self.c_kf = g.tensor((T, self.h, self.w, 4), name='chunk_kf_raw') if emit_metrics else None
        self.chunk_scan = KalmanScanOp(self.c_flow, self.c_sigma, self.c_meas, self.state, self.c_rec,
                                       self.c_temp, self.c_nis, S=1, T=1, H=self.h, W=self.w,
                                       reset_period=self.reset_period, min_uncertainty=self.net.min_uncertainty,
                                       nis_gate=self.nis_gate, transform=self.transform, kf_raw=self.c_kf,
                                       raw_on_reset=emit_metrics)
        g.finalize(device)
        g.load_weights(weights)
        self.lib = _lib.load()
        self.device = g.device
        self.hw = hw
        self.heavy_ops = self.net.frame_ops + self.net.pair_ops
        self.handover = self.net.scan_ops[1]
        self._staging = None
        # Two-stream schedule of the heavy phase: the measurement tower (SCoordNet, MFMA-dense
        # big tiles) on the main stream, the flow-feature tower + OFlowNet (many small
        # launches that cannot fill the chip alone) on a side stream; they only share the
        # fused first-layer kernel.
        first = self.net.frame_ops[0]
        side = [op for op in self.net.frame_ops[1:] if op in self.net.feat_tower.ops] + self.net.pair_ops
        self.main_ops = [op for op in self.net.frame_ops[1:] if op not in side]
        self.side_ops = side
        self.first_op = first
        self.side_stream = torch.cuda.Stream(device=self.device)
        self.ev_first = torch.cuda.Event()
        self.ev_side = torch.cuda.Event()
        self.two_streams = True
        # Optional hipGraph replay of the (full-batch) heavy phase: every C-ABI launch is
        # stream-ordered and allocation-free, so the whole two-stream schedule of a batch is
        # captured once and replayed with a single host call (matters when batches are
        # small / launch-bound, e.g. batch=1 latency mode).
        self.use_graph = bool(use_graph)
        self._graph = None
        self.tuned = None
        if autotune:
            g.active = (self.B, self.B)
            self.tuned = g.autotune(self.heavy_ops)
            g.active = (1, 1)

    # ------------------------------------------------------------------------------
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def upload_frames(self, frames):
        """uint8 [T,H,W,3] host array -> device tensor (H2D once; the timed region of
        bench.py starts with frames resident in HBM)."""
        frames = np.ascontiguousarray(frames)
        assert frames.dtype == np.uint8 and frames.shape[1:] == (self.H, self.W, 3)
        return self.torch.from_numpy(frames).to(self.device)

    def _set_batch_images(self, dev_frames, start, count, stream):
        """Copy `count` frames starting at `start` into the batch input (pad with the last)."""
        fb = self.H * self.W * 3
        dst = self.images.ptr
        src = dev_frames.data_ptr() + start * fb
        _lib.check(self.lib.kfn_memcpy_d2d(dst, src, count * fb, stream), 'memcpy images')

    def prime(self, dev_prev_frame):
        """Compute the flow features of the frame preceding this chunk and park them in
        ring slot 0 (multi-GPU: rank r recomputes them from the image -- 5.6 GFLOP --
        instead of receiving 614 KB from rank r-1)."""
        stream = self._stream()
        fb = self.H * self.W * 3
        _lib.check(self.lib.kfn_memcpy_d2d(self.images.ptr, dev_prev_frame.data_ptr(), fb, stream), 'prime')
        tower_ops = [op for op in self.net.frame_ops if op in self.net.feat_tower.ops]
        ring = self.net.temp_feat_maps
        ring.slide(0)
        self.graph.run(stream, tower_ops, active=(1, self.B))   # one frame only -> ring slot 1
        self._ring_pos = 1

    def heavy(self, dev_frames, T=None, dst0=0):
        """State-independent phase for frames [0,T) of `dev_frames` -> chunk scan buffers
        (frame t lands in slot dst0 + t)."""
        T = dev_frames.shape[0] if T is None else T
        if dst0 + T > self.max_chunk:
            raise ValueError('chunk of %d frames (at slot %d) exceeds max_chunk=%d' % (T, dst0, self.max_chunk))
        stream = self._stream()
        lib = self.lib
        ring = self.net.temp_feat_maps
        hw = self.hw
        per_map = hw * ring.C
        if self._ring_pos != 0:
            # once per chunk: the map of the frame before this chunk (the last slot written, or prime()'s) -> slot 0
            ring.slide(0)
            _lib.check(lib.kfn_memcpy_d2d(ring.ptr, ring.ptr + self._ring_pos * per_map * 4, per_map * 4, stream), 'ring hand-over')
            self._ring_pos = 0
        for s0 in range(0, T, self.B):
            cnt = min(self.B, T - s0)
            d0 = dst0 + s0
            self._set_batch_images(dev_frames, s0, cnt, stream)
            fixed = self.use_graph            # hipGraph replay: the captured pointers cannot move
            slot = self.max_chunk if fixed else d0
            self.meas.slide(slot * hw * 4)
            self.flow.slide(slot * hw * 2)
            self.sigma_t.slide(slot * hw)
            ring.slide(0 if fixed else self._ring_pos * per_map)
            if self.use_graph and cnt == self.B:
                self._replay_heavy_graph()
            elif self.two_streams:
                main = self.torch.cuda.current_stream(self.device)
                self.graph.run(stream, [self.first_op], active=(cnt, self.B))
                self.ev_first.record(main)
                self.side_stream.wait_event(self.ev_first)
                self.graph.run(self.side_stream.cuda_stream, self.side_ops, active=(cnt, self.B))
                self.ev_side.record(self.side_stream)
                self.graph.run(stream, self.main_ops, active=(cnt, self.B))
                main.wait_event(self.ev_side)
            else:
                self.graph.run(stream, self.heavy_ops, active=(cnt, self.B))   # partial batches cost their share
            if fixed:
                _lib.check(lib.kfn_memcpy_d2d(self.c_flow.ptr + d0 * hw * 8, self.flow.ptr, cnt * hw * 8, stream), 'cp flow')
                _lib.check(lib.kfn_memcpy_d2d(self.c_sigma.ptr + d0 * hw * 4, self.sigma_t.ptr, cnt * hw * 4, stream), 'cp sig')
                _lib.check(lib.kfn_memcpy_d2d(self.c_meas.ptr + d0 * hw * 16, self.meas.ptr, cnt * hw * 16, stream), 'cp meas')
                _lib.check(lib.kfn_memcpy_d2d(ring.ptr, ring.ptr + cnt * per_map * 4, per_map * 4, stream), 'ring hand-over')
            else:
                self._ring_pos += cnt

    def _launch_heavy_full(self, stream_ptr, main):
        """Full-batch heavy phase on (main, side) streams; used directly and under capture."""
        B = self.B
        self.graph.run(stream_ptr, [self.first_op], active=(B, B))
        self.ev_first.record(main)
        self.side_stream.wait_event(self.ev_first)
        self.graph.run(self.side_stream.cuda_stream, self.side_ops, active=(B, B))
        self.ev_side.record(self.side_stream)
        self.graph.run(stream_ptr, self.main_ops, active=(B, B))
        main.wait_event(self.ev_side)

    def _replay_heavy_graph(self):
        torch = self.torch
        if self._graph is None:
            main = torch.cuda.current_stream(self.device)
            self._launch_heavy_full(main.cuda_stream, main)   # warm-up outside capture (one-time attributes)
            torch.cuda.synchronize()
            cap = torch.cuda.Stream(device=self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    cur = torch.cuda.current_stream(self.device)
                    self._launch_heavy_full(cur.cuda_stream, cur)
            self._graph = g
        self._graph.replay()

    def scan(self, T, t0=0):
        """Sequential phase: one launch over the T frames staged by `heavy`."""
        self.chunk_scan.T = int(T)
        self.chunk_scan.t0 = int(t0)
        self.chunk_scan.launch(self.lib, self._stream())

    def process(self, dev_frames, t0=0):
        """heavy + scan; returns the device records tensor view [T,h,w,4] (torch)."""
        T = dev_frames.shape[0]
        if T == 0:   # empty chunk: nothing to do, state untouched
            return self.records(0)
        self.heavy(dev_frames, T)
        self.scan(T, t0)
        return self.records(T)

    def process_sequences(self, dev_seqs):
        """Batch-of-sequences (BASELINE config 5): `dev_seqs` is a uint8 device tensor
        [S,T,H,W,3] of S independent sequences.  The heavy phase runs sequence after
        sequence (each restarts the feature ring at its reset frame 0); then ONE scan launch
        advances all S Kalman filters in lockstep, one workgroup per sequence.
        Returns the records [S,T,h,w,4] (torch device tensor)."""
        torch = self.torch
        S, T = int(dev_seqs.shape[0]), int(dev_seqs.shape[1])
        if S * T > self.max_chunk:
            raise ValueError('S*T = %d exceeds max_chunk=%d' % (S * T, self.max_chunk))
        if self.reset_period <= 0 or 0 % self.reset_period != 0:
            raise ValueError('batch-of-sequences needs every sequence to start on a reset frame')
        hw = self.hw
        lib, stream = self.lib, self._stream()
        # stage every sequence's scan inputs at [s*T, (s+1)*T) of the chunk buffers
        for s in range(S):
            self.heavy(dev_seqs[s], T, dst0=s * T)
        # scan buffers of this (S, T): allocated once and kept (no allocation inside a timed / captured region); the returned
        # records are a view of the cached buffer, valid until the next call -- the contract of records() for process()
        key = (S, T)
        if getattr(self, '_seq_key', None) != key:
            d = _lib.KalmanDesc(S=S, T=T, H=self.h, W=self.w, t0=0, reset_period=self.reset_period,
                                min_uncertainty=self.net.min_uncertainty, nis_gate=self.nis_gate,
                                has_transform=int(self.transform is not None))
            if self.transform is not None:
                for i, v in enumerate(np.asarray(self.transform, np.float32)[:3, :4].reshape(-1)):
                    d.transform[i] = float(v)
            import ctypes as C
            need = C.c_size_t(0)
            _lib.check(lib.kfn_kalman_scan_scratch_bytes(C.byref(d), C.byref(need)), 'kfn_kalman_scan_scratch_bytes')
            self._seq_bufs = (d, torch.zeros(S * hw * 4, device=self.device), torch.empty(S * T * hw * 4, device=self.device),
                              torch.empty((need.value + 3) // 4, device=self.device) if need.value else None)
            self._seq_key = key
        import ctypes as C
        d, states, rec, scratch = self._seq_bufs
        _lib.check(lib.kfn_memset(states.data_ptr(), 0, S * hw * 16, stream), 'kfn_memset states')
        _lib.check(lib.kfn_kalman_scan(C.byref(d), self.c_flow.ptr, self.c_sigma.ptr, self.c_meas.ptr,
                                       states.data_ptr(), rec.data_ptr(), None, None,
                                       scratch.data_ptr() if scratch is not None else None, stream), 'kfn_kalman_scan')
        return rec.view(S, T, self.h, self.w, 4)

    def records(self, T):
        buf = self.c_rec.root_storage.buf
        return buf[:T * self.hw * 4].view(T, self.h, self.w, 4)

    def debug(self, T):
        out = {}
        if self.c_temp is not None:
            out['temp'] = self.c_temp.root_storage.buf[:T * self.hw * 4].view(T, self.h, self.w, 4).cpu().numpy()
            out['nis'] = self.c_nis.root_storage.buf[:T * self.hw * 3].view(T, self.h, self.w, 3).cpu().numpy()
        out['flow'] = self.c_flow.root_storage.buf[:T * self.hw * 2].view(T, self.h, self.w, 2).cpu().numpy()
        out['sigma_trans'] = self.c_sigma.root_storage.buf[:T * self.hw].view(T, self.h, self.w, 1).cpu().numpy()
        out['meas'] = self.c_meas.root_storage.buf[:T * self.hw * 4].view(T, self.h, self.w, 4).cpu().numpy()
        return out

    def get_state(self):
        return self.state.root_storage.buf  # torch [hw*4] (x,y,z,sigma), the message rank->rank

    def flops_per_frame(self):
        return sum(op.flops() for op in self.heavy_ops if hasattr(op, 'flops')) / self.B
