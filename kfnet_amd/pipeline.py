"""Host side of the image stream: decode -> pinned host chunks -> HBM -> records -> host.

The reference feeds its graph through TF queue runners (KFNet/train.py:195-239:
`string_input_producer`, `tf.image.decode_png(channels=3)`, `tf.train.batch`, a 2-frame
batch per step) and fetches every result synchronously (`sess.run`, KFNet/eval.py:77-104).
Here a sequence is cut into chunks:

* `ChunkLoader`  -- a background thread fills page-locked uint8 buffers [n,H,W,3], decoding
  the PNGs of chunk k+1 on a small thread pool (PIL releases the GIL while inflating) while
  chunk k is on the GPU;
* `StreamedSequence` -- three HIP streams: the upload of chunk k+1 (0.92 MB per frame) and
  the download of chunk k-1's records (76.8 KB per frame) run beside the compute of chunk
  k, so PCIe never sits on the critical path.  The Kalman state and the flow-feature ring
  stay in HBM from chunk to chunk (KFNetEngine).

Nothing here touches the numbers: records are bit-identical to `KFNetEngine.process` on a
resident sequence (tests/test_gpu_e2e.py).
"""
import os
import queue
import threading
import time

import numpy as np


def decode_image(path, image_size):
    """tf.image.decode_png(channels=3) (KFNet/train.py:213-217): uint8 RGB [H,W,3]."""
    from PIL import Image
    H, W = image_size
    with Image.open(path) as im:
        a = np.asarray(im.convert('RGB'))
    if a.shape[:2] != (H, W):
        raise ValueError('%s is %dx%d, expected %dx%d' % (path, a.shape[0], a.shape[1], H, W))
    return a


def decode_png_batch(paths, dst, image_size, threads):
    """PNG files -> dst [n,H,W,3] uint8 (a numpy view of the staging buffer) on the library's own threads
    (`kfn_decode_png_rgb8`, csrc/kfn_png.hip): one call per chunk, no interpreter lock while it runs.  Files the native
    decoder does not take (interlaced, 16-bit) go through PIL; a broken file raises ValueError naming it."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(paths)
    if n == 0:
        return
    H, W = image_size
    assert dst.dtype == np.uint8 and dst.shape[1:] == (H, W, 3) and dst.shape[0] >= n and dst.flags['C_CONTIGUOUS']
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    status = (C.c_int * n)()
    rc = lib.kfn_decode_png_rgb8(arr, n, H, W, dst.ctypes.data, status, int(threads))
    if rc != 0:
        raise ValueError(lib.kfn_last_error().decode())
    for i in range(n):
        if status[i] == _lib.PNG_UNSUPPORTED:
            try:
                dst[i] = decode_image(paths[i], image_size)
            except ValueError:
                raise
            except Exception as e:      # PIL's own error types (UnidentifiedImageError, OSError, ...): one type for the caller
                raise ValueError('%s: %s' % (paths[i], e))


def _native_png_available():
    if os.environ.get('KFN_PNG_DECODER', 'native') == 'pil':     # A/B switch (bench runs, tools/experiments)
        return False
    try:
        from . import _lib
        _lib.load()
        return True
    except Exception:      # library not built (a source checkout without hipcc): PIL decodes
        return False


def _host_buffer(shape, dtype, pinned):
    """uint8/float32 staging buffer; page-locked when a GPU runtime is there to lock it."""
    import torch
    if pinned and torch.cuda.is_available():
        return torch.empty(shape, dtype=dtype, pin_memory=True)     # (not empty().pin_memory(): that allocates twice and copies)
    return torch.empty(shape, dtype=dtype)


class ChunkLoader(object):
    """Iterates over (first_index, uint8 host tensor [n,H,W,3]) for consecutive chunks of a
    sequence.  `source` is a list of image paths (decoded here) or a uint8 array
    [T,H,W,3] already in memory (copied through the same staging buffers so that both kinds
    of input take the same path).  `depth` staging buffers rotate: a chunk handed out stays
    valid until `depth - 1` further chunks have been requested (a StreamedSequence with N chunks in
    flight needs depth >= N + 1: a buffer is then recycled only after the records of the chunk uploaded
    from it have been handed out)."""

    def __init__(self, source, image_size, chunk, workers=8, depth=4, pinned=True, decode=decode_image, first_chunk=None,
                 native=True):
        """`first_chunk` (< chunk): length of the FIRST chunk only -- its decode is the one nothing overlaps, so a short
        one gets the GPU going while the first full chunk is still decoding -- or a sequence of lengths, a RAMP (e.g. (8, 16)
        in front of chunks of 32: decoding n frames on n threads takes about as long as decoding one, computing them takes n
        frame times, so each ramp chunk's compute covers the next one's decode); every later chunk is `chunk` frames (deep
        launch queues: the consumer thread shares the interpreter with the decode and writer threads)."""
        import torch
        self.image_size = tuple(image_size)
        self.chunk = int(chunk)
        if self.chunk <= 0:
            raise ValueError('chunk must be positive')
        ramp = [] if not first_chunk else ([first_chunk] if isinstance(first_chunk, (int, np.integer)) else list(first_chunk))
        self.ramp = [max(1, min(int(r), self.chunk)) for r in ramp]
        self.ramp = [r for r in self.ramp if r < self.chunk]           # a "ramp" chunk as long as a chunk is just a chunk
        self.first_chunk = self.ramp[0] if self.ramp else self.chunk
        self.source = source
        self.T = len(source)
        self.workers = max(1, int(workers))
        self.decode = decode
        # PNG lists are decoded by the C-ABI library when it is there (kfn_decode_png_rgb8); `native=False` or a custom `decode`
        # callable selects the Python thread pool (PIL)
        self.native = bool(native) and _native_png_available()
        H, W = self.image_size
        if isinstance(source, np.ndarray):
            if source.dtype != np.uint8 or source.shape[1:] != (H, W, 3):
                raise ValueError('frames must be uint8 [T,%d,%d,3]' % (H, W))
        self.depth = max(2, int(depth))
        # Page-locking costs ~0.3 ms per MB: three 128-frame buffers (354 MB) would add > 0.1 s in front of the first frame.
        # Only the FIRST chunk's own small buffer is locked here; the rotating full-size buffers are allocated by the producer
        # thread when it first needs them, i.e. beside the first chunk's compute.
        self.pinned = bool(pinned)
        self._torch = torch
        self.bufs = [None] * self.depth
        # ramp chunks: their own small buffers, outside the rotation (valid for the loader's whole life); the first is locked
        # here, the others by the producer thread
        self.ramp_bufs = [None] * len(self.ramp)
        if self.ramp and self.T > 0:
            self.ramp_bufs[0] = _host_buffer((self.ramp[0], H, W, 3), torch.uint8, pinned)
        self.free = queue.Queue()
        for i in range(self.depth):
            self.free.put(i)
        self.ready = queue.Queue()
        self.thread = None
        self._held = []
        self._pool = None

    def bounds(self):
        """[(lo, hi)] of the chunks, in order."""
        out, lo = [], 0
        while lo < self.T:
            hi = min(self.T, lo + (self.ramp[len(out)] if len(out) < len(self.ramp) else self.chunk))
            out.append((lo, hi))
            lo = hi
        return out

    def __len__(self):
        return len(self.bounds())

    def _fill(self, buf, lo, hi):
        dst = buf.numpy()
        if isinstance(self.source, np.ndarray):
            dst[:hi - lo] = self.source[lo:hi]
            return
        if self.decode is decode_image and self.native:      # the default decoder: the library's threads, one call
            decode_png_batch(self.source[lo:hi], dst, self.image_size, self.workers)
            return
        def one(i):
            dst[i - lo] = self.decode(self.source[i], self.image_size)
        if self.workers == 1:
            for i in range(lo, hi):
                one(i)
        else:
            if self._pool is None:      # one pool for the whole sequence (not one per chunk)
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.workers, thread_name_prefix='kfnet-decode')
            list(self._pool.map(one, range(lo, hi)))   # list(): re-raise decode errors here

    def _produce(self):
        # the producer thread's wall time (seconds): page-locking staging buffers, decoding / copying, waiting for a free buffer
        st = self.stats = {'alloc': 0.0, 'fill': 0.0, 'free_wait': 0.0}
        clock = time.perf_counter
        try:
            H, W = self.image_size
            for k, (lo, hi) in enumerate(self.bounds()):
                if k < len(self.ramp):                          # its own buffer, outside the rotation
                    t0 = clock()
                    if self.ramp_bufs[k] is None:
                        self.ramp_bufs[k] = _host_buffer((self.ramp[k], H, W, 3), self._torch.uint8, self.pinned)
                    t1 = clock()
                    self._fill(self.ramp_bufs[k], lo, hi)
                    st['alloc'] += t1 - t0
                    st['fill'] += clock() - t1
                    self.ready.put((lo, hi - lo, -1 - k, None))
                    continue
                t0 = clock()
                b = self.free.get()
                if b is None:
                    return
                t1 = clock()
                if self.bufs[b] is None:
                    self.bufs[b] = _host_buffer((self.chunk, H, W, 3), self._torch.uint8, self.pinned)
                t2 = clock()
                self._fill(self.bufs[b], lo, hi)
                st['free_wait'] += t1 - t0
                st['alloc'] += t2 - t1
                st['fill'] += clock() - t2
                self.ready.put((lo, hi - lo, b, None))
            self.ready.put((None, 0, None, None))
        except BaseException as e:   # hand decode errors to the consumer thread
            self.ready.put((None, 0, None, e))
        finally:
            if self._pool is not None:
                self._pool.shutdown(wait=False)
                self._pool = None

    def __iter__(self):
        if self.thread is not None:
            raise RuntimeError('a ChunkLoader can be iterated once')
        self.thread = threading.Thread(target=self._produce, name='kfnet-chunk-loader', daemon=True)
        self.thread.start()
        try:
            while True:
                lo, n, b, err = self.ready.get()
                if err is not None:
                    raise err
                if lo is None:
                    return
                if b < 0:
                    yield lo, self.ramp_bufs[-1 - b][:n]
                    continue
                # recycle the buffer handed out depth-1 chunks ago
                self._held.append(b)
                if len(self._held) >= self.depth:
                    self.free.put(self._held.pop(0))
                yield lo, self.bufs[b][:n]
        finally:
            self.free.put(None)   # unblock the producer if the consumer stopped early


class StreamedSequence(object):
    """Runs host-resident chunks through a `KFNetEngine` with uploads, compute and downloads
    overlapped.  `run(chunks)` takes an iterable of (first_index, uint8 host tensor
    [n,H,W,3]) -- e.g. a `ChunkLoader` -- and yields (first_index, float32 numpy records
    [n,h,w,4]); a yielded array is a view of a rotating pinned buffer and is valid until the
    generator is advanced again."""

    def __init__(self, eng, chunk=None, depth=3):
        """`depth` chunks may be in flight on the GPU (uploaded / computing / downloading) before the oldest one's records
        are handed out: 2 = rounds 2-4 (the host fetches and enqueues chunk k+1 while chunk k computes); 3 keeps a second
        chunk queued behind the running one, so a host stall of one chunk's compute time (the consumer thread shares the
        interpreter with the decode and writer threads) no longer idles the GPU."""
        torch = eng.torch
        self.eng = eng
        self.chunk = int(chunk or eng.max_chunk)
        if self.chunk > eng.max_chunk:
            raise ValueError('chunk %d exceeds the engine\'s max_chunk %d' % (self.chunk, eng.max_chunk))
        self.depth = max(2, int(depth))
        dev = eng.device
        n = self.depth
        self.up = torch.cuda.Stream(device=dev)
        self.down = torch.cuda.Stream(device=dev)
        self.dev_frames = [torch.empty((self.chunk, eng.H, eng.W, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
        self.dev_rec = [torch.empty((self.chunk, eng.h, eng.w, 4), dtype=torch.float32, device=dev) for _ in range(n)]
        self.host_rec = [_host_buffer((self.chunk, eng.h, eng.w, 4), torch.float32, True) for _ in range(n)]
        self.ev_up = [torch.cuda.Event() for _ in range(n)]
        self.ev_done = [None] * n
        self.ev_down = [None] * n

    def run(self, chunks, after_process=None):
        """`after_process(k, first_index, n)` (optional) is called right after chunk k's compute has been
        enqueued, on the compute stream -- e.g. to enqueue a reduction over the chunk's scan buffers."""
        torch = self.eng.torch
        eng = self.eng
        main = torch.cuda.current_stream(eng.device)
        pending = []     # [(first_index, n, slot)] whose downloads are in flight, oldest first
        # where the consumer thread's wall time goes (seconds): waiting for the loader's next chunk, enqueueing a chunk's work,
        # waiting for a chunk's records; `consumer` (the caller's time between two yields) is the rest of the wall time
        st = self.stats = {'loader_wait': 0.0, 'enqueue': 0.0, 'records_wait': 0.0, 'chunks': 0, 'loader_wait_first': 0.0}
        # a staging buffer of the loader must not be refilled while its upload is still in flight: with `depth` chunks in flight
        # here the loader needs depth + 1 rotating buffers (ChunkLoader's docstring) -- checked, not assumed
        ld = getattr(chunks, 'depth', None)
        if ld is not None and int(ld) < self.depth + 1:
            raise ValueError('StreamedSequence(depth=%d) needs a loader with at least %d staging buffers, got depth=%d: a buffer '
                             'would be refilled while its host-to-device copy is still running' % (self.depth, self.depth + 1, int(ld)))
        it = iter(chunks)
        k = -1
        while True:
            t0 = time.perf_counter()
            try:
                lo, host = next(it)
            except StopIteration:
                break
            k += 1
            t1 = time.perf_counter()
            st['loader_wait'] += t1 - t0
            if k == 0:
                st['loader_wait_first'] = t1 - t0
            st['chunks'] += 1
            n = int(host.shape[0])
            if n > self.chunk:
                raise ValueError('chunk of %d frames exceeds %d' % (n, self.chunk))
            b = k % self.depth
            with torch.cuda.stream(self.up):
                if self.ev_done[b] is not None:
                    self.up.wait_event(self.ev_done[b])      # compute of chunk k-depth has read dev_frames[b]
                self.dev_frames[b][:n].copy_(host, non_blocking=True)
                self.ev_up[b].record(self.up)
            main.wait_event(self.ev_up[b])
            if self.ev_down[b] is not None:
                main.wait_event(self.ev_down[b])             # download of chunk k-depth has read dev_rec[b]
            rec = eng.process(self.dev_frames[b][:n], t0=lo)
            if after_process is not None:
                after_process(k, lo, n)
            self.dev_rec[b][:n].copy_(rec)                   # the engine's record buffer is reused by chunk k+1
            self.ev_done[b] = torch.cuda.Event()
            self.ev_done[b].record(main)
            with torch.cuda.stream(self.down):
                self.down.wait_event(self.ev_done[b])
                self.host_rec[b][:n].copy_(self.dev_rec[b][:n], non_blocking=True)
                self.ev_down[b] = torch.cuda.Event()
                self.ev_down[b].record(self.down)
            pending.append((lo, n, b))
            t2 = time.perf_counter()
            st['enqueue'] += t2 - t1
            if len(pending) >= self.depth:                   # the slot chunk k+1 will take must be handed out first
                plo, pn, pb = pending.pop(0)
                self.ev_down[pb].synchronize()
                st['records_wait'] += time.perf_counter() - t2
                yield plo, self.host_rec[pb][:pn].numpy()
        while pending:
            plo, pn, pb = pending.pop(0)
            t2 = time.perf_counter()
            self.ev_down[pb].synchronize()
            st['records_wait'] += time.perf_counter() - t2
            yield plo, self.host_rec[pb][:pn].numpy()
