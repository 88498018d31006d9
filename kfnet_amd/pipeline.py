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
import queue
import threading

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


def _host_buffer(shape, dtype, pinned):
    """uint8/float32 staging buffer; page-locked when a GPU runtime is there to lock it."""
    import torch
    t = torch.empty(shape, dtype=dtype)
    if pinned and torch.cuda.is_available():
        t = t.pin_memory()
    return t


class ChunkLoader(object):
    """Iterates over (first_index, uint8 host tensor [n,H,W,3]) for consecutive chunks of a
    sequence.  `source` is a list of image paths (decoded here) or a uint8 array
    [T,H,W,3] already in memory (copied through the same staging buffers so that both kinds
    of input take the same path).  `depth` staging buffers rotate: a chunk handed out stays
    valid until `depth - 1` further chunks have been requested (a StreamedSequence with N chunks in
    flight needs depth >= N + 1: a buffer is then recycled only after the records of the chunk uploaded
    from it have been handed out)."""

    def __init__(self, source, image_size, chunk, workers=8, depth=4, pinned=True, decode=decode_image, first_chunk=None):
        """`first_chunk` (< chunk): length of the FIRST chunk only -- its decode is the one nothing overlaps, so a short
        one (a tower batch) gets the GPU going while the first full chunk is still decoding; every later chunk is `chunk`
        frames (deep launch queues: the consumer thread shares the interpreter with the decode and writer threads)."""
        import torch
        self.image_size = tuple(image_size)
        self.chunk = int(chunk)
        if self.chunk <= 0:
            raise ValueError('chunk must be positive')
        self.first_chunk = self.chunk if not first_chunk else max(1, min(int(first_chunk), self.chunk))
        self.source = source
        self.T = len(source)
        self.workers = max(1, int(workers))
        self.decode = decode
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
        self.first_buf = None
        if self.first_chunk < self.chunk and self.T > 0:
            self.first_buf = _host_buffer((self.first_chunk, H, W, 3), torch.uint8, pinned)
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
            hi = min(self.T, lo + (self.first_chunk if lo == 0 else self.chunk))
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
        try:
            H, W = self.image_size
            for lo, hi in self.bounds():
                if lo == 0 and self.first_buf is not None:      # its own buffer, outside the rotation
                    self._fill(self.first_buf, lo, hi)
                    self.ready.put((lo, hi - lo, -1, None))
                    continue
                b = self.free.get()
                if b is None:
                    return
                if self.bufs[b] is None:
                    self.bufs[b] = _host_buffer((self.chunk, H, W, 3), self._torch.uint8, self.pinned)
                self._fill(self.bufs[b], lo, hi)
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
                    yield lo, self.first_buf[:n]
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
        for k, (lo, host) in enumerate(chunks):
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
            if len(pending) >= self.depth:                   # the slot chunk k+1 will take must be handed out first
                plo, pn, pb = pending.pop(0)
                self.ev_down[pb].synchronize()
                yield plo, self.host_rec[pb][:pn].numpy()
        while pending:
            plo, pn, pb = pending.pop(0)
            self.ev_down[pb].synchronize()
            yield plo, self.host_rec[pb][:pn].numpy()
