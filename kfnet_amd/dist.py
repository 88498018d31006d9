"""Frame-sharded multi-GPU execution (one process per GPU, torch.distributed; backend
'nccl' is RCCL over xGMI on ROCm, 'gloo' on CPU for tests).

The reference has no parallelism at all.  The path shards naturally
(F7): everything heavy depends on images only, so a T-frame stream is cut into `world`
contiguous chunks; the only exchange is the recurrent Kalman state -- one [h,w,4] fp32
message (76.8 KB at 60x80) from rank r to rank r+1, sent when r has finished its scan.
A chunk that starts on a reset boundary (global index % reset_period == 0) needs no
message at all.  There is no collective on the data path.
"""


def needs_state(first_frame, reset_period):
    """Does a chunk starting at global frame `first_frame` depend on the previous chunk?"""
    if first_frame == 0:
        return False
    return not (reset_period > 0 and first_frame % reset_period == 0)


def chunk_bounds(total_frames, world, rank):
    """Contiguous, balanced chunks: the first (total % world) ranks get one more frame."""
    q, r = divmod(total_frames, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def run_chunk(eng, dev_frames, first_frame, rank, world, dist=None, dev_prev_frame=None):
    """Process this rank's chunk.  `dist` is the initialised torch.distributed module (or
    None for a single process)."""
    T = dev_frames.shape[0]
    dep = needs_state(first_frame, eng.reset_period)
    if dep:
        if dev_prev_frame is None:
            raise ValueError('chunk starting at frame %d needs the preceding frame' % first_frame)
        eng.prime(dev_prev_frame)          # flow features of frame first_frame-1, recomputed locally
    eng.heavy(dev_frames, T)               # state-independent: no waiting on other ranks
    # RCCL ('nccl') moves the device tensor directly over xGMI; with the 'gloo' backend (CPU
    # tests, or several ranks sharing one GPU) the 76.8 KB message is staged through the host.
    via_host = dist is not None and world > 1 and dist.get_backend() != 'nccl'
    if dist is not None and world > 1 and rank > 0:
        state = eng.get_state()
        buf = state.cpu() if via_host else (state if dep else state.clone())
        dist.recv(buf, src=rank - 1)       # the 76.8 KB hand-off
        if dep and via_host:
            state.copy_(buf)
        # (a chunk that resets on its first frame still receives, to keep the send/recv
        #  pairing uniform, but ignores the message)
    eng.scan(T, first_frame)
    if dist is not None and world > 1 and rank + 1 < world:
        state = eng.get_state()
        dist.send(state.cpu() if via_host else state, dst=rank + 1)
    return eng.records(T)


def scan_sharded_host(states_in, chunk_scan_fn, rank, world, dist, state_buf, first_frame, reset_period):
    """Backend-agnostic skeleton of the hand-off used by the gloo CPU tests: receive the
    state if the chunk depends on it, run `chunk_scan_fn(state_buf)` (which updates
    state_buf in place), then send it on."""
    dep = needs_state(first_frame, reset_period)
    if world > 1 and rank > 0:
        if dep:
            dist.recv(state_buf, src=rank - 1)
        else:
            tmp = state_buf.clone()
            dist.recv(tmp, src=rank - 1)
    chunk_scan_fn(state_buf)
    if world > 1 and rank + 1 < world:
        dist.send(state_buf, dst=rank + 1)
    return state_buf
