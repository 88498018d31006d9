"""Frame-sharded multi-GPU execution (one process per GPU).

The reference has no parallelism at all (its only device placement is
KFNet/train.py:375).  The path shards naturally (F7): everything heavy depends on images
only, so a T-frame stream is cut into `world` CONTIGUOUS chunks, rank r owning chunk r; the
only exchange is the recurrent Kalman state -- one [h,w,4] fp32 message (76.8 KB at 60x80,
what KFNet/eval.py:103-104 feeds back through SetVariableByName) from rank r to rank r+1,
sent when r has finished its scan.  There is no collective on the data path.

Pairing rule (both sides evaluate it on the same frame index, so no message is ever
unmatched): the chunk that starts at global frame f receives iff `needs_state(f)`; the
chunk that ends at f (exclusive) sends iff rank+1 exists and `needs_state(f)`.  A chunk that
starts on a reset boundary (f % reset_period == 0, KFNet/eval.py:94) is independent of its
predecessor: it posts no receive and its predecessor no send, so it never waits.

Transports (`StateLink`): `RcclLink` = the C ABI (kfn_comm_init / kfn_send_state /
kfn_recv_state: ncclSend/ncclRecv over xGMI, stream-ordered with the scan, device to device);
`TorchLink` = torch.distributed send/recv (backend 'nccl' is RCCL too; 'gloo' stages the
message through the host -- CPU tests, or several ranks sharing one GPU).
"""
import ctypes as C

from . import _lib


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


def handoff_plan(first_frame, n_frames, rank, world, reset_period):
    """(recv_from_prev, send_to_next) for the chunk [first_frame, first_frame + n_frames)."""
    recv = world > 1 and rank > 0 and needs_state(first_frame, reset_period)
    send = world > 1 and rank + 1 < world and needs_state(first_frame + n_frames, reset_period)
    return recv, send


class TorchLink(object):
    """State hand-off through an initialised torch.distributed process group."""

    def __init__(self, dist):
        self.dist = dist
        self.via_host = dist.get_backend() != 'nccl'
        self.name = 'torch.distributed/' + dist.get_backend()

    def recv(self, state, src, eng=None):
        if self.via_host:
            buf = state.cpu()
            self.dist.recv(buf, src=src)
            state.copy_(buf)
        else:
            self.dist.recv(state, src=src)

    def send(self, state, dst, eng=None):
        self.dist.send(state.cpu() if self.via_host else state, dst=dst)

    def close(self):
        pass


class RcclLink(object):
    """State hand-off through the C ABI (include/kfnet_hip.h: kfn_comm_*, kfn_send_state,
    kfn_recv_state).  `dist` (any initialised backend) is used once, to distribute rank 0's
    128-byte RCCL unique id; the data path never touches torch.distributed."""

    def __init__(self, rank, world, device_index, dist=None, unique_id=None, grid_hw=None):
        self.lib = _lib.load()
        self.rank, self.world = rank, world
        self.name = 'C-ABI kfn_send_state/kfn_recv_state (RCCL ncclSend/ncclRecv)'
        if unique_id is None:
            ids = [None]
            if rank == 0:
                buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
                _lib.check(self.lib.kfn_comm_unique_id(buf, _lib.COMM_ID_BYTES), 'kfn_comm_unique_id')
                ids[0] = bytes(buf.raw)
            if world > 1:
                if dist is None:
                    raise ValueError('RcclLink needs torch.distributed (or an explicit unique_id) to share the id')
                dist.broadcast_object_list(ids, src=0)
            unique_id = ids[0]
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError('unique id must be %d bytes' % _lib.COMM_ID_BYTES)
        self.comm = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        _lib.check(self.lib.kfn_comm_init(C.byref(self.comm), rank, world, idbuf, int(device_index)), 'kfn_comm_init')
        self.grid_hw = grid_hw

    def _hw(self, eng):
        return (eng.h, eng.w) if eng is not None else self.grid_hw

    def recv(self, state, src, eng=None):
        h, w = self._hw(eng)
        stream = eng._stream() if eng is not None else None
        _lib.check(self.lib.kfn_recv_state(self.comm, src, state.data_ptr(), h, w, stream), 'kfn_recv_state')

    def send(self, state, dst, eng=None):
        h, w = self._hw(eng)
        stream = eng._stream() if eng is not None else None
        _lib.check(self.lib.kfn_send_state(self.comm, dst, state.data_ptr(), h, w, stream), 'kfn_send_state')

    def close(self):
        if self.comm:
            _lib.check(self.lib.kfn_comm_destroy(self.comm), 'kfn_comm_destroy')
            self.comm = C.c_void_p()


class LoopbackLink(object):
    """N "ranks" inside ONE process (SURVEY.md §4.2 'multi-GPU without a cluster'): the ranks'
    chunks are run one after the other and the state travels through a dict -- the same
    run_chunk code path and pairing rule as the real transports, used to check that a sharded
    run is bit-identical to a single pass."""

    def __init__(self, mailbox, rank):
        self.mailbox, self.rank = mailbox, rank
        self.name = 'in-process loopback'

    def recv(self, state, src, eng=None):
        state.copy_(self.mailbox.pop((src, self.rank)))   # KeyError = unmatched receive

    def send(self, state, dst, eng=None):
        assert (self.rank, dst) not in self.mailbox, 'unmatched send'
        self.mailbox[(self.rank, dst)] = state.clone()

    def close(self):
        pass


def _all_ranks_ok(dist, ok, device_index):
    """MIN over ranks of a success flag (collective: every rank of the group must call it)."""
    import torch
    on_gpu = dist.get_backend() == 'nccl' and torch.cuda.is_available()
    dev = 'cpu'
    if on_gpu:
        # The flag must reach the all_reduce on EVERY rank, above all on the rank whose device index is the thing that
        # failed: an index that is no integer or out of range falls back to the process's current device (where the
        # nccl process group lives anyway) instead of raising here and leaving the peers blocked in the reduction.
        try:
            idx = int(device_index)
        except (TypeError, ValueError):
            idx = -1
        if not (0 <= idx < torch.cuda.device_count()):
            idx = torch.cuda.current_device()
        dev = torch.device('cuda', idx)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def make_link(dist, rank, world, device_index, prefer='auto'):
    """The transport for this process group: the C-ABI RCCL link when every rank owns a GPU
    (backend 'nccl'), torch.distributed otherwise.  prefer = 'auto' | 'cabi' | 'torch'.

    The choice is COLLECTIVE: every rank runs the same sequence of group operations whatever
    fails where, so the ranks can never end up on different transports (a rank that cannot bind
    librccl must not leave the others blocked in the id broadcast, and a group that mixes
    ncclSend with torch.distributed.recv hangs at the first hand-off).
      1. every rank checks its LOCAL preconditions (library loads, device index valid; rank 0 obtains the unique id
         -- the only call that starts RCCL's bootstrap listener) -> all_reduce(MIN);
      2. rank 0 broadcasts its unique id, every rank runs kfn_comm_init -> all_reduce(MIN);
      3. only if every rank succeeded is RcclLink used; otherwise every rank closes what it
         opened and takes TorchLink -- or, with prefer='cabi', every rank raises.
    Remaining window: ncclCommInitRank is a collective inside RCCL; a rank that dies INSIDE it (not before: step 1
    covers everything local) leaves the others to RCCL's own timeout."""
    if dist is None or world <= 1:
        return None
    if prefer == 'torch' or dist.get_backend() != 'nccl':
        if prefer == 'cabi':
            raise _lib.KfnError("the C-ABI RCCL link needs backend 'nccl' (one GPU per rank), got %r"
                                % dist.get_backend())
        return TorchLink(dist)
    why = None
    lib = None
    uid = None
    try:
        # every LOCAL precondition is checked here, before the first group operation, so that step 2
        # (ncclCommInitRank, itself collective) can only fail collectively: the library loads with the entry points
        # bound, the device index is valid, and -- on rank 0 only: every call opens a bootstrap listener -- RCCL
        # hands out a unique id (the other ranks only bind librccl: kfn_comm_available)
        import torch
        lib = _lib.load()
        if not (0 <= int(device_index) < torch.cuda.device_count()):
            raise _lib.KfnError('device index %r out of range' % (device_index,))
        _lib.check(lib.kfn_comm_available(), 'kfn_comm_available')      # dlopen(librccl) + every symbol; no socket, no device
        if rank == 0:
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(lib.kfn_comm_unique_id(buf, _lib.COMM_ID_BYTES), 'kfn_comm_unique_id')
            uid = bytes(buf.raw)
    except (_lib.KfnError, OSError, TypeError, ValueError) as e:     # (int(None) / int('x'): a bad index is a local failure too)
        why = e
    link = None
    if _all_ranks_ok(dist, why is None, device_index):
        ids = [uid if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            if not isinstance(ids[0], (bytes, bytearray)) or len(ids[0]) != _lib.COMM_ID_BYTES:
                raise ValueError('rank 0 sent no usable RCCL id')     # the same on every rank: fails collectively
            link = RcclLink(rank, world, device_index, unique_id=ids[0])
        except (_lib.KfnError, OSError, ValueError) as e:
            why = e
        if _all_ranks_ok(dist, link is not None, device_index):
            return link
        if link is not None:
            link.close()
    if prefer == 'cabi':
        raise _lib.KfnError('C-ABI RCCL link unavailable on at least one rank (this rank: %s)'
                            % (why if why is not None else 'ok'))
    return TorchLink(dist)


# ---- block-cyclic sharding (round 5) -----------------------------------------------------------------------------------------
# Contiguous chunks put the whole serial part at the END of the job: every rank finishes its heavy phase at the same time and
# then the `world` scans run one after another (each waits for its predecessor's state).  With blocks of `block` frames dealt
# round-robin (block j -> rank j % world) the state hops once per block and a rank scans block j while the others are still
# in the heavy phase of theirs: what is left exposed is the start-up skew, (world - 1) x (one block's scan + one hop), instead
# of (world - 1) x (one chunk's scan + one hop).  Same arithmetic, same pairing rule -- evaluated per block -- so the records
# are bit-identical to a single pass and to the contiguous sharding.
def cyclic_blocks(total_frames, block, rank, world):
    """[(j, lo, hi)] of the blocks rank `rank` owns, ascending."""
    if block <= 0:
        raise ValueError('block must be positive')
    nblocks = -(-total_frames // block)
    return [(j, j * block, min(total_frames, (j + 1) * block)) for j in range(rank, nblocks, world)]


def cyclic_handoff_plan(lo, hi, total_frames, rank, world, reset_period):
    """(rank to receive the state from | None, rank to send it to | None) for the block [lo, hi): the same predicate on the
    same frame index on both sides of every boundary, as handoff_plan."""
    recv = world > 1 and needs_state(lo, reset_period)
    send = world > 1 and hi < total_frames and needs_state(hi, reset_period)
    return ((rank - 1) % world if recv else None), ((rank + 1) % world if send else None)


def iter_cyclic(eng, frames_of, total_frames, block, rank, world, link=None, stamp=None):
    """Generator form of run_cyclic: yields (lo, records VIEW [n,h,w,4]) after each of this rank's blocks (the view is valid
    until the generator is advanced).  In-process tests drive the generators of N "ranks" in global block order."""
    if link is not None and hasattr(link, 'get_backend'):
        link = TorchLink(link)
    if block > eng.max_chunk:
        raise ValueError('block %d exceeds the engine\'s max_chunk %d' % (block, eng.max_chunk))
    if world > 1 and link is None:
        #  without a link the blocks of the other ranks would be skipped and no state handed over: silently wrong
        # records.  A single process that wants the whole sequence passes world = 1.
        raise ValueError('iter_cyclic: world = %d needs a state link (kfnet_amd.dist.make_link); pass world = 1 to run every block here' % world)
    w = world
    for j, lo, hi in cyclic_blocks(total_frames, block, rank, w):
        n = hi - lo
        need = needs_state(lo, eng.reset_period)
        frames = frames_of(lo - 1 if need else lo, hi)
        if need:
            eng.prime(frames[0])           # flow features of frame lo-1, recomputed locally (5.6 GFLOP instead of a 614 KB message)
            frames = frames[1:]
        eng.heavy(frames, n)
        if stamp is not None:
            stamp('heavy_end', j)
        src, dst = cyclic_handoff_plan(lo, hi, total_frames, rank, w, eng.reset_period)
        if src is not None:
            link.recv(eng.get_state(), src, eng)
        if stamp is not None:
            stamp('scan_start', j)
        eng.scan(n, lo)
        if stamp is not None:
            stamp('scan_end', j)
        if dst is not None:
            link.send(eng.get_state(), dst, eng)
        yield lo, eng.records(n)


def run_cyclic(eng, frames_of, total_frames, block, rank, world, link=None, on_block=None, stamp=None):
    """Process this rank's blocks of a `total_frames`-frame sequence dealt round-robin in blocks of `block` frames.
    `frames_of(lo, hi)` returns the device uint8 frames [lo, hi) (a rank only ever asks for its own blocks and the frame in
    front of each).  Per block: heavy phase (state-independent) -> receive the Kalman state from rank-1 unless the block
    starts on a reset frame -> scan -> send the state to rank+1 unless the next block starts on a reset frame.
    Returns [(lo, records tensor [n,h,w,4])] (copies: the engine's record buffer is reused by the next block);
    `on_block(lo, records_view)` instead consumes each block's records in place.  `stamp(name, j)`: phase hook of bench.py."""
    out = []
    for lo, rec in iter_cyclic(eng, frames_of, total_frames, block, rank, world, link, stamp):
        if on_block is not None:
            on_block(lo, rec)
        else:
            out.append((lo, rec.clone()))
    return out


def scan_cyclic_host(chunk_scan_fn, rank, world, dist, state_buf, total_frames, block, reset_period):
    """Backend-agnostic skeleton of run_cyclic's hand-off for the gloo CPU tests: `chunk_scan_fn(state_buf, lo, hi)` advances
    the state over frames [lo, hi) in place and returns that block's outputs.  Returns [(lo, outputs)]."""
    out = []
    for j, lo, hi in cyclic_blocks(total_frames, block, rank, world):
        src, dst = cyclic_handoff_plan(lo, hi, total_frames, rank, world, reset_period)
        if src is not None:
            dist.recv(state_buf, src=src)
        out.append((lo, chunk_scan_fn(state_buf, lo, hi)))
        if dst is not None:
            dist.send(state_buf, dst=dst)
    return out


class ChunkTimer(object):
    """HIP-event stamps of one `run_chunk` pass on the engine's stream, for the serial chain of the sharded
    configuration: heavy phase | recv (waits for the predecessor's scan) | scan | send.  `origin()` is called right
    after a barrier + synchronize, so event times relative to it can be placed on the node's monotonic clock and
    compared ACROSS ranks (bench.py: scan_chain_ms = last rank's scan end - rank 0's scan start)."""
    NAMES = ('origin', 'heavy_end', 'recv_end', 'scan_end', 'send_end')

    def __init__(self, torch, device):
        self.torch, self.device = torch, device
        self.ev = {n: torch.cuda.Event(enable_timing=True) for n in self.NAMES}
        self.t_origin = None
        self.did = {'recv': False, 'send': False}

    def origin(self):
        import time
        self.torch.cuda.synchronize(self.device)
        self.t_origin = time.monotonic()
        self.stamp('origin')

    def stamp(self, name):
        self.ev[name].record(self.torch.cuda.current_stream(self.device))

    def summary(self):
        """ms since origin of every stamp + absolute monotonic seconds; call after a synchronize."""
        rel = {n: float(self.ev['origin'].elapsed_time(self.ev[n])) for n in self.NAMES[1:]}
        return {'t_origin_monotonic_s': self.t_origin, 'ms_since_origin': rel,
                'heavy_ms': rel['heavy_end'],
                'recv_wait_us': (rel['recv_end'] - rel['heavy_end']) * 1e3 if self.did['recv'] else None,
                'scan_ms': rel['scan_end'] - rel['recv_end'],
                'send_us': (rel['send_end'] - rel['scan_end']) * 1e3 if self.did['send'] else None}


def run_chunk(eng, dev_frames, first_frame, rank, world, link=None, dev_prev_frame=None, timer=None):
    """Process this rank's chunk [first_frame, first_frame + T).  `link` is a StateLink (or an
    initialised torch.distributed module, wrapped on the fly; None for a single process).  `timer`: a ChunkTimer
    whose origin() has been called; stamps the phase boundaries on the stream (no host synchronisation)."""
    if link is not None and hasattr(link, 'get_backend'):   # the torch.distributed module itself
        link = TorchLink(link)
    T = int(dev_frames.shape[0])
    recv, send = handoff_plan(first_frame, T, rank, world if link is not None else 1, eng.reset_period)
    if T > 0:
        if needs_state(first_frame, eng.reset_period):
            if dev_prev_frame is None:
                raise ValueError('chunk starting at frame %d needs the preceding frame' % first_frame)
            eng.prime(dev_prev_frame)      # flow features of frame first_frame-1, recomputed locally
        eng.heavy(dev_frames, T)           # state-independent: no waiting on other ranks
    if timer is not None:
        timer.stamp('heavy_end')
    if recv:
        link.recv(eng.get_state(), rank - 1, eng)      # the 76.8 KB hand-off, into the live state
    if timer is not None:
        timer.did['recv'] = bool(recv)
        timer.stamp('recv_end')
    if T > 0:
        eng.scan(T, first_frame)
    if timer is not None:
        timer.stamp('scan_end')
    if send:                               # an empty chunk just forwards what it received
        link.send(eng.get_state(), rank + 1, eng)
    if timer is not None:
        timer.did['send'] = bool(send)
        timer.stamp('send_end')
    return eng.records(T)


def scan_sharded_host(states_in, chunk_scan_fn, rank, world, dist, state_buf, first_frame, reset_period,
                      n_frames=None):
    """Backend-agnostic skeleton of the hand-off used by the gloo CPU tests: receive the
    state if the chunk depends on it, run `chunk_scan_fn(state_buf)` (which updates
    state_buf in place), then send it on if the next chunk depends on it.  `n_frames` is
    this chunk's length (needed for the send rule)."""
    if n_frames is None:
        raise ValueError('scan_sharded_host needs the chunk length to apply the pairing rule')
    recv, send = handoff_plan(first_frame, n_frames, rank, world, reset_period)
    if recv:
        dist.recv(state_buf, src=rank - 1)
    if n_frames > 0:
        chunk_scan_fn(state_buf)
    if send:
        dist.send(state_buf, dst=rank + 1)
    return state_buf
