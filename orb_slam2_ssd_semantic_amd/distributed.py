"""Batched keyframe mode across the GPUs of one node.

Frames are independent, so a keyframe batch is sharded in contiguous blocks, one process per GPU, with NO
collective on the data path.  The single exchange step is one all-gather of the three fixed-size padded
result buffers (counts, keypoints, descriptors) so that every rank holds the whole batch's descriptors
(what KeyFrameDatabase / LoopClosing consume serially in the reference, src/LoopClosing.cc:312-342).
torch.distributed is plumbing: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests.
"""
import torch
import torch.distributed as dist


def shard_range(nframes, rank, world):
    """Contiguous block of frames owned by `rank`: [lo, hi). Remainder frames go to the lowest ranks."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(nframes, world):
    return -(-nframes // world)


def all_gather_keyframes(n, kps, desc, nframes_total=None, group=None, host_staged=False):
    """All-gather per-rank results.

    n    : int32 [S]            keypoint counts of this rank's S frames (S = max_shard; unused slots 0)
    kps  : int32 [S, cap, 7]    orbfe_keypoint records, bit-cast to int32
    desc : uint8 [S, cap, 32]
    Returns (n_all [W*S], kps_all [W*S, cap, 7], desc_all [W*S, cap, 32]) in rank-major order; with
    `nframes_total` the padding slots of uneven shards are removed so the result is in frame order.
    host_staged: device tensors over a CPU-only backend (gloo) -- D2H, gather, H2D.  This is how several ranks that share ONE
    GPU exchange their blocks (RCCL refuses two ranks per device): a test transport, never a measurement.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = (n, kps, desc)
    else:
        out = []
        for t in (n, kps, desc):
            src = t.contiguous()
            if host_staged and src.is_cuda:
                hg = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
                dist.all_gather_into_tensor(hg, src.cpu(), group=group)
                g = hg.to(t.device)
            else:
                g = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(g, src, group=group)
            out.append(g)
        out = tuple(out)
    if nframes_total is None:
        return out
    S = n.shape[0]
    idx = []
    for r in range(world):
        lo, hi = shard_range(nframes_total, r, world)
        idx.extend(range(r * S, r * S + (hi - lo)))
    idx = torch.as_tensor(idx, dtype=torch.long, device=n.device)
    return tuple(t.index_select(0, idx) for t in out)


class _HostStagedWork:
    """the pending gather of one tensor on the host-staged transport: wait() = host waits for the worker thread, then the
    CURRENT stream waits (stream level) for the upload of the gathered block"""

    def __init__(self):
        import threading
        self.done = threading.Event()
        self.uploaded = None
        self.error = None

    def wait(self):
        self.done.wait()
        if self.error is not None:
            raise self.error
        if self.uploaded is not None:
            torch.cuda.current_stream().wait_event(self.uploaded)


class _StreamWork:
    """the pending gather of one launch on the RCCL transport: wait() = the CURRENT stream waits (stream level) for the event
    recorded behind the collectives on the gather stream"""

    def __init__(self, done):
        self.done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


class OverlappedKeyframeGather:
    """Double-buffered, asynchronous form of `all_gather_keyframes` for a steady stream of batches.

    The producer alternates between two output sets (n, kps, desc).  `acquire(k)` makes the current stream wait until
    the gather that last read set k has finished (stream-level on RCCL, host-level on gloo), `launch(k)` starts the
    all-gather of set k after the work already enqueued on the current stream; it then overlaps whatever is enqueued
    next.  `result(k)` waits for and returns the gathered (n, kps, desc) of set k in rank-major order.

    RCCL transport (device tensors, backend "nccl"): the collectives of a launch are issued under a GATHER STREAM of this object that
    first waits for an event recorded behind the producer's kernels; HIP events recorded on that stream around them time the
    exchange itself (`timing()`), whatever it overlaps.

    gather_cap: slots per frame that travel (None: all `cap` of them).  The extractor fills count(f) <= cap slots of a frame and zero
    pads the rest; a caller that knows a bound on the counts (bench.py: the maximum over a probe step, rounded up to 64 -- 1024 of
    1088 slots at 1000 features) sends the valid prefix only: the blocks are compacted on the gather stream ([S, cap, .] ->
    [S, gather_cap, .]), the gathered blocks have gather_cap slots per frame, and the counts (always gathered whole) let the
    receiver verify that nothing was cut (`truncated(k)`).

    host_staged=True: device blocks over a CPU-only group (gloo).  A worker thread waits for an event recorded behind the
    producer's launches, copies the blocks to pinned host memory, gathers them over `group`, uploads the result on a side stream.
    The producer keeps enqueuing the next step meanwhile, exactly as with the asynchronous RCCL collective -- this is the
    transport for several ranks on ONE device (tests; RCCL refuses two ranks per GPU).  `group` must then be a group used by
    nobody else (the worker thread issues its collectives in launch order on every rank).
    """

    def __init__(self, sets, group=None, host_staged=False, gather_cap=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sets = sets
        self.pending = [[] for _ in sets]
        cap = max((t.shape[1] for s in sets for t in s if t.dim() >= 2), default=0)
        self.cap = cap
        self.gcap = int(gather_cap) if gather_cap and 0 < int(gather_cap) < cap else None

        def travel_shape(t):   # what one rank sends of tensor t
            return (t.shape[0], self.gcap) + tuple(t.shape[2:]) if (self.gcap and t.dim() >= 2) else tuple(t.shape)
        self.compact = [tuple(torch.empty(travel_shape(t), dtype=t.dtype, device=t.device) if (self.gcap and t.dim() >= 2) else None
                              for t in s) for s in sets]
        self.gathered = [tuple(torch.empty((self.world * t.shape[0],) + travel_shape(t)[1:], dtype=t.dtype, device=t.device)
                               for t in s) for s in sets]
        import math
        self.bytes_per_rank = int(sum(t.element_size() * math.prod(travel_shape(t)) for t in sets[0]))
        self.host_staged = bool(host_staged) and self.world > 1
        self.times = []        # RCCL: (event before, event after) per launch; host-staged: seconds per launch
        self.comm = None
        on_gpu = any(t.is_cuda for s in sets for t in s)
        if on_gpu and not self.host_staged:
            self.comm = torch.cuda.Stream()
        if self.host_staged:
            import queue
            import threading
            self.h_src = [tuple(torch.empty(travel_shape(t), dtype=t.dtype).pin_memory() for t in s) for s in sets]
            self.h_dst = [tuple(torch.empty(g.shape, dtype=g.dtype).pin_memory() for g in gs) for gs in self.gathered]
            self.side = torch.cuda.Stream()
            self.device = torch.cuda.current_device()
            self.q = queue.Queue()
            self.thread = threading.Thread(target=self._worker, daemon=True)
            self.thread.start()

    def _travel(self, k, i, src):
        """tensor i of set k as it travels: the valid prefix, compacted into this object's buffer on the CURRENT stream"""
        c = self.compact[k][i]
        if c is None:
            return src.contiguous()
        c.copy_(src[:, :self.gcap])
        return c

    def _worker(self):
        import time
        torch.cuda.set_device(self.device)   # the current device is per thread
        while True:
            job = self.q.get()
            if job is None:
                return
            k, ready, work = job
            try:
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ready)
                    t0 = time.perf_counter()
                    for i, (hs, src) in enumerate(zip(self.h_src[k], self.sets[k])):
                        hs.copy_(self._travel(k, i, src), non_blocking=True)
                    self.side.synchronize()
                    for hd, hs in zip(self.h_dst[k], self.h_src[k]):
                        dist.all_gather_into_tensor(hd, hs, group=self.group)
                    for dst, hd in zip(self.gathered[k], self.h_dst[k]):
                        dst.copy_(hd, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.side.synchronize()   # the pinned blocks are re-used by the next job
                    self.times.append(time.perf_counter() - t0)
                work.uploaded = ev
            except BaseException as e:   # noqa: BLE001 -- surfaced by wait()
                work.error = e
            work.done.set()

    def acquire(self, k):
        for w in self.pending[k]:
            w.wait()
        self.pending[k] = []

    def launch(self, k):
        if self.world == 1 and self.comm is None:
            for i, (dst, src) in enumerate(zip(self.gathered[k], self.sets[k])):
                dst.copy_(self._travel(k, i, src))
            return
        if self.host_staged:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            work = _HostStagedWork()
            self.pending[k].append(work)
            self.q.put((k, ready, work))
            return
        if self.comm is None:   # CPU tensors (gloo): asynchronous work handles
            for i, (dst, src) in enumerate(zip(self.gathered[k], self.sets[k])):
                self.pending[k].append(dist.all_gather_into_tensor(dst, self._travel(k, i, src), group=self.group, async_op=True))
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            e0.record(self.comm)
            for i, (dst, src) in enumerate(zip(self.gathered[k], self.sets[k])):
                t = self._travel(k, i, src)
                if self.world == 1 and not dist.is_initialized():
                    dst.copy_(t)
                else:   # synchronous form: the gather stream (not the producer's) waits for the collective
                    dist.all_gather_into_tensor(dst, t, group=self.group)
            e1.record(self.comm)
        self.pending[k].append(_StreamWork(e1))
        self.times.append((e0, e1))

    def result(self, k):
        self.acquire(k)
        return self.gathered[k]

    def truncated(self, k):
        """frames of the gathered set k whose count exceeds the slots that travelled (0 unless gather_cap was chosen too small)"""
        if not self.gcap:
            return 0
        return int((self.result(k)[0] > self.gcap).sum().item())

    def timing(self, last=None):
        """mean / max milliseconds of the exchange of one launch (the device must be idle: events are read), over the last `last`
        launches; None when nothing was timed"""
        ts = self.times[-last:] if last else self.times
        if not ts:
            return None
        ms = [t * 1e3 if isinstance(t, float) else t[0].elapsed_time(t[1]) for t in ts]
        return {"mean_ms": sum(ms) / len(ms), "max_ms": max(ms), "min_ms": min(ms), "launches": len(ms)}

    def close(self):
        if self.host_staged and self.thread is not None:
            self.q.put(None)
            self.thread.join(timeout=60)
            self.thread = None


class KeyframeGroup:
    """ctypes mirror of the C-ABI's multi-device layer (include/orbfe.h, orbfe_group_*): the batched keyframe mode for
    C / C++ hosts -- contiguous shards, one in-place ncclAllGather (RCCL) of the padded count / keypoint / descriptor blocks,
    and the consumer of the gather (own frames against candidate frames anywhere in the batch).  `devices` = the devices
    ONE process drives (ncclCommInitAll); `rank_of_world=(rank, world, id_bytes)` = one process per device."""

    RCCL, COPY = 0, 1   # transports of the exchange step (include/orbfe.h ORBFE_GROUP_RCCL / ORBFE_GROUP_COPY)

    def __init__(self, nfeatures, scale_factor, nlevels, ini_th, min_th, max_width, max_height, max_batch, devices=(0,),
                 rank_of_world=None, device=None, transport=None):
        import ctypes as C
        from . import _ffi
        self._ffi, self._C = _ffi, C
        L = _ffi.lib()
        p = _ffi.OrbfeParams(nfeatures, scale_factor, nlevels, ini_th, min_th, max_width, max_height, max_batch, -1, 0)
        h = C.c_void_p()
        if rank_of_world is None:
            devs = (C.c_int32 * len(devices))(*devices)
            if transport is None:   # the C entry point's own default (RCCL)
                _ffi.check(L.orbfe_group_create_local(C.byref(p), devs, len(devices), C.byref(h)), "orbfe_group_create_local")
            else:
                _ffi.check(L.orbfe_group_create_local_ex(C.byref(p), devs, len(devices), int(transport), C.byref(h)),
                           "orbfe_group_create_local_ex")
        else:
            rank, world, idb = rank_of_world
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(idb))
            _ffi.check(L.orbfe_group_create_rank(C.byref(p), -1 if device is None else device, rank, world, buf, C.byref(h)),
                       "orbfe_group_create_rank")
        self.handle = h
        self.L = L
        self.world = L.orbfe_group_world(h)
        self.members = L.orbfe_group_members(h)
        self.transport = L.orbfe_group_transport(h)
        self.cap = L.orbfe_group_capacity(h)
        self.frames_padded = L.orbfe_group_frames_padded(h)

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _ffi
        buf = (C.c_uint8 * 128)()
        _ffi.check(_ffi.lib().orbfe_group_unique_id(buf), "orbfe_group_unique_id")
        return bytes(buf)

    @staticmethod
    def shard_range_c(nframes, rank, world):
        import ctypes as C
        from . import _ffi
        lo, hi = C.c_int32(), C.c_int32()
        _ffi.lib().orbfe_group_shard_range(nframes, rank, world, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def close(self):
        if getattr(self, "handle", None):
            self.L.orbfe_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract_batch(self, frames):
        """frames: uint8 [n, h, w] numpy (host).  Every member extracts its shard."""
        import numpy as np
        C = self._C
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        ptrs = (C.c_void_p * n)(*[frames[i].ctypes.data for i in range(n)])
        self._ffi.check(self.L.orbfe_group_extract_batch(self.handle, ptrs, n, w, h, w), "orbfe_group_extract_batch")
        self.nframes = n

    def extract_shard_device(self, member, d_gray_ptr, nframes_global, w, h, stride, frame_stride):
        self._ffi.check(self.L.orbfe_group_extract_shard_device(self.handle, member, d_gray_ptr, nframes_global, w, h, stride, frame_stride),
                        "orbfe_group_extract_shard_device")
        self.nframes = nframes_global

    def allgather(self):
        self._ffi.check(self.L.orbfe_group_allgather(self.handle), "orbfe_group_allgather")

    def synchronize(self):
        self._ffi.check(self.L.orbfe_group_synchronize(self.handle), "orbfe_group_synchronize")

    def block_index(self, frame):
        return self.L.orbfe_group_block_index(self.handle, self.nframes, frame)

    def blocks(self, member=0):
        C = self._C
        dn, dk, dd, st = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ffi.check(self.L.orbfe_group_blocks(self.handle, member, C.byref(dn), C.byref(dk), C.byref(dd), C.byref(st)), "orbfe_group_blocks")
        return dn.value, dk.value, dd.value, st.value

    def get_frame(self, frame, member=0):
        import numpy as np
        C = self._C
        kps = np.zeros(self.cap, self._ffi.KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int32()
        self._ffi.check(self.L.orbfe_group_get_frame_from(self.handle, member, frame, self._ffi.ptr(kps), self._ffi.ptr(desc), self.cap,
                                                          C.byref(n)), "orbfe_group_get_frame_from")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def counts(self, member=0):
        import numpy as np
        n = np.zeros(self.frames_padded, np.int32)
        self._ffi.check(self.L.orbfe_group_get_counts(self.handle, member, self._ffi.ptr(n)), "orbfe_group_get_counts")
        return n

    @staticmethod
    def owner_rank_c(nframes, world, frame):
        from . import _ffi
        return _ffi.lib().orbfe_group_owner_rank(nframes, world, frame)

    @staticmethod
    def block_index_c(nframes, world, shard, frame):
        from . import _ffi
        return _ffi.lib().orbfe_group_block_index_of(nframes, world, shard, frame)

    def match_device(self, member, d_qblock_ptr, d_tblock_ptr, npairs, d_match_ptr, d_nm_ptr, nnratio=0.9, th=100, check_ori=True):
        self._ffi.check(self.L.orbfe_group_match_device(self.handle, member, d_qblock_ptr, d_tblock_ptr, npairs, nnratio, th, int(check_ori),
                                                        d_match_ptr, d_nm_ptr), "orbfe_group_match_device")

    def match(self, qframe, tframe, nnratio=0.9, th=100, check_ori=True):
        import numpy as np
        q = np.ascontiguousarray(qframe, np.int32)
        t = np.ascontiguousarray(tframe, np.int32)
        m = np.full((len(q), self.cap), -1, np.int32)
        nm = np.zeros(len(q), np.int32)
        self._ffi.check(self.L.orbfe_group_match(self.handle, self._ffi.ptr(q), self._ffi.ptr(t), len(q), nnratio, th, int(check_ori),
                                                 self._ffi.ptr(m), self._ffi.ptr(nm)), "orbfe_group_match")
        return m, nm
