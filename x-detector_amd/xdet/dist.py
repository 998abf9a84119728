"""Multi-GPU plumbing: one process per GPU, images sharded by rank, ONE all-gather of the
fixed-size padded detections per step (BASELINE config 4).  The reference has no
distributed code at all (a single tf.estimator session at batch 1,
light_head_rfcn_eval.py:212,466-499); this layer is new.

Transport: RCCL behind the C-ABI (xdet_comm_* in include/xdet.h, csrc/comm.hip) -- ncclAllGather
over xGMI on the communicator's own HIP stream, so the gather of step k runs under the forward of
step k+1.  Ranks rendezvous through a file that carries rank 0's ncclUniqueId; no PyTorch, no MPI.
The payload per image is 20 classes x 200 slots x (score + 4 box coords) x 4 B = 80 KB, so the
exchange is latency-bound (tens of microseconds), never xGMI-link-bound.

The pure-NumPy helpers (shard_range, pack_detections, unpack_detections, gathered_layout) define
the record layout; the device pack kernel and the tests are checked against them.
"""
import ctypes
import os

import numpy as np

RECORD = 5      # floats per detection slot: score | ymin xmin ymax xmax


def shard_range(global_batch, rank, world):
    """contiguous block partition of a global batch: rank r owns [lo, hi)."""
    per = global_batch // world
    rem = global_batch % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def pack_detections(scores, boxes):
    """scores [B,C,K], boxes [B,C,K,4] -> one contiguous [B,C,K,5] record (score | box)."""
    scores = np.asarray(scores, np.float32)
    boxes = np.asarray(boxes, np.float32)
    out = np.empty(scores.shape + (RECORD,), np.float32)
    out[..., 0] = scores
    out[..., 1:] = boxes
    return out


def unpack_detections(packed):
    return packed[..., 0], packed[..., 1:]


def gathered_layout(world, per_rank, num_fg_classes=20, topk=200):
    """shape of the all-gather result: rank-major, i.e. global image index = rank * per_rank + local index
    (the contiguous-block sharding of shard_range with equal shards)."""
    return (world * per_rank, num_fg_classes, topk, RECORD)


def rendezvous_path(environ=None):
    """where rank 0 publishes the ncclUniqueId.  XDET_COMM_ID_FILE if set (xdet.launch sets it);
    under `python -m torch.distributed.run` (which only hands out RANK/WORLD_SIZE/MASTER_*) a name
    that all workers of ONE launch agree on and that no earlier launch can have left behind: master
    port + the launcher's pid (every worker's parent) + the restart count."""
    env = os.environ if environ is None else environ
    if env.get('XDET_COMM_ID_FILE'):
        return env['XDET_COMM_ID_FILE']
    tag = '%s_%s_%s_%s' % (env.get('MASTER_PORT', '0'), os.getppid(), env.get('TORCHELASTIC_RUN_ID', 'x'),
                           env.get('TORCHELASTIC_RESTART_COUNT', '0'))
    tag = ''.join(ch if ch.isalnum() or ch in '_-' else '_' for ch in tag)
    return os.path.join(env.get('TMPDIR', '/tmp'), 'xdet_rccl_id_' + tag)


class Communicator(object):
    """RCCL communicator of this rank (call after the device has been selected).

    allgather_detections() is asynchronous and double-buffered: the result of call k stays valid
    until call k+2, so a consumer can read step k while step k+1 is gathered."""

    def __init__(self, rank, world, id_path=None, timeout_s=120):
        from ._lib import lib, check, c_void_p
        self.rank, self.world = int(rank), int(world)
        if self.world > 1 and not id_path:
            id_path = rendezvous_path()
        h = c_void_p()
        # librccl prints a version banner to STDOUT when it initialises; a caller that promises "one JSON line on
        # stdout" (bench.py) must not inherit it: fd 1 points at stderr for the duration of the init
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            check(lib().xdet_comm_init(ctypes.byref(h), self.rank, self.world,
                                       id_path.encode() if id_path else None, int(timeout_s)))
        finally:
            try:
                ctypes.CDLL(None).fflush(None)     # the banner sits in libc's stdout buffer until flushed
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)
        self.handle = h
        self._bufs = None
        self._turn = 0
        self._last = None

    def info(self):
        from ._lib import lib, check
        v = [ctypes.c_int() for _ in range(4)]
        check(lib().xdet_comm_info(self.handle, *[ctypes.byref(x) for x in v]))
        return dict(zip(('rank', 'world', 'device', 'rccl_version'), [x.value for x in v]))

    @staticmethod
    def library():
        """(path of the shared object the collective entry points were bound from, named-by-XDET_RCCL_LIB?)"""
        from ._lib import lib, check
        buf = ctypes.create_string_buffer(4096)
        over = ctypes.c_int()
        check(lib().xdet_comm_library(buf, 4096, ctypes.byref(over)))
        return buf.value.decode('utf-8', 'replace'), bool(over.value)

    def _ensure(self, n_images, nc, topk):
        from .runtime import DeviceBuffer
        key = (n_images, nc, topk)
        if self._bufs is None or self._bufs[0] != key:
            slot = n_images * nc * topk * RECORD * 4
            self._bufs = (key, DeviceBuffer(slot, zero=True),
                          [DeviceBuffer(slot * self.world, zero=True) for _ in range(2)])
        return self._bufs

    def allgather_detections(self, det_scores_ptr, det_boxes_ptr, n_images, num_fg_classes, topk, streams=(),
                             double_buffered=False):
        """det_scores [B,C,K] / det_boxes [B,C,K,4] (device pointers of THIS rank's shard) -> enqueue
        pack + ncclAllGather behind the `streams` that produce them; returns the device buffer that
        will hold [world*B, C, K, 5].  double_buffered: the caller alternates between two det buffer
        pairs from call to call (removes the per-step join of the producer streams)."""
        from ._lib import lib, check, c_void_p
        _, packed, outs = self._ensure(n_images, num_fg_classes, topk)
        out = outs[self._turn]
        self._turn ^= 1
        hs = [s.handle if hasattr(s, 'handle') else s for s in streams]
        arr = (c_void_p * max(len(hs), 1))(*[h.value if isinstance(h, c_void_p) else h for h in hs])
        check(lib().xdet_comm_allgather_detections(self.handle, det_scores_ptr, det_boxes_ptr, n_images,
                                                   num_fg_classes, topk, packed.ptr, out.ptr, arr, len(hs),
                                                   1 if double_buffered else 0))
        self._last = (out, gathered_layout(self.world, n_images, num_fg_classes, topk))
        return out

    def wait(self, stream=None):
        """the host (stream=None) or `stream` waits for the last gather."""
        from ._lib import lib, check
        check(lib().xdet_comm_wait(self.handle, stream.handle if stream is not None else None))

    def gathered(self):
        """host copy of the last gather: [world*B, C, K, 5]."""
        from .runtime import to_host
        self.wait()
        buf, shape = self._last
        return to_host(buf.ptr, shape, np.float32)

    def max_over_ranks(self, value):
        from ._lib import lib, check
        v = ctypes.c_double(float(value))
        check(lib().xdet_comm_allreduce_max(self.handle, ctypes.byref(v)))
        return v.value

    def barrier(self):
        from ._lib import lib, check
        check(lib().xdet_comm_barrier(self.handle))

    def set_timeout(self, seconds):
        """watchdog of every host wait on the communicator's stream (default 300 s / XDET_COMM_TIMEOUT_S): past it
        the communicator is aborted and the call raises instead of hanging behind a dead peer."""
        from ._lib import lib, check
        check(lib().xdet_comm_set_timeout(self.handle, float(seconds)))

    def allgather_bytes(self, payload):
        """`payload` (bytes, the same length on every rank, <= 1 MiB) -> list of world payloads in rank order,
        moved by ncclAllGather on the communicator's stream."""
        from ._lib import lib, check
        n = len(payload)
        send = ctypes.create_string_buffer(bytes(payload), n)
        recv = ctypes.create_string_buffer(n * self.world)
        check(lib().xdet_comm_allgather_bytes(self.handle, send, recv, n))
        return [recv.raw[i * n:(i + 1) * n] for i in range(self.world)]

    def device_records(self, extra=None):
        """every rank's (rank, hip device, PCI bus id, host, pid[, extra]) gathered THROUGH the collective: N distinct
        (host, pci_bus_id) pairs in the result prove that N distinct GPUs took part.  `extra`: a small dict of
        numbers per rank (e.g. its images/s)."""
        import json
        import socket
        from ._lib import lib, check
        info = self.info()
        buf = ctypes.create_string_buffer(32)
        check(lib().xdet_device_pci_bus_id(info['device'], buf, 32))
        rec = {'rank': self.rank, 'hip_device': info['device'], 'pci_bus_id': buf.value.decode(),
               'host': socket.gethostname(), 'pid': os.getpid()}
        if extra:
            rec.update(extra)
        raw = json.dumps(rec).encode()
        if len(raw) > 500:
            raise ValueError('device record too large')
        out = [json.loads(b.rstrip(b'\0').decode()) for b in self.allgather_bytes(raw.ljust(512, b'\0'))]
        return out

    def close(self):
        if getattr(self, 'handle', None) is not None:
            from ._lib import lib
            lib().xdet_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
