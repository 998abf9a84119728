"""Device memory, streams and the NHWC device tensor the host API passes around.

Plain ctypes over libxdet_hip.so / libamdhip64 -- no PyTorch in the product path.
"""
import ctypes

import numpy as np

from ._lib import lib, check, c_void_p


def _ptr(x):
    """device pointer (int / c_void_p / DeviceTensor / DeviceBuffer / None) -> c_void_p"""
    if x is None:
        return c_void_p(None)
    if isinstance(x, (DeviceTensor, DeviceBuffer)):
        return c_void_p(x.ptr)
    if isinstance(x, c_void_p):
        return x
    return c_void_p(int(x))


def _host(a):
    return c_void_p(a.ctypes.data)


def round_up(x, m):
    return (x + m - 1) // m * m


def channel_ld(c):
    """channel stride of an activation with c channels (include/xdet.h)."""
    return 4 if c <= 4 else round_up(c, 32)


PRECISIONS = {'f32': 0, 'f16x3': 1, 'f16': 2}


def set_precision(mode):
    """arithmetic of the conv/dense contractions for layers and nets created afterwards:
    'f32' (exact f32 MFMA, default), 'f16x3' (split-precision f16 MFMA, ~f32 accuracy) or
    'f16' (plain f16 operands: speed mode, outside the 1e-3 parity budget)."""
    check(lib().xdet_set_default_precision(PRECISIONS[mode]))


def get_precision():
    v = lib().xdet_get_default_precision()
    return [k for k, x in PRECISIONS.items() if x == v][0]


class Stream(object):
    def __init__(self):
        h = c_void_p()
        check(lib().xdet_stream_create(ctypes.byref(h)))
        self.handle = h

    def synchronize(self):
        check(lib().xdet_stream_sync(self.handle))

    def __del__(self):
        try:
            lib().xdet_stream_destroy(self.handle)
        except Exception:
            pass


class Event(object):
    def __init__(self):
        h = c_void_p()
        check(lib().xdet_event_create(ctypes.byref(h)))
        self.handle = h

    def record(self, stream=None):
        check(lib().xdet_event_record(self.handle, stream.handle if stream else None))

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        check(lib().xdet_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().xdet_event_destroy(self.handle)
        except Exception:
            pass


def synchronize(stream=None):
    check(lib().xdet_stream_sync(stream.handle if stream else None))


class DeviceBuffer(object):
    """An owned hipMalloc allocation."""
    def __init__(self, nbytes, zero=False):
        p = c_void_p()
        check(lib().xdet_malloc(ctypes.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes
        if zero:
            check(lib().xdet_memset(self.ptr, 0, nbytes, None))
            synchronize()

    def free(self):
        if self.ptr:
            lib().xdet_free(c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_device(a, dtype=None):
    a = np.ascontiguousarray(a, dtype or a.dtype)
    buf = DeviceBuffer(max(a.nbytes, 16))
    check(lib().xdet_memcpy_h2d(buf.ptr, _host(a), a.nbytes, None))
    synchronize()
    return buf


def to_host(ptr, shape, dtype=np.float32, stream=None):
    out = np.empty(shape, dtype)
    check(lib().xdet_memcpy_d2h(_host(out), _ptr(ptr), out.nbytes, stream.handle if stream else None))
    return out


class DeviceTensor(object):
    """f32 NHWC activation on the GPU: logical shape (N,H,W,C), channel stride ld >= C
    (padding channels are zero).  May own its memory or be a view of a net workspace buffer."""
    def __init__(self, ptr, shape, ld, owner=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.ld = int(ld)
        self._owner = owner

    @classmethod
    def from_numpy(cls, a, layout='NHWC'):
        a = np.asarray(a, np.float32)
        if layout == 'NCHW':
            a = np.transpose(a, (0, 2, 3, 1))
        n, h, w, c = a.shape
        ld = channel_ld(c)
        pad = np.zeros((n, h, w, ld), np.float32)
        pad[..., :c] = a
        # slack: the conv loader reads whole 32-channel slices
        buf = DeviceBuffer(pad.nbytes + 512, zero=True)
        check(lib().xdet_memcpy_h2d(buf.ptr, _host(pad), pad.nbytes, None))
        synchronize()
        return cls(buf.ptr, (n, h, w, c), ld, buf)

    @classmethod
    def empty(cls, shape, ld=None):
        n, h, w, c = shape
        ld = ld or channel_ld(c)
        buf = DeviceBuffer(n * h * w * ld * 4 + 512, zero=True)
        return cls(buf.ptr, shape, ld, buf)

    def numpy(self, layout='NHWC', n=None, stream=None):
        N, H, W, C = self.shape
        N = n or N
        raw = to_host(self.ptr, (N, H, W, self.ld), np.float32, stream)
        a = np.ascontiguousarray(raw[..., :C])
        return np.transpose(a, (0, 3, 1, 2)) if layout == 'NCHW' else a

    def channels(self, start, stop):
        """view of a channel range (e.g. the cls / box halves of the fused RPN output)."""
        N, H, W, C = self.shape
        return DeviceTensor(self.ptr + 4 * start, (N, H, W, stop - start), self.ld, self._owner or self)

    def copy_from(self, other, n=None):
        """device-to-device copy of the first n images (same H,W,C,ld)."""
        assert self.shape[1:] == other.shape[1:] and self.ld == other.ld, (self.shape, other.shape)
        n = n or min(self.shape[0], other.shape[0])
        nb = n * self.shape[1] * self.shape[2] * self.ld * 4
        check(lib().xdet_memcpy_d2d(self.ptr, other.ptr, nb, None))
        synchronize()
