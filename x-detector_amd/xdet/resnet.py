"""ResNet-50 v2 trunk (net/resnet_v2.py:311-345 up to the final batch_norm_relu) -- BASELINE
config 2: a stand-alone backbone benchmark of the same MFMA conv kernels."""
import ctypes

import numpy as np

from ._lib import lib, check, c_void_p
from .runtime import DeviceBuffer, Stream, to_host, _host


class ResNet50Trunk(object):
    def __init__(self, weights, image_size=480, max_batch=8, options=None):
        """options: {key: 'on' | 'off'} for xdet_resnet_set_option (include/xdet.h), e.g. {'bneck': 'off'}"""
        h = c_void_p()
        check(lib().xdet_resnet_create(ctypes.byref(h), image_size, max_batch))
        self.handle = h
        for k, v in (options or {}).items():
            check(lib().xdet_resnet_set_option(self.handle, k.encode(), v.encode()))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, np.float32)
            dims = (ctypes.c_int64 * a.ndim)(*a.shape)
            check(lib().xdet_resnet_set_weight(self.handle, name.encode(), _host(a), a.ndim, dims))
        check(lib().xdet_resnet_build(self.handle))
        ho, wo, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().xdet_resnet_out_shape(self.handle, ctypes.byref(ho), ctypes.byref(wo), ctypes.byref(c)))
        self.out_shape = (ho.value, wo.value, c.value)
        self.image_size, self.max_batch = image_size, max_batch
        self.stream = Stream()
        self._images = DeviceBuffer(max_batch * 3 * image_size * image_size * 4)
        self._out = DeviceBuffer(max_batch * ho.value * wo.value * c.value * 4)

    def set_images(self, images_nchw):
        a = np.ascontiguousarray(images_nchw, np.float32)
        assert a.shape[0] <= self.max_batch and a.shape[1:] == (3, self.image_size, self.image_size)
        check(lib().xdet_memcpy_h2d(self._images.ptr, _host(a), a.nbytes, self.stream.handle))
        return a.shape[0]

    def forward_device(self, n, use_graph=False):
        fn = lib().xdet_resnet_forward_graph if use_graph else lib().xdet_resnet_forward
        check(fn(self.handle, self._images.ptr, n, self._out.ptr, self.stream.handle))

    def forward(self, images_nchw):
        """[N,3,S,S] -> NHWC [N,h,w,2048] (C = 2048 is already a multiple of 32: no padding)."""
        n = self.set_images(images_nchw)
        self.forward_device(n)
        self.stream.synchronize()
        return to_host(self._out.ptr, (n,) + self.out_shape, np.float32)

    def calibrate(self, images_nchw):
        """choose the activation pre-scale exponents of the split-precision operands from a calibration batch
        (xdet_resnet_calibrate); returns {tensor name: exponent} of the tensors that got one."""
        n = self.set_images(images_nchw)
        k = ctypes.c_int()
        check(lib().xdet_resnet_calibrate(self.handle, self._images.ptr, n, ctypes.byref(k), self.stream.handle))
        return {name: e for name, e in self.plane_scales().items() if e}

    def plane_scales(self):
        n = ctypes.c_int()
        check(lib().xdet_net_plane_scales(self.handle, 0, ctypes.byref(n), None))
        exps = (ctypes.c_int * max(n.value, 1))()
        check(lib().xdet_net_plane_scales(self.handle, n.value, ctypes.byref(n), exps))
        out = {}
        buf = ctypes.create_string_buffer(256)
        for i in range(n.value):
            check(lib().xdet_net_plane_scale_name(self.handle, i, buf, 256))
            out['%d: %s' % (i, buf.value.decode())] = exps[i]
        return out

    def flops_per_image(self):
        f = ctypes.c_double()
        check(lib().xdet_resnet_flops_per_image(self.handle, ctypes.byref(f)))
        return f.value

    def __del__(self):
        try:
            lib().xdet_resnet_destroy(self.handle)
        except Exception:
            pass
