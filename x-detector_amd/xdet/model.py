"""The Light-Head R-CNN eval graph behind the reference's graph-builder API.

`LightHeadDetector` owns the native net (weights + workspace); the module-level
functions carry the reference's names and argument order (net/xception_body.py:236,
381,402,450,477) and operate on the detector that is current (`with det.scope():`,
the counterpart of `tf.variable_scope(params['model_scope'])`,
light_head_rfcn_eval.py:382).  Tensors are DeviceTensor views of the net's workspace.
"""
import contextlib
import ctypes

import numpy as np

from ._lib import lib, check, c_void_p, LightHeadConfig, XdetError, InvalidArgumentError
from .runtime import DeviceBuffer, DeviceTensor, Stream, to_device, to_host, synchronize, _host

_current = []


class LightHeadDetector(object):
    def __init__(self, weights, image_size=480, max_batch=1, num_classes=21, rpn_pre_nms_top_n=5000,
                 rpn_post_nms_top_n=1000, rpn_nms_thres=0.7, rpn_min_size=None, select_threshold=0.01,
                 nms_threshold=0.3, nms_topk=200, device=None, large_sep='auto', sepconv='fused', rpn_stream='side',
                 conv3x3='patch', pool='split', check_range=False, ksplit=True, cross='f16', workspace=None, pool_sub=None):
        """check_range=True: every activation tensor is validated against the f16 range of the split-precision convs
        after each forward (|x| <= 65504, no NaN); a violation raises in detections() / forward().  For validating a
        new checkpoint once: the pass re-reads every activation (~+30 % time).
        workspace='reuse' (default) | 'ssa': dead tensors' workspace blocks are handed to later tensors of the same size, or
        every tensor keeps its own block (check_range=True selects 'ssa' by itself); memory() reports both figures.
        cross='fp8': the cross terms of the split-precision products of the depthwise -> pointwise layers (28 of the 46
        contractions, the dominant ones) are computed from fp8 copies of the operands (the "x8" form, include/xdet.h) -- once
        calibrate() has measured the tensors; until then, and with 'f16', everything is f16x3."""
        if device is not None:
            check(lib().xdet_set_device(int(device)))
        self.cfg = LightHeadConfig(image_size=image_size, max_batch=max_batch, num_classes=num_classes,
                                   rpn_pre_nms_top_n=rpn_pre_nms_top_n, rpn_post_nms_top_n=rpn_post_nms_top_n,
                                   rpn_nms_thres=rpn_nms_thres,
                                   rpn_min_size=(16. / 480) if rpn_min_size is None else rpn_min_size,   # flag default, eval.py:112
                                   select_threshold=select_threshold, nms_threshold=nms_threshold, nms_topk=nms_topk)
        h = c_void_p()
        check(lib().xdet_net_create(ctypes.byref(h), ctypes.byref(self.cfg)))
        self.handle = h
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, np.float32)
            dims = (ctypes.c_int64 * a.ndim)(*a.shape)
            check(lib().xdet_net_set_weight(self.handle, name.encode(), _host(a), a.ndim, dims))
        check(lib().xdet_net_set_option(self.handle, b'large_sep', large_sep.encode()))
        check(lib().xdet_net_set_option(self.handle, b'sepconv', sepconv.encode()))
        check(lib().xdet_net_set_option(self.handle, b'rpn_stream', rpn_stream.encode()))
        check(lib().xdet_net_set_option(self.handle, b'conv3x3', conv3x3.encode()))
        check(lib().xdet_net_set_option(self.handle, b'pool', pool.encode()))
        check(lib().xdet_net_set_option(self.handle, b'check_range', b'on' if check_range else b'off'))
        if workspace is not None:
            check(lib().xdet_net_set_option(self.handle, b'workspace', workspace.encode()))
        if pool_sub is not None:
            check(lib().xdet_net_set_option(self.handle, b'pool_sub', pool_sub.encode()))
        check(lib().xdet_net_set_option(self.handle, b'ksplit', ksplit.encode() if isinstance(ksplit, str) else (b'on' if ksplit else b'off')))
        check(lib().xdet_net_set_option(self.handle, b'cross', cross.encode()))
        check(lib().xdet_net_build(self.handle))
        self.max_batch = max_batch
        self.image_size = image_size
        self.num_classes = num_classes
        self.R = rpn_post_nms_top_n
        self.nms_topk = nms_topk
        self.stream = Stream()
        B, nc, k = max_batch, num_classes - 1, nms_topk
        self._images = DeviceBuffer(B * 3 * image_size * image_size * 4)
        self._det_scores = DeviceBuffer(B * nc * k * 4)
        self._det_boxes = DeviceBuffer(B * nc * k * 16)
        self._N = 0

    # ---- plumbing -------------------------------------------------------------------
    def buffer(self, name, n=None):
        p = c_void_p()
        dims = (ctypes.c_int64 * 4)()
        ld = ctypes.c_int()
        check(lib().xdet_net_buffer(self.handle, name.encode(), ctypes.byref(p), ctypes.byref(dims), ctypes.byref(ld)))
        d = list(dims)
        d[0] = n or self._N or d[0]
        return DeviceTensor(p.value, d, ld.value, owner=self)

    def flat(self, name, shape, dtype=np.float32):
        p = c_void_p()
        dims = (ctypes.c_int64 * 4)()
        ld = ctypes.c_int()
        check(lib().xdet_net_buffer(self.handle, name.encode(), ctypes.byref(p), ctypes.byref(dims), ctypes.byref(ld)))
        return to_host(p.value, shape, dtype)

    def write(self, name, array):
        """overwrite a workspace buffer from the host (tests: feed one stage the oracle's tensors)."""
        t = self.buffer(name, n=array.shape[0])
        a = np.asarray(array, np.float32)
        if name in ('objectness', 'rpn_boxes', 'proposals', 'head_boxes'):
            flat = np.ascontiguousarray(a)
        else:
            n, h, w, c = a.shape
            flat = np.zeros((n, h, w, t.ld), np.float32)
            flat[..., :c] = a
        check(lib().xdet_memcpy_h2d(t.ptr, _host(flat), flat.nbytes, None))
        synchronize()
        self._N = a.shape[0]

    @contextlib.contextmanager
    def scope(self):
        _current.append(self)
        try:
            yield self
        finally:
            _current.pop()

    def set_images(self, images_nchw):
        a = np.ascontiguousarray(images_nchw, np.float32)
        assert a.ndim == 4 and a.shape[1] == 3 and a.shape[2] == a.shape[3] == self.image_size, a.shape
        assert a.shape[0] <= self.max_batch
        check(lib().xdet_memcpy_h2d(self._images.ptr, _host(a), a.nbytes, self.stream.handle))
        self._N = a.shape[0]
        return self._N

    # ---- the whole forward (lighr_head_model_fn, eval) -----------------------------------
    def forward_device(self, n=None, use_graph=False, images_ptr=None, image_shapes_ptr=None, bbox_img_ptr=None,
                       det_scores_ptr=None, det_boxes_ptr=None):
        """asynchronous: images already resident (set_images or caller-owned device pointer)."""
        n = n or self._N
        check(lib().xdet_net_forward(self.handle, images_ptr or self._images.ptr, n, image_shapes_ptr, bbox_img_ptr,
                                     det_scores_ptr or self._det_scores.ptr, det_boxes_ptr or self._det_boxes.ptr,
                                     1 if use_graph else 0, self.stream.handle))

    def calibrate(self, images_nchw):
        """Choose the activation pre-scale of the split-precision operands from a calibration batch (whitened f32
        [N,3,S,S], N <= max_batch): tensors whose magnitude comes near the f16 range of the hi/lo planes get a
        power-of-two exponent (planes hold x * 2^-e, folded back exactly by the consumer).  Returns {name: e} of the
        tensors that were scaled -- empty for a net whose activations are small, which then stays bit-identical."""
        n = self.set_images(images_nchw)
        k = ctypes.c_int()
        check(lib().xdet_net_calibrate(self.handle, self._images.ptr, n, ctypes.byref(k), self.stream.handle))
        return {name: e for name, e in self.plane_scales().items() if e}

    def x8_planes(self):
        """tensors in the x8 form (cross='fp8', after calibrate()): xdet_net_x8_planes"""
        n = ctypes.c_int()
        check(lib().xdet_net_x8_planes(self.handle, ctypes.byref(n)))
        return n.value

    def plane_scales(self):
        cnt = ctypes.c_int()
        check(lib().xdet_net_plane_scales(self.handle, 0, ctypes.byref(cnt), None))
        ex = (ctypes.c_int * max(cnt.value, 1))()
        check(lib().xdet_net_plane_scales(self.handle, cnt.value, ctypes.byref(cnt), ex))
        buf = ctypes.create_string_buffer(256)
        out = {}
        for i in range(cnt.value):
            check(lib().xdet_net_plane_scale_name(self.handle, i, buf, 256))
            out['%d: %s' % (i, buf.value.decode())] = ex[i]
        return out

    def detections(self, n=None):
        n = n or self._N
        nc, k = self.num_classes - 1, self.nms_topk
        self.stream.synchronize()
        s = to_host(self._det_scores.ptr, (n, nc, k), np.float32)
        b = to_host(self._det_boxes.ptr, (n, nc, k, 4), np.float32)
        if np.isnan(s[:, :, 0]).any():
            # bboxes_eval marks an (image, class) slot NaN when a head logit was not finite (csrc/detect.hip)
            raise XdetError(-4, 'non-finite network outputs / out-of-range activations for image(s) %s: an activation '
                                'left the f16 range of the split-precision convs (|x| > 65504) or the weights are '
                                'broken; use set_precision("f32")'
                            % sorted(set(np.argwhere(np.isnan(s[:, :, 0]))[:, 0].tolist())))
        return s, b

    def forward(self, images_nchw, use_graph=False):
        """images [N,3,S,S] whitened f32 -> list of {class: (scores[topk], boxes[topk,4])}, the
        per-class zero-padded output of bboxes_eval (light_head_rfcn_eval.py:263-287)."""
        n = self.set_images(images_nchw)
        self.forward_device(n, use_graph)
        s, b = self.detections(n)
        return [{c + 1: (s[i, c], b[i, c]) for c in range(self.num_classes - 1)} for i in range(n)]

    def predictions(self, n=None):
        """the `predictions` dict of the EstimatorSpec (light_head_rfcn_eval.py:429-433)."""
        n = n or self._N
        nc = self.num_classes
        cr = self.buffer('cls_reg', n).numpy().reshape(n, self.R, -1)
        logits = cr[..., :nc]
        e = np.exp(logits - logits.max(-1, keepdims=True))
        prob = e / e.sum(-1, keepdims=True)
        hb = self.flat('head_boxes', (n, self.R, 4))
        return {'classes': prob.argmax(-1), 'probabilities': prob.max(-1), 'bboxes_predict': hb}

    def memory(self):
        """{'allocated_bytes': device memory of workspace + weights, 'recycled_bytes': tensors placed into recycled blocks}"""
        a, r = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().xdet_net_memory(self.handle, ctypes.byref(a), ctypes.byref(r)))
        return {'allocated_bytes': int(a.value), 'recycled_bytes': int(r.value)}

    def flops_per_image(self):
        v = [ctypes.c_double() for _ in range(4)]
        check(lib().xdet_net_flops_per_image(self.handle, *[ctypes.byref(x) for x in v]))
        return dict(zip(('backbone', 'rpn', 'large_sep', 'head'), [x.value for x in v]))

    def __del__(self):
        try:
            lib().xdet_net_destroy(self.handle)
        except Exception:
            pass


class PipelinedDetector(object):
    """`ways` LightHeadDetector instances, each on its own HIP stream, that process contiguous slices of
    one batch concurrently.  Images are independent units and a detection does not depend on the batch it
    was computed in (tests/test_gpu_fullsize.py::test_batch_invariance), so the result equals the single
    detector's bit for bit; what changes is throughput: the partial last round of workgroups of one
    launch is filled by the other stream's kernels (about +5 % at 2 x 64 images on an MI355X)."""

    def __init__(self, weights, ways=2, max_batch=2, **kw):
        assert ways >= 1 and max_batch >= ways
        self.ways = ways
        self.sub = -(-max_batch // ways)
        self.max_batch = self.sub * ways
        self.nets = [LightHeadDetector(weights, max_batch=self.sub, **kw) for _ in range(ways)]
        n0 = self.nets[0]
        self.image_size, self.num_classes, self.R, self.nms_topk = n0.image_size, n0.num_classes, n0.R, n0.nms_topk
        self._counts = [0] * ways

    def set_images(self, images_nchw):
        n = images_nchw.shape[0]
        assert n <= self.max_batch
        self._counts = []
        for i, net in enumerate(self.nets):
            part = images_nchw[i * self.sub:min(n, (i + 1) * self.sub)]
            self._counts.append(part.shape[0])
            if part.shape[0]:
                net.set_images(part)
        return n

    def forward_device(self, use_graph=True):
        """asynchronous: one launch sequence (or graph replay) per sub-batch, each on its own stream"""
        for net, c in zip(self.nets, self._counts):
            if c:
                net.forward_device(c, use_graph=use_graph)

    def synchronize(self):
        for net in self.nets:
            net.stream.synchronize()

    def detections(self):
        parts = [net.detections(c) for net, c in zip(self.nets, self._counts) if c]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def forward(self, images_nchw, use_graph=True):
        """same contract as LightHeadDetector.forward"""
        n = self.set_images(images_nchw)
        self.forward_device(use_graph)
        s, b = self.detections()
        return [{c + 1: (s[i, c], b[i, c]) for c in range(self.num_classes - 1)} for i in range(n)]

    def flops_per_image(self):
        return self.nets[0].flops_per_image()


def _det():
    if not _current:
        raise XdetError(-3, 'no current LightHeadDetector: use `with detector.scope():`')
    return _current[-1]


def _sync(d):
    d.stream.synchronize()


def _same(t, view):
    return isinstance(t, DeviceTensor) and t.ptr == view.ptr


def XceptionBody(input_image, num_classes, is_training=False, data_format='channels_last'):
    """net/xception_body.py:236-379 -> (mid_outputs, outputs) as DeviceTensors (NHWC)."""
    assert not is_training, 'forward-only path'
    d = _det()
    a = np.asarray(input_image, np.float32)
    if data_format == 'channels_last':
        a = np.transpose(a, (0, 3, 1, 2))
    n = d.set_images(a)
    check(lib().xdet_net_xception_body(d.handle, d._images.ptr, n, d.stream.handle))
    _sync(d)
    return d.buffer('mid', n), d.buffer('out', n)


def get_rpn(net_input, num_anchors, is_training, data_format, var_scope):
    """net/xception_body.py:381-400 -> (rpn_cls_score [N,h,w,2A], rpn_bbox_pred [N,h,w,4A])."""
    d = _det()
    n = net_input.shape[0]
    if not _same(net_input, d.buffer('mid', n)):
        raise XdetError(-1, 'get_rpn expects the mid_outputs tensor returned by XceptionBody')
    check(lib().xdet_net_get_rpn(d.handle, n, d.stream.handle))
    _sync(d)
    out = d.buffer('rpn_out', n)
    return out.channels(0, 2 * num_anchors), out.channels(2 * num_anchors, 6 * num_anchors)


def large_sep_kernel(net_input, depth_mid, depth_output, is_training, data_format, var_scope):
    """net/xception_body.py:450-475 -> thin feature map [N,h,w,depth_output]."""
    d = _det()
    n = net_input.shape[0]
    view = d.buffer('out', n)
    if not _same(net_input, view):
        d.write('out', net_input.numpy() if isinstance(net_input, DeviceTensor) else net_input)
    check(lib().xdet_net_large_sep(d.handle, n, d.stream.handle))
    _sync(d)
    return d.buffer('feat', n)


def rpn_decode(rpn_cls_score, rpn_bbox_pred):
    """light_head_rfcn_eval.py:389-397 + labels['rpn_decode_fn'] -> (objectness [N,HWA], boxes [N,HWA,4])
    as numpy arrays; the device copies stay in the net for get_proposals."""
    d = _det()
    n = rpn_cls_score.shape[0]
    check(lib().xdet_net_rpn_decode(d.handle, n, d.stream.handle))
    _sync(d)
    na = d.buffer('objectness', n).shape[1]
    return d.flat('objectness', (n, na)), d.flat('rpn_boxes', (n, na, 4))


def get_proposals(object_score, bboxes_pred, encode_fn, rpn_pre_nms_top_n, rpn_post_nms_top_n, nms_threshold,
                  rpn_min_size, is_training, data_format):
    """net/xception_body.py:402-448 (eval branch) -> proposals [N,post_n,4] numpy."""
    assert not is_training, 'forward-only path'
    d = _det()
    # the native net bakes these in at build time (xdet_lighthead_config): a caller porting reference code
    # with other values must build the detector with them, not get silently different proposals
    want = (d.cfg.rpn_pre_nms_top_n, d.cfg.rpn_post_nms_top_n, np.float32(d.cfg.rpn_nms_thres), np.float32(d.cfg.rpn_min_size))
    got = (rpn_pre_nms_top_n, rpn_post_nms_top_n, np.float32(nms_threshold), np.float32(rpn_min_size))
    if want != got:
        raise InvalidArgumentError(-1, 'get_proposals: (pre_n, post_n, nms_threshold, rpn_min_size) = %r but the '
                                       'detector was built with %r' % (got, want))
    n = object_score.shape[0]
    d.write('objectness', np.asarray(object_score, np.float32))
    d.write('rpn_boxes', np.asarray(bboxes_pred, np.float32))
    check(lib().xdet_net_get_proposals(d.handle, n, d.stream.handle))
    _sync(d)
    return d.flat('proposals', (n, d.R, 4))


def get_head(net_input, pooling_op, grid_width, grid_height, loss_func, proposals_bboxes, num_classes, is_training,
             using_ohem, ohem_roi_one_image, data_format, var_scope):
    """net/xception_body.py:477-560 (eval, no OHEM) -> (cls_score [N,R,nc], bboxes_reg [N,R,4]) numpy.
    `pooling_op` is accepted for signature parity; the fused HIP PsRoiAlign is always used."""
    assert not is_training and not using_ohem, 'forward-only path'
    d = _det()
    if (grid_width, grid_height) != (d.cfg.grid, d.cfg.grid) or num_classes != d.cfg.num_classes:
        raise InvalidArgumentError(-1, 'get_head: grid %dx%d / %d classes but the detector was built with %dx%d / %d'
                                   % (grid_width, grid_height, num_classes, d.cfg.grid, d.cfg.grid, d.cfg.num_classes))
    n = proposals_bboxes.shape[0]
    view = d.buffer('feat', n)
    if not _same(net_input, view):
        d.write('feat', net_input.numpy() if isinstance(net_input, DeviceTensor) else net_input)
    d.write('proposals', np.asarray(proposals_bboxes, np.float32))
    check(lib().xdet_net_get_head(d.handle, n, d.stream.handle))
    _sync(d)
    cr = d.buffer('cls_reg', n).numpy().reshape(n, d.R, -1)
    return cr[..., :num_classes], cr[..., num_classes:num_classes + 4]
