"""Operator-level host API: the reference's custom op and the box algebra around it,
same names / argument meaning / error behaviour as the reference, running on HIP.

Inputs may be numpy arrays (copied to the GPU, results copied back -- convenient for
tests) or DeviceTensor / DeviceBuffer objects (stay on the GPU).
"""
import ctypes

import numpy as np

from ._lib import lib, check, c_void_p, InvalidArgumentError
from .runtime import DeviceBuffer, DeviceTensor, to_device, to_host, synchronize, _ptr, _host, channel_ld


def ps_roi_align(inputs, rois, grid_dim_width, grid_dim_height, pool_method, stream=None):
    """op_module.ps_roi_align (light_head_rfcn_eval.py:143-155; REGISTER_OP
    cpp/PSROIPooling/ps_roi_align_op.cc:38-76).

    inputs [N,C,H,W] f32 NCHW, rois [N,R,4] (cy,cx,h,w) in [0,1]
    -> (pooled_features [N,R,gh*gw,C/(gh*gw)] f32, pooled_index same shape i32).
    Raises InvalidArgumentError for the cases the reference's OP_REQUIRES reject
    (ps_roi_align_op.cc:209-226)."""
    if not isinstance(pool_method, str) or ('mean' not in pool_method and 'max' not in pool_method):
        raise InvalidArgumentError(-1, "Need Attr pool_method to be either 'mean' or 'max', got %r" % (pool_method,))
    if grid_dim_width < 0 or grid_dim_height < 0:
        raise InvalidArgumentError(-1, 'Need Attr grid_dim_width/grid_dim_height >= 0')
    inputs = np.asarray(inputs, np.float32)
    rois = np.asarray(rois, np.float32)
    if inputs.ndim != 4:
        raise InvalidArgumentError(-1, "inputs must be in 'NCHW' format.")
    if rois.ndim != 3 or rois.shape[2] != 4:
        raise InvalidArgumentError(-1, "rois must be in 'batch_size x num_rois x 4' format.")
    if inputs.shape[0] != rois.shape[0]:
        raise InvalidArgumentError(-1, "'batch_size' in inputs and rois don't match.")
    N, C, H, W = inputs.shape
    R = rois.shape[1]
    gs = grid_dim_width * grid_dim_height
    if gs == 0 or C % gs != 0:
        raise InvalidArgumentError(-1, 'channels must be divisible by grid_dim_width * grid_dim_height')
    d_in, d_roi = to_device(inputs), to_device(rois)
    n_out = N * R * C
    d_pool, d_idx = DeviceBuffer(max(n_out * 4, 16)), DeviceBuffer(max(n_out * 4, 16))
    check(lib().xdet_psroialign_fwd(d_in.ptr, d_roi.ptr, d_pool.ptr, d_idx.ptr, N, C, H, W, R, grid_dim_width,
                                    grid_dim_height, 1 if 'max' in pool_method else 0, 0, C, C, 0,
                                    stream.handle if stream else None))
    shape = (N, R, gs, C // gs)
    return to_host(d_pool.ptr, shape, np.float32, stream), to_host(d_idx.ptr, shape, np.int32, stream)


def ps_roi_align_grad(inputs, rois, pooled_features_grad, pooled_index, grid_dim_width, grid_dim_height, pool_method,
                      stream=None):
    """op_module.ps_roi_align_grad (REGISTER_OP cpp/PSROIPooling/ps_roi_align_grad_op.cc:39-57; registered as
    the gradient of PsRoiAlign in cpp/PSROIPooling/test_op.py:93-104).

    inputs [N,C,H,W] (only its shape is used, as in the reference), rois [N,R,4],
    pooled_features_grad / pooled_index [N,R,gh*gw,C/(gh*gw)] -> grad_output [N,C,H,W] f32."""
    if not isinstance(pool_method, str) or ('mean' not in pool_method and 'max' not in pool_method):
        raise InvalidArgumentError(-1, "Need Attr pool_method to be either 'mean' or 'max', got %r" % (pool_method,))
    shape = tuple(inputs.shape)
    rois = np.asarray(rois, np.float32)
    if len(shape) != 4:
        raise InvalidArgumentError(-1, "inputs must be in 'NCHW' format.")
    if rois.ndim != 3 or rois.shape[2] != 4:
        raise InvalidArgumentError(-1, "rois must be in 'batch_size x num_rois x 4' format.")
    if shape[0] != rois.shape[0]:
        raise InvalidArgumentError(-1, "'batch_size' in inputs and rois don't match.")
    N, C, H, W = shape
    R = rois.shape[1]
    gs = grid_dim_width * grid_dim_height
    if gs <= 0 or C % gs != 0:
        raise InvalidArgumentError(-1, 'channels must be divisible by grid_dim_width * grid_dim_height')
    grad = np.ascontiguousarray(pooled_features_grad, np.float32)
    index = np.ascontiguousarray(pooled_index, np.int32)
    if grad.size != N * R * C or index.size != N * R * C:
        raise InvalidArgumentError(-1, 'pooled_features_grad / pooled_index must hold batch_size*num_rois*channels '
                                       'elements')
    d_roi, d_grad, d_idx = to_device(rois), to_device(grad), to_device(index)
    d_out = DeviceBuffer(max(N * C * H * W * 4, 16))
    check(lib().xdet_psroialign_grad(d_roi.ptr, d_grad.ptr, d_idx.ptr, d_out.ptr, N, C, H, W, R, grid_dim_width,
                                     grid_dim_height, 1 if 'max' in pool_method else 0, 0, C,
                                     stream.handle if stream else None))
    return to_host(d_out.ptr, shape, np.float32, stream)


PAD_VALID, PAD_SAME, PAD_EXPLICIT = 0, 1, 2


class Conv2D(object):
    """tf.layers.conv2d / dense with folded inference BN or bias (+ReLU) on the f32 MFMA pipe."""
    def __init__(self, kernel_hwio, stride=1, padding='SAME', dilation=1, scale=None, shift=None, relu=False,
                 explicit_pad=0):
        k = np.ascontiguousarray(kernel_hwio, np.float32)
        self.kh, self.kw, self.cin, self.cout = k.shape
        mode = {'VALID': PAD_VALID, 'SAME': PAD_SAME, 'EXPLICIT': PAD_EXPLICIT}[padding]
        sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
        sh = None if shift is None else np.ascontiguousarray(shift, np.float32)
        h = c_void_p()
        check(lib().xdet_conv_create(ctypes.byref(h), self.kh, self.kw, self.cin, self.cout, stride, dilation, mode,
                                     explicit_pad, explicit_pad, _host(k), _host(sc) if sc is not None else None,
                                     _host(sh) if sh is not None else None, 1 if relu else 0))
        self.handle = h

    def set_ksplit(self, ksplit, mode=0, max_parallel_tiles=None):
        """fixed split of the reduction (xdet_conv_set_ksplit): the planes path then runs on the split-K kernel;
        mode 0 = by grid size, 1 = ranges in parallel, 2 = one workgroup per tile -- all bit-identical."""
        if max_parallel_tiles is None:       # the parallel form is chosen for grids up to 448 (tile, range) workgroups
            max_parallel_tiles = 448 // max(int(ksplit), 1)
        check(lib().xdet_conv_set_ksplit(self.handle, int(ksplit), int(mode), int(max_parallel_tiles)))

    def __call__(self, x, residual=None, relu_in=False, stream=None, planes=False, staged_tile=False, x8_exp=None):
        """planes=True (split-precision modes only): split x into f16 hi/lo planes first and run the
        LDS-DMA kernel -- the path every big contraction takes inside a net.  staged_tile=True (with planes; 3x3 VALID
        stride 1 over 32 channels, <= 64 outputs): the kernel that stages the input tile once in LDS
        (xdet_conv3x3_patch_forward, block1_conv2 inside a net).  x8_exp (with planes; 1x1 stride-1 layers): the x8 form of
        the planes -- the cross terms of the split-precision product from fp8 copies (xdet_conv_forward_planes_x8); pass the
        exponent e with max|x| * 2^-e in (128, 256]."""
        N, H, W, C = x.shape
        assert C == self.cin, (C, self.cin)
        ho, wo = ctypes.c_int(), ctypes.c_int()
        check(lib().xdet_conv_out_shape(self.handle, H, W, ctypes.byref(ho), ctypes.byref(wo)))
        out = DeviceTensor.empty((N, ho.value, wo.value, self.cout))
        if planes:
            n = -(-N * H * W // 16) * 16 * x.ld
            hi, lo = DeviceBuffer(n * 2 + 512, zero=True), DeviceBuffer(n * 2 + 512, zero=True)
            st = stream.handle if stream else None
            if x8_exp is not None:
                # the x8 form of the planes (pointwise layers): cross terms from fp8 copies scaled by 2^-x8_exp
                assert not staged_tile
                check(lib().xdet_split_f32_x8(x.ptr, hi.ptr, lo.ptr, N * H * W, x.ld, 1 if relu_in else 0, int(x8_exp), st))
                check(lib().xdet_conv_forward_planes_x8(self.handle, hi.ptr, lo.ptr, N, H, W, x.ld, out.ptr, out.ld,
                                                        residual.ptr if residual is not None else None, int(x8_exp), st))
                synchronize(stream)
                return out
            check(lib().xdet_split_f32(x.ptr, hi.ptr, lo.ptr, N * H * W, x.ld, 1 if relu_in else 0, st))
            if staged_tile:
                assert residual is None
                check(lib().xdet_conv3x3_patch_forward(self.handle, hi.ptr, lo.ptr, N, H, W, out.ptr, out.ld, st))
            else:
                check(lib().xdet_conv_forward_planes(self.handle, hi.ptr, lo.ptr, N, H, W, x.ld, out.ptr, out.ld,
                                                     residual.ptr if residual is not None else None, st))
            synchronize(stream)
            return out
        check(lib().xdet_conv_forward(self.handle, x.ptr, N, H, W, x.ld, out.ptr, out.ld,
                                      residual.ptr if residual is not None else None, 1 if relu_in else 0,
                                      stream.handle if stream else None))
        return out

    def __del__(self):
        try:
            lib().xdet_layer_destroy(self.handle)
        except Exception:
            pass


class DepthwiseConv2D(object):
    """depthwise half of tf.layers.separable_conv2d (3x3, stride 1, SAME)."""
    def __init__(self, dw_kernel, dilation=1):
        k = np.ascontiguousarray(dw_kernel, np.float32)
        assert k.shape[:2] == (3, 3) and k.shape[3] == 1
        self.C = k.shape[2]
        h = c_void_p()
        check(lib().xdet_depthwise_create(ctypes.byref(h), self.C, dilation, _host(k)))
        self.handle = h

    def __call__(self, x, relu_in=False, stream=None):
        N, H, W, C = x.shape
        out = DeviceTensor.empty((N, H, W, C))
        check(lib().xdet_depthwise_forward(self.handle, x.ptr, N, H, W, x.ld, out.ptr, 1 if relu_in else 0,
                                           stream.handle if stream else None))
        return out

    def __del__(self):
        try:
            lib().xdet_layer_destroy(self.handle)
        except Exception:
            pass


class SeparableConvBN(object):
    """relu_separable_bn_block (net/xception_body.py:220-234): (ReLU ->) depthwise 3x3 -> pointwise 1x1 -> BN
    (folded into scale/shift) (-> ReLU), stride 1, SAME.  fused=True runs the one-kernel form
    (xdet_sepconv_fused_forward; split-precision modes, <= 256 input channels, 128 / 256 outputs), fused=False
    depthwise -> split planes -> pointwise, the form the wide layers of a net use; both give the same bits."""
    def __init__(self, dw_kernel, pw_kernel, scale=None, shift=None, relu=False, dilation=1):
        self.dw = DepthwiseConv2D(dw_kernel, dilation)
        self.pw = Conv2D(pw_kernel, 1, 'SAME', 1, scale, shift, relu)

    def __call__(self, x, relu_in=False, fused=True, stream=None):
        N, H, W, C = x.shape
        if not fused:
            return self.pw(self.dw(x, relu_in=relu_in, stream=stream), stream=stream, planes=True)
        out = DeviceTensor.empty((N, H, W, self.pw.cout))
        check(lib().xdet_sepconv_fused_forward(self.dw.handle, self.pw.handle, x.ptr, N, H, W, x.ld, out.ptr, out.ld,
                                               1 if relu_in else 0, stream.handle if stream else None))
        synchronize(stream)
        return out


def separable_block_then_pool_add(op, x, residual, relu_in=False, stream=None):
    """SeparableConvBN `op` -> max_pooling2d(3, 2, 'same') -> + residual with the pool split between the block's epilogue
    (horizontal half) and a light vertical pass (xdet_sepconv_fused_hpool_forward + xdet_maxpool_v3s2_add)."""
    N, H, W, C = x.shape
    Ho, Wo = -(-H // 2), -(-W // 2)
    hp = DeviceTensor.empty((N, H, Wo, op.pw.cout))
    out = DeviceTensor.empty((N, Ho, Wo, op.pw.cout))
    st = stream.handle if stream else None
    check(lib().xdet_sepconv_fused_hpool_forward(op.dw.handle, op.pw.handle, x.ptr, N, H, W, x.ld, hp.ptr, hp.ld,
                                                 1 if relu_in else 0, st))
    check(lib().xdet_maxpool_v3s2_add(hp.ptr, residual.ptr if residual is not None else None, out.ptr, N, H, Wo,
                                      op.pw.cout, hp.ld, st))
    synchronize(stream)
    return out


def max_pool_3x3_s2_same_add(x, residual=None, stream=None):
    N, H, W, C = x.shape
    out = DeviceTensor.empty((N, -(-H // 2), -(-W // 2), C))
    check(lib().xdet_maxpool3x3s2_add(x.ptr, residual.ptr if residual is not None else None, out.ptr, N, H, W, C,
                                      x.ld, stream.handle if stream else None))
    return out


class AnchorCreator(object):
    """preprocessing/anchor_manipulator.py:686-757 (single feature layer)."""
    def __init__(self, img_shape, layers_shapes, anchor_scales, extra_anchor_scales, anchor_ratios, layer_steps):
        self._img_shape = img_shape
        self._layers_shapes = layers_shapes
        self._anchor_scales = anchor_scales
        self._extra_anchor_scales = extra_anchor_scales
        self._anchor_ratios = anchor_ratios
        self._layer_steps = layer_steps

    def get_layer_anchors(self, layer_shape, anchor_scale, extra_anchor_scale, anchor_ratio, layer_step, offset=0.5):
        import math
        f = np.float32
        xs, ys = np.meshgrid(np.arange(layer_shape[1]), np.arange(layer_shape[0]))
        y = ((ys.astype(f) + f(offset)) * f(layer_step) / f(self._img_shape[0])).astype(f)
        x = ((xs.astype(f) + f(offset)) * f(layer_step) / f(self._img_shape[1])).astype(f)
        hs = [s for s in extra_anchor_scale] + [s / math.sqrt(r) for s in anchor_scale for r in anchor_ratio]
        ws = [s for s in extra_anchor_scale] + [s * math.sqrt(r) for s in anchor_scale for r in anchor_ratio]
        return y, x, np.array(hs, f), np.array(ws, f), len(hs)

    def get_all_anchors(self):
        all_anchors, num = [], []
        for i, shp in enumerate(self._layers_shapes):
            a = self.get_layer_anchors(shp, self._anchor_scales[i], self._extra_anchor_scales[i],
                                       self._anchor_ratios[i], self._layer_steps[i])
            all_anchors.append(a[:-1])
            num.append(a[-1])
        return all_anchors, num


def rpn_decode(rpn_cls, rpn_box, anchors, stream=None):
    """RPN glue + AnchorEncoder.decode_all_anchors(squeeze_inner=True)
    (light_head_rfcn_eval.py:389-397, anchor_manipulator.py:641-669).
    rpn_cls [N,Hh,Ww,2A], rpn_box [N,Hh,Ww,4A] numpy NHWC -> objectness [N,HWA], boxes [N,HWA,4]."""
    rpn_cls = np.asarray(rpn_cls, np.float32)
    rpn_box = np.asarray(rpn_box, np.float32)
    N, Hh, Ww, A2 = rpn_cls.shape
    A = A2 // 2
    both = np.concatenate([rpn_cls, rpn_box], axis=-1)
    t = DeviceTensor.from_numpy(both)
    yref, xref, href, wref = anchors
    yx = to_device(np.stack([yref.reshape(-1), xref.reshape(-1)], axis=1).astype(np.float32))
    hw = to_device(np.stack([href, wref], axis=1).astype(np.float32))
    n_anchor = Hh * Ww * A
    d_obj, d_box = DeviceBuffer(N * n_anchor * 4), DeviceBuffer(N * n_anchor * 16)
    check(lib().xdet_rpn_decode(t.ptr, t.ld, 0, 2 * A, N, Hh, Ww, A, yx.ptr, hw.ptr, d_obj.ptr, d_box.ptr,
                                stream.handle if stream else None))
    return to_host(d_obj.ptr, (N, n_anchor), np.float32, stream), to_host(d_box.ptr, (N, n_anchor, 4), np.float32, stream)


def get_proposals(object_score, bboxes_pred, encode_fn=None, rpn_pre_nms_top_n=5000, rpn_post_nms_top_n=1000,
                  nms_threshold=0.7, rpn_min_size=16. / 480, is_training=False, data_format='channels_first',
                  return_counts=False, stream=None):
    """net/xception_body.py:402-448, eval branch (is_training must be False).
    object_score [N,n], bboxes_pred [N,n,4] -> proposals [N,post_n,4]."""
    if is_training:
        raise NotImplementedError('forward-only path: is_training=True is out of scope')
    s = np.ascontiguousarray(object_score, np.float32)
    b = np.ascontiguousarray(bboxes_pred, np.float32)
    N, n = s.shape
    d_s, d_b = to_device(s), to_device(b)
    ws = DeviceBuffer(lib().xdet_proposals_workspace_bytes(N, n, rpn_pre_nms_top_n, rpn_post_nms_top_n), zero=True)
    d_r = DeviceBuffer(N * rpn_post_nms_top_n * 16)
    d_c = DeviceBuffer(N * 16)
    check(lib().xdet_get_proposals(d_s.ptr, d_b.ptr, N, n, rpn_pre_nms_top_n, rpn_post_nms_top_n, nms_threshold,
                                   rpn_min_size, ws.ptr, d_r.ptr, d_c.ptr, stream.handle if stream else None))
    rois = to_host(d_r.ptr, (N, rpn_post_nms_top_n, 4), np.float32, stream)
    if return_counts:
        return rois, to_host(d_c.ptr, (N, 4), np.int32, stream)
    return rois


def ext_decode_rois(proposals_roi, pred_location, head_prior_scaling=(1., 1., 1., 1.), stream=None):
    """AnchorEncoder.ext_decode_rois (anchor_manipulator.py:671-683); scaling is the eval default 1."""
    assert tuple(head_prior_scaling) == (1., 1., 1., 1.)
    r = np.ascontiguousarray(proposals_roi, np.float32)
    p = np.ascontiguousarray(pred_location, np.float32)
    n = int(np.prod(r.shape[:-1]))
    d_r, d_p, d_o = to_device(r), to_device(p), DeviceBuffer(max(n * 16, 16))
    check(lib().xdet_ext_decode_rois(d_r.ptr, d_p.ptr, 4, n, d_o.ptr, stream.handle if stream else None))
    return to_host(d_o.ptr, r.shape, np.float32, stream)


def bboxes_eval(cls_pred_logits, bboxes_pred, image_shape=(480, 480), bbox_img=(0., 0., 1., 1.), num_classes=21,
                select_threshold=0.01, nms_threshold=0.3, nms_topk=200, train_image_size=480, stream=None):
    """Detection part of bboxes_eval (light_head_rfcn_eval.py:263-287).
    cls_pred_logits [R,num_classes] (or [N,R,nc]), bboxes_pred [R,4] -> {c: (scores[topk], boxes[topk,4])}
    (a list of such dicts for batched input)."""
    c = np.asarray(cls_pred_logits, np.float32)
    b = np.asarray(bboxes_pred, np.float32)
    single = c.ndim == 2
    if single:
        c, b = c[None], b[None]
    N, R, nc = c.shape
    shapes = np.broadcast_to(np.asarray(image_shape, np.int32).reshape(-1, 2), (N, 2))
    bimg = np.broadcast_to(np.asarray(bbox_img, np.float32).reshape(-1, 4), (N, 4))
    d_c, d_b = to_device(c), to_device(b)
    d_s, d_i = to_device(np.ascontiguousarray(shapes)), to_device(np.ascontiguousarray(bimg))
    d_os, d_ob = DeviceBuffer(N * (nc - 1) * nms_topk * 4), DeviceBuffer(N * (nc - 1) * nms_topk * 16)
    check(lib().xdet_bboxes_eval(d_c.ptr, nc, d_b.ptr, N, R, nc, d_s.ptr, d_i.ptr, train_image_size,
                                 train_image_size, select_threshold, nms_threshold, nms_topk, d_os.ptr, d_ob.ptr,
                                 stream.handle if stream else None))
    sc = to_host(d_os.ptr, (N, nc - 1, nms_topk), np.float32, stream)
    bx = to_host(d_ob.ptr, (N, nc - 1, nms_topk, 4), np.float32, stream)
    out = [{k + 1: (sc[n, k], bx[n, k]) for k in range(nc - 1)} for n in range(N)]
    return out[0] if single else out


def light_head_preprocess_for_eval(image, labels=None, bboxes=None, out_shape=(480, 480), data_format='NHWC',
                                   difficults=None, stream=None):
    """preprocessing/common_preprocessing.py:383-440 (Resize.WARP_RESIZE, the eval default):
    uint8 [H,W,3] -> (image f32 [S,S,3] or [3,S,S], labels, bboxes, bbox_img=[0,0,1,1])."""
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError('Input must be of size [height, width, C>0]')
    assert out_shape[0] == out_shape[1], 'square network input'
    S = int(out_shape[0])
    d_in = to_device(img)
    d_out = DeviceBuffer(3 * S * S * 4)
    check(lib().xdet_preprocess_eval(d_in.ptr, img.shape[0], img.shape[1], d_out.ptr, S,
                                     stream.handle if stream else None))
    chw = to_host(d_out.ptr, (3, S, S), np.float32, stream)
    out = chw if data_format == 'NCHW' else np.ascontiguousarray(chw.transpose(1, 2, 0))
    return out, labels, bboxes, np.array([0., 0., 1., 1.], np.float32)


def light_head_preprocess_for_test(image, out_shape, data_format='NHWC', stream=None):
    """preprocessing/common_preprocessing.py:442-458."""
    return light_head_preprocess_for_eval(image, None, None, out_shape, data_format, stream=stream)[0]
