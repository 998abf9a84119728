"""ctypes binding of libxdet_hip.so (the C-ABI declared in include/xdet.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the
ops raises -- the product path never routes through oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XDET_LIB: another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get('XDET_LIB') or os.path.join(_HERE, 'libxdet_hip.so')

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_int64 = ctypes.c_int64
c_double = ctypes.c_double
PF = c_void_p      # device float* / host float* passed as integers
PI = c_void_p


class XdetError(RuntimeError):
    """Any non-zero return code of the C-ABI."""
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'xdet error %d: %s' % (code, msg))
        self.code = code


class InvalidArgumentError(XdetError, ValueError):
    """XDET_ERR_INVALID_ARG: the counterpart of tf.errors.InvalidArgumentError raised by the
    reference op's OP_REQUIRES checks (cpp/PSROIPooling/ps_roi_align_op.cc:209-226)."""


class LightHeadConfig(ctypes.Structure):
    """xdet_lighthead_config; defaults are the reference's eval flags
    (light_head_rfcn_eval.py:76,98-113)."""
    _fields_ = [('image_size', c_int), ('max_batch', c_int), ('num_classes', c_int), ('num_anchors', c_int),
                ('rpn_pre_nms_top_n', c_int), ('rpn_post_nms_top_n', c_int), ('rpn_nms_thres', c_float),
                ('rpn_min_size', c_float), ('select_threshold', c_float), ('nms_threshold', c_float),
                ('nms_topk', c_int), ('grid', c_int), ('bank', c_int)]

    def __init__(self, image_size=480, max_batch=1, num_classes=21, num_anchors=22, rpn_pre_nms_top_n=5000,
                 rpn_post_nms_top_n=1000, rpn_nms_thres=0.7, rpn_min_size=16. / 480, select_threshold=0.01,
                 nms_threshold=0.3, nms_topk=200, grid=7, bank=10):
        ctypes.Structure.__init__(self, image_size, max_batch, num_classes, num_anchors, rpn_pre_nms_top_n,
                                  rpn_post_nms_top_n, rpn_nms_thres, rpn_min_size, select_threshold, nms_threshold,
                                  nms_topk, grid, bank)


# name -> (restype, argtypes); must list every symbol include/xdet.h declares
SIGNATURES = {
    'xdet_last_error': (ctypes.c_char_p, []),
    'xdet_version': (c_int, []),
    'xdet_device_count': (c_int, [ctypes.POINTER(c_int)]),
    'xdet_set_device': (c_int, [c_int]),
    'xdet_device_pci_bus_id': (c_int, [c_int, ctypes.c_char_p, c_int]),
    'xdet_probe_ipc': (c_int, []),
    'xdet_set_default_precision': (c_int, [c_int]),
    'xdet_get_default_precision': (c_int, []),
    'xdet_malloc': (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    'xdet_free': (c_int, [c_void_p]),
    'xdet_memset': (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    'xdet_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'xdet_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'xdet_memcpy_d2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'xdet_stream_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'xdet_stream_destroy': (c_int, [c_void_p]),
    'xdet_stream_sync': (c_int, [c_void_p]),
    'xdet_event_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'xdet_event_destroy': (c_int, [c_void_p]),
    'xdet_event_record': (c_int, [c_void_p, c_void_p]),
    'xdet_event_elapsed_ms': (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float)]),
    'xdet_psroialign_fwd': (c_int, [PF, PF, PF, PI] + [c_int] * 12 + [c_void_p]),
    'xdet_psroialign_grad': (c_int, [PF, PF, PI, PF] + [c_int] * 10 + [c_void_p]),
    'xdet_conv_create': (c_int, [ctypes.POINTER(c_void_p)] + [c_int] * 9 + [PF, PF, PF, c_int]),
    'xdet_conv_forward': (c_int, [c_void_p, PF, c_int, c_int, c_int, c_int, PF, c_int, PF, c_int, c_void_p]),
    'xdet_conv_out_shape': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'xdet_conv_set_ksplit': (c_int, [c_void_p, c_int, c_int, c_int]),
    'xdet_split_f32': (c_int, [PF, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'xdet_conv_forward_planes': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, PF, c_int, PF,
                                         c_void_p]),
    'xdet_split_f32_x8': (c_int, [PF, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'xdet_conv_forward_planes_x8': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, PF, c_int, PF, c_int,
                                            c_void_p]),
    'xdet_layer_destroy': (c_int, [c_void_p]),
    'xdet_depthwise_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, PF]),
    'xdet_depthwise_forward': (c_int, [c_void_p, PF, c_int, c_int, c_int, c_int, PF, c_int, c_void_p]),
    'xdet_sepconv_fused_forward': (c_int, [c_void_p, c_void_p, PF, c_int, c_int, c_int, c_int, PF, c_int, c_int, c_void_p]),
    'xdet_conv3x3_patch_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, PF, c_int, c_void_p]),
    'xdet_resnet_bneck_forward': (c_int, [c_void_p, c_void_p, c_void_p, PF, PF, PF, c_int, c_int, c_int, PF, PF, PF,
                                          c_void_p, c_void_p, c_void_p]),
    'xdet_sepconv_fused_hpool_forward': (c_int, [c_void_p, c_void_p, PF, c_int, c_int, c_int, c_int, PF, c_int, c_int,
                                                 c_void_p]),
    'xdet_maxpool_v3s2_add': (c_int, [PF, PF, PF, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'xdet_maxpool3x3s2_add': (c_int, [PF, PF, PF, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'xdet_preprocess_eval': (c_int, [c_void_p, c_int, c_int, PF, c_int, c_void_p]),
    'xdet_nchw_to_nhwc4': (c_int, [PF, PF, c_int, c_int, c_int, c_int, c_void_p]),
    'xdet_rpn_decode': (c_int, [PF, c_int, c_int, c_int, c_int, c_int, c_int, c_int, PF, PF, PF, PF, c_void_p]),
    'xdet_proposals_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'xdet_get_proposals': (c_int, [PF, PF, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, PF, PI, c_void_p]),
    'xdet_ext_decode_rois': (c_int, [PF, PF, c_int, c_int64, PF, c_void_p]),
    'xdet_bboxes_eval': (c_int, [PF, c_int, PF, c_int, c_int, c_int, PI, PF, c_int, c_int, c_float, c_float, c_int,
                                 PF, PF, c_void_p]),
    'xdet_net_create': (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(LightHeadConfig)]),
    'xdet_net_set_weight': (c_int, [c_void_p, ctypes.c_char_p, PF, c_int, ctypes.POINTER(c_int64)]),
    'xdet_net_set_option': (c_int, [c_void_p, ctypes.c_char_p, ctypes.c_char_p]),
    'xdet_net_build': (c_int, [c_void_p]),
    'xdet_net_destroy': (c_int, [c_void_p]),
    'xdet_net_buffer': (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64 * 4),
                                ctypes.POINTER(c_int)]),
    'xdet_net_xception_body': (c_int, [c_void_p, PF, c_int, c_void_p]),
    'xdet_net_get_rpn': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_large_sep': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_rpn_decode': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_get_proposals': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_get_head': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_head_decode': (c_int, [c_void_p, c_int, c_void_p]),
    'xdet_net_bboxes_eval': (c_int, [c_void_p, c_int, PI, PF, PF, PF, c_void_p]),
    'xdet_net_forward': (c_int, [c_void_p, PF, c_int, PI, PF, PF, PF, c_int, c_void_p]),
    'xdet_net_calibrate': (c_int, [c_void_p, PF, c_int, ctypes.POINTER(c_int), c_void_p]),
    'xdet_net_plane_scales': (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    'xdet_net_x8_planes': (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    'xdet_net_plane_scale_name': (c_int, [c_void_p, c_int, ctypes.c_char_p, c_int]),
    'xdet_net_graph_count': (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    'xdet_net_memory': (c_int, [c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'xdet_net_flops_per_image': (c_int, [c_void_p] + [ctypes.POINTER(c_double)] * 4),
    'xdet_profile_enable': (c_int, [c_void_p, c_int, c_int]),
    'xdet_profile_read': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_double),
                                  ctypes.POINTER(c_int), ctypes.POINTER(c_double)]),
    'xdet_profile_mfma_flops': (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_double)]),
    'xdet_profile_op_name': (c_int, [c_void_p, c_int, c_int, ctypes.c_char_p, c_int]),
    'xdet_resnet_create': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int]),
    'xdet_resnet_set_weight': (c_int, [c_void_p, ctypes.c_char_p, PF, c_int, ctypes.POINTER(c_int64)]),
    'xdet_resnet_set_option': (c_int, [c_void_p, ctypes.c_char_p, ctypes.c_char_p]),
    'xdet_resnet_build': (c_int, [c_void_p]),
    'xdet_resnet_forward': (c_int, [c_void_p, PF, c_int, PF, c_void_p]),
    'xdet_resnet_forward_graph': (c_int, [c_void_p, PF, c_int, PF, c_void_p]),
    'xdet_resnet_calibrate': (c_int, [c_void_p, PF, c_int, ctypes.POINTER(c_int), c_void_p]),
    'xdet_resnet_out_shape': (c_int, [c_void_p] + [ctypes.POINTER(c_int)] * 3),
    'xdet_resnet_flops_per_image': (c_int, [c_void_p, ctypes.POINTER(c_double)]),
    'xdet_resnet_destroy': (c_int, [c_void_p]),
    'xdet_comm_init': (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, ctypes.c_char_p, c_int]),
    'xdet_comm_destroy': (c_int, [c_void_p]),
    'xdet_comm_info': (c_int, [c_void_p] + [ctypes.POINTER(c_int)] * 4),
    'xdet_pack_detections': (c_int, [PF, PF, c_int64, PF, c_void_p]),
    'xdet_comm_allgather_detections': (c_int, [c_void_p, PF, PF, c_int, c_int, c_int, PF, PF,
                                               ctypes.POINTER(c_void_p), c_int, c_int]),
    'xdet_comm_wait': (c_int, [c_void_p, c_void_p]),
    'xdet_comm_allreduce_max': (c_int, [c_void_p, ctypes.POINTER(c_double)]),
    'xdet_comm_barrier': (c_int, [c_void_p]),
    'xdet_comm_allgather_bytes': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'xdet_comm_set_timeout': (c_int, [c_void_p, c_double]),
    'xdet_comm_library': (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_int)]),
}

_lib = None


def lib():
    """The loaded library; raises if libxdet_hip.so has not been built (xdet/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('libxdet_hip.so is missing (%s): run `python __graft_entry__.py` / '
                              'xdet/build.py; there is no CPU fallback' % LIB_PATH)
        # HSA_ENABLE_IPC_MODE_LEGACY is read by the HSA runtime when it starts, i.e. it must be decided before the
        # first HIP call.  Only a multi-rank process needs device-memory IPC (RCCL's intra-node transport); whether
        # the host driver wants the legacy or the dmabuf mode is probed once per launch (xdet.launch.ipc_env) --
        # never forced on a single-GPU user.
        if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ and int(os.environ.get('WORLD_SIZE', '1') or 1) > 1:
            from .launch import ipc_env
            os.environ.update(ipc_env())
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)  # AttributeError if the symbol is not exported
            except AttributeError:
                if os.environ.get('XDET_LIB'):
                    continue           # an older build loaded for an A/B measurement: it simply lacks the newer entries
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().xdet_last_error().decode('utf-8', 'replace')
    if rc == -1:
        raise InvalidArgumentError(rc, msg)
    raise XdetError(rc, msg)
