"""xdet: MI355X-native Light-Head R-CNN forward path (host side of libxdet_hip.so).

The names below mirror the interfaces of HiKapok/X-Detector's eval path:
  ps_roi_align            <- op_module.ps_roi_align          (light_head_rfcn_eval.py:143-155)
  ps_roi_align_grad       <- op_module.ps_roi_align_grad     (cpp/PSROIPooling/test_op.py:93-104)
  XceptionBody, get_rpn, get_proposals, large_sep_kernel, get_head
                          <- net/xception_body.py:236,381,402,450,477
  AnchorCreator, ext_decode_rois
                          <- preprocessing/anchor_manipulator.py:686-757, 671-683
  bboxes_eval             <- light_head_rfcn_eval.py:263-287
Importing this package does not load the HIP library; the first op call does and fails
loudly if it is missing (no CPU fallback).
"""
from ._lib import XdetError, InvalidArgumentError, LightHeadConfig, lib      # noqa: F401
from . import weights                                                         # noqa: F401


def __getattr__(name):
    import importlib
    for mod in ('ops', 'model', 'resnet', 'runtime'):
        m = importlib.import_module('.' + mod, __name__)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
