"""Single-image front end: the contract of the reference's light_head_simple_demo.py.

    labels, scores, bboxes = light_head_simple_demo(np_image_uint8, detector)

mirrors `sess.run([all_labels, all_scores, all_bboxes], {image_input: np_image, shape_input: np_image.shape[:-1]})`
(light_head_simple_demo.py:110-199): uint8 [H,W,3] image -> light_head_preprocess_for_test (whiten + warp to
480x480, channels_last) -> the Light-Head R-CNN forward with the demo's flags (select_threshold 0.5,
nms_threshold 0.3, nms_topk 20, rpn_post_nms_top_n 1000: :63-82) -> for every class 1..20 the nms_topk zero-padded
slots of tf_bboxes_select / bboxes_clip / filter_boxes(0.03, image shape) / bboxes_sort(2*nms_topk) /
bboxes_nms_batch, concatenated over the classes in ascending order (:180-189): three arrays of 20*nms_topk rows.
Everything between the uint8 image and those arrays runs on the GPU (F1 kernel, captured forward)."""
import numpy as np

from ._lib import lib, check
from .runtime import DeviceBuffer, _host, to_device, synchronize

DEMO_FLAGS = dict(num_classes=21, image_size=480, select_threshold=0.5, nms_threshold=0.3, nms_topk=20,
                  rpn_pre_nms_top_n=5000, rpn_post_nms_top_n=1000, rpn_nms_thres=0.7, rpn_min_size=16. / 480)


def make_demo_detector(weights, **overrides):
    """a LightHeadDetector built with the demo script's flags (light_head_simple_demo.py:41-82)."""
    from .model import LightHeadDetector
    kw = dict(DEMO_FLAGS, max_batch=1)
    kw.update(overrides)
    return LightHeadDetector(weights, **kw)


def light_head_simple_demo(np_image, detector, use_graph=False):
    """np_image uint8 [H,W,3] -> (labels int32 [20*topk], scores f32 [20*topk], bboxes f32 [20*topk,4]); boxes are
    (ymin, xmin, ymax, xmax) normalised to the image, unused slots have score 0 and a zero box."""
    img = np.ascontiguousarray(np_image, np.uint8)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError('Input must be of size [height, width, C>0]')
    S = detector.image_size
    d_img = to_device(img)
    # F1 writes the whitened, warped CHW planes straight into the detector's input buffer
    check(lib().xdet_preprocess_eval(d_img.ptr, img.shape[0], img.shape[1], detector._images.ptr, S,
                                     detector.stream.handle))
    # the image-shape operand lives in ONE device buffer per detector, refreshed on the detector's stream in front of
    # the forward: a captured graph bakes the pointer in (the cache is keyed on it), so a buffer allocated and freed per
    # call would capture a new graph for every image and leave the old ones pointing at freed memory
    if getattr(detector, '_demo_shape', None) is None:
        detector._demo_shape = DeviceBuffer(8)
    host_shape = np.array([[img.shape[0], img.shape[1]]], np.int32)
    check(lib().xdet_memcpy_h2d(detector._demo_shape.ptr, _host(host_shape), 8, detector.stream.handle))
    detector._N = 1
    detector.forward_device(1, use_graph=use_graph, image_shapes_ptr=detector._demo_shape.ptr)
    scores, boxes = detector.detections(1)           # [1, 20, topk], [1, 20, topk, 4]
    nc, k = scores.shape[1], scores.shape[2]
    labels = np.repeat(np.arange(1, nc + 1, dtype=np.int32), k)
    synchronize()
    return labels, scores[0].reshape(-1).copy(), boxes[0].reshape(-1, 4).copy()
