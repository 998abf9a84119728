#!/usr/bin/env python
"""Experiment: does running two half-batches on two streams (two net instances) fill the partial last
round of each conv launch?  python tools/dual_stream_bench.py --batch 64 --ways 2"""
import argparse
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'x-detector_amd'))
import xdet
from xdet import weights as W
from xdet.runtime import set_precision

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--ways', type=int, default=2)
ap.add_argument('--steps', type=int, default=10)
a = ap.parse_args()
set_precision('f16x3')
w = W.make_lighthead_weights()
b = a.batch // a.ways
dets = [xdet.LightHeadDetector(w, image_size=480, max_batch=b, rpn_post_nms_top_n=300) for _ in range(a.ways)]
imgs = W.synthetic_images(b, 480, seed=5)
for d in dets:
    d.set_images(imgs)
for _ in range(3):
    for d in dets:
        d.forward_device(b, use_graph=True)
for d in dets:
    d.stream.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    for d in dets:
        d.forward_device(b, use_graph=True)
for d in dets:
    d.stream.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print('ways %d x batch %d: %.3f ms/step  %.1f images/s' % (a.ways, b, dt * 1e3, a.batch / dt))
