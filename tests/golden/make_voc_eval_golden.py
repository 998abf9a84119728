#!/usr/bin/env python
"""Golden vectors for F4 (eval bookkeeping) from the REFERENCE's own NumPy evaluation, run in this container:
the functions parse_rec / voc_ap / voc_eval are taken verbatim out of /root/reference/voc_eval.py (the module
itself cannot be imported: its `from dataset import dataset_common` pulls in TensorFlow, which the three
functions never touch) and executed on a small synthetic PASCAL-VOC style data set written to a temp dir
(XML annotations + per-class detection files).  Inputs and the reference's outputs (recall / precision curves,
VOC07 11-point AP and VOC12 area AP per class) are stored in tests/golden/voc_eval_golden.npz.

The synthetic set avoids the one case where the reference's TF streaming matcher (utility/eval_helper.py:700-781,
which xdet.evaluation restates) and voc_eval.py differ by design: a detection whose best-IoU ground truth is
`difficult` but below the matching threshold is skipped by the former and a false positive for the latter."""
import ast
import os
import pickle
import shutil
import sys
import tempfile
import xml.etree.ElementTree as ET

import numpy as np

REF = '/root/reference/voc_eval.py'
HERE = os.path.dirname(os.path.abspath(__file__))
CLASSES = ['boat', 'person', 'dog']


def reference_functions():
    src = open(REF).read()
    tree = ast.parse(src)
    ns = {'np': np, 'os': os, 'pickle': pickle, 'ET': ET, 'print': lambda *a, **k: None}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('parse_rec', 'voc_ap', 'voc_eval'):
            exec(compile(ast.Module([node], []), REF, 'exec'), ns)
    return ns


def synth(rng):
    """per image: gt [(class, difficult, xmin, ymin, xmax, ymax)]; per class: detections [(image, conf, box)]"""
    images = ['%06d' % (i + 1) for i in range(14)]
    gts, dets = {}, {c: [] for c in CLASSES}
    conf = iter(rng.permutation(np.linspace(0.05, 0.99, 400)))          # distinct confidences
    for im in images:
        objs = []
        n = int(rng.integers(1, 6))
        for k in range(n):
            x0, y0 = int(rng.integers(0, 300)), int(rng.integers(0, 300))
            w, h = int(rng.integers(40, 160)), int(rng.integers(40, 160))
            cls = CLASSES[int(rng.integers(0, 3))]
            difficult = 0 if k == 0 else int(rng.random() < 0.25)        # the first object is never difficult
            objs.append((cls, difficult, x0, y0, x0 + w, y0 + h))
        gts[im] = objs
        for (cls, difficult, x0, y0, x1, y1) in objs:
            r = rng.random()
            if r < 0.75:                                                 # a good detection (TP, or ignored if difficult)
                j = rng.integers(-6, 7, 4)
                dets[cls].append((im, float(next(conf)), [x0 + j[0], y0 + j[1], x1 + j[2], y1 + j[3]]))
                if rng.random() < 0.3 and not difficult:                 # and a duplicate of it (FP)
                    j = rng.integers(-8, 9, 4)
                    dets[cls].append((im, float(next(conf)), [x0 + j[0], y0 + j[1], x1 + j[2], y1 + j[3]]))
            elif r < 0.9 and not difficult:                              # a poorly localised one (IoU < 0.5 -> FP)
                dets[cls].append((im, float(next(conf)), [x0 + (x1 - x0) * 0.6, y0, x1 + (x1 - x0) * 0.6, y1]))
        if rng.random() < 0.5:                                           # a detection far from every object
            cls = CLASSES[int(rng.integers(0, 3))]
            dets[cls].append((im, float(next(conf)), [460., 460., 499., 499.]))
    return images, gts, dets


def write_voc(root, images, gts, dets):
    os.makedirs(os.path.join(root, 'Annotations'))
    os.makedirs(os.path.join(root, 'ImageSets', 'Main'))
    os.makedirs(os.path.join(root, 'pred'))
    open(os.path.join(root, 'ImageSets', 'Main', 'test.txt'), 'w').write('\n'.join(images) + '\n')
    for im in images:
        ann = ET.Element('annotation')
        for (cls, difficult, x0, y0, x1, y1) in gts[im]:
            o = ET.SubElement(ann, 'object')
            ET.SubElement(o, 'name').text = cls
            ET.SubElement(o, 'pose').text = 'Unspecified'
            ET.SubElement(o, 'truncated').text = '0'
            ET.SubElement(o, 'difficult').text = str(difficult)
            b = ET.SubElement(o, 'bndbox')
            for tag, v in zip(('xmin', 'ymin', 'xmax', 'ymax'), (x0 + 1, y0 + 1, x1 + 1, y1 + 1)):   # parse_rec subtracts 1
                ET.SubElement(b, tag).text = str(v)
        ET.ElementTree(ann).write(os.path.join(root, 'Annotations', im + '.xml'))
    for i, cls in enumerate(CLASSES):
        with open(os.path.join(root, 'pred', 'results_%d.txt' % (i + 1)), 'w') as f:
            for (im, c, b) in dets[cls]:
                f.write('%s %.6f %.3f %.3f %.3f %.3f\n' % (im, c, b[0], b[1], b[2], b[3]))


def main():
    ns = reference_functions()
    rng = np.random.default_rng(20260928)
    images, gts, dets = synth(rng)
    root = tempfile.mkdtemp(prefix='voc_golden_')
    try:
        write_voc(root, images, gts, dets)
        out = {}
        for i, cls in enumerate(CLASSES):
            for tag, use07 in (('07', True), ('12', False)):
                cache = os.path.join(root, 'cache_%s_%s' % (cls, tag))
                rec, prec, ap = ns['voc_eval'](os.path.join(root, 'pred', 'results_%d.txt' % (i + 1)),
                                               os.path.join(root, 'Annotations', '{}.xml'),
                                               os.path.join(root, 'ImageSets', 'Main', 'test.txt'), cls, cache,
                                               ovthresh=0.5, use_07_metric=use07)
                out['%s_rec' % cls], out['%s_prec' % cls] = np.asarray(rec), np.asarray(prec)
                out['%s_ap%s' % (cls, tag)] = np.float64(ap)
        # the same inputs as arrays (what the detections file held after its %.3f / %.6f formatting)
        gt_rows = [(images.index(im), CLASSES.index(c) + 1, d, x0, y0, x1, y1) for im in images for (c, d, x0, y0, x1, y1) in gts[im]]
        out['gt'] = np.asarray(gt_rows, np.float64)                       # image, label, difficult, xmin, ymin, xmax, ymax
        for i, cls in enumerate(CLASSES):
            rows = [(images.index(im), float('%.6f' % c)) + tuple(float('%.3f' % v) for v in b) for (im, c, b) in dets[cls]]
            out['%s_det' % cls] = np.asarray(rows, np.float64)           # image, confidence, xmin, ymin, xmax, ymax
        out['classes'] = np.asarray(CLASSES)
        np.savez(os.path.join(HERE, 'voc_eval_golden.npz'), **out)
        for cls in CLASSES:
            print(cls, 'dets', len(dets[cls]), 'ap07 %.6f ap12 %.6f' % (out[cls + '_ap07'], out[cls + '_ap12']))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
