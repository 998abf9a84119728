"""Decode the reference's demo input (demo/test.jpg, the image light_head_simple_demo.py:198 reads;
BASELINE config 1) once, in the build container, into a uint8 array fixture so the GPU box needs
neither the reference tree nor a JPEG decoder.  A fixture is data: pixels only.

    python tests/golden/make_demo_fixture.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    img = np.asarray(Image.open('/root/reference/demo/test.jpg').convert('RGB'), np.uint8)
    assert img.shape == (333, 500, 3), img.shape
    np.savez_compressed(os.path.join(HERE, 'demo_test_u8.npz'), image=img)
    print('wrote demo_test_u8.npz', img.shape, os.path.getsize(os.path.join(HERE, 'demo_test_u8.npz')))
