#!/usr/bin/env python
"""How many host cores does the GPU box really give this container?  Prints the cgroup CPU quota / affinity and the C++ CPU
baseline's rate at several thread counts (image-parallel: one image per thread per call).   python tools/cpu_sweep.py"""
import os
import sys
import time

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, 'x-detector_amd'))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us',
          '/sys/fs/cgroup/cpuset.cpus.effective', '/sys/fs/cgroup/memory.max'):
    try:
        print(f, '=', open(f).read().strip())
    except OSError as e:
        print(f, ':', e.strerror)
print('os.cpu_count()', os.cpu_count(), ' sched_getaffinity', len(os.sched_getaffinity(0)))
try:
    print('loadavg', open('/proc/loadavg').read().strip())
except OSError:
    pass
from oracle import lighthead_oracle as O          # noqa: E402
from xdet import weights as W                     # noqa: E402
w = W.make_lighthead_weights(1234)
f = O.CppForward(w, 480, 300)
base = W.synthetic_images(8, 480, seed=20)
import numpy as np                                # noqa: E402
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    imgs = np.concatenate([base] * (-(-nt // 8)))[:nt]
    f.set_threads(nt)
    f(imgs)
    t = time.time()
    f(imgs)
    dt = time.time() - t
    print('threads %4d, %4d images per call: %7.2f s  %7.2f images/s  %6.3f images/s per thread' % (nt, nt, dt, nt / dt, 1.0 / dt))
