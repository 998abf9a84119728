"""build the RCCL test double (tests/fake_rccl/fake_rccl.cpp) next to its source; used by tests/test_gpu_two_ranks.py"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libfake_rccl.so')


def build(force=False):
    src = os.path.join(HERE, 'fake_rccl.cpp')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip',
                               '--offload-arch=gfx950', src, '-o', LIB, '-lrt', '-lpthread'])
    return LIB


if __name__ == '__main__':
    print(build(force=True))
