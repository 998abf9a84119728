"""Build libxdet_hip.so for gfx950 with hipcc (in-tree, so the .so travels with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
LIB = os.path.join(HERE, 'libxdet_hip.so')

# (source, extra flags).  Box/ROI arithmetic must round exactly like the reference's separately
# rounded f32 operations, so those translation units are compiled without FMA contraction.
SOURCES = [
    ('conv_mfma.hip', []),
    ('conv_mfma_split.hip', []),
    ('conv_mfma_dma.hip', []),
    ('conv_mfma_ksplit.hip', []),
    ('elementwise.hip', []),
    ('spectral.hip', []),
    ('sepconv_fused.hip', []),
    ('conv3x3_patch.hip', []),
    ('resnet_bneck.hip', []),
    ('resnet_stem.hip', []),
    ('resnet_preconv.hip', []),
    ('psroialign.hip', ['-ffp-contract=off']),
    ('proposals.hip', ['-ffp-contract=off']),
    ('detect.hip', ['-ffp-contract=off']),
    ('preprocess.hip', ['-ffp-contract=off']),
    ('net.hip', []),
    ('comm.hip', []),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-result']


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    dirty = force or not os.path.exists(LIB)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'xdet.h'))
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [hipcc] + COMMON + extra + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            dirty = True
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    # (an object compiled by hand -- a resource-usage or -S run with -o <the object> -- is newer than the library too)
    if not dirty and max(os.path.getmtime(o) for o in objs) > os.path.getmtime(LIB):
        dirty = True
    if dirty:
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
