#!/usr/bin/env python
"""bench.py -- images/sec of the Light-Head R-CNN forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload lighthead|resnet50]

One "step" = one pass of the whole hot path (backbone + RPN + proposals + PsRoiAlign + light
head + per-class NMS) over one batch of B synthetic 480x480 images per GPU, inputs already
resident in HBM.  The batch runs as --ways concurrent sub-batches (default 2 x 128), each a net
instance replaying its hipGraph on its own stream: while one stream is in an HBM-bound launch
(depthwise, pools, DFT passes) the other's GEMMs have the matrix pipes (measured +3 % over one
stream of 256; 4 x 64 is slower again: profiles/NOTES_r04.md).

Multi-GPU (--gpus N > 1): one process per GPU.  Started plainly, this script spawns the N ranks
itself (xdet.launch); started by `python -m torch.distributed.run --nproc-per-node N ...` it IS one
rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment) -- either way no PyTorch is imported:
images are sharded by rank (independent units, weak scaling) and the only exchange is one
ncclAllGather of the fixed-size padded detections per step, issued through the C-ABI
(xdet_comm_*) on the communicator's own stream so that it overlaps the next step's forward.
Rank 0 prints ONE JSON line.

Timing: barrier + device sync on both sides of exactly K steps, MAX over ranks; the median of the
per-step device times (event per step on stream 0) is reported next to the mean.
roofline: the conv/dense MFMA kernels, bracketed by HIP event pairs on their launch stream in an
eager repeat of K steps right after the graph-replayed
timed region (events cannot be recorded inside a replayed graph).  achieved / frac = the DOMINANT
kernel alone (the 256 x 256 pointwise GEMM): its executed FLOPs / products per term over its own
launch time; backbone_frac = the backbone's dense FLOPs over all of the backbone's kernel time
(north_star's quantity); frac_algorithmic_credit = SURVEY 8d's count over every conv kernel, which
credits the DFT-domain convs with their direct-form FLOPs.  That leg runs ONE sub-batch stream alone ("one_stream"): with two
concurrent streams a launch's duration would include the other stream's kernels.  frac_whole_step
is the un-instrumented view: all algorithmic FLOPs of a step over the timed wall clock.
cpu_baseline: the CPU restatement of the reference graph under oracle/ (a port, not the TF1
runtime, which cannot run here) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'x-detector_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA (no sparsity)
PEAK_HBM_GBS = 8000.0
# SURVEY.md 8d "Bytes (minimum, each tensor once)": activations sum(in+out) per layer (136.4 + 1.7 + 5.9) M elements x 4 B,
# PSROIAlign 1.76 + 1.2 MB, detections 80 KB; weights (44.7 M x 4 B) are amortised over the batch
HBM_MIN_BYTES_PER_IMAGE = (136.4e6 + 1.7e6 + 5.9e6) * 4 + 1.76e6 + 1.2e6 + 80e3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: ~7 s of timed GPU work at the default batch (a 1.4 s region is over before a 5 s utilisation sampler
    # looks once), still well within a minute with the instrumented repeat and the CPU leg
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--ways', type=int, default=2,
                    help='the batch runs as this many concurrent sub-batches (net instances on their own HIP '
                         'streams): the partial last round of one launch is filled by the other stream')
    ap.add_argument('--workload', default='lighthead', choices=['lighthead', 'resnet50'])
    ap.add_argument('--resnet-ways', type=int, default=2, help='resnet50 workload: concurrent sub-batches (see --ways)')
    ap.add_argument('--image-size', type=int, default=480,
                    help='network input size: 480 (the metric) or 800 (BASELINE config 5 shape; implies --no-parity '
                         '--no-cpu-baseline, whose legs are 480x480)')
    ap.add_argument('--proposals', type=int, default=300, help='rpn_post_nms_top_n (BASELINE config 3: 300)')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3', 'f16'],
                    help='conv/dense arithmetic: exact f32 MFMA, split-precision f16 MFMA (~f32 accuracy), plain f16')
    ap.add_argument('--voc-stream', action='store_true',
                    help='BASELINE config 4 input: raw uint8 VOC-shape images resident in HBM (shapes cycling '
                         '375x500, 500x375, 333x500, 500x333); every step runs the F1 pre-processing kernel per '
                         'image (whiten + TF-legacy bilinear warp to 480x480) in front of the forward')
    ap.add_argument('--eager', action='store_true',
                    help='launch kernel by kernel in the timed region (default: replay the captured hipGraph)')
    ap.add_argument('--comm', action='store_true',
                    help='create the RCCL communicator and all-gather the detections every step even at one rank '
                         '(always on for N > 1 and under torch.distributed.run)')
    ap.add_argument('--dry-run', action='store_true',
                    help='multi-GPU pre-flight: rendezvous, ONE all-gather of the per-rank device records over the '
                         'collective library, print them (ranks_seen, distinct_gpus, which library) and exit -- seconds')
    ap.add_argument('--serial-rpn', action='store_true',
                    help='keep the RPN / proposal branch on the main stream (default: side stream under the '
                         'large-separable convs): per-kernel rocprofv3 durations without cross-stream sharing')
    ap.add_argument('--conv3x3', default='patch', choices=['patch', 'gemm'],
                    help='block1_conv2: LDS-staged input tile (default) or the implicit-GEMM kernel (A/B measurements)')
    ap.add_argument('--pool', default='split', choices=['split', 'whole', 'split_all'],
                    help='entry-flow pools: horizontal half in the producing block (default, 237x237 block) or one kernel')
    ap.add_argument('--sustain-seconds', type=float, default=10.0,
                    help='after the K timed steps: an UN-timed-for-value leg of back-to-back steps of about this many '
                         'seconds (reported as sustained_images_per_sec), so that a utilisation sampler with a period of '
                         'seconds sees the GPU busy whatever --steps is; 0 = off')
    ap.add_argument('--ksplit', default='on', choices=['on', 'off', 'all'],
                    help='fixed split-K: narrow head GEMM (on, default), nothing (off), RPN conv as well (all) -- A/B runs')
    ap.add_argument('--cross', default='f16', choices=['f16', 'fp8'],
                    help="cross terms of the split-precision products of the depthwise -> pointwise layers: f16 (f16x3 everywhere) "
                         "or fp8 copies of the operands (the x8 form, after the calibration pass)")
    ap.add_argument('--large-sep', default='auto', choices=['auto', 'direct', 'spectral'],
                    help='large-separable convs: DFT-domain GEMMs (auto, where the feature map allows) or the direct (15,1)/(1,15) convs')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=2000, help='images of the bounded CPU-baseline sample (~10-20 s)')
    ap.add_argument('--cpu-batch', type=int, default=0,
                    help='images per call of the C++ CPU baseline (0 = two per CPU the container may use -- affinity cut by '
                         'the cgroup quota --, at most 256: the baseline runs image-parallel, oracle/lighthead_cpu.cpp)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='time budget of the CPU-baseline sample')
    ap.add_argument('--pool-sub', choices=['on', 'off'], default='on',
                    help='blocks 2-3: the pool pass also writes the next projection\'s subsampled planes and relu(sum) '
                         '(default) or not (A/B measurements)')
    ap.add_argument('--stagger', type=float, default=0.0,
                    help='EXPERIMENT (profiles/NOTES_r05.md): phase-shift the second sub-batch stream by this fraction of a '
                         'step (it first runs a forward over that fraction of its images, inside the timed region), so that '
                         'its bandwidth-bound entry flow meets the first stream\'s GEMM-bound middle flow; compare '
                         'median_ms_per_step')
    ap.add_argument('--ops', action='store_true', help='also print the per-op table to stderr')
    ap.add_argument('--no-parity', action='store_true', help='skip the live f16x3-vs-f32 GPU cross-check')
    ap.add_argument('--no-roofline', action='store_true', help='skip the instrumented eager repeat')
    return ap.parse_args()


def _code_only(src):
    """C/C++ source text without comments and without whitespace: what a profile depends on (a comment edit or a
    re-indent must not invalidate the committed counters)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"' or c == "'":                       # string / char literal: copied verbatim
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == '\\' else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith('//', i):
            j = src.find('\n', i)
            i = n if j < 0 else j
        elif src.startswith('/*', i):
            j = src.find('*/', i + 2)
            i = n if j < 0 else j + 2
        elif c.isspace():
            i += 1
        else:
            out.append(c)
            i += 1
    return ''.join(out)


def kernel_source_hashes():
    """{file: hash of its code} for every kernel source / header (the GPU box has no .git)."""
    d = os.path.join(ROOT, 'x-detector_amd', 'csrc')
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            out[f] = hashlib.sha256(_code_only(open(os.path.join(d, f), errors='replace').read()).encode()).hexdigest()[:16]
    return out


def kernel_source_hash():
    """one hash over all of them: identifies the kernel sources a profile was taken with."""
    h = hashlib.sha256()
    for f, v in sorted(kernel_source_hashes().items()):
        h.update(f.encode())
        h.update(v.encode())
    return h.hexdigest()[:16]


# the sources the dominant conv kernel is compiled from: the per-launch counters of a committed profile are quoted
# whenever THESE are unchanged (net.hip decides which layers are launched; it is listed with the per-image total)
CONV_KERNEL_FILES = ('conv_mfma_dma.hip', 'conv_epilogue.h', 'conv_params.h')


def read_profile(handle, kind):
    from xdet._lib import lib, check
    maxo = 512
    n = ctypes.c_int()
    ms = (ctypes.c_double * maxo)()
    cnt = (ctypes.c_int * maxo)()
    fl = (ctypes.c_double * maxo)()
    check(lib().xdet_profile_read(handle, kind, maxo, ctypes.byref(n), ms, cnt, fl))
    issued = (ctypes.c_double * maxo)()
    n2 = ctypes.c_int()
    check(lib().xdet_profile_mfma_flops(handle, kind, maxo, ctypes.byref(n2), issued))
    rows = []
    buf = ctypes.create_string_buffer(256)
    for i in range(n.value):
        check(lib().xdet_profile_op_name(handle, kind, i, buf, 256))
        rows.append((buf.value.decode(), ms[i], cnt[i], fl[i], issued[i]))
    return rows


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def net_memory(nets, sb):
    try:
        m = nets[0].memory()
    except Exception:                                # an older build behind XDET_LIB has no xdet_net_memory
        return None
    return {'net_instances': len(nets), 'images_per_instance': sb,
            'allocated_gb_per_instance': round(m['allocated_bytes'] / 1e9, 3),
            'recycled_gb_per_instance': round(m['recycled_bytes'] / 1e9, 3)}


def host_memory_images(bytes_per_image):
    """how many images' working sets fit into half of what the host (and the container's cgroup) has available"""
    avail = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for f in ('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory/memory.limit_in_bytes'):
        try:
            v = open(f).read().strip()
            if v.isdigit():
                avail = int(v) if avail is None else min(avail, int(v))
        except OSError:
            pass
    return 1 << 20 if avail is None else int(avail * 0.5 / bytes_per_image)


def cpu_baseline(args, weights):
    """The oracle timed on this box's host cores (kind "port")."""
    from oracle import lighthead_oracle as O
    from xdet import weights as W
    n = args.cpu_images
    if args.workload == 'lighthead':
        fwd = getattr(O, 'lighthead_forward_fast', None) or O.lighthead_forward
        how = getattr(O, 'FAST_PATH_DESCRIPTION', 'NumPy fp32 + OpenBLAS')
        # images per call: two per CPU the container can really use (its cgroup quota, not the host's thread count)
        cb = args.cpu_batch if args.cpu_batch > 0 else min(256, 2 * O.effective_cpus())
        cb = max(1, min(cb, host_memory_images(0.4e9)))            # every image in flight holds ~0.35 GB of f32 tensors
        # inputs are synthesised BEFORE the clock starts (four different batches, cycled): the window times the forward only
        base = W.synthetic_images(min(cb, 16), 480, seed=20)           # 16 distinct images, repeated to fill a call
        full = np.concatenate([base] * (-(-cb // len(base))))[:cb]
        batches = [np.ascontiguousarray(np.roll(full, i, axis=0)) for i in range(2)]
        fwd(batches[0], weights, rpn_post_nms_top_n=args.proposals)   # warm-up (page-in, weight packing, thread sweep)
        t = time.perf_counter()
        done = calls = 0
        while done < n and (done < 3 or time.perf_counter() - t < args.cpu_seconds):   # bounded sample: ~cpu_seconds of CPU work
            fwd(batches[calls % len(batches)], weights, rpn_post_nms_top_n=args.proposals)
            done += cb
            calls += 1
        dt = time.perf_counter() - t
        n = done
        what = ('%d 480x480 images (%d calls of %d, inputs prepared outside the timed window) through the full forward '
                '(R=%d), %s' % (n, calls, cb, args.proposals, how))
    else:
        xs = [np.ascontiguousarray(np.transpose(W.synthetic_images(1, 480, seed=20 + i), (0, 2, 3, 1))) for i in range(4)]
        O.resnet50_trunk(xs[0], weights)
        t = time.perf_counter()
        done = 0
        while done < n and (done < 3 or time.perf_counter() - t < args.cpu_seconds):
            O.resnet50_trunk(xs[done % len(xs)], weights)
            done += 1
        dt = time.perf_counter() - t
        n = done
        what = '%d x one 480x480 image through the ResNet-50 v2 trunk, NumPy fp32 + OpenBLAS' % n
    cores = None
    if args.workload == 'lighthead' and getattr(O, '_fast_cache', None):
        cores = next(iter(O._fast_cache.values())).threads            # OpenMP threads of the C++ restatement
    if cores is None:
        try:
            from threadpoolctl import threadpool_info
            cores = max([d.get('num_threads', 1) for d in threadpool_info()] or [os.cpu_count()])
        except Exception:
            cores = os.cpu_count()
    out = {'value': round(n / dt, 3), 'unit': 'images/sec', 'cores': int(cores), 'kind': 'port', 'sample': what,
           'host': {'nproc': os.cpu_count(), 'cpu': cpu_model(),
                    'usable_cpus': O.effective_cpus() if hasattr(O, 'effective_cpus') else None}}
    if args.workload == 'lighthead' and getattr(O, '_fast_cache', None):
        tun = getattr(next(iter(O._fast_cache.values())), 'tuning', None)
        if tun:
            # the warm-up's thread-count sweep: seconds per CALL (of cpu_batch images) at each OpenMP thread count
            out['thread_sweep_s_per_call'] = {str(k): round(v, 3) for k, v in tun.items()}
            out['thread_sweep_images_per_call'] = cb
    return out


def counters_from_profiles(precision, sub_batch):
    """HBM bytes per conv launch and counter-based MFMA utilisation from the committed rocprofv3 --pmc
    passes (profiles/<tag>_summary.json, tools/summarize_profile.py; PMC collection cannot run inside the
    bench itself).  A summary is quoted if it was taken with launches of the same size (one sub-batch stream of
    `sub_batch` images: the command the summary records) and with the same CODE (comments / whitespace ignored) of the
    files the quoted kernel is compiled from (CONV_KERNEL_FILES).  The whole-forward figure (hbm_bytes_per_image) sums
    over every kernel: the files that changed since the profile are listed next to it (empty list = the profile is of
    exactly these sources)."""
    if precision != 'f16x3':      # the committed PMC passes are of the default configuration only
        return None
    want = 'conv_dma_f16'
    now = kernel_source_hashes()
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_summary.json')), reverse=True):     # newest tag first (r04b > r04a > r03b)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        cmd = (d.get('command') or '') + ' '
        if '--ways 1 --batch %d ' % sub_batch not in cmd or any(k in cmd for k in ('--proposals', '--image-size', '--workload')):
            continue                                  # (only the default configuration's summary: 300 proposals, 480 x 480)
        then = d.get('source_hashes')
        if not then or any(then.get(f) != now.get(f) for f in CONV_KERNEL_FILES):
            continue
        changed = sorted(f for f in set(then) | set(now) if then.get(f) != now.get(f))
        tot = cnt = 0
        busy = act = 0.0
        for k, e in d.get('kernels', {}).items():
            if want in k and 'hbm_bytes_per_launch' in e:
                tot += e['hbm_bytes_per_launch'] * e['calls']
                cnt += e['calls']
            if want in k and 'mfma_busy_cycles' in e and 'gui_active_cycles' in e:
                busy += e['mfma_busy_cycles']
                act += e['gui_active_cycles']
        if cnt:
            return {'traffic': int(tot / cnt), 'mfma_busy_frac': round(busy / (act / 8.0 * 1024.0), 4) if act else None,
                    'hbm_bytes_per_image': d.get('hbm_bytes_per_image'), 'file': os.path.relpath(path, ROOT),
                    'files_changed_since': changed}
    return None


def live_parity(weights, proposals):
    """Cross-check of the split-precision path against the exact-f32 MFMA path of the same library on
    2 seeded images (both are product code; the CPU oracle comparison lives in tests/)."""
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import set_precision, get_precision
    imgs = W.synthetic_images(2, 480, seed=7)
    cur = get_precision()
    res = {}
    for mode in (cur, 'f32'):
        set_precision(mode)
        det = LightHeadDetector(weights, image_size=480, max_batch=2, rpn_post_nms_top_n=proposals)
        det.forward(imgs)
        res[mode] = (det.buffer('feat', 2).numpy(), det.detections(2))
        del det
    set_precision(cur)
    fa, (sa, ba) = res[cur]
    fb, (sb, bb) = res['f32']
    nd = int((sb > 0).sum())
    same = int(((np.abs(sa - sb) < 1e-3) & (np.abs(ba - bb).max(-1) < 1e-3) & (sb > 0)).sum())
    return {'vs': 'f32 MFMA path, same inputs', 'feat_max_abs_err': float(np.abs(fa - fb).max()),
            'feat_absmax': float(np.abs(fb).max()), 'detections': nd, 'same_slot_within_1e-3': same}


def main():
    args = parse()
    if 'RANK' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU)
        from xdet.launch import launch_ranks
        sys.exit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks with --gpus N)' % (args.gpus, world))
    numa_cpus = None
    if 'RANK' in os.environ:
        # a rank of a multi-process job (xdet.launch or torch.distributed.run): RCCL's warnings go to stderr (stdout
        # carries the JSON line), and the process is bound to the CPUs of its GPU's NUMA node
        os.environ.setdefault('NCCL_DEBUG', 'WARN')
        os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')
        if os.environ.get('XDET_BIND_NUMA', '1') != '0':
            from xdet.launch import bind_to_gpu_numa
            numa_cpus = bind_to_gpu_numa(local_rank)   # (before the device count is known: a rank without a GPU binds nothing)

    from xdet import weights as W
    from xdet import dist as xdist
    from xdet._lib import lib, check
    from xdet.runtime import Event, DeviceBuffer, set_precision
    ndev = ctypes.c_int()
    check(lib().xdet_device_count(ctypes.byref(ndev)))
    device = local_rank
    if local_rank >= ndev.value:
        # XDET_OVERSUBSCRIBE_GPUS=1 (tests/test_gpu_two_ranks.py: two ranks on the one GPU of the box, over the RCCL test
        # double -- real RCCL refuses two ranks per device): ranks wrap around the visible devices
        if os.environ.get('XDET_OVERSUBSCRIBE_GPUS') != '1':
            raise SystemExit('bench.py: rank %d wants GPU %d but only %d visible' % (rank, local_rank, ndev.value))
        device = local_rank % ndev.value
    check(lib().xdet_set_device(device))
    set_precision(args.precision)
    # under a launcher (RANK set) the collective path is exercised even for one rank
    use_comm = world > 1 or args.comm or 'RANK' in os.environ
    comm = xdist.Communicator(rank, world) if use_comm else None

    if args.dry_run:
        # pre-flight of a multi-GPU node: every rank's device record THROUGH the collective, nothing else
        if comm is None:
            comm = xdist.Communicator(rank, world)
        t0 = time.perf_counter()
        recs = comm.device_records({'numa_cpus': len(numa_cpus) if numa_cpus else None})
        comm.barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            path, over = comm.library()
            print(json.dumps({'dry_run': True, 'n_gpus': world,
                              'comm': {'library': path, 'library_overridden': over,
                                       'rccl_version': comm.info()['rccl_version'], 'world': world, 'devices': recs,
                                       'ranks_seen': sorted(r['rank'] for r in recs),
                                       'distinct_gpus': len({(r['host'], r['pci_bus_id']) for r in recs}),
                                       'records_plus_barrier_s': round(dt, 3)}}))
            sys.stdout.flush()
        comm.close()
        return

    B, K, Wm = args.batch, args.steps, args.warmup
    S = args.image_size
    if S != 480:
        args.no_parity = args.no_cpu_baseline = True
    gathered = None
    if args.workload == 'lighthead':
        from xdet.model import LightHeadDetector
        weights = W.make_lighthead_weights(1234)
        ways = max(1, min(args.ways, B))
        if B % ways:
            raise SystemExit('--batch must be a multiple of --ways')
        sb = B // ways                               # images per sub-batch / net instance
        nets = [LightHeadDetector(weights, image_size=S, max_batch=sb, rpn_post_nms_top_n=args.proposals,
                                  rpn_stream='main' if args.serial_rpn else 'side', conv3x3=args.conv3x3,
                                  pool=args.pool, ksplit=args.ksplit, cross=args.cross, large_sep=args.large_sep,
                                  pool_sub=None if args.pool_sub == 'on' else args.pool_sub)   # (None: the library's default;
                                  # an older build behind XDET_LIB, tools/ab_bench.sh, does not know the option)
                for _ in range(ways)]
        net = nets[0]
        kind = 0
        fl = net.flops_per_image()
        flops_img = sum(fl.values())
        imgs = W.synthetic_images(B, S, seed=100 + rank)
        for i, nt in enumerate(nets):
            if args.cross == 'fp8':
                # the x8 form is switched on by the calibration pass (it measures the tensors whose fp8 copies it scales):
                # a seeded batch of the same distribution, not the timed inputs
                nt.calibrate(W.synthetic_images(min(sb, 8), S, seed=4242))
            nt.set_images(imgs[i * sb:(i + 1) * sb])
        raw = None
        if args.voc_stream:
            # raw uint8 images stay resident; F1 writes the whitened 480x480 planes into the net's input buffer
            from xdet.runtime import to_device
            shapes = [(375, 500), (500, 375), (333, 500), (500, 333)]
            rng = np.random.default_rng(7 + rank)
            raw = [(to_device(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)), h, w) for h, w in shapes]
        nc, topk = net.num_classes - 1, net.nms_topk
        det = None
        if comm is not None:
            # the rank's detections live in ONE pair of buffers per parity (each sub-batch net writes its slice),
            # which the communicator packs and all-gathers in place; two pairs used alternately, so a forward never
            # has to wait for the pack of the step before it
            det = [(DeviceBuffer(B * nc * topk * 4, zero=True), DeviceBuffer(B * nc * topk * 16, zero=True))
                   for _ in range(2)]
        turn = [0]

        use_graph = not args.eager

        def step(graph=None, only_first=False):
            g = use_graph if graph is None else graph
            if only_first:                           # the roofline leg: ONE sub-batch stream alone on the chip
                nets[0].forward_device(sb, use_graph=g)
                return
            if raw is not None:
                for nt in nets:
                    for j in range(sb):
                        buf, h, w = raw[j % len(raw)]
                        check(lib().xdet_preprocess_eval(buf.ptr, h, w, nt._images.ptr + j * 3 * S * S * 4, S,
                                                         nt.stream.handle))
            if comm is None:
                for nt in nets:
                    nt.forward_device(sb, use_graph=g)
            else:
                ds, db = det[turn[0]]
                turn[0] ^= 1
                for i, nt in enumerate(nets):
                    nt.forward_device(sb, use_graph=g, det_scores_ptr=ds.ptr + i * sb * nc * topk * 4,
                                      det_boxes_ptr=db.ptr + i * sb * nc * topk * 16)
                # asynchronous: waits for the nets' streams on the device, overlaps the next step
                comm.allgather_detections(ds.ptr, db.ptr, B, nc, topk, streams=[nt.stream for nt in nets],
                                          double_buffered=True)
    else:
        from xdet.resnet import ResNet50Trunk
        weights = W.make_resnet50_weights(4321)
        # --resnet-ways N: the batch as N concurrent sub-batches (trunk instances on their own streams), as the detector's
        # --ways: at batch 8 most launches are 60-230 workgroups of 30-50 us, and two independent streams fill each
        # other's tails and launch gaps (batch 8: 3697 -> 3931 images/s, same box).  Default 2; 1 = the configuration of
        # every ResNet line before round 4.
        ways = max(1, min(args.resnet_ways, B))
        if B % ways:
            raise SystemExit('--batch must be a multiple of --resnet-ways')
        sb = B // ways
        nets = [ResNet50Trunk(weights, image_size=480, max_batch=sb) for _ in range(ways)]
        net = nets[0]
        kind = 1
        flops_img = net.flops_per_image()
        fl = {'backbone': flops_img}
        imgs = W.synthetic_images(B, 480, seed=100 + rank)
        for i, nt in enumerate(nets):
            nt.set_images(imgs[i * sb:(i + 1) * sb])

        use_graph = not args.eager

        def step(graph=None, only_first=False):
            g = use_graph if graph is None else graph
            for nt in (nets[:1] if only_first else nets):
                nt.forward_device(sb, use_graph=g)

    def sync_all():
        for nt in nets:
            nt.stream.synchronize()
        if comm is not None:
            comm.wait()
            comm.barrier()

    for _ in range(Wm):
        step()
    sync_all()
    # per-op HIP events cannot be recorded into a replayed graph: with graph replay the roofline leg is an
    # eager, instrumented repeat of K steps of ONE sub-batch stream after the timed region
    profile = not use_graph and not args.no_roofline and ways == 1
    if profile:
        check(lib().xdet_profile_enable(net.handle, kind, 1))
    evs = [Event() for _ in range(K + 1)]
    sync_all()
    t0 = time.perf_counter()
    evs[0].record(net.stream)
    for k in range(K):
        if k == 0 and args.stagger > 0 and args.workload == 'lighthead' and len(nets) > 1:
            nets[1].forward_device(max(1, int(sb * args.stagger)), use_graph=False)
        step()
        evs[k + 1].record(net.stream)
    sync_all()
    dt = time.perf_counter() - t0
    step_ms = [evs[k].elapsed_ms(evs[k + 1]) for k in range(K)]
    dev_ms = evs[0].elapsed_ms(evs[K])
    if comm is not None and args.workload == 'lighthead':
        gathered = comm.gathered()                   # [world*B, C, K, 5]: proves every rank's shard arrived
    dt_local = dt
    if comm is not None:
        dt = comm.max_over_ranks(dt)                 # the slowest rank's wall clock: what `value` is computed from
    # sustained leg (not part of `value`): the same step back to back for ~sustain_seconds.  The step count is derived from
    # the max-over-ranks time, i.e. identical on every rank (each step carries a collective).
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(1, int(np.ceil(args.sustain_seconds / (dt / K))))
        sync_all()
        t1 = time.perf_counter()
        for k in range(n_sus):
            step()
            if k % 16 == 15:                         # bound the launch queue; a stream sync, not a collective
                for nt in nets:
                    nt.stream.synchronize()
        sync_all()
        ds = time.perf_counter() - t1
        if comm is not None:
            ds = comm.max_over_ranks(ds)
        sustained = {'sustained_images_per_sec': round(world * B * n_sus / ds, 2), 'sustained_steps': n_sus,
                     'sustained_seconds': round(ds, 2)}
    rows = []
    KI = K                                           # steps the per-op event pairs cover
    if not args.no_roofline:
        if not profile:
            step(graph=False, only_first=True)       # eager warm-up of the instrumented leg
            sync_all()
            check(lib().xdet_profile_enable(net.handle, kind, 1))
            KI = min(K, 20)                          # the instrumented repeat: 20 steps are plenty for per-op means
            for _ in range(KI):
                step(graph=False, only_first=True)
            sync_all()
        rows = read_profile(net.handle, kind)
        check(lib().xdet_profile_enable(net.handle, kind, 0))

    dev_records = None
    if comm is not None:
        # every rank's (rank, hip device, PCI bus id, host, its own images/s) through ncclAllGather: the line then
        # proves how many DISTINCT GPUs took part, and shows the per-rank rates behind the max-over-ranks value
        dev_records = comm.device_records({'images_per_sec': round(B * K / dt_local, 1),
                                           'ms_per_step': round(dt_local / K * 1e3, 3),
                                           'numa_cpus': len(numa_cpus) if numa_cpus else None})

    if rank == 0:
        ms_per_step = dt / K * 1e3
        value = world * B * K / dt
        # contraction ops: flops > 0; their auxiliary passes (DFT around a spectral GEMM): flops < 0 -- time counts, no FLOPs
        conv_ms = sum(r[1] for r in rows if r[3] != 0)
        conv_launches = sum(r[2] for r in rows if r[3] != 0)
        conv_flops = sum(max(r[3], 0.0) * r[2] for r in rows) * sb      # flops are per image, a launch covers one sub-batch
        issued_flops = sum(r[4] * r[2] for r in rows if r[3] != 0) * sb  # what the matrix cores actually executed
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == 'f32' else PEAK_F16_MFMA_TFLOPS
        nprod = 3 if args.precision == 'f16x3' else 1
        roof = None
        if conv_ms > 0:
            ach = conv_flops / (conv_ms * 1e-3) / 1e12
            kname = 'conv_mfma_f32_kernel' if args.precision == 'f32' else 'conv_dma_f16_kernel (+conv_mfma_f16_kernel for the 5 small/strided convs)'
            src = kernel_source_hash()
            default_cfg = args.workload == 'lighthead' and args.proposals == 300
            ctr = counters_from_profiles(args.precision, sb) if default_cfg else None
            # frac counts SURVEY 8d's algorithmic FLOPs (the direct-form count also for the two large-separable convs,
            # which run in the DFT domain and execute ~5x fewer).  The kernel-quality views next to it:
            #   frac_executed    = FLOPs actually executed / products per term (i.e. the algorithmic FLOPs of the form
            #                      that ran) over the same conv time -- no spectral credit;
            #   frac_direct_only = the directly computed contractions alone: their FLOPs over their own time.
            spec_rows = [r for r in rows if r[3] > 0 and r[4] > 0 and abs(r[4] - nprod * r[3]) > 1e-6 * r[3]]
            prefixes = [r[0].split(' [')[0] for r in spec_rows]       # their DFT passes carry the same name prefix
            d_rows = [r for r in rows if r[3] > 0 and not any(r[0].startswith(q) for q in prefixes)]
            d_ms = sum(r[1] for r in d_rows)
            d_fl = sum(r[3] * r[2] for r in d_rows) * sb
            # The DOMINANT kernel alone (what `frac` is): conv_dma_f16_kernel<256, 256, ...> in its pointwise form runs the
            # direct 1x1 GEMMs of the separable convs (ops named */pointwise): ~half of the GPU time.  Their algorithmic
            # FLOPs (= executed / products per term: no algorithm credit is involved) over their own summed launch time.
            dom_rows = [r for r in rows if r[3] > 0 and r[0].endswith('/pointwise')] if args.workload == 'lighthead' else []
            if not dom_rows:                                  # (ResNet-50: every contraction launch)
                dom_rows = [r for r in rows if r[3] > 0]
            dom_ms = sum(r[1] for r in dom_rows)
            dom_launches = sum(r[2] for r in dom_rows)
            dom_fl = sum(r[4] / nprod * r[2] for r in dom_rows) * sb
            dom_ach = dom_fl / (dom_ms * 1e-3) / 1e12
            # north_star's own quantity: the BACKBONE's dense FLOPs over ALL of the backbone's time (depthwise, pool,
            # split and add passes included), one sub-batch stream alone
            if args.workload == 'lighthead':
                bb_rows = [r for r in rows if r[0].startswith('block') or r[0].startswith('conv2d_')]
            else:
                bb_rows = list(rows)
            bb_ms = sum(r[1] for r in bb_rows)
            bb_fl = sum(max(r[3], 0.0) * r[2] for r in bb_rows) * sb
            dom_name = ('conv_dma_f16_kernel<256, 256, ...> pointwise form: the %d direct 1x1 GEMMs of the separable convs'
                        % (dom_launches // KI) if args.workload == 'lighthead' and args.precision != 'f32' else kname)
            roof = {'bound': 'mfma', 'kernel': dom_name, 'achieved': round(dom_ach, 2),
                    'peak': peak, 'unit': 'TFLOP/s', 'frac': round(dom_ach / peak, 4),
                    'frac_scope': 'dominant kernel only: executed FLOPs / products per term over its own launch time',
                    'launches_per_step': dom_launches // KI,
                    'avg_launch_us': round(dom_ms * 1e3 / max(dom_launches, 1), 2),
                    'kernel_ms_per_step': round(dom_ms / KI, 3),
                    'backbone_frac': round(bb_fl / (bb_ms * 1e-3) / 1e12 / peak, 4) if bb_ms > 0 else None,
                    'backbone_tflops': round(bb_fl / (bb_ms * 1e-3) / 1e12, 2) if bb_ms > 0 else None,
                    'backbone_ms_per_step': round(bb_ms / KI, 3),
                    'backbone_scope': 'dense FLOPs of the backbone (SURVEY 8d) over the time of EVERY backbone kernel',
                    # every conv / dense kernel of the step together (the round 1-5 headline)
                    'all_conv_kernels': {'kernels': kname, 'launches_per_step': conv_launches // KI,
                                         'kernel_ms_per_step': round(conv_ms / KI, 3),
                                         'avg_launch_us': round(conv_ms * 1e3 / max(conv_launches, 1), 2),
                                         'achieved_algorithmic_credit': round(ach, 2)},
                    # SURVEY 8d's algorithmic count over all conv kernels: the two spectral launches are credited with their
                    # direct-form FLOPs (they execute ~5x fewer) -- an algorithm saving, not kernel quality
                    'frac_algorithmic_credit': round(ach / peak, 4),
                    'frac_executed': round(issued_flops / nprod / (conv_ms * 1e-3) / 1e12 / peak, 4),
                    'frac_direct_only': round(d_fl / (d_ms * 1e-3) / 1e12 / peak, 4) if d_ms > 0 else None,
                    'frac_cap': round(1.0 / nprod, 4),
                    # the shader clock the chip sustains INSIDE the dominant kernel's K loop with all CUs busy (s_memtime /
                    # s_memrealtime around the loop, profiles/r04_kloop_clock.txt: 1.55-1.69 GHz; `peak` is priced at 2.4 GHz)
                    # NOT measured by this run: the constant read off the committed stamp file, named as such
                    'assumed_shader_clock_ghz_from_profiles_r04_kloop_clock': 1.6 if default_cfg and args.precision != 'f32' else None,
                    'frac_of_peak_at_assumed_clock': (round(dom_ach / (peak * 1.6 / 2.4), 4)
                                                      if default_cfg and args.precision != 'f32' else None),
                    'frac_note': ('%d MFMA products per term cap frac at %.3f; spectral ops are credited with their direct-form '
                                  'FLOPs in frac_algorithmic_credit only' % (nprod, 1.0 / nprod)),
                    'mfma_issued_tflops': round(issued_flops / (conv_ms * 1e-3) / 1e12, 2),
                    'mfma_util': round(issued_flops / (conv_ms * 1e-3) / 1e12 / peak, 4),
                    'mfma_util_how': 'FLOPs executed on the matrix cores (%d products per term; the spectral GEMMs execute '
                                     '~5x fewer than their algorithmic count x %d) / conv kernel time / peak' % (nprod, nprod),
                    'mfma_busy_frac_pmc': ctr['mfma_busy_frac'] if ctr else None,
                    'traffic': ctr['traffic'] if ctr else None,
                    'traffic_unit': ('HBM bytes per conv launch (PMC FETCH_SIZE x2 + WRITE_SIZE), from %s: same code of %s'
                                     % (ctr['file'], ' / '.join(CONV_KERNEL_FILES))) if ctr
                    else 'no PMC pass committed for this code of %s (%s) / this configuration' % (' / '.join(CONV_KERNEL_FILES), src),
                    'hbm_bytes_per_image': ctr['hbm_bytes_per_image'] if ctr else None,
                    # sources whose code changed since that profile (the per-image total sums over every kernel)
                    'hbm_profile_files_changed_since': ctr['files_changed_since'] if ctr else None,
                    'hbm_min_bytes_per_image': int(HBM_MIN_BYTES_PER_IMAGE + 44.7e6 * 4 / sb) if args.workload == 'lighthead' and S == 480 else None,
                    'hbm_over_min': (round(ctr['hbm_bytes_per_image'] / (HBM_MIN_BYTES_PER_IMAGE + 44.7e6 * 4 / sb), 3)
                                     if ctr and ctr.get('hbm_bytes_per_image') and args.workload == 'lighthead' and S == 480 else None),
                    'gflop_per_image': round(flops_img / 1e9, 2),
                    # the un-instrumented view: every algorithmic FLOP of the step over the timed wall clock
                    'frac_whole_step': round(B * flops_img / (ms_per_step * 1e-3) / 1e12 / peak, 4),
                    'config': 'one_stream: one sub-batch of %d images alone on the chip' % sb,
                    'how': ('HIP event pair around every conv/dense launch on its launch stream, ' +
                            ('inside the timed region' if profile else
                             'in an eager repeat of K steps of ONE sub-batch stream right after the timed region (a '
                             'replayed graph cannot carry event pairs, and with %d concurrent streams a launch\'s '
                             'duration would include the other stream\'s kernels); frac_whole_step is the '
                             'un-instrumented figure of the timed region itself' % ways))}
        out = {
            'metric': 'images/sec at %dx%d Light-Head R-CNN, 1/2/4/8 MI355X + backbone MFMA util%%' % (S, S),
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'f16x3': 'f16x3 (split-precision f16 MFMA, f32 accumulate, f32 activations)',
                      'f16': 'f16'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('Full Light-Head R-CNN (Xception backbone + RPN + GPU proposals/NMS + PSROIAlign + '
                                    'light head + per-class NMS), %d proposals, %dx%d' % (args.proposals, S, S))
                       if args.workload == 'lighthead' else 'ResNet-50 v2 trunk only (BASELINE config 2), 480x480',
                       'batch_per_gpu': B, 'global_batch': B * world, 'image_size': S,
                       'concurrent_sub_batches': ways,
                       'cross_terms': args.cross if args.workload == 'lighthead' else 'f16',
                       'input': ('uint8 VOC-shape stream + F1 pre-processing kernel in the step' if args.voc_stream
                                 else 'whitened f32 [B,3,%d,%d] resident in HBM' % (S, S)),
                       'parallelism': 'image-sharded dp%d%s' % (world, ', RCCL all-gather of detections per step '
                                                                '(C-ABI, overlapped with the next forward)'
                                                                if comm is not None else ''),
                       'weights': 'seeded random init (no checkpoint exists)', 'graph_replay': bool(use_graph)},
            # device memory of ONE sub-batch net (weights + workspace) and the bytes of tensors placed into recycled blocks
            # (option "workspace" = "reuse", include/xdet.h xdet_net_memory)
            'memory': net_memory(nets, sb) if args.workload == 'lighthead' else None,
            'device_ms_per_step': round(dev_ms / K, 3),
            'median_ms_per_step': round(float(np.median(step_ms)), 3),
            'min_ms_per_step': round(float(np.min(step_ms)), 3),
            'gflop_per_image': {k: round(v / 1e9, 2) for k, v in fl.items()},
            'roofline': roof,
        }
        if sustained:
            out.update(sustained)
        if comm is not None:
            info = comm.info()
            lib_path, lib_over = comm.library()
            rates = [r.get('images_per_sec') for r in dev_records] if dev_records else None
            out['comm'] = {'transport': ('%s (version %d) via libxdet_hip.so (xdet_comm_*), no torch'
                                         % ('XDET_RCCL_LIB stand-in, NOT RCCL' if lib_over else 'RCCL', info['rccl_version'])),
                           'library': lib_path, 'library_overridden': lib_over,
                           'world': info['world'],
                           # value / (N x the median of the ranks' own rates): 1.0 = the job runs at the rate of its typical
                           # rank (the max-over-ranks time and the collectives cost nothing); the driver computes the
                           # efficiency against its own N=1 run, this is the same curve read off ONE line
                           'weak_scaling_efficiency_vs_rank_median': (round(value / (world * float(np.median(rates))), 4)
                                                                      if rates and all(rates) else None),
                           # gathered through ncclAllGather: one record per rank
                           'devices': dev_records,
                           'ranks_seen': sorted(r['rank'] for r in dev_records) if dev_records else None,
                           'distinct_gpus': len({(r['host'], r['pci_bus_id']) for r in dev_records}) if dev_records else None,
                           'per_rank_images_per_sec': [r.get('images_per_sec') for r in dev_records] if dev_records else None,
                           'gathered_shape': list(gathered.shape) if gathered is not None else None,
                           'gathered_images_with_detections': int((gathered[..., 0].reshape(gathered.shape[0], -1) > 0)
                                                                  .any(1).sum()) if gathered is not None else None}
        if args.workload == 'lighthead' and args.precision != 'f32' and not args.no_parity:
            out['parity'] = live_parity(weights, args.proposals)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, weights)
        else:
            out['cpu_baseline'] = None
        if args.ops and rows:
            tot = sum(r[1] for r in rows)
            for name, ms, cnt, f, _ in sorted(rows, key=lambda r: -r[1]):
                tf = (f * sb * cnt / (ms * 1e-3) / 1e12) if (ms > 0 and f > 0) else 0
                sys.stderr.write('%-52s %8.3f ms/step %5.1f%%  %7.1f TFLOP/s\n' % (name, ms / KI, 100 * ms / tot, tf))
            sys.stderr.write('planned ops %.3f ms/step of %.3f ms/step\n' % (tot / KI, ms_per_step))
        print(json.dumps(out))
        sys.stdout.flush()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == '__main__':
    main()
