#!/usr/bin/env python
"""bench.py -- images/sec of the Light-Head R-CNN forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--workload lighthead|resnet50]

One "step" = one pass of the whole hot path (backbone + RPN + proposals + PsRoiAlign + light
head + per-class NMS) over one batch of B synthetic 480x480 images per GPU, inputs already
resident in HBM.  The batch runs as --ways concurrent sub-batches (default 2 x 64), each a net
instance replaying its hipGraph on its own stream, so the partial last round of workgroups of one
launch is filled by the other stream's kernels.  N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N`
(one rank per GPU); images are sharded by rank (independent units, weak scaling) and the only
exchange is one all-gather of the fixed-size padded detections per step over RCCL
(torch.distributed backend "nccl").  Rank 0 prints ONE JSON line.

Per-step timing: barrier + device sync on both sides of exactly K steps, MAX over ranks.
roofline: the conv/dense MFMA kernel, bracketed by HIP events on its launch stream inside
the timed region (xdet_profile_*); achieved = algorithmic dense FLOPs / summed kernel time.
cpu_baseline: the NumPy/OpenBLAS oracle (a port of the reference graph, not the TF1 runtime,
which cannot run here) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import ctypes
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU per step')
    ap.add_argument('--ways', type=int, default=2,
                    help='the batch runs as this many concurrent sub-batches (net instances on their own HIP '
                         'streams): the partial last round of one launch is filled by the other stream')
    ap.add_argument('--workload', default='lighthead', choices=['lighthead', 'resnet50'])
    ap.add_argument('--proposals', type=int, default=300, help='rpn_post_nms_top_n (BASELINE config 3: 300)')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3', 'f16'],
                    help='conv/dense arithmetic: exact f32 MFMA, split-precision f16 MFMA (~f32 accuracy), plain f16')
    ap.add_argument('--voc-stream', action='store_true',
                    help='BASELINE config 4 input: raw uint8 VOC-shape images resident in HBM (shapes cycling '
                         '375x500, 500x375, 333x500, 500x333); every step runs the F1 pre-processing kernel per '
                         'image (whiten + TF-legacy bilinear warp to 480x480) in front of the forward')
    ap.add_argument('--eager', action='store_true',
                    help='launch kernel by kernel in the timed region (default: replay the captured hipGraph)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=12, help='images of the bounded CPU-oracle sample (~1.4 s each)')
    ap.add_argument('--ops', action='store_true', help='also print the per-op table to stderr')
    ap.add_argument('--no-parity', action='store_true', help='skip the live f16x3-vs-f32 GPU cross-check')
    return ap.parse_args()


def read_profile(handle, kind):
    from xdet._lib import lib, check
    maxo = 512
    n = ctypes.c_int()
    ms = (ctypes.c_double * maxo)()
    cnt = (ctypes.c_int * maxo)()
    fl = (ctypes.c_double * maxo)()
    check(lib().xdet_profile_read(handle, kind, maxo, ctypes.byref(n), ms, cnt, fl))
    rows = []
    buf = ctypes.create_string_buffer(256)
    for i in range(n.value):
        check(lib().xdet_profile_op_name(handle, kind, i, buf, 256))
        rows.append((buf.value.decode(), ms[i], cnt[i], fl[i]))
    return rows


def cpu_baseline(args, weights):
    """The oracle timed on this box's host cores (kind "port")."""
    from oracle import lighthead_oracle as O
    from xdet import weights as W
    n = args.cpu_images
    if args.workload == 'lighthead':
        imgs = W.synthetic_images(1, 480, seed=11)
        O.lighthead_forward(imgs, weights, rpn_post_nms_top_n=args.proposals)      # warm-up (page-in, BLAS threads)
        t = time.time()
        for i in range(n):
            O.lighthead_forward(W.synthetic_images(1, 480, seed=20 + i), weights, rpn_post_nms_top_n=args.proposals)
        dt = time.time() - t
        what = '%d x one 480x480 image through the full forward (R=%d), NumPy fp32 + OpenBLAS' % (n, args.proposals)
    else:
        x = np.transpose(W.synthetic_images(1, 480, seed=11), (0, 2, 3, 1))
        O.resnet50_trunk(x, weights)
        t = time.time()
        for i in range(n):
            O.resnet50_trunk(np.transpose(W.synthetic_images(1, 480, seed=20 + i), (0, 2, 3, 1)), weights)
        dt = time.time() - t
        what = '%d x one 480x480 image through the ResNet-50 v2 trunk, NumPy fp32 + OpenBLAS' % n
    try:
        from threadpoolctl import threadpool_info
        cores = max([d.get('num_threads', 1) for d in threadpool_info()] or [os.cpu_count()])
    except Exception:
        cores = os.cpu_count()
    return {'value': round(n / dt, 3), 'unit': 'images/sec', 'cores': int(cores), 'kind': 'port', 'sample': what}


def traffic_from_profiles(precision):
    """HBM bytes per conv launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/<tag>_summary.json, written by tools/summarize_profile.py with the gfx950 x2 read
    correction); PMC collection cannot run inside the bench itself."""
    import glob
    if precision != 'f16x3':      # the committed PMC passes are of the default configuration only
        return None, None
    want = 'conv_dma_f16'
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_summary.json')), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        tot = cnt = 0
        for k, e in d.get('kernels', {}).items():
            if want in k and 'hbm_bytes_per_launch' in e:
                tot += e['hbm_bytes_per_launch'] * e['calls']
                cnt += e['calls']
        if cnt:
            return int(tot / cnt), os.path.relpath(path, ROOT)
    return None, None


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
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    torch = None
    # under torch.distributed.run (RANK set) the collective path is exercised even for one rank
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('XDET_BENCH_NO_DIST') != '1')
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', local_rank))

    from xdet import weights as W
    from xdet import dist as xdist
    from xdet._lib import lib, check
    from xdet.runtime import Event
    check(lib().xdet_set_device(local_rank))
    from xdet.runtime import set_precision
    set_precision(args.precision)

    B, K, Wm = args.batch, args.steps, args.warmup
    if args.workload == 'lighthead':
        from xdet.model import LightHeadDetector
        weights = W.make_lighthead_weights(1234)
        ways = max(1, args.ways)
        if B % ways:
            raise SystemExit('--batch must be a multiple of --ways')
        sb = B // ways                               # images per sub-batch / net instance
        nets = [LightHeadDetector(weights, image_size=480, max_batch=sb, rpn_post_nms_top_n=args.proposals)
                for _ in range(ways)]
        net = nets[0]
        kind = 0
        fl = net.flops_per_image()
        flops_img = sum(fl.values())
        imgs = W.synthetic_images(B, 480, seed=100 + rank)
        for i, nt in enumerate(nets):
            nt.set_images(imgs[i * sb:(i + 1) * sb])
        raw = None
        if args.voc_stream:
            # raw uint8 images stay resident; F1 writes the whitened 480x480 planes into the net's input buffer
            from xdet.runtime import to_device
            shapes = [(375, 500), (500, 375), (333, 500), (500, 333)]
            rng = np.random.default_rng(7 + rank)
            raw = [(to_device(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)), h, w) for h, w in shapes]
        nc, topk = net.num_classes - 1, net.nms_topk
        gather = None
        if use_dist:
            # detections land in torch-owned device memory so RCCL can gather them in place
            sc = torch.zeros((B, nc, topk), dtype=torch.float32, device='cuda')
            bx = torch.zeros((B, nc, topk, 4), dtype=torch.float32, device='cuda')
            allb = torch.zeros((world * B, nc, topk, 5), dtype=torch.float32, device='cuda')
            gather = (sc, bx, allb)

        use_graph = not args.eager

        def step(graph=None, only_first=False):
            g = use_graph if graph is None else graph
            run = nets[:1] if only_first else nets
            if raw is not None:
                for nt in run:
                    for j in range(sb):
                        buf, h, w = raw[j % len(raw)]
                        check(lib().xdet_preprocess_eval(buf.ptr, h, w, nt._images.ptr + j * 3 * 480 * 480 * 4, 480,
                                                         nt.stream.handle))
            if gather is None:
                for nt in run:
                    nt.forward_device(sb, use_graph=g)
            else:
                sc, bx, allb = gather
                for i, nt in enumerate(run):
                    nt.forward_device(sb, use_graph=g, det_scores_ptr=sc[i * sb:].data_ptr(),
                                      det_boxes_ptr=bx[i * sb:].data_ptr())
                for nt in run:
                    nt.stream.synchronize()
                xdist.gather_detections(xdist.pack_detections(sc, bx), world, allb)
    else:
        from xdet.resnet import ResNet50Trunk
        weights = W.make_resnet50_weights(4321)
        net = ResNet50Trunk(weights, image_size=480, max_batch=B)
        kind = 1
        flops_img = net.flops_per_image()
        fl = {'backbone': flops_img}
        net.set_images(W.synthetic_images(B, 480, seed=100 + rank))

        use_graph = False
        nets, sb, ways = [net], B, 1

        def step(graph=None, only_first=False):
            net.forward_device(B)

    def sync_all():
        for nt in nets:
            nt.stream.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(Wm):
        step()
    sync_all()
    # per-op HIP events cannot be recorded into a captured graph: with graph replay the roofline leg
    # is an eager, instrumented repeat of the same K steps right after the timed region
    profile = not use_graph
    if profile:
        check(lib().xdet_profile_enable(net.handle, kind, 1))
    ev0, ev1 = Event(), Event()
    sync_all()
    t0 = time.perf_counter()
    ev0.record(net.stream)
    for _ in range(K):
        step()
    ev1.record(net.stream)
    sync_all()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_ms(ev1)
    if not profile:
        # roofline leg: one sub-batch alone, eager and instrumented (per-launch HIP event pairs)
        step(graph=False, only_first=True)
        sync_all()
        check(lib().xdet_profile_enable(net.handle, kind, 1))
        for _ in range(K):
            step(graph=False, only_first=True)
        sync_all()
    rows = read_profile(net.handle, kind)
    check(lib().xdet_profile_enable(net.handle, kind, 0))

    if use_dist:
        dt = xdist.max_over_ranks(dt, device='cuda')

    if rank == 0:
        ms_per_step = dt / K * 1e3
        value = world * B * K / dt
        conv_ms = sum(r[1] for r in rows if r[3] > 0)
        conv_launches = sum(r[2] for r in rows if r[3] > 0)
        conv_flops = sum(r[3] * r[2] for r in rows if r[3] > 0) * sb    # flops are per image, a launch covers one sub-batch
        roof = None
        if conv_ms > 0:
            ach = conv_flops / (conv_ms * 1e-3) / 1e12
            peak = PEAK_F32_MFMA_TFLOPS if args.precision == 'f32' else PEAK_F16_MFMA_TFLOPS
            kname = 'conv_mfma_f32_kernel' if args.precision == 'f32' else 'conv_dma_f16_kernel (+conv_mfma_f16_kernel for the 5 small/strided convs)'
            default_cfg = args.workload == 'lighthead' and sb == 64 and args.proposals == 300
            traffic = traffic_from_profiles(args.precision) if default_cfg else (None, None)
            roof = {'bound': 'mfma', 'kernel': kname, 'achieved': round(ach, 2),
                    'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                    'mfma_issued_tflops': round(ach * (3 if args.precision == 'f16x3' else 1), 2),
                    'mfma_util': round(ach * (3 if args.precision == 'f16x3' else 1) / peak, 4),
                    'traffic': traffic[0],
                    'traffic_unit': ('HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE), from ' + str(traffic[1]))
                    if traffic[0] else 'no PMC pass committed for this configuration',
                    'launches_per_step': conv_launches // K,
                    'avg_launch_us': round(conv_ms * 1e3 / max(conv_launches, 1), 2),
                    'kernel_ms_per_step': round(conv_ms / K, 3), 'gflop_per_image': round(flops_img / 1e9, 2),
                    'how': ('HIP event pair around every conv/dense launch on its launch stream, ' +
                            ('in an eager repeat of the same K steps right after the graph-replayed timed region'
                             if use_graph else 'inside the timed region'))}
        out = {
            'metric': 'images/sec at 480x480 Light-Head R-CNN, 1/2/4/8 MI355X + backbone MFMA util%',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': K, 'warmup': Wm,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'f16x3': 'f16x3 (split-precision f16 MFMA, f32 accumulate, f32 activations)',
                      'f16': 'f16'}[args.precision], 'data': 'synthetic',
            'config': {'workload': ('Full Light-Head R-CNN (Xception backbone + RPN + GPU proposals/NMS + PSROIAlign + '
                                    'light head + per-class NMS), %d proposals, 480x480' % args.proposals)
                       if args.workload == 'lighthead' else 'ResNet-50 v2 trunk only (BASELINE config 2), 480x480',
                       'batch_per_gpu': B, 'global_batch': B * world, 'image_size': 480,
                       'concurrent_sub_batches': ways,
                       'input': ('uint8 VOC-shape stream + F1 pre-processing kernel in the step' if args.voc_stream
                                 else 'whitened f32 [B,3,480,480] resident in HBM'),
                       'parallelism': 'image-sharded dp%d, all-gather of detections' % world,
                       'weights': 'seeded random init (no checkpoint exists)', 'graph_replay': bool(use_graph)},
            'device_ms_per_step': round(dev_ms / K, 3),
            'gflop_per_image': {k: round(v / 1e9, 2) for k, v in fl.items()},
            'roofline': roof,
        }
        if args.workload == 'lighthead' and args.precision != 'f32' and not args.no_parity:
            out['parity'] = live_parity(weights, args.proposals)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, weights)
        else:
            out['cpu_baseline'] = None
        if args.ops and rows:
            tot = sum(r[1] for r in rows)
            for name, ms, cnt, f in sorted(rows, key=lambda r: -r[1]):
                tf = (f * sb * cnt / (ms * 1e-3) / 1e12) if (ms > 0 and f > 0) else 0
                sys.stderr.write('%-52s %8.3f ms/step %5.1f%%  %7.1f TFLOP/s\n' % (name, ms / K, 100 * ms / tot, tf))
            sys.stderr.write('planned ops %.3f ms/step of %.3f ms/step\n' % (tot / K, ms_per_step))
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
