"""(e) multi-GPU on the 1-GPU box: a world-size-1 RCCL communicator through the C-ABI
(xdet_comm_init -> ncclCommInitRank, xdet_comm_allgather_detections -> pack kernel + ncclAllGather on
the communicator's stream), the device pack kernel against the NumPy record layout, the stream
protocol (the gather waits for the producer streams; the producers wait for the pack), and the
hipGraph cache keyed on every baked-in pointer."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pack_kernel_matches_the_numpy_layout():
    from xdet import dist as xd
    from xdet._lib import lib, check
    from xdet.runtime import DeviceBuffer, to_device, to_host, synchronize
    rng = np.random.default_rng(0)
    s = rng.random((3, 20, 200), dtype=np.float32)
    b = rng.random((3, 20, 200, 4), dtype=np.float32)
    ds, db = to_device(s), to_device(b)
    out = DeviceBuffer(s.size * 5 * 4, zero=True)
    check(lib().xdet_pack_detections(ds.ptr, db.ptr, s.size, out.ptr, None))
    synchronize()
    got = to_host(out.ptr, s.shape + (5,), np.float32)
    assert np.array_equal(got, xd.pack_detections(s, b))


def test_world1_rccl_allgather_through_the_cabi(lh_weights):
    from xdet import dist as xd
    from xdet import weights as W
    from xdet.model import LightHeadDetector
    from xdet.runtime import DeviceBuffer, set_precision
    comm = xd.Communicator(0, 1)
    info = comm.info()
    assert info['world'] == 1 and info['rank'] == 0 and info['rccl_version'] > 20000
    assert comm.max_over_ranks(3.25) == 3.25
    comm.barrier()
    set_precision('f16x3')
    try:
        nets = [LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=300) for _ in range(2)]
    finally:
        set_precision('f32')
    imgs = W.synthetic_images(2, 480, seed=5)
    nc, k = 20, 200
    det_s, det_b = DeviceBuffer(2 * nc * k * 4, zero=True), DeviceBuffer(2 * nc * k * 16, zero=True)
    ref = []
    for i, nt in enumerate(nets):
        nt.set_images(imgs[i:i + 1])
        nt.forward_device(1, use_graph=False)
        ref.append(nt.detections(1))
    ref_s = np.concatenate([r[0] for r in ref])
    ref_b = np.concatenate([r[1] for r in ref])
    assert (ref_s > 0).sum() > 50
    outs = []
    for step in range(3):                 # back-to-back steps, no host sync in between
        for i, nt in enumerate(nets):
            nt.forward_device(1, use_graph=True, det_scores_ptr=det_s.ptr + i * nc * k * 4,
                              det_boxes_ptr=det_b.ptr + i * nc * k * 16)
        outs.append(comm.allgather_detections(det_s.ptr, det_b.ptr, 2, nc, k, streams=[nt.stream for nt in nets]))
    assert outs[0].ptr == outs[2].ptr != outs[1].ptr          # double-buffered results
    g = comm.gathered()
    assert g.shape == xd.gathered_layout(1, 2) == (2, 20, 200, 5)
    s, b = xd.unpack_detections(g)
    assert np.array_equal(s, ref_s) and np.array_equal(b, ref_b)
    # two alternating det buffer pairs (what bench.py does): forwards never wait for the pack of their own step
    pairs = [(det_s, det_b), (DeviceBuffer(2 * nc * k * 4, zero=True), DeviceBuffer(2 * nc * k * 16, zero=True))]
    for step in range(4):
        ds, db = pairs[step & 1]
        for i, nt in enumerate(nets):
            nt.forward_device(1, use_graph=True, det_scores_ptr=ds.ptr + i * nc * k * 4,
                              det_boxes_ptr=db.ptr + i * nc * k * 16)
        comm.allgather_detections(ds.ptr, db.ptr, 2, nc, k, streams=[nt.stream for nt in nets], double_buffered=True)
    s, b = xd.unpack_detections(comm.gathered())
    assert np.array_equal(s, ref_s) and np.array_equal(b, ref_b)
    comm.close()


def test_graph_cache_is_keyed_on_every_pointer(lh_weights):
    """round-1 bug: a replay with other output buffers silently wrote into the first capture's."""
    from xdet import weights as W
    from xdet._lib import lib, check
    from xdet.model import LightHeadDetector
    from xdet.runtime import DeviceBuffer, to_host, to_device, set_precision
    set_precision('f16x3')
    try:
        det = LightHeadDetector(lh_weights, image_size=480, max_batch=1, rpn_post_nms_top_n=300)
    finally:
        set_precision('f32')
    nc, k = 20, 200
    det.set_images(W.synthetic_images(1, 480, seed=9))
    det.forward_device(1, use_graph=True)
    s0, b0 = det.detections(1)
    assert (s0 > 0).sum() > 20
    cnt = ctypes.c_int()
    check(lib().xdet_net_graph_count(det.handle, ctypes.byref(cnt)))
    assert cnt.value == 1
    # second pair of output buffers: must be written (own graph), the first pair left alone
    s2, b2 = DeviceBuffer(nc * k * 4, zero=True), DeviceBuffer(nc * k * 16, zero=True)
    check(lib().xdet_memset(det._det_scores.ptr, 0, nc * k * 4, det.stream.handle))
    det.forward_device(1, use_graph=True, det_scores_ptr=s2.ptr, det_boxes_ptr=b2.ptr)
    det.stream.synchronize()
    assert np.array_equal(to_host(s2.ptr, (1, nc, k)), s0) and np.array_equal(to_host(b2.ptr, (1, nc, k, 4)), b0)
    assert not to_host(det._det_scores.ptr, (1, nc, k)).any()
    check(lib().xdet_net_graph_count(det.handle, ctypes.byref(cnt)))
    assert cnt.value == 2
    # other image_shapes / bbox_img pointers are part of the key too
    shp = to_device(np.array([[375, 500]], np.int32))
    bb = to_device(np.array([[0, 0, 1, 1]], np.float32))
    det.forward_device(1, use_graph=True, image_shapes_ptr=shp.ptr, bbox_img_ptr=bb.ptr)
    det.stream.synchronize()
    check(lib().xdet_net_graph_count(det.handle, ctypes.byref(cnt)))
    assert cnt.value == 3
    # replays of a cached tuple add nothing; the cache is bounded
    for _ in range(3):
        det.forward_device(1, use_graph=True, det_scores_ptr=s2.ptr, det_boxes_ptr=b2.ptr)
    extra = [(DeviceBuffer(nc * k * 4), DeviceBuffer(nc * k * 16)) for _ in range(8)]
    for a, b in extra:
        det.forward_device(1, use_graph=True, det_scores_ptr=a.ptr, det_boxes_ptr=b.ptr)
    det.stream.synchronize()
    check(lib().xdet_net_graph_count(det.handle, ctypes.byref(cnt)))
    assert cnt.value == 8
    assert np.array_equal(to_host(extra[-1][0].ptr, (1, nc, k)), s0)


def test_layers_remember_their_device():
    """entry points make the layer's device current for the call and restore the caller's (one device
    here, so this checks the guard is a no-op that leaves device 0 current and results intact)."""
    from xdet._lib import lib, check
    import xdet
    n = ctypes.c_int()
    check(lib().xdet_device_count(ctypes.byref(n)))
    assert n.value >= 1
    plane = np.arange(1, 26, dtype=np.float32).reshape(5, 5)
    inp = np.tile(plane, (1, 16, 1, 1)).astype(np.float32)
    rois = np.array([[[0.2, 0.2, 0.7, 0.7]]], np.float32)
    p, _ = xdet.ps_roi_align(inp, rois, 2, 2, 'mean')
    assert np.allclose(p[0, 0, :, 0], [5.125, 6.5, 12., 13.375])


def test_byte_allgather_and_device_records():
    """bench.py's comm.devices: every rank's (rank, hip device, PCI bus id, host) moved by ncclAllGather."""
    import re
    from xdet import dist as xd
    comm = xd.Communicator(0, 1)
    comm.set_timeout(60)
    got = comm.allgather_bytes(b'0123456789abcdef' * 4)
    assert got == [b'0123456789abcdef' * 4]
    recs = comm.device_records({'images_per_sec': 12.5})
    assert len(recs) == 1 and recs[0]['rank'] == 0 and recs[0]['hip_device'] == 0 and recs[0]['images_per_sec'] == 12.5
    assert re.match(r'^[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9]$', recs[0]['pci_bus_id']), recs[0]
    comm.close()


def test_ipc_probe_and_numa_binding():
    """what xdet.launch decides per launch: the IPC probe runs (this image exports HSA_ENABLE_IPC_MODE_LEGACY=0, under
    which exporting a device allocation must work), and binding to the GPU's NUMA node leaves a non-empty CPU set."""
    import os
    from xdet._lib import lib
    from xdet import launch
    assert lib().xdet_probe_ipc() == 0
    before = os.sched_getaffinity(0)
    try:
        cpus = launch.bind_to_gpu_numa(0)
        assert cpus is None or (len(cpus) > 0 and cpus <= before)
        assert len(os.sched_getaffinity(0)) > 0
    finally:
        os.sched_setaffinity(0, before)


def test_bench_through_the_rank_launcher(tmp_path):
    """`python bench.py --gpus 1 --comm` through xdet.launch.launch_ranks (the spawn path `bench.py --gpus N` takes):
    rank 0 prints exactly one JSON line with the SCALE-ready keys."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys; sys.path.insert(0, %r); from xdet.launch import launch_ranks; '
            'sys.exit(launch_ranks([sys.executable, %r, "--gpus", "1", "--comm", "--steps", "2", "--warmup", "1", '
            '"--batch", "16", "--no-cpu-baseline", "--no-parity"], 1, timeout=900))'
            % (os.path.join(root, 'x-detector_amd'), os.path.join(root, 'bench.py')))
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1000)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0
    c = d['comm']
    assert c['world'] == 1 and c['ranks_seen'] == [0] and c['distinct_gpus'] == 1
    assert c['devices'][0]['pci_bus_id'] and c['per_rank_images_per_sec'][0] > 0
    assert c['gathered_shape'] == [16, 20, 200, 5] and c['gathered_images_with_detections'] == 16
    r = d['roofline']
    assert 0 < r['frac_executed'] <= r['frac_algorithmic_credit'] and r['frac_direct_only'] > 0
    assert 0 < r['backbone_frac'] < r['frac'] <= r['frac_cap']       # frac: the dominant kernel alone, executed FLOPs
