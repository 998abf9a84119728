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
