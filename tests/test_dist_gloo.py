"""The N>1 path on CPU: two gloo ranks shard a global batch, pack fixed-size detection records
and all-gather them -- the same xdet.dist functions bench.py uses over RCCL."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'x-detector_amd'))
    import torch
    import torch.distributed as dist
    from xdet import dist as xd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G, C, K = 6, 20, 200
    lo, hi = xd.shard_range(G, rank, world)
    # deterministic "detections" of the global batch; each rank fills only its own shard
    g = torch.Generator().manual_seed(0)
    all_scores = torch.rand((G, C, K), generator=g)
    all_boxes = torch.rand((G, C, K, 4), generator=g)
    packed = xd.pack_detections(all_scores[lo:hi].clone(), all_boxes[lo:hi].clone())
    out = xd.gather_detections(packed, world)
    s, b = xd.unpack_detections(out)
    ok = bool(torch.equal(s, all_scores) and torch.equal(b, all_boxes))
    t = xd.max_over_ranks(1.0 + rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, t, (lo, hi), tuple(out.shape)))


def test_two_rank_gather_of_detections():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] for r in res)                      # every rank sees the whole gathered batch, in order
    assert all(r[2] == 2.0 for r in res)               # max-over-ranks timing
    assert res[0][3] == (0, 3) and res[1][3] == (3, 6)
    assert res[0][4] == (6, 20, 200, 5)


def test_shard_range_covers_ragged_batches():
    from xdet import dist as xd
    for G in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            spans = [xd.shard_range(G, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_payload_is_latency_bound():
    """80 KB per image: at batch 8 per rank x 8 ranks ~5 MB total, far below one xGMI link-second."""
    per_image = 20 * 200 * 5 * 4
    assert per_image == 80000
    assert 8 * 8 * per_image < 153e9 * 1e-3
