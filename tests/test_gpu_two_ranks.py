"""The N > 1 code of libxdet_hip.so with TWO rank processes on the one GPU of the box.  RCCL refuses two ranks on one
device, so the ranks load a test double (tests/fake_rccl: the eight RCCL entry points over a shared-memory segment,
collectives enqueued on the caller's stream like the real ones) through XDET_RCCL_LIB.  Everything else is the product:
xdet.launch (rank environment, id file), csrc/comm.hip (rendezvous, ncclCommInitRank, pack kernel + all-gather with the
event protocol, scalar collectives, byte all-gather, watchdog), xdet.dist, bench.py's aggregation.  What this cannot
show is that RCCL's own transports work between GPUs -- only that nothing on OUR side of the collective is wrong when
world > 1 (a world-size-1 all-gather degenerates to a copy and hides rank-major layout / rendezvous / timing bugs)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def fake_rccl():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'fake_rccl'))
    import build as fb
    return fb.build()


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(pkg)r)
from xdet import dist as xd, weights as W
from xdet._lib import lib, check
from xdet.model import LightHeadDetector
from xdet.runtime import DeviceBuffer, set_precision
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
check(lib().xdet_set_device(0))                      # both ranks share the box's one GPU
comm = xd.Communicator(rank, world, timeout_s=120)
assert comm.info()['rccl_version'] == 99999          # the test double, not RCCL
B, S, R, nc, k = 3, 256, 50, 20, 200
G = world * B
imgs = W.synthetic_images(G, S, seed=31)
lo, hi = xd.shard_range(G, rank, world)
w = W.make_lighthead_weights(1234)
set_precision('f16x3')
det = LightHeadDetector(w, image_size=S, max_batch=G, rpn_post_nms_top_n=R)
# reference: this rank also computes the WHOLE global batch by itself
det.forward(imgs, use_graph=True)
ref_s, ref_b = det.detections(G)
pairs = [(DeviceBuffer(B * nc * k * 4, zero=True), DeviceBuffer(B * nc * k * 16, zero=True)) for _ in range(2)]
det.set_images(imgs[lo:hi])
for step in range(4):                                # back-to-back steps, alternating buffer pairs, no host sync
    ds, db = pairs[step & 1]
    det.forward_device(hi - lo, use_graph=True, det_scores_ptr=ds.ptr, det_boxes_ptr=db.ptr)
    comm.allgather_detections(ds.ptr, db.ptr, hi - lo, nc, k, streams=[det.stream], double_buffered=True)
g = comm.gathered()
s, b = xd.unpack_detections(g)
ok = g.shape == (G, nc, k, 5) and np.array_equal(s, ref_s) and np.array_equal(b, ref_b)   # rank-major == global order
mx = comm.max_over_ranks(1.5 + rank)
comm.barrier()
recs = comm.device_records({'images_per_sec': 100.0 + rank})
blobs = comm.allgather_bytes(bytes([rank]) * 64)
out = {'rank': rank, 'ok': bool(ok), 'max': mx, 'ranks': [r['rank'] for r in recs], 'pci': [r['pci_bus_id'] for r in recs],
       'pids': [r['pid'] for r in recs], 'rates': [r['images_per_sec'] for r in recs], 'blobs': [list(set(x)) for x in blobs],
       'n_det': int((s > 0).sum())}
open(os.path.join(%(out)r, 'rank_%%d.json' %% rank), 'w').write(json.dumps(out))
comm.close()
'''


def _env(fake):
    env = dict(os.environ, XDET_RCCL_LIB=fake, XDET_ALLOW_RCCL_OVERRIDE='1', XDET_BIND_NUMA='0', XDET_OVERSUBSCRIBE_GPUS='1')
    env.pop('RANK', None)
    return env


# world 8 = the node the SCALE run uses: eight oversubscribed ranks on the box's one GPU ("make N = 8
# boring before hardware shows up") -- id-file rendezvous with eight ranks, rank-major gather order, the launcher's
# NUMA / visibility handling when local_rank exceeds what the box has
WORLDS = [2, 8]


@pytest.mark.parametrize('world', WORLDS)
def test_ranks_gather_their_shards_in_rank_major_order(fake_rccl, tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'pkg': os.path.join(ROOT, 'x-detector_amd'), 'out': str(tmp_path)})
    code = ('import sys; sys.path.insert(0, %r); from xdet.launch import launch_ranks; '
            'sys.exit(launch_ranks([sys.executable, %r], %d, timeout=900))' % (os.path.join(ROOT, 'x-detector_amd'), str(script), world))
    p = subprocess.run([sys.executable, '-c', code], env=_env(fake_rccl), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    res = [json.load(open(tmp_path / ('rank_%d.json' % r))) for r in range(world)]
    for r in res:
        assert r['ok'], r                                    # every rank holds the whole global batch, in global order
        assert r['max'] == 1.5 + world - 1                   # max over ranks
        assert r['ranks'] == list(range(world)) and r['blobs'] == [[k] for k in range(world)]
        assert r['rates'] == [100.0 + k for k in range(world)]
        assert len(set(r['pci'])) == 1 and len(set(r['pids'])) == world         # `world` processes, one physical GPU
        assert r['n_det'] > 100


@pytest.mark.parametrize('world,batch', [(2, 16), (8, 4)])
def test_bench_with_ranks(fake_rccl, world, batch):
    """`python bench.py --gpus N`: the script spawns its ranks itself; rank 0 prints ONE JSON line whose value is the
    images of ALL ranks over the max-over-ranks time."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '3', '--warmup', '1',
                        '--batch', str(batch), '--no-cpu-baseline', '--no-parity', '--no-roofline'], env=_env(fake_rccl),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    G = world * batch
    assert d['n_gpus'] == world and d['config']['global_batch'] == G and d['scaling'] == 'weak'
    c = d['comm']
    assert c['world'] == world and c['ranks_seen'] == list(range(world)) and c['distinct_gpus'] == 1   # honest: one physical GPU here
    assert c['gathered_shape'] == [G, 20, 200, 5] and c['gathered_images_with_detections'] == G
    assert len(c['per_rank_images_per_sec']) == world and min(c['per_rank_images_per_sec']) > 0
    assert c['library_overridden'] is True and 'fake_rccl' in c['library']     # the line says what carried the collectives
    assert 0.1 < c['weak_scaling_efficiency_vs_rank_median'] <= 1.001
    # value = all ranks' images / the slowest rank's time: never above the sum of the per-rank rates
    assert d['value'] <= sum(c['per_rank_images_per_sec']) * 1.001
    assert abs(d['value'] - G * 3 / (d['ms_per_step'] * 3e-3)) < 1e-2 * d['value']


def test_a_dead_peer_is_a_timeout_not_a_hang(fake_rccl, tmp_path):
    """rank 1 disappears between two collectives; rank 0's next gather can never complete.  The communicator's watchdog
    turns that into an error within XDET_COMM_TIMEOUT_S, the rank exits non-zero and the launcher returns -- instead
    of rank 0 sitting in hipStreamSynchronize until someone notices."""
    script = tmp_path / 'worker.py'
    script.write_text(r'''
import os, sys, time
sys.path.insert(0, %(pkg)r)
from xdet import dist as xd
from xdet._lib import lib, check, XdetError
from xdet.runtime import DeviceBuffer
rank = int(os.environ['RANK'])
check(lib().xdet_set_device(0))
comm = xd.Communicator(rank, 2, timeout_s=120)
ds, db = DeviceBuffer(20 * 200 * 4, zero=True), DeviceBuffer(20 * 200 * 16, zero=True)
comm.allgather_detections(ds.ptr, db.ptr, 1, 20, 200)
comm.wait()                                           # one healthy collective
if rank == 1:
    os._exit(0)                                       # gone, without a goodbye (exit code 0: only the hang can tell)
time.sleep(1.0)
comm.allgather_detections(ds.ptr, db.ptr, 1, 20, 200)
t0 = time.time()
try:
    comm.wait()
except XdetError as e:
    open(os.path.join(%(out)r, 'watchdog.txt'), 'w').write('%%.1f %%s' %% (time.time() - t0, e))
    os._exit(17)
os._exit(0)
''' % {'pkg': os.path.join(ROOT, 'x-detector_amd'), 'out': str(tmp_path)})
    code = ('import sys; sys.path.insert(0, %r); from xdet.launch import launch_ranks; '
            'sys.exit(launch_ranks([sys.executable, %r], 2, timeout=300))' % (os.path.join(ROOT, 'x-detector_amd'), str(script)))
    t0 = time.time()
    p = subprocess.run([sys.executable, '-c', code], env=dict(_env(fake_rccl), XDET_COMM_TIMEOUT_S='5'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    dt = time.time() - t0
    assert p.returncode == 17, (p.returncode, p.stderr.decode()[-2000:])
    msg = open(tmp_path / 'watchdog.txt').read()
    assert 'exceeded its timeout of 5 s' in msg and 'aborted its communicator' in msg, msg
    assert 4.0 <= float(msg.split()[0]) < 30.0 and dt < 120, (msg, dt)


@pytest.mark.parametrize('call', ['barrier', 'max_over_ranks', 'allgather_bytes'])
def test_a_dead_peer_under_the_host_buffer_collectives(fake_rccl, tmp_path, call):
    """barrier / allreduce_max / allgather_bytes copy to and from HOST memory around their collective.  With
    pageable host buffers the D2H copy itself blocks until the stream drains -- behind a dead peer, forever, before the
    watchdog is polled.  They stage through pinned memory now: the same timeout as comm.wait()."""
    script = tmp_path / 'worker.py'
    script.write_text(r'''
import os, sys, time
sys.path.insert(0, %(pkg)r)
from xdet import dist as xd
from xdet._lib import lib, check, XdetError
rank = int(os.environ['RANK'])
check(lib().xdet_set_device(0))
comm = xd.Communicator(rank, 2, timeout_s=120)
comm.barrier()                                        # one healthy collective
assert comm.max_over_ranks(1.0 + rank) == 2.0
assert comm.allgather_bytes(bytes([rank]) * 8) == [bytes([0]) * 8, bytes([1]) * 8]
if rank == 1:
    os._exit(0)
time.sleep(1.0)
t0 = time.time()
try:
    %(call)s
except XdetError as e:
    open(os.path.join(%(out)r, 'watchdog.txt'), 'w').write('%%.1f %%s' %% (time.time() - t0, e))
    os._exit(17)
os._exit(0)
''' % {'pkg': os.path.join(ROOT, 'x-detector_amd'), 'out': str(tmp_path),
       'call': {'barrier': 'comm.barrier()', 'max_over_ranks': 'comm.max_over_ranks(3.0)',
                'allgather_bytes': 'comm.allgather_bytes(b"x" * 100)'}[call]})
    code = ('import sys; sys.path.insert(0, %r); from xdet.launch import launch_ranks; '
            'sys.exit(launch_ranks([sys.executable, %r], 2, timeout=300))' % (os.path.join(ROOT, 'x-detector_amd'), str(script)))
    t0 = time.time()
    p = subprocess.run([sys.executable, '-c', code], env=dict(_env(fake_rccl), XDET_COMM_TIMEOUT_S='5'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    dt = time.time() - t0
    assert p.returncode == 17, (p.returncode, p.stderr.decode()[-2000:])
    msg = open(tmp_path / 'watchdog.txt').read()
    assert 'exceeded its timeout' in msg and 'aborted its communicator' in msg, msg
    assert 4.0 <= float(msg.split()[0]) < 30.0 and dt < 120, (msg, dt)


def test_a_stand_in_for_rccl_must_be_asked_for_twice(fake_rccl):
    """XDET_RCCL_LIB alone is refused: a bench must not silently run its collectives on something that is not RCCL."""
    env = _env(fake_rccl)
    env.pop('XDET_ALLOW_RCCL_OVERRIDE')
    code = ('import sys; sys.path.insert(0, %r); from xdet import dist as xd; from xdet._lib import lib, check, XdetError\n'
            'check(lib().xdet_set_device(0))\n'
            'try:\n    xd.Communicator(0, 1)\nexcept XdetError as e:\n    print(e); sys.exit(9)\n' % os.path.join(ROOT, 'x-detector_amd'))
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 9 and b'XDET_ALLOW_RCCL_OVERRIDE' in p.stdout, (p.returncode, p.stdout, p.stderr[-2000:])


@pytest.mark.parametrize('world', WORLDS)
def test_dry_run_with_ranks(fake_rccl, world):
    """`bench.py --gpus N --dry-run`: rendezvous + one all-gather of the device records + exit, in seconds -- what an
    8-GPU node is asked first, before the long run."""
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--dry-run'], env=_env(fake_rccl),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['n_gpus'] == world and d['comm']['ranks_seen'] == list(range(world))
    assert d['comm']['distinct_gpus'] == 1 and d['comm']['library_overridden'] is True
    assert 'fake_rccl' in d['comm']['library'] and time.time() - t0 < 300


@pytest.mark.parametrize('world,batch', [(2, 16), (8, 4)])
def test_bench_under_torch_distributed_run(fake_rccl, world, batch):
    """the driver's SCALE command line, verbatim, at N = 2: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- torch only launches; every rank is
    bench.py, which imports no torch and finds its peers through the id file named after MASTER_PORT + the agent's pid."""
    pytest.importorskip('torch')
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '3',
                        '--warmup', '1', '--batch', str(batch), '--no-cpu-baseline', '--no-parity', '--no-roofline'],
                       env=_env(fake_rccl), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == world and d['comm']['world'] == world and d['comm']['ranks_seen'] == list(range(world))
    assert d['comm']['gathered_shape'] == [world * batch, 20, 200, 5] and d['value'] > 0
