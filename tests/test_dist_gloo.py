"""The N>1 path on CPU.  The product transport is RCCL behind the C-ABI (csrc/comm.hip), which needs
GPUs; what CAN be checked here is everything around it: the shard partition, the detection record
layout and rank-major gather layout (two gloo ranks move the records exactly as ncclAllGather
would -- torch is test infrastructure only, the product imports none of it), the rendezvous file
naming, and the rank launcher that `bench.py --gpus N` uses."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'x-detector_amd'))
    import torch
    import torch.distributed as dist
    from xdet import dist as xd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    G, C, K = 6, 20, 200
    lo, hi = xd.shard_range(G, rank, world)
    # deterministic "detections" of the global batch; each rank fills only its own shard
    rng = np.random.default_rng(0)
    all_scores = rng.random((G, C, K), dtype=np.float32)
    all_boxes = rng.random((G, C, K, 4), dtype=np.float32)
    packed = xd.pack_detections(all_scores[lo:hi], all_boxes[lo:hi])
    shape = xd.gathered_layout(world, hi - lo, C, K)
    out = torch.empty(shape, dtype=torch.float32)
    # all-gather with ncclAllGather semantics: rank r's send buffer lands at offset r * sendcount
    dist.all_gather(list(out.chunk(world, dim=0)), torch.from_numpy(packed))
    s, b = xd.unpack_detections(out.numpy())
    ok = bool(np.array_equal(s, all_scores) and np.array_equal(b, all_boxes))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the max-over-ranks timing rule of bench.py
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok, float(t.item()), (lo, hi), tuple(out.shape)))


def test_two_rank_gather_of_detections():
    pytest.importorskip('torch')
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


def test_pack_unpack_round_trip():
    from xdet import dist as xd
    rng = np.random.default_rng(3)
    s = rng.random((3, 20, 200), dtype=np.float32)
    b = rng.random((3, 20, 200, 4), dtype=np.float32)
    p = xd.pack_detections(s, b)
    assert p.shape == (3, 20, 200, 5) and p.flags['C_CONTIGUOUS']
    s2, b2 = xd.unpack_detections(p)
    assert np.array_equal(s, s2) and np.array_equal(b, b2)


def test_payload_is_latency_bound():
    """80 KB per image: at batch 8 per rank x 8 ranks ~5 MB total, far below one xGMI link-second."""
    per_image = 20 * 200 * 5 * 4
    assert per_image == 80000
    assert 8 * 8 * per_image < 153e9 * 1e-3


def test_rendezvous_path_is_shared_by_the_workers_of_one_launch():
    from xdet import dist as xd
    assert xd.rendezvous_path({'XDET_COMM_ID_FILE': '/x/y'}) == '/x/y'
    env = {'MASTER_PORT': '29511', 'TORCHELASTIC_RUN_ID': 'none', 'TORCHELASTIC_RESTART_COUNT': '0', 'TMPDIR': '/tmp'}
    a, b = xd.rendezvous_path(env), xd.rendezvous_path(dict(env))
    assert a == b and a.startswith('/tmp/xdet_rccl_id_29511_%d_' % os.getppid())
    assert xd.rendezvous_path(dict(env, MASTER_PORT='29512')) != a
    assert xd.rendezvous_path(dict(env, TORCHELASTIC_RESTART_COUNT='1')) != a


def test_launcher_hands_out_ranks_and_one_id_file(tmp_path):
    """xdet.launch.launch_ranks == what `bench.py --gpus N` does when started plainly."""
    from xdet.launch import launch_ranks
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, sys\n'
        'r = os.environ["RANK"]\n'
        'open(os.path.join(%r, "out_" + r), "w").write(" ".join([r, os.environ["LOCAL_RANK"], '
        'os.environ["WORLD_SIZE"], os.environ["XDET_COMM_ID_FILE"]]))\n'
        'print("rank", r)\n' % str(tmp_path))
    assert launch_ranks([sys.executable, str(script)], 3) == 0
    got = [open(tmp_path / ('out_%d' % r)).read().split() for r in range(3)]
    assert [g[0] for g in got] == ['0', '1', '2'] and [g[1] for g in got] == ['0', '1', '2']
    assert all(g[2] == '3' for g in got)
    assert len({g[3] for g in got}) == 1 and not os.path.exists(os.path.dirname(got[0][3]))   # cleaned up


def test_launcher_propagates_a_failing_rank(tmp_path):
    from xdet.launch import launch_ranks
    script = tmp_path / 'rank.py'
    script.write_text('import os, sys, time\n'
                      'if os.environ["RANK"] == "1": sys.exit(7)\n'
                      'time.sleep(30)\n')
    assert launch_ranks([sys.executable, str(script)], 2) == 7


def test_bench_rejects_a_world_size_mismatch():
    """--gpus must equal WORLD_SIZE when a launcher provides ranks (round-1 bug: --gpus was ignored)."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b'WORLD_SIZE=2' in p.stderr


def test_no_torch_in_the_product_or_the_bench():
    import re
    files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
    for d, _, fs in os.walk(os.path.join(ROOT, 'x-detector_amd')):
        files += [os.path.join(d, f) for f in fs if f.endswith('.py')]
    for f in files:
        src = open(f).read()
        assert not re.search(r'^\s*(import|from)\s+torch\b', src, flags=re.M), f


# ---- hardening of the N-rank path (round 3): none of it needs a GPU ------------------------------
def _gone(pid, timeout=10.0):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            os.kill(pid, 0)
        except OSError:
            return True
        time.sleep(0.05)
    return False


def test_launcher_returns_promptly_when_a_rank_is_killed_mid_run(tmp_path):
    """rank 1 dies by SIGKILL while rank 0 sits in a (simulated) collective that would never finish: the launcher
    must come back with a failure well inside the timeout and leave no rank behind."""
    import signal
    import time
    from xdet.launch import launch_ranks
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, signal, sys, time\n'
        'r = os.environ["RANK"]\n'
        'open(os.path.join(%r, "pid_" + r), "w").write(str(os.getpid()))\n'
        'if r == "1":\n'
        '    time.sleep(0.5)\n'
        '    os.kill(os.getpid(), signal.SIGKILL)\n'
        'signal.signal(signal.SIGTERM, signal.SIG_IGN)   # a rank stuck in a collective does not react to SIGTERM\n'
        'time.sleep(600)\n' % str(tmp_path))
    t0 = time.time()
    rc = launch_ranks([sys.executable, str(script)], 2, timeout=120)
    dt = time.time() - t0
    assert rc == -signal.SIGKILL
    assert dt < 30, dt                       # SIGTERM grace (10 s) + SIGKILL, not the 120 s timeout
    assert _gone(int(open(tmp_path / 'pid_0').read()))     # rank 0 is gone too


def test_launcher_kills_its_ranks_when_it_is_interrupted(tmp_path):
    """an exception inside the poll loop (KeyboardInterrupt) must not leave ranks holding GPUs."""
    from xdet import launch
    script = tmp_path / 'rank.py'
    script.write_text('import os, time\n'
                      'open(os.path.join(%r, "pid_" + os.environ["RANK"]), "w").write(str(os.getpid()))\n'
                      'time.sleep(600)\n' % str(tmp_path))
    real_sleep = launch.time.sleep
    calls = {'n': 0}

    def sleep(t):
        calls['n'] += 1
        if calls['n'] > 20 and not calls.get('raised') and all(os.path.exists(tmp_path / ('pid_%d' % r)) for r in range(2)):
            calls['raised'] = True           # once: Popen.wait(timeout) inside the cleanup sleeps through time.sleep too
            raise KeyboardInterrupt
        real_sleep(t)
    launch.time.sleep = sleep
    try:
        with pytest.raises(KeyboardInterrupt):
            launch.launch_ranks([sys.executable, str(script)], 2, ipc={})
    finally:
        launch.time.sleep = real_sleep
    for r in range(2):
        assert _gone(int(open(tmp_path / ('pid_%d' % r)).read()))


def test_rank_env_defaults(monkeypatch):
    from xdet.launch import rank_env
    monkeypatch.delenv('NCCL_DEBUG', raising=False)
    monkeypatch.delenv('NCCL_DEBUG_FILE', raising=False)
    env = rank_env(2, 4, '/tmp/idf', base={'PATH': '/bin'}, ipc={})
    assert env['RANK'] == '2' and env['LOCAL_RANK'] == '2' and env['WORLD_SIZE'] == '4'
    assert env['XDET_COMM_ID_FILE'] == '/tmp/idf'
    assert env['NCCL_DEBUG'] == 'WARN' and env['NCCL_DEBUG_FILE'] == '/dev/stderr'   # RCCL warnings -> stderr, per rank
    assert env['XDET_BIND_NUMA'] == '1'
    assert 'HSA_ENABLE_IPC_MODE_LEGACY' not in env                                    # nothing forced
    env = rank_env(0, 2, '/x', base={'NCCL_DEBUG': 'INFO'}, ipc={'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    assert env['NCCL_DEBUG'] == 'INFO' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_ipc_mode_is_set_only_when_the_probe_says_the_default_fails(monkeypatch):
    from xdet import launch
    monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
    monkeypatch.delenv('XDET_IPC_PROBE', raising=False)
    monkeypatch.delenv('XDET_IPC_PROBED', raising=False)
    seen = []

    def probe_factory(default_ok, dmabuf_ok):
        def probe(extra):
            seen.append(dict(extra))
            return dmabuf_ok if extra.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0' else default_ok
        return probe
    assert launch.ipc_env(probe_factory(True, True)) == {}                                   # default works: untouched
    assert launch.ipc_env(probe_factory(False, True)) == {'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    assert launch.ipc_env(probe_factory(False, False)) == {}                                 # nothing works: do not guess
    assert launch.ipc_env(probe_factory(None, True)) == {}                                   # probe could not run (no GPU)
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '1')
    n = len(seen)
    assert launch.ipc_env(probe_factory(False, True)) == {'HSA_ENABLE_IPC_MODE_LEGACY': '1'}  # the caller's word wins
    assert len(seen) == n                                                                     # ... without probing
    # a rank started by xdet.launch: the launcher probed once for all ranks and says so (no N x 2 probe processes)
    monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY')
    monkeypatch.setenv('XDET_IPC_PROBED', '1')
    assert launch.ipc_env(probe_factory(False, True)) == {} and len(seen) == n
    assert launch.rank_env(1, 2, '/x', base={}, ipc={})['XDET_IPC_PROBED'] == '1'


def test_numa_cpus_of_a_pci_device(tmp_path):
    from xdet.launch import numa_cpus_of_pci
    dev = tmp_path / 'bus/pci/devices/0000:c1:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    node = tmp_path / 'devices/system/node/node1'
    node.mkdir(parents=True)
    (node / 'cpulist').write_text('64-67,192-193\n')
    assert numa_cpus_of_pci('0000:C1:00.0', str(tmp_path)) == {64, 65, 66, 67, 192, 193}
    (dev / 'numa_node').write_text('-1\n')
    assert numa_cpus_of_pci('0000:c1:00.0', str(tmp_path)) is None          # single-node box: leave the affinity alone
    assert numa_cpus_of_pci('0000:ff:00.0', str(tmp_path)) is None
