"""Single-node launcher: one process per GPU, no torch.distributed.run needed.

    launch_ranks([sys.executable, 'bench.py', ...], n) starts n copies of the command with
    RANK / LOCAL_RANK / WORLD_SIZE set and a fresh XDET_COMM_ID_FILE (where rank 0 publishes the
    ncclUniqueId, xdet.dist.rendezvous_path).  Rank 0 inherits stdout; the other ranks' stdout
    goes to stderr so that a "rank 0 prints ONE JSON line" contract survives.  If any rank fails
    the others are terminated (exact PIDs) and its exit code is returned.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time


def rank_env(rank, world, id_file, base=None):
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world),
                'LOCAL_WORLD_SIZE': str(world), 'XDET_COMM_ID_FILE': id_file,
                # the host driver only supports dmabuf IPC (RCCL's intra-node P2P needs it)
                'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
    return env


def launch_ranks(cmd, n, timeout=None, poll=0.05):
    tmp = tempfile.mkdtemp(prefix='xdet_ranks_')
    id_file = os.path.join(tmp, 'rccl_unique_id')
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen(cmd, env=rank_env(r, n, id_file),
                                          stdout=None if r == 0 else sys.stderr))
        t0 = time.time()
        rc = 0
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout and time.time() - t0 > timeout):
                for p in alive:
                    p.terminate()
                for p in alive:
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                return rc or 124
            time.sleep(poll)
        return 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
