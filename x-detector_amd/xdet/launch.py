"""Single-node launcher: one process per GPU, no torch.distributed.run needed.

    launch_ranks([sys.executable, 'bench.py', ...], n) starts n copies of the command with
    RANK / LOCAL_RANK / WORLD_SIZE set and a fresh XDET_COMM_ID_FILE (where rank 0 publishes the
    ncclUniqueId, xdet.dist.rendezvous_path).  Rank 0 inherits stdout; the other ranks' stdout
    goes to stderr so that a "rank 0 prints ONE JSON line" contract survives.  If any rank fails
    (or the timeout expires, or the launcher itself is interrupted) every rank still alive is
    terminated -- exact PIDs, SIGTERM then SIGKILL -- and the failing exit code is returned: a peer
    that died inside a collective cannot leave the others holding their GPUs.

What a rank's environment carries besides the rank variables:
  * HSA_ENABLE_IPC_MODE_LEGACY -- only what ipc_env() decides: the caller's value if it set one, nothing
    if a probe process can export device memory with the runtime's default, '0' if only the dmabuf mode works;
  * NCCL_DEBUG=WARN with NCCL_DEBUG_FILE=/dev/stderr (RCCL's warnings of every rank reach stderr, never
    the JSON line on stdout); both only as defaults;
  * CPU affinity: each rank is bound to the CPUs of its GPU's NUMA node (bind_to_gpu_numa, applied in the child
    before exec through XDET_BIND_NUMA=1 -> xdet.launch.apply_affinity_from_env(), which bench.py calls first).
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_ipc_cache = None


def _probe_ipc(extra_env, timeout=60):
    """run xdet_probe_ipc() in a FRESH process (the HSA runtime reads the variable once, when it starts).
    Returns True / False, or None when the probe could not run at all (no GPU, no library)."""
    env = dict(os.environ)
    env.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)
    env.update(extra_env)
    env['XDET_IPC_PROBE'] = '1'
    code = ('import sys; sys.path.insert(0, %r); from xdet._lib import lib; l = lib(); '
            'import ctypes; n = ctypes.c_int(); '
            'sys.exit(3) if l.xdet_device_count(ctypes.byref(n)) != 0 or n.value < 1 else None; '
            'sys.exit(0 if l.xdet_probe_ipc() == 0 else 1)' % os.path.dirname(_HERE))
    try:
        rc = subprocess.call([sys.executable, '-c', code], env=env, stdout=subprocess.DEVNULL,
                             stderr=subprocess.DEVNULL, timeout=timeout)
    except Exception:
        return None
    return True if rc == 0 else False if rc == 1 else None


def ipc_env(probe=_probe_ipc):
    """{} or {'HSA_ENABLE_IPC_MODE_LEGACY': v}: what a multi-rank launch must export so that RCCL's intra-node
    transport can share device memory between the rank processes.  An explicit setting of the caller always wins;
    otherwise the runtime's default is kept if a probe process can export a device allocation with it, and the
    dmabuf mode ('0': the only one the host driver of the MI355X boxes this was built on supports) is chosen only
    when the default fails and '0' works."""
    global _ipc_cache
    if 'HSA_ENABLE_IPC_MODE_LEGACY' in os.environ:
        return {'HSA_ENABLE_IPC_MODE_LEGACY': os.environ['HSA_ENABLE_IPC_MODE_LEGACY']}
    if os.environ.get('XDET_IPC_PROBE'):          # we ARE the probe process: never recurse
        return {}
    if os.environ.get('XDET_IPC_PROBED') == '1':  # a rank of xdet.launch: the launcher probed once for all of them (N
        return {}                                 # ranks x two HIP-initialising probe processes would contend for the GPUs)
    if probe is _probe_ipc and _ipc_cache is not None:
        return dict(_ipc_cache)
    out = {}
    if probe({}) is False and probe({'HSA_ENABLE_IPC_MODE_LEGACY': '0'}):
        out = {'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    if probe is _probe_ipc:
        _ipc_cache = dict(out)
    return out


def numa_cpus_of_pci(bus_id, sysfs='/sys'):
    """CPUs of the NUMA node the PCI device `bus_id` ("0000:c1:00.0") hangs off, or None (single node / unknown)."""
    try:
        node = int(open(os.path.join(sysfs, 'bus/pci/devices', bus_id.lower(), 'numa_node')).read().strip())
        if node < 0:
            return None
        txt = open(os.path.join(sysfs, 'devices/system/node/node%d/cpulist' % node)).read().strip()
    except (OSError, ValueError):
        return None
    cpus = set()
    for part in txt.split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def bind_to_gpu_numa(local_rank, sysfs='/sys'):
    """pin this process to the CPUs next to GPU `local_rank` (launch threads, RCCL proxy threads and the staging
    copies then stay on the socket the GPU's PCIe root port belongs to).  Returns the CPU set or None."""
    try:
        import ctypes
        from ._lib import lib
        buf = ctypes.create_string_buffer(32)
        if lib().xdet_device_pci_bus_id(int(local_rank), buf, 32) != 0:
            return None
        cpus = numa_cpus_of_pci(buf.value.decode(), sysfs)
        if cpus:
            allowed = os.sched_getaffinity(0)
            cpus = (cpus & allowed) or None
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:
        return None


def apply_affinity_from_env():
    """called first thing by a rank process (bench.py): XDET_BIND_NUMA=1 -> bind to the GPU's NUMA node."""
    if os.environ.get('XDET_BIND_NUMA', '') == '1' and 'LOCAL_RANK' in os.environ:
        return bind_to_gpu_numa(int(os.environ['LOCAL_RANK']))
    return None


def rank_env(rank, world, id_file, base=None, ipc=None):
    env = dict(os.environ if base is None else base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world),
                'LOCAL_WORLD_SIZE': str(world), 'XDET_COMM_ID_FILE': id_file})
    env.update(ipc_env() if ipc is None else ipc)
    env['XDET_IPC_PROBED'] = '1'        # the launcher decided the IPC mode once: the ranks' lib() must not probe again
    env.setdefault('XDET_BIND_NUMA', '1')
    env.setdefault('NCCL_DEBUG', 'WARN')
    env.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')
    return env


def _stop(procs, grace=10):
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            p.terminate()
        except OSError:
            pass
    t0 = time.time()
    for p in alive:
        try:
            p.wait(max(0.1, grace - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
                p.wait(5)
            except (OSError, subprocess.TimeoutExpired):
                pass


def launch_ranks(cmd, n, timeout=None, poll=0.05, ipc=None):
    tmp = tempfile.mkdtemp(prefix='xdet_ranks_')
    id_file = os.path.join(tmp, 'rccl_unique_id')
    procs = []
    try:
        if ipc is None:
            ipc = ipc_env()
        for r in range(n):
            procs.append(subprocess.Popen(cmd, env=rank_env(r, n, id_file, ipc=ipc),
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
                    sys.stderr.write('xdet.launch: rank %d exited with code %d; stopping the other ranks\n'
                                     % (procs.index(p), code))
            if rc != 0 or (timeout and time.time() - t0 > timeout):
                _stop(alive)
                return rc or 124
            time.sleep(poll)
        return 0
    finally:
        # KeyboardInterrupt or any exception in the loop above: no rank may outlive the launcher
        _stop(procs)
        shutil.rmtree(tmp, ignore_errors=True)
