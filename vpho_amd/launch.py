"""One process per GPU: self-launch for the ``--gpus N`` entry points (bench.py, train.py, force_optim.py).

The reference starts its ranks with ``accelerate launch --config_file lib/configs/ddp*.yaml main.py ...`` (README.md:61-72,
lib/configs/ddp01.yaml:3-12: one node, static rendezvous).  Here ``python <entry>.py --gpus N`` does the same from a bare
shell: when no launcher has set WORLD_SIZE and N > 1, the parent starts ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> <entry>.py <same argv>`` as a CHILD process, forwards its
output and exits with its code.  The decision is taken before anything touches the GPU (no torch.cuda call, no HIP call):
a process that has initialised the GPU must never be replaced or forked into rank processes.

Host threads: every rank's launch threads share one host.  Unless the caller already set them, the children get
``OMP_NUM_THREADS = max(1, min(4, cpus // N))`` and ``OMP_WAIT_POLICY=PASSIVE`` so that N ranks x (launch thread + sampler
threads + OpenMP workers of the CPU prior draw) do not oversubscribe the node (8 x 4.3 busy threads were measured at N=1).
"""
import os
import signal
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def child_env(n, base=None):
    env = dict(os.environ if base is None else base)
    from .hostcpu import usable_cpus
    cpus = usable_cpus()                                       # affinity AND cgroup quota (a GPU box grants 16 CPUs per GPU)
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(4, cpus // max(n, 1)))))
    env.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    return env


def launch_command(script, n, argv, port=None):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port or free_port()), script] + list(argv)


def launched_by_torchrun():
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def _die_with_parent():
    """preexec_fn of the launcher child (runs between fork and exec, nothing has touched a GPU): prctl(PR_SET_PDEATHSIG, SIGTERM)"""
    try:
        import ctypes
        ctypes.CDLL('libc.so.6', use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)
    except Exception:
        pass


def maybe_spawn(gpus, script=None, argv=None):
    """Call FIRST in main(), before importing anything that may touch the GPU.  Returns normally in a rank process (or for
    N == 1); in the parent of an N > 1 job it never returns: it runs the ranks as a child process group and exits with
    the child's return code."""
    if gpus <= 1 or launched_by_torchrun():
        return
    script = script or os.path.abspath(sys.argv[0])
    argv = sys.argv[1:] if argv is None else argv
    cmd = launch_command(script, gpus, argv)
    # the ranks run in their own session (process group): a SIGTERM / SIGINT / SIGHUP / SIGQUIT that reaches only this parent (a
    # scheduler or watchdog killing one PID, a terminal or ssh hang-up -- the ranks' session has no controlling tty, so they would never
    # see it) is forwarded to the whole group, so no rank is left behind holding a GPU; and should this parent die without running its
    # handlers (SIGKILL), the kernel sends the launcher child SIGTERM (PR_SET_PDEATHSIG), which torch.distributed.run passes on to its
    # ranks.  Still a CHILD, never an exec
    child = subprocess.Popen(cmd, env=child_env(gpus), start_new_session=True, preexec_fn=_die_with_parent)

    def forward(signum, frame):
        try:
            os.killpg(child.pid, signum)
        except ProcessLookupError:
            pass

    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP, signal.SIGQUIT)}
    try:
        rc = child.wait()
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
        if child.poll() is None:                                   # parent is leaving for another reason: take the ranks along
            forward(signal.SIGTERM, None)
            try:
                child.wait(timeout=30)
            except subprocess.TimeoutExpired:
                forward(signal.SIGKILL, None)
    if rc != 0:
        print(f'[vpho_amd.launch] the {gpus}-rank job ended with code {rc}: {" ".join(cmd)}', file=sys.stderr)
    sys.exit(rc if rc >= 0 else 128 - rc)


def init_process_group(dev=None, timeout_s=None):
    """torch.distributed rendezvous of a rank process, loud on failure: backend 'nccl' (= RCCL over xGMI on ROCm) bound to ``dev``,
    or gloo for the one-GPU rehearsal (VPHO_REHEARSE_ONE_GPU=1: every rank on cuda:0, timings meaningless).  A bounded timeout
    (VPHO_DIST_TIMEOUT_S, default 180 s) instead of torch's 10-30 minutes, the world that was EXPECTED next to the one observed, and a
    first collective right away so that a broken fabric shows here and not in the middle of a run."""
    import datetime
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world <= 1 and not force_group():
        return 'none'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world <= 1:                     # VPHO_FORCE_NCCL=1 without a launcher: a ONE-rank communicator is still a real RCCL communicator
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('LOCAL_RANK', '0')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    rehearse = os.environ.get('VPHO_REHEARSE_ONE_GPU') == '1'
    backend = 'gloo' if (rehearse or dev is None) else 'nccl'
    timeout = datetime.timedelta(seconds=float(timeout_s or os.environ.get('VPHO_DIST_TIMEOUT_S', '180')))
    where = f'rank {rank}/{world}, backend {backend}, device {dev}, MASTER {os.environ.get("MASTER_ADDR")}:{os.environ.get("MASTER_PORT")}'
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=timeout)
        else:
            dist.init_process_group('gloo', timeout=timeout)
        probe = torch.ones(1, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(probe)
        if backend == 'nccl':
            torch.cuda.synchronize(dev)
        seen = int(probe.item())
    except Exception as e:
        raise SystemExit(f'[vpho_amd] process-group initialisation FAILED ({where}): {type(e).__name__}: {e}')
    if seen != world or dist.get_world_size() != world:
        raise SystemExit(f'[vpho_amd] process group is incomplete ({where}): a first all-reduce saw {seen} ranks, '
                         f'get_world_size() = {dist.get_world_size()}, expected {world}')
    if rank == 0:
        print(f'[vpho_amd] process group up: {world} ranks, backend {dist.get_backend()}' + (' (one-GPU rehearsal)' if rehearse else ''),
              file=sys.stderr, flush=True)
    return backend


def force_group():
    """VPHO_FORCE_NCCL=1: build the process group and take every collective's call site also at world size 1 (a one-rank 'nccl' group is
    a real RCCL communicator: it loads the library, binds the device and runs the collectives) -- how the RCCL branches are executed
    on a one-GPU box (tests/test_gpu_rccl.py)."""
    return os.environ.get('VPHO_FORCE_NCCL') == '1'


def group_active():
    """True where the collectives' call sites must run: a process group exists and spans more than one rank (or VPHO_FORCE_NCCL=1)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_group())


def world_from_env(gpus):
    """(world, rank, local_rank) of this process; raises when a launcher's WORLD_SIZE contradicts --gpus."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != gpus:
        raise SystemExit(f'--gpus {gpus} but the launcher set WORLD_SIZE={world}')
    return world, rank, local
