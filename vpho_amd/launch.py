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


def maybe_spawn(gpus, script=None, argv=None):
    """Call FIRST in main(), before importing anything that may touch the GPU.  Returns normally in a rank process (or for
    N == 1); in the parent of an N > 1 job it never returns: it runs the ranks as a child process group and exits with
    the child's return code."""
    if gpus <= 1 or launched_by_torchrun():
        return
    script = script or os.path.abspath(sys.argv[0])
    argv = sys.argv[1:] if argv is None else argv
    cmd = launch_command(script, gpus, argv)
    r = subprocess.run(cmd, env=child_env(gpus))
    sys.exit(r.returncode)


def world_from_env(gpus):
    """(world, rank, local_rank) of this process; raises when a launcher's WORLD_SIZE contradicts --gpus."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != gpus:
        raise SystemExit(f'--gpus {gpus} but the launcher set WORLD_SIZE={world}')
    return world, rank, local
