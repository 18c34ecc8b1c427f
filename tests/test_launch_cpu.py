"""`python <entry>.py --gpus N` from a bare shell starts its own N rank processes (vpho_amd/launch.py) -- the driver's scaling leg
calls bench.py exactly like that.  CPU / gloo only."""
import json
import os
import subprocess
import sys

import pytest

from vpho_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _bare_env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'OMP_NUM_THREADS')}
    env['PYTHONPATH'] = ROOT
    return env


def test_command_and_env():
    cmd = launch.launch_command('/x/bench.py', 4, ['--gpus', '4', '--steps', '3'], port=29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-5:] == ['/x/bench.py', '--gpus', '4', '--steps', '3']
    env = launch.child_env(8, base={'PATH': '/bin'})
    assert 1 <= int(env['OMP_NUM_THREADS']) <= 4 and env['OMP_WAIT_POLICY'] == 'PASSIVE' and env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert launch.child_env(2, base={'OMP_NUM_THREADS': '7'})['OMP_NUM_THREADS'] == '7'      # the caller's setting wins


def test_world_mismatch_is_an_error(monkeypatch):
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    with pytest.raises(SystemExit):
        launch.world_from_env(4)


def test_single_process_does_not_spawn():
    r = subprocess.run([sys.executable, os.path.join(HERE, '_launch_probe.py'), '--gpus', '1'], env=_bare_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['n_gpus'] == 1 and not out['pid_is_child']


def test_bare_shell_gpus_2_starts_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(HERE, '_launch_probe.py'), '--gpus', '2', '--tag', 'abc'], env=_bare_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                      # ONE JSON line for the whole job
    out = json.loads(lines[0])
    assert out == {'n_gpus': 2, 'sum': 3.0, 'tag': 'abc', 'omp': out['omp'], 'pid_is_child': True}
    assert 1 <= int(out['omp']) <= 4


def test_entry_points_spawn_before_touching_the_gpu():
    """bench.py / train.py / train_score.py / force_optim.py call maybe_spawn() before `import torch`."""
    for f in ('bench.py', 'train.py', 'train_score.py', 'force_optim.py'):
        src = open(os.path.join(ROOT, f)).read()
        body = src[src.index('def main'):]
        assert 'maybe_spawn(args.gpus)' in body, f
        assert body.index('maybe_spawn(args.gpus)') < body.index('import torch'), f


def test_accelerate_launch_starts_the_ranks_like_the_reference_does():
    """The reference is started with `accelerate launch --config_file lib/configs/ddp01.yaml main.py ...` (README.md:61-72).  accelerate
    exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like torch.distributed.run; an entry point started that way must NOT spawn again
    and must form the process group from that environment (CPU / gloo here; `accelerate` is installed in the build container)."""
    pytest.importorskip('accelerate')
    port = launch.free_port()
    cmd = [sys.executable, '-m', 'accelerate.commands.launch', '--cpu', '--multi_gpu', '--num_processes', '2', '--num_machines', '1',
           '--mixed_precision', 'no', '--dynamo_backend', 'no', '--main_process_ip', '127.0.0.1', '--main_process_port', str(port),
           os.path.join(HERE, '_launch_probe.py'), '--gpus', '2', '--tag', 'acc']
    r = subprocess.run(cmd, env=_bare_env(), capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and '--multi_gpu' in r.stderr:                     # accelerate versions differ on how a CPU multi-process job is spelt
        cmd.remove('--multi_gpu')
        r = subprocess.run(cmd, env=_bare_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['sum'] == 3.0 and out['tag'] == 'acc'


def test_process_group_failure_is_loud_and_quick():
    """a rank whose peers never arrive: SystemExit naming rank / world / backend / rendezvous address after the bounded timeout,
    not torch's 10-30 minutes of silence"""
    env = _bare_env()
    env.update(WORLD_SIZE='2', RANK='1', LOCAL_RANK='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(launch.free_port()), VPHO_DIST_TIMEOUT_S='5')
    code = 'import sys; sys.path.insert(0, %r); from vpho_amd.launch import init_process_group; init_process_group(None)' % ROOT
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert 'process-group initialisation FAILED' in r.stderr and 'rank 1/2' in r.stderr and 'backend gloo' in r.stderr, r.stderr[-1500:]
