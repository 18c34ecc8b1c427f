"""RCCL executed on the GPU box: the 'nccl' branches of launch.init_process_group, evaluate.gather_rows, grad_buckets.GradBuckets and
train_score.allreduce_mean_scale run on a real RCCL communicator (world size 1 on the one-GPU box; VPHO_FORCE_NCCL=1, launch.group_active).
The child is started before anything touches the GPU -- a process that has initialised HIP is never forked into ranks.
Reference: accelerate's process group / prepared loaders / gather_for_metrics / DDP (lib/engine/base_trainer.py:22,
lib/engine/train_diff_hand_obj.py:121-124, 180, 333-335)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, 'tests', '_rccl_probe.py')


def _env(**extra):
    from vpho_amd.launch import free_port
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'VPHO_REHEARSE_ONE_GPU')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), VPHO_DIST_TIMEOUT_S='120', **extra)
    return env


def test_rccl_world_size_one_runs_every_collective_call_site():
    env = _env(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', VPHO_FORCE_NCCL='1', NCCL_DEBUG='VERSION')
    r = subprocess.run([sys.executable, PROBE, 'one'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['backend'] == 'nccl' and d['destroyed'] is True
    assert 'process group up: 1 ranks, backend nccl' in r.stderr, r.stderr[-2000:]
    major = int(d['rccl_version'].split('.')[0])
    assert major >= 2, d
    print('RCCL', d['rccl_version'], '|', d['gather_rows'], '|', d['train_step'], '|', d['score_step'])
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'rccl_world1.json'), 'w') as f:
            json.dump(dict(d, stderr_tail=r.stderr[-1500:]), f, indent=1)


@pytest.mark.skipif(os.environ.get('VPHO_TRY_DUP_GPU') != '1', reason='a record, made once per round by hand (VPHO_TRY_DUP_GPU=1): profiles/r05_rccl.txt')
def test_two_ranks_on_one_gpu_over_rccl_is_reported():
    """Not a requirement, a record: RCCL is given two ranks that both bind cuda:0.  A refusal ('Duplicate GPU detected') is the documented
    outcome; a run is fine too.  What must hold: the attempt ENDS (bounded timeout), and nothing is left behind."""
    env = _env(NCCL_DEBUG='WARN')
    from vpho_amd.launch import launch_command
    cmd = launch_command(PROBE, 2, ['dup'], port=env['MASTER_PORT'])
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240, start_new_session=True)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, so, se = 'timeout', (e.stdout or b'').decode(errors='replace'), (e.stderr or b'').decode(errors='replace')
    lines = [json.loads(l) for l in so.splitlines() if l.startswith('{')]
    rec = {'returncode': rc, 'ranks': lines, 'stderr_tail': se[-3000:]}
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'rccl_two_ranks_one_gpu.json'), 'w') as f:
            json.dump(rec, f, indent=1)
    print(json.dumps(rec)[:1500])
    assert rc != 'timeout', 'the duplicate-GPU attempt did not end'
