"""The N > 1 entry points as their own processes, rehearsed with TWO ranks on the one GPU of the box (VPHO_REHEARSE_ONE_GPU=1: both
ranks on cuda:0, gloo instead of RCCL -- the launch path, the rank / shard arithmetic, the collectives' call sites and the single JSON
line are the real ones; timings are meaningless).  `python <entry>.py --gpus 2` from a bare shell starts its ranks itself
(vpho_amd/launch.py), like `accelerate launch` does for the reference (README.md:61-72, lib/configs/ddp01.yaml)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(VPHO_REHEARSE_ONE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-3000:]                   # ONE JSON line, from rank 0
    assert 'process group up: 2 ranks, backend gloo' in r.stderr, r.stderr[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_two_ranks(scaling):
    d = _run('bench.py', ['--gpus', '2', '--steps', '2', '--warmup', '1', '--no_cpu_baseline', '--no_kernel_timing', '--scaling', scaling])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == scaling and d['unit'] == 'images/s'
    per_rank = 64 if scaling == 'weak' else 32                 # strong: rank r takes images [32 r, 32 (r + 1)) of ONE 64-image batch
    assert d['config']['per_gpu_batch'] == per_rank and d['config']['global_batch_per_step'] == 2 * per_rank
    assert d['metrics_rows_gathered'] == 2 * 2 * per_rank      # ranks x steps x local batch: every image's row arrived exactly once
    assert d['value'] > 0 and abs(d['value'] - 2 * 2 * per_rank / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']
    assert d['config']['parallelism'] == 'dp2'


def test_train_two_ranks_keep_replicas_in_sync():
    d = _run('train.py', ['--gpus', '2', '--steps', '2', '--warmup', '1', '--bs', '8', '--repeat_num', '4'])
    assert d['n_gpus'] == 2 and d['replicas_in_sync'] is True and d['trained_tensors'] == 569
    assert d['loss_last']['total_loss'] == d['loss_last']['total_loss']          # finite, not NaN


def test_force_optim_two_ranks_shard_the_pairs():
    d = _run('force_optim.py', ['--gpus', '2', '--pairs', '512', '--iters', '300', '--phase1', '30'])
    assert d['n_gpus'] == 2 and d['pairs'] == 512 and d['value'] > 0


def test_train_score_two_ranks():
    d = _run('train_score.py', ['--gpus', '2', '--steps', '3', '--warmup', '1'])
    assert d['n_gpus'] == 2 and d['loss_hand_first_last'][1] < d['loss_hand_first_last'][0]
