"""`main.py --mode eval --model vpho_net ...` end to end (BASELINE.json configs[0] sizes: sample_num=4, sampling_steps=5, topk 8/3),
as its own process like the reference's `accelerate launch main.py` (README.md:61-72), with a checkpoint in accelerate's
directory layout (base_trainer.py:81-89).  The table it prints must equal an in-process evaluation with the same weights."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--sample_num', '4', '--sampling_steps', '5', '--topk_hand', '8', '--topk_obj', '3', '--sample_T0', '0.2',
        '--eval_batch_size', '2', '--num_batches', '2', '--random_seed', '7']


def _run_main(extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--mode', 'eval', '--model', 'vpho_net'] + ARGS + extra,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('EVAL_JSON ')]
    assert len(line) == 1, r.stdout[-2000:]
    return json.loads(line[0][len('EVAL_JSON '):]), r.stdout


def test_main_eval_with_accelerate_checkpoint(tmp_path, assets):
    from safetensors.torch import save_file
    from vpho_amd.model.VPHO import vpho_net
    from vpho_amd.synth import synth_state_dict
    torch.manual_seed(7)                                     # base_trainer.py:39-50 (rank 0): the seed is set BEFORE the model is
    m = vpho_net(assets)                                     # built, and building it draws from the same CPU generator (Trainer order)
    rng_after_build = torch.get_rng_state()
    sd = synth_state_dict(m, seed=3)                         # NOT the default seed: the numbers below prove the file was used
    d = tmp_path / 'checkpoint' / 'epoch_45.state'
    d.mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / 'model.safetensors'))
    (d / 'optimizer.bin').write_bytes(b'')                   # accelerate writes these next to the model; they must be ignored
    (d / 'random_states_0.pkl').write_bytes(b'')
    got, stdout = _run_main(['--checkpoint', str(d)])
    assert 'model.safetensors: 0 missing, 0 unexpected keys' in stdout, stdout[-1500:]
    assert got['images'] == 4 and got['world'] == 1
    # in-process evaluation with the same weights, batches and seeds (Trainer.eval's loop, sequential)
    from vpho_amd.configs.args import cfg
    from vpho_amd import evaluate as E
    from vpho_amd.synth import synth_batch
    saved = {k: getattr(cfg, k) for k in ('sample_num', 'sampling_steps', 'topk_hand', 'topk_obj', 'sample_T0')}
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 4, 5, 8, 3, 0.2
    try:
        m.load_state_dict(sd)
        m = m.cuda().eval()
        torch.set_rng_state(rng_after_build)
        rows, gt = [], None
        for i in range(2):
            b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(2, assets, seed=7 + i, rank=0).items()}
            out = m(b, mode='predict')
            if gt is None:
                gt = (out['reg_hand_joint'] + b['root_joint'][:, None], out['reg_hand_vert'] + b['root_joint'][:, None])
            rows.append(E.metric_rows(out, b, gt[0], gt[1], i * 2, assets))
        want = E.summarize(torch.cat(rows, 0).cpu())
    finally:
        for k, v in saved.items():
            setattr(cfg, k, v)
    for side in want:
        for k, v in want[side].items():
            assert got['table'][side][k] == pytest.approx(v, rel=1e-5, abs=1e-6), (side, k, got['table'][side][k], v)
    # and the default weights give a different table: the checkpoint was really loaded
    other, _ = _run_main([])
    assert abs(other['table']['both']['MJE_agg'] - got['table']['both']['MJE_agg']) > 1e-3


def test_main_eval_bad_checkpoint_path_fails_loudly(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--mode', 'eval'] + ARGS + ['--checkpoint', str(tmp_path / 'missing.state')],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'FileNotFoundError' in r.stderr


def test_trainer_eval_consumes_any_iterable_of_batch_dicts(assets):
    """``Trainer.eval(loader)`` is the reference's ``evaluate(testing_dataloader)`` (train_diff_hand_obj.py:202-258): fed the SAME synthetic
    batches as dicts that carry their ground truth (gt_joint / gt_hand_vert, dexycb6.py:471-509) it prints the table of the default
    synthetic run, row for row; a ragged last batch and host-resident tensors are fine; a batch without ground truth is refused."""
    from vpho_amd.configs.args import cfg
    from vpho_amd.synth import synth_batch
    from vpho_amd.trainer import Trainer
    keys = ('sample_num', 'sampling_steps', 'topk_hand', 'topk_obj', 'sample_T0', 'eval_batch_size', 'num_batches', 'random_seed', 'checkpoint')
    saved = {k: getattr(cfg, k) for k in keys}
    cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 4, 5, 8, 3, 0.2
    cfg.eval_batch_size, cfg.num_batches, cfg.random_seed, cfg.checkpoint = 2, 3, 7, None
    try:
        t = Trainer(cfg)
        torch.manual_seed(11)
        want = t.eval()
        batches = [synth_batch(2, t.assets, seed=7 + i, rank=0) for i in range(3)]
        b0 = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batches[0].items()}
        out0 = t.model(b0, mode='predict')
        gt_j, gt_v = (out0['reg_hand_joint'] + b0['root_joint'][:, None]).cpu(), (out0['reg_hand_vert'] + b0['root_joint'][:, None]).cpu()
        for b in batches:
            b['gt_joint'], b['gt_hand_vert'] = gt_j.clone(), gt_v.clone()
        torch.manual_seed(11)
        got = t.eval(loader=iter(batches))                     # a one-shot iterator, tensors on the host
        assert torch.equal(got[:, 1:], want[:, 1:])
        # column 0 without an `index` column in the batches: rank-unique negative ids -(rank + world * running count) - 1
        assert got[:, 0].tolist() == [-1.0, -2.0, -3.0, -4.0, -5.0, -6.0]
        for i, b in enumerate(batches):                        # with the data set's own index column (a run): that index
            b['index'] = torch.arange(100 + 2 * i, 102 + 2 * i)
        torch.manual_seed(11)
        got = t.eval(loader=iter(batches))
        assert got[:, 0].tolist() == [100.0, 101.0, 102.0, 103.0, 104.0, 105.0] and torch.equal(got[:, 1:], want[:, 1:])
        for i, b in enumerate(batches):                        # a shuffled loader: the indices are kept image by image
            b['index'] = torch.tensor([[905, 17], [3, 4400], [12, 11]][i])
        torch.manual_seed(11)
        got = t.eval(loader=iter(batches))
        assert got[:, 0].tolist() == [905.0, 17.0, 3.0, 4400.0, 12.0, 11.0] and torch.equal(got[:, 1:], want[:, 1:])
        for b in batches:
            del b['index']
        with pytest.raises(ValueError, match='no batch on any rank'):
            t.eval(loader=[])
        # ragged last batch: one image
        last = {k: (v[:1] if torch.is_tensor(v) else v[:1]) for k, v in batches[2].items()}
        rows = t.eval(loader=[batches[0], batches[1], last])
        assert rows.shape[0] == 5 and torch.isfinite(rows).all()
        with pytest.raises(KeyError, match='gt_joint'):
            t.eval(loader=[synth_batch(2, t.assets, seed=1, rank=0)])
    finally:
        for k, v in saved.items():
            setattr(cfg, k, v)
