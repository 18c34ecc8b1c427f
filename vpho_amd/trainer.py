"""Entry-point glue that mirrors the reference's ``Trainer`` surface for the inference path
(lib/engine/base_trainer.py:18-96, lib/engine/train_diff_hand_obj.py:27-357): ``Trainer(cfg).eval()`` builds
``vpho_net``, optionally loads a checkpoint with the reference's key layout, runs ``evaluate`` over per-rank batches and
prints the MJE/MVE tables after ONE all-gather of fixed-layout metric rows.  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N main.py --mode eval ...`` (or ``accelerate launch``, which sets the
same RANK/LOCAL_RANK/WORLD_SIZE variables).  There is no dataset in this build: batches are synthetic (vpho_amd.synth).
``run`` (``--mode train``): ``--train_scope full`` (default) runs the end-to-end training step (train_step.DiffusionTrainStep:
backbone, heat-map heads, encoders, MANO head and score networks are all updated); ``--train_scope score`` trains the two score
networks on frozen features (``ScoreTrainer.step``: the reference's ``repeat_num`` DSM draws, backward, data-parallel gradient
average and AdamW per denoiser).
"""
import os
import time

import torch
import torch.distributed as dist

from . import evaluate as E
from .assets import load_assets
from .synth import synth_state_dict, synth_batch


def load_checkpoint_state_dict(path):
    """The state_dict of a checkpoint written by the reference: ``accel.save_state(<save_dir>/checkpoint/epoch_N.state)`` --
    a DIRECTORY holding ``model.safetensors`` (accelerate's default) or ``pytorch_model.bin`` next to optimizer / scheduler /
    RNG files (base_trainer.py:81-89) -- or ``final_model.pt`` / any plain ``state_dict`` file (base_trainer.py:91-96).
    Keys of a DistributedDataParallel wrapper (``module.`` prefix) are accepted."""
    if os.path.isdir(path):
        cand = [os.path.join(path, f) for f in ('model.safetensors', 'pytorch_model.bin', 'model.bin', 'pytorch_model.safetensors')]
        found = [c for c in cand if os.path.exists(c)]
        if not found:
            raise FileNotFoundError(f'{path} is a directory but holds no model file of an accelerate state (looked for: '
                                    + ', '.join(os.path.basename(c) for c in cand) + f'); it contains: {sorted(os.listdir(path))[:20]}')
        path = found[0]
    elif not os.path.exists(path):
        raise FileNotFoundError(f'checkpoint {path} does not exist')
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location='cpu', weights_only=True)
    if isinstance(sd, dict) and 'state_dict' in sd and not any(torch.is_tensor(v) for v in sd.values()):
        sd = sd['state_dict']
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    return sd, path


def synthetic_mano_targets(mano, gt_rot6d, gt_shape, is_right):
    """ground-truth vertices / joints of a synthetic hand pose: the MANO forward of the training kernel with zero loss weights"""
    bs = gt_rot6d.shape[0]
    z = lambda *s: torch.zeros(s, device=gt_rot6d.device)
    _, _, _, gv, gj = mano.train(gt_rot6d.contiguous(), gt_shape.contiguous(), z(bs, 778, 3), z(bs, 21, 3), gt_rot6d.contiguous(), gt_shape.contiguous(),
                                 is_right.to(torch.uint8).contiguous(), (0.0, 0.0, 0.0, 0.0), want_outputs=True)
    return dict(gt_hand_vert_flip=gv, gt_hand_jt3d_flip=gj, gt_mano=torch.cat([z(bs, 48), gt_shape], 1))


def loader_indices_are_consecutive(index):
    """a collated ``index`` column (dexycb6.py:472) can serve as the row's image index only if the batch is a run"""
    i = index.reshape(-1).long()
    return bool((i[1:] - i[:-1] == 1).all()) if i.numel() > 1 else True


class Trainer:
    def __init__(self, cfg):
        self.cfg = cfg
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        if self.world > 1 and not dist.is_initialized():
            from .launch import init_process_group
            init_process_group(self.device)                # loud on failure: bounded timeout, expected vs observed world
        self.setup_seed()
        self.assets = load_assets(cfg.asset_root)
        self.model = self.get_model()

    def setup_seed(self):
        """base_trainer.py:39-50: seed = random_seed + rank * 1e8."""
        torch.manual_seed(int(self.cfg.random_seed + self.rank * 1e8))

    def get_model(self):
        from .model.VPHO import vpho_net
        model = vpho_net(self.assets)
        if self.cfg.checkpoint:
            sd, path = load_checkpoint_state_dict(self.cfg.checkpoint)
            missing, unexpected = model.load_state_dict(sd, strict=False)      # base_trainer.py:81-83 (strict=False)
            if self.rank == 0:
                print(f'loaded {path}: {len(missing)} missing, {len(unexpected)} unexpected keys')
        else:
            model.load_state_dict(synth_state_dict(model, seed=1))
        return model.to(self.device).eval()

    def _synthetic_targets(self, bs, i):
        g = torch.Generator().manual_seed(int(self.cfg.random_seed + 7919 * i + self.rank))
        rot = torch.linalg.qr(torch.randn(bs, 17, 3, 3, generator=g))[0]                   # synthetic ground truth: random rotations
        gt_hand = rot[:, :16, :2, :].reshape(bs, 96).to(self.device)                       # 16 x rot6d (mano_aa_to_6D(...)[..., :-10])
        gt_obj = torch.cat([rot[:, 16, :2, :].reshape(bs, 6), torch.randn(bs, 3, generator=g) * 0.05], -1).to(self.device)
        return gt_hand, gt_obj, g

    def _train_batches(self, loader, n_batches, mano):
        """yields (batch on the device, gt_hand (bs,96) rot6d, gt_obj (bs,9)).  ``loader`` None: synthetic batches with synthetic targets.
        Otherwise any iterable of collated Appendix-A batch dicts with the training keys of lib/dataset/dexycb6.py:471-509 (hm_hand,
        hm_obj, gt_mano, gt_obj, gt_hand_vert_flip, gt_hand_jt3d_flip, force_local, ...): the targets are derived as the reference's
        forward does, gt_hand = mano_aa_to_6D(gt_mano)[..., :96] (VPHO.py:190, head_mano.py:10-18), gt_obj as it stands."""
        cfg, bs = self.cfg, self.cfg.batch_size
        if loader is not None:
            from .model.VPHO import _aa_to_rot6d
            for i, b in enumerate(loader):
                if n_batches is not None and i >= n_batches:
                    return
                b = self._to_device(b)
                missing = [k for k in ('gt_mano', 'gt_obj', 'hm_hand', 'hm_obj') if k not in b]
                if missing:
                    raise KeyError(f'training batch lacks {missing} (lib/dataset/dexycb6.py:471-509)')
                yield b, _aa_to_rot6d(b['gt_mano'][:, :48].float()), b['gt_obj'].float().contiguous()
            return
        for i in range(cfg.num_batches if n_batches is None else n_batches):
            batch = self._to_device(synth_batch(bs, self.assets, seed=cfg.random_seed + i, rank=self.rank))
            gt_hand, gt_obj, g = self._synthetic_targets(bs, i)
            batch['hm_hand'] = (torch.rand(bs, 21, cfg.heatmap_size, cfg.heatmap_size, generator=g) * 0.2).to(self.device)
            batch['hm_obj'] = (torch.rand(bs, 27, cfg.heatmap_size, cfg.heatmap_size, generator=g) * 0.2).to(self.device)
            batch.update(synthetic_mano_targets(mano, gt_hand, (torch.randn(bs, 10, generator=g) * 0.5).to(self.device), batch['is_right']))
            batch['force_local'] = (torch.randn(bs, 32, 3, generator=g) * 0.1).to(self.device)      # pseudo-force labels (force_optim.py's output)
            yield batch, gt_hand, gt_obj

    def run_full(self, n_batches=None, loader=None):
        """End-to-end training (train_diff_hand_obj.py:169-199) with all 13 losses of VPHO.py:190-212 (train_step.DiffusionTrainStep):
        backbone, heat-map heads, encoders, score networks, head_mano, both cross modules and head_physics are updated.  Batches:
        ``_train_batches`` (synthetic by default, or any iterable of Appendix-A dicts).  Returns the per-batch loss dicts (floats)."""
        from .train_step import DiffusionTrainStep
        cfg = self.cfg
        step = DiffusionTrainStep(self.model.state_dict(), self.device, assets=self.assets)
        hist = []
        for i, (batch, gt_hand, gt_obj) in enumerate(self._train_batches(loader, n_batches, step.mano_head.mano)):
            L = step.step(batch, gt_hand, gt_obj)
            hist.append({k: float(v) for k, v in L.items()})
            if self.rank == 0 and i % max(1, getattr(cfg, 'print_freq', 10)) == 0:
                print(f'[{i:04d}] ' + '  '.join(f'{k.replace("_loss", "")} {v:.3e}' for k, v in hist[-1].items()))
        self.model.load_state_dict(step.state_dict(), strict=False)
        return hist

    def run(self, n_batches=None, loader=None):
        """``Trainer.run`` (train_diff_hand_obj.py:126-153).  ``loader``: an iterable of Appendix-A batch dicts; default synthetic."""
        if getattr(self.cfg, 'train_scope', 'full') == 'full':
            return self.run_full(n_batches, loader)
        if loader is not None:
            raise NotImplementedError("--train_scope score trains on frozen features of synthetic batches only; use --train_scope full with a loader")
        return self.run_score(n_batches)

    def run_score(self, n_batches=None):
        from .model.engine import Engine
        from .train_score import ScoreTrainer
        cfg, bs = self.cfg, self.cfg.eval_batch_size
        n_batches = cfg.num_batches if n_batches is None else n_batches
        eng = Engine(self.model)
        sd = self.model.state_dict()
        hand, obj = ScoreTrainer(sd, 'denoiser_hand', self.device), ScoreTrainer(sd, 'denoiser_obj', self.device)
        losses = []
        for i in range(n_batches):
            batch = {k: (v.to(self.device) if torch.is_tensor(v) else v)
                     for k, v in synth_batch(bs, self.assets, seed=cfg.random_seed + i, rank=self.rank).items()}
            with torch.no_grad():
                f = eng.features(batch)
            gt_hand, gt_obj, _ = self._synthetic_targets(bs, i)
            lh, _ = hand.step(f['encoding_hand'], gt_hand, repeat_num=getattr(cfg, 'repeat_num', 20))
            lo, _ = obj.step(f['encoding_obj'], gt_obj, repeat_num=getattr(cfg, 'repeat_num', 20))
            losses.append((float(lh), float(lo)))
            if self.rank == 0 and i % max(1, getattr(cfg, 'print_freq', 10)) == 0:
                print(f'[{i:04d}/{n_batches}] diff_hand_loss {losses[-1][0]:.3e}  diff_obj_loss {losses[-1][1]:.3e}')
        self.model.load_state_dict({**hand.state_dict(), **obj.state_dict()}, strict=False)
        return losses

    def _to_device(self, batch):
        return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def _eval_batches(self, loader):
        """yields (batch on the device, (gt_joint, gt_vert), index of its first image).  ``loader`` None: ``cfg.num_batches`` synthetic
        batches whose ground truth is batch 0's own regression output (there is no data set in this build).  Otherwise ANY iterable of
        collated Appendix-A batch dicts (lib/dataset/dexycb6.py:471-509; what ``accel.prepare`` hands each rank,
        train_diff_hand_obj.py:211-217): ground truth = ``gt_joint`` / ``gt_hand_vert`` in the camera frame (:237-238), the object's
        ``gt_obj_rt`` or, like the reference's postprocess of the batch (:578-597), ``gt_obj`` (9-D) + root joint."""
        cfg, bs = self.cfg, self.cfg.eval_batch_size
        if loader is None:
            gt = None
            for i in range(cfg.num_batches):
                b = self._to_device(synth_batch(bs, self.assets, seed=cfg.random_seed + i, rank=self.rank))
                gt = yield b, gt, (self.rank * cfg.num_batches + i) * bs
            return
        seen = 0
        for b in loader:
            # column 0 of the metric rows, decided on the HOST batch (a device tensor here would cost two syncs per batch inside the
            # three-deep pipeline): the data set's own indices (one arange when the batch is a run of them, else the index tensor itself); batches WITHOUT an `index` key get rank-unique NEGATIVE ids
            # -(rank + world * running count) - 1 per image -- they can collide neither with another rank's fallback nor with a real data-set index
            idx = b.get('index')
            first = ids = None
            if torch.is_tensor(idx) and idx.numel():
                hidx = (idx.cpu() if idx.is_cuda else idx).reshape(-1)
                if loader_indices_are_consecutive(hidx):
                    first = int(hidx[0])
                else:
                    ids = hidx.clone()            # a shuffled loader / DistributedSampler: the data set's own indices, image by image
            b = self._to_device(b)
            missing = [k for k in ('gt_joint', 'gt_hand_vert') if k not in b]
            if missing:
                raise KeyError(f'evaluation batch lacks {missing}: Trainer.eval(loader) needs the ground truth of lib/dataset/dexycb6.py:471-509')
            if 'gt_obj_rt' not in b and 'gt_obj' in b:
                from . import ops
                b['gt_obj_rt'] = ops.obj_9d_to_rt(b['gt_obj'].double().contiguous(), b['root_joint'].float().contiguous()).float()
            n = b['rgb'].shape[0]
            yield b, (b['gt_joint'].float(), b['gt_hand_vert'].float()), (first if first is not None else ids if ids is not None else -(self.rank + self.world * (seen + torch.arange(n))) - 1)
            seen += n

    @torch.no_grad()
    def eval(self, loader=None):
        """``evaluate(testing_dataloader)`` (train_diff_hand_obj.py:202-357).  ``loader``: see ``_eval_batches``; default synthetic."""
        rows = []
        t0 = time.perf_counter()
        # three batches in flight (independent images; see evaluate.PipelinedPredictor)
        pipe = E.PipelinedPredictor(self.model, depth=3)
        futs = []
        gen = self._eval_batches(loader)
        try:
            item = next(gen)
            while True:
                b, gt, first = item
                if gt is None:
                    # synthetic run: batch 0 provides the ground truth (its own regression output), so it is evaluated first
                    out0 = pipe.submit(b).result()
                    gt = (out0['reg_hand_joint'] + b['root_joint'][:, None], out0['reg_hand_vert'] + b['root_joint'][:, None])
                    rows.append(E.metric_rows(out0, b, gt[0], gt[1], first, self.assets))
                else:
                    futs.append(pipe.submit(b, lambda out, batch, eng, first=first, gt=gt: E.metric_rows(out, batch, gt[0], gt[1], first, self.assets)))
                item = gen.send(gt)
        except StopIteration:
            pass
        rows += [f.result() for f in futs]
        pipe.close()
        # a rank whose shard is empty still takes part in the collective (with zero rows: the ragged gather carries the counts first);
        # raising here would leave the other ranks blocked in their all-gather.  Only an evaluation without ANY image is an error
        mine = torch.cat(rows, 0) if rows else torch.zeros((0, E.ROW), device=self.device, dtype=torch.float32)
        rows = E.gather_rows(mine)
        if rows.shape[0] == 0:
            raise ValueError('Trainer.eval: the loader yielded no batch on any rank')
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.rank == 0:
            what = 'synthetic images' if loader is None else 'images'
            print(f'evaluated {rows.shape[0]} {what} on {self.world} GPU(s) in {dt:.2f} s ({rows.shape[0] / dt:.1f} images/s)')
            table = E.summarize(rows.cpu())
            for name, r in table.items():
                if name != 'object':
                    print(f'{name:>5s}: n={r["n"]:5d}  MJE reg {r["MJE_reg"]:.2f}  first {r["MJE_first"]:.2f}  agg {r["MJE_agg"]:.2f}  MVE agg {r["MVE_agg"]:.2f}  (mm)')
            print('object (aggregated pose): ' + '  '.join(f'{k} {v:.2f}' for k, v in table['object'].items()))
            import json
            print('EVAL_JSON ' + json.dumps({'images': int(rows.shape[0]), 'world': self.world, 'table': table}))
        if self.world > 1:
            dist.barrier()
        return rows

    def infer(self):
        return self.eval()
