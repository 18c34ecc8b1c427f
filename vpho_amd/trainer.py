"""Entry-point glue that mirrors the reference's ``Trainer`` surface for the inference path
(lib/engine/base_trainer.py:18-96, lib/engine/train_diff_hand_obj.py:27-357): ``Trainer(cfg).eval()`` builds
``vpho_net``, optionally loads a checkpoint with the reference's key layout, runs ``evaluate`` over per-rank batches and
prints the MJE/MVE tables after ONE all-gather of fixed-layout metric rows.  One process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N main.py --mode eval ...`` (or ``accelerate launch``, which sets the
same RANK/LOCAL_RANK/WORLD_SIZE variables).  There is no dataset in this build: batches are synthetic (vpho_amd.synth).
Training (``run``) is out of scope of the hot path and raises.
"""
import os
import time

import torch
import torch.distributed as dist

from . import evaluate as E
from .assets import load_assets
from .synth import synth_state_dict, synth_batch


class Trainer:
    def __init__(self, cfg):
        self.cfg = cfg
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('nccl', device_id=self.device)
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
            path = self.cfg.checkpoint
            if os.path.isdir(path):                                    # accelerate.save_state directory (epoch_N.state)
                cand = [os.path.join(path, f) for f in ('pytorch_model.bin', 'model.safetensors')]
                path = next(p for p in cand if os.path.exists(p))
            if path.endswith('.safetensors'):
                from safetensors.torch import load_file
                sd = load_file(path)
            else:
                sd = torch.load(path, map_location='cpu')
            missing, unexpected = model.load_state_dict(sd, strict=False)      # base_trainer.py:81-83 (strict=False)
            if self.rank == 0:
                print(f'loaded {path}: {len(missing)} missing, {len(unexpected)} unexpected keys')
        else:
            model.load_state_dict(synth_state_dict(model, seed=1))
        return model.to(self.device).eval()

    def run(self):
        raise NotImplementedError('vpho_amd builds the inference hot path; training (--mode train) is not part of it')

    @torch.no_grad()
    def eval(self):
        cfg, bs = self.cfg, self.cfg.eval_batch_size
        rows = []
        t0 = time.perf_counter()
        # two batches in flight (independent images; see evaluate.PipelinedPredictor); batch 0 also provides the synthetic
        # ground truth (its own regression output), so it is evaluated first
        pipe = E.PipelinedPredictor(self.model, depth=2)
        make = lambda i: {k: (v.to(self.device) if torch.is_tensor(v) else v)
                          for k, v in synth_batch(bs, self.assets, seed=cfg.random_seed + i, rank=self.rank).items()}
        b0 = make(0)
        out0 = pipe.submit(b0).result()
        gt = (out0['reg_hand_joint'] + b0['root_joint'][:, None], out0['reg_hand_vert'] + b0['root_joint'][:, None])
        rows.append(E.metric_rows(out0, b0, gt[0], gt[1], self.rank * cfg.num_batches * bs, self.assets))
        futs = []
        for i in range(1, cfg.num_batches):
            first = (self.rank * cfg.num_batches + i) * bs
            futs.append(pipe.submit(make(i), lambda out, batch, eng, first=first: E.metric_rows(out, batch, gt[0], gt[1], first, self.assets)))
        rows += [f.result() for f in futs]
        pipe.close()
        rows = E.gather_rows(torch.cat(rows, 0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.rank == 0:
            print(f'evaluated {rows.shape[0]} synthetic images on {self.world} GPU(s) in {dt:.2f} s ({rows.shape[0] / dt:.1f} images/s)')
            for name, r in E.summarize(rows.cpu()).items():
                print(f'{name:>5s}: n={r["n"]:5d}  MJE reg {r["MJE_reg"]:.2f}  first {r["MJE_first"]:.2f}  agg {r["MJE_agg"]:.2f}  MVE agg {r["MVE_agg"]:.2f}  (mm)')
        if self.world > 1:
            dist.barrier()
        return rows

    def infer(self):
        return self.eval()
