"""RCCL process group alive (watchdog thread running) while the execution plans capture their HIP graphs: one rank, cuda:0."""
import os, sys, warnings
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import torch, torch.distributed as dist
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
from vpho_amd import evaluate as E
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=dev)
t = torch.ones(1, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, a).items()}
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    pipe = E.PipelinedPredictor(m, 2)
    futs = [pipe.submit(data, lambda out, batch, eng: float(out['agg_hand_joint'].abs().sum())) for _ in range(6)]
    res = [f.result() for f in futs]
    dist.barrier(); torch.cuda.synchronize()
    print('results', res[:3], 'graph entries per slot', [len(e._features_graph.entries) for e in pipe.engines],
          'disabled', [e._features_graph.disabled for e in pipe.engines], 'warnings', [str(x.message)[:80] for x in w])
dist.destroy_process_group()
