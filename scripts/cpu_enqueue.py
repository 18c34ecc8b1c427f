import sys, time, torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.engine import Engine
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, a).items()}
eng = Engine(m)
for _ in range(3): eng.features(data)
torch.cuda.synchronize()
for _ in range(5):
    t0 = time.perf_counter(); f = eng.features(data); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'features: CPU enqueue {1e3*(t1-t0):.1f} ms, until GPU done {1e3*(t2-t0):.1f} ms')
import os
print('cpus', os.cpu_count(), 'loadavg', os.getloadavg())
