import sys, time, gc, os
import torch
sys.argv = sys.argv[:1]; sys.path.insert(0, '.')
from vpho_amd.configs.args import cfg
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict, synth_batch
from vpho_amd.assets import synthetic_assets
cfg.sample_num, cfg.sampling_steps, cfg.topk_hand, cfg.topk_obj, cfg.sample_T0 = 100, 50, 30, 10, 0.65
a = synthetic_assets(0)
m = vpho_net(a); m.load_state_dict(synth_state_dict(m, 1)); m = m.cuda().eval()
data = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth_batch(64, a).items()}
mode = os.environ.get('MODE', 'default')
if mode == 'nogc': gc.disable()
for i in range(3): m(data, mode='predict')
torch.cuda.synchronize()
ts = []
for i in range(30):
    if mode == 'collect': gc.collect()
    t0 = time.perf_counter(); out = m(data, mode='predict'); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
st = torch.cuda.memory_stats()
print(mode, ' '.join(f'{t:.0f}' for t in ts), '| device allocs', st.get('num_device_alloc'), 'retries', st.get('num_alloc_retries'), 'reserved GB', st['reserved_bytes.all.peak'] / 1e9, 'gc counts', gc.get_count())
