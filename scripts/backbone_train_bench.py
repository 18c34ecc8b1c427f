"""Training-mode forward + backward of the backbone at the training batch shape (bs x 3 x 256 x 256): wall time per phase."""
import sys, time, torch
sys.argv = ['x']; sys.path.insert(0, '.')
from vpho_amd.assets import synthetic_assets
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import synth_state_dict
from vpho_amd.train_blocks import FPNTrain
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
import os
bs = int(os.environ.get('BS', '32'))
sd = synth_state_dict(vpho_net(synthetic_assets(0)), seed=1)
net = FPNTrain(sd, 'feature_extractor', 'cuda')
x = torch.randn(bs, 3, 256, 256, device='cuda')
A = torch.randn(bs, 64, 64, 256, device='cuda'); B = torch.randn(bs, 64, 64, 256, device='cuda')
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ph, po = net.forward(x); torch.cuda.synchronize(); t1 = time.perf_counter()
    g = net.backward(A, B); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'bs={bs}: forward(train) {1e3*(t1-t0):.1f} ms, backward {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f} ms -> {bs/(t2-t0):.0f} images/s; peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB')
