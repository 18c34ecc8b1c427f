"""Isolated score evaluations (time_embed + pose encoder + score head, one stream, nothing else on the GPU) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o pe -- python3 scripts/pe_bench.py"""
import sys, torch
import os; sys.argv = ['x']; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vpho_amd.model.VPHO import vpho_net
from vpho_amd.synth import bench_state_dict
from vpho_amd.assets import synthetic_assets
from vpho_amd import ops
a = synthetic_assets(0); m = vpho_net(a); sd = bench_state_dict(m, 1)
dev = 'cuda'
for name, D in (('hand', 96), ('obj', 9)):
    net = ops.ScoreNet(sd, f'denoiser_{name}', dev)
    bs, S = int(os.environ.get('PE_BS', 64)), int(os.environ.get('PE_S', 100))          # cfg4: PE_BS=128 PE_S=256
    feat = torch.randn(bs, 1024, device=dev) * 0.3
    x = torch.randn(bs * S, D, device=dev)
    for _ in range(40): net.score(feat, x, 0.3, S)
    torch.cuda.synchronize()
