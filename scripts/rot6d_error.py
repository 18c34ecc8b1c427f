"""rot6d -> axis-angle of the sampler's hypotheses (VPHO.py:306-331): the HIP kernel and the reference arithmetic (oracle/rotations.py in
torch fp32) against float64, on hypothesis-like inputs (unit-ish 6-D columns with noise); rms / max |axis-angle error| in radians."""
import os, sys
sys.argv = sys.argv[:1]; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vpho_amd import ops
from oracle import rotations as R
g = torch.Generator().manual_seed(1)
n = 200000
for name, scale in (('small rotations (fingers)', 0.3), ('medium', 1.0), ('large (wrist)', 2.5)):
    aa = torch.randn(n, 3, generator=g, dtype=torch.float64) * scale / 3 ** 0.5
    m = R.axis_angle_to_matrix(aa)
    x6 = (m[:, :2, :].reshape(n, 6) * (1 + 0.05 * torch.randn(n, 1, generator=g, dtype=torch.float64)) + 0.01 * torch.randn(n, 6, generator=g, dtype=torch.float64)).float()
    t64 = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(x6.double()))
    t32 = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(x6))
    hip = ops.rot6d_to_axis_angle(x6.cuda().contiguous(), 1).cpu()
    eh, eo = (hip.double() - t64).norm(dim=-1), (t32.double() - t64).norm(dim=-1)
    print(f'{name:28s} HIP rms {eh.pow(2).mean().sqrt():.2e} max {eh.max():.2e}   torch fp32 rms {eo.pow(2).mean().sqrt():.2e} max {eo.max():.2e}')
