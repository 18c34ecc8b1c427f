#!/usr/bin/env python
"""HBM rate of the training step's element-wise / column-reduction kernels on two activation sizes (HIP events, 10 repeats)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from vpho_amd import ops


def timeit(f, n=10):
    f(); f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for shape in ((64, 64, 64, 256), (64, 64, 64, 64), (64, 32, 32, 512), (64, 16, 16, 1024), (64, 8, 8, 2048)):
    x, dy = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda')
    C = shape[-1]
    g, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    mb = x.numel() * 4 / 1e6
    y, saved = ops.bn_train_forward(x, g, b, rm, rv, slope=0.01)
    rows = [('colsum (1 read)', lambda: ops.colsum(x.view(-1, C)), 1),
            ('bn_train_forward (stats 1 read + apply 1 read 1 write)', lambda: ops.bn_train_forward(x, g, b, rm, rv, slope=0.01), 3),
            ('bn_train_backward (reduce 2 reads + apply 2 reads 1 write)', lambda: ops.bn_train_backward(x, dy, g, saved), 5),
            ('lrelu_bwd (2 reads 1 write)', lambda: ops.lrelu_bwd(dy, y, 0.01), 3),
            ('add_lrelu (2 reads 1 write)', lambda: ops.add_lrelu(x, dy), 3)]
    for name, f, passes in rows:
        ms = timeit(f)
        print(f'{str(shape):22s} {mb:7.1f} MB  {name:58s} {ms*1e3:8.1f} us  {passes * mb / ms / 1e3:6.2f} TB/s', flush=True)
