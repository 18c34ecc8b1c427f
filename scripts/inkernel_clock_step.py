"""In-kernel clock of the chip-filling kernels INSIDE the predict step (the companion of scripts/inkernel_clock.py, which launches each
kernel back to back on its own): the stamps build runs bench.py's timed loop for >= 2 s -- pipelined (several batches in flight, kernels of
different streams sharing the chip) or sequential (--pipeline 1) -- and the stamp buffers are read afterwards.  A slot holds the record of
the last workgroup that had that block index, whichever launch it belonged to; every record carries its own main-loop span in shader cycles
and in 100-MHz ticks, so clock = d memtime / d memrealtime x 100 MHz per workgroup, median over the records of each kernel family.

    bash scripts/build_stamps.sh          (here)
    VPHO_HIP_LIB=scripts/_ab/libvpho_hip_stamps.so python scripts/inkernel_clock_step.py [--pipeline 1]      (GPU box)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert 'stamps' in os.environ.get('VPHO_HIP_LIB', ''), 'run with VPHO_HIP_LIB=scripts/_ab/libvpho_hip_stamps.so (scripts/build_stamps.sh)'
extra = sys.argv[1:]
sys.argv = ['bench.py', '--steps', '80', '--warmup', '5', '--no_cpu_baseline', '--no_opt_in', '--no_kernel_timing'] + extra
import numpy as np
import torch
import bench

bench.main()
from vpho_amd import ops

WORDS, SLOTS = 10, 65536
print(f'# in-kernel clock inside the step: bench.py {" ".join(sys.argv[1:])}   ({torch.cuda.get_device_name(0)}, diagnostic build {os.environ["VPHO_HIP_LIB"]})')
for name, label in (('conv', 'conv_igemm (one-tile + persistent)'), ('wino', 'conv_winograd'), ('head', 'score_head')):
    fn = getattr(ops.lib, f'vpho_diag_stamps_{name}')
    fn.restype = C.c_int
    buf = np.zeros(WORDS * SLOTS, dtype=np.uint64)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data_as(C.c_void_p), C.c_int(SLOTS), C.c_int(0)) == 0
    b = buf.reshape(SLOTS, WORDS)
    b = b[b[:, 7] > 0]
    loop_real = (b[:, 8] >> np.uint64(32)).astype(np.float64)
    cyc = (b[:, 3] - b[:, 2]).astype(np.float64)
    ok = (loop_real >= 500) & (cyc > 0)                               # main loops of >= 5 us: a 100-MHz tick is 1 % of the shortest span kept
    clk = cyc[ok] / loop_real[ok] * 0.1
    print(f'{label}: {int(ok.sum())} workgroup records with a main loop >= 5 us; in-kernel clock median {np.median(clk):.3f} GHz '
          f'(p05 {np.percentile(clk, 5):.3f}, p95 {np.percentile(clk, 95):.3f}); main loop median {np.median(loop_real[ok]) / 100:.1f} us', flush=True)
