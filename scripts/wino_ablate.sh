#!/bin/bash
# Timing ablations of the Winograd kernel: builds scripts/_ab/libvpho_hip_w<mask>.so (conv_winograd.hip compiled with -DWINO_ABLATE=<mask>,
# every other object of the product build).  Run on the GPU box:  for m in 0 1 2 4 8 16 31; do VPHO_HIP_LIB=scripts/_ab/libvpho_hip_w$m.so python scripts/wino_bench.py; done
set -e
cd "$(dirname "$0")/.."
python -m vpho_amd.build > /dev/null
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DWINO_ABLATE=$m -x hip -c vpho_amd/csrc/conv_winograd.hip -o scripts/_ab/wino_$m.o
  objs=$(ls vpho_amd/csrc/_obj/*.o | grep -v conv_winograd)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libvpho_hip_w$m.so $objs scripts/_ab/wino_$m.o
done
ls -la scripts/_ab/*.so
