#!/bin/bash
# Timing ablations of one kernel file: builds scripts/_ab/libvpho_hip_<name>_<mask>.so (csrc/<name>.hip compiled with -D<MACRO>=<mask>, every
# other object of the product build).  usage: scripts/kernel_ablate.sh mano FK_ABLATE 1 2 4 8
# then on the GPU box:  VPHO_HIP_LIB=scripts/_ab/libvpho_hip_mano_1.so python scripts/mano_bench.py
set -e
cd "$(dirname "$0")/.."
name=$1; macro=$2; shift 2
python -m vpho_amd.build > /dev/null
mkdir -p scripts/_ab
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -D$macro=$m -x hip -c vpho_amd/csrc/$name.hip -o scripts/_ab/${name}_$m.o
  objs=$(ls vpho_amd/csrc/_obj/*.o | grep -v "/$name.hip.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libvpho_hip_${name}_$m.so $objs scripts/_ab/${name}_$m.o
done
ls -la scripts/_ab/libvpho_hip_${name}_*.so
