#!/bin/bash
# DIAGNOSTIC build for the in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): the three chip-filling kernels compiled with
# -DVPHO_CLOCK_STAMPS (s_memtime / s_memrealtime once in front of and once behind the main loop, written to a buffer of their own), every
# other object of the product build -> scripts/_ab/libvpho_hip_stamps.so.  Never the product library.  Then on the GPU box:
#   VPHO_HIP_LIB=scripts/_ab/libvpho_hip_stamps.so python scripts/inkernel_clock.py
set -e
cd "$(dirname "$0")/.."
python -m vpho_amd.build > /dev/null
mkdir -p scripts/_ab
objs=""
for name in conv_igemm conv_winograd score_ode; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVPHO_CLOCK_STAMPS=1 -x hip -c vpho_amd/csrc/$name.hip -o scripts/_ab/${name}_stamps.o &
done
wait
objs=$(ls vpho_amd/csrc/_obj/*.o | grep -v -e "/conv_igemm.hip.o" -e "/conv_winograd.hip.o" -e "/score_ode.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libvpho_hip_stamps.so $objs scripts/_ab/conv_igemm_stamps.o scripts/_ab/conv_winograd_stamps.o scripts/_ab/score_ode_stamps.o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o scripts/_ab/mfma_f32_clock scripts/microbench/mfma_f32_peak_random.hip
ls -la scripts/_ab/libvpho_hip_stamps.so scripts/_ab/mfma_f32_clock
