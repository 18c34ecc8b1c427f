#!/bin/bash
# A/B build: the chip-filling kernels with s_setprio <P> around their main loops (-DVPHO_MAINLOOP_PRIO=<P>), every other object of the
# product build -> scripts/_ab/libvpho_hip_prio<P>.so;  VPHO_HIP_LIB=scripts/_ab/libvpho_hip_prio<P>.so selects it.   bash scripts/build_prio.sh 1 2 3
set -e
cd "$(dirname "$0")/.."
python -m vpho_amd.build > /dev/null
mkdir -p scripts/_ab
for P in "$@"; do
  for name in conv_igemm conv_winograd score_ode; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVPHO_MAINLOOP_PRIO=$P -x hip -c vpho_amd/csrc/$name.hip -o scripts/_ab/${name}_prio$P.o &
  done
  wait
  objs=$(ls vpho_amd/csrc/_obj/*.o | grep -v -e "/conv_igemm.hip.o" -e "/conv_winograd.hip.o" -e "/score_ode.hip.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/_ab/libvpho_hip_prio$P.so $objs scripts/_ab/conv_igemm_prio$P.o scripts/_ab/conv_winograd_prio$P.o scripts/_ab/score_ode_prio$P.o
done
ls -la scripts/_ab/libvpho_hip_prio*.so
