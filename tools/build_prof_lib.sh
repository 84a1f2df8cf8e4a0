#!/bin/bash
# lib/libdvslam_hip_prof.so: the product library with -DDVS_QT_PROF in orb.hip (time stamps of the level-0 quad-tree; tools/qt_phase_profile.py)
set -e
cd "$(dirname "$0")/../dynamic-visual-slam_amd"
make -j8 > /tmp/make_prof.log 2>&1
mkdir -p build/prof
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -DDVS_QT_PROF -c csrc/orb.hip -o build/prof/orb.o > /tmp/make_prof2.log 2>&1
OBJ=$(ls build/*.o | grep -v "build/orb.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libdvslam_hip_prof.so build/prof/orb.o $OBJ -ldl
echo built
