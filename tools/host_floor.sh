#!/bin/bash
# Run ON THE GPU BOX: the C++ host program on SMALL frames (320x240) at the batch sizes of the three schedules — the GPU work shrinks ~12x, the
# HIP calls per step stay, so the step time that is left is the host's enqueue floor of each schedule.
#   usage: tools/host_floor.sh <tag> "<B list>" <steps>
set -e -o pipefail
TAG=${1:-hostfloor}; BS=${2:-"1 4 8 16 64"}; STEPS=${3:-300}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
LIBDIR=$R/dynamic-visual-slam_amd/lib
g++ -std=c++17 -O2 -I$R/include $R/tests/cpp/pipeline_stream.cpp -o /tmp/pipeline_stream -L$LIBDIR -ldvslam_hip -Wl,-rpath,$LIBDIR -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lpthread
for B in $BS; do
  for WH in ${SIZES:-"320 240" "1280 720"}; do
    set -- ${WH/x/ }
    python3 - $R $B $1 $2 <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/dynamic-visual-slam_amd")
from dvslam_amd import synth
B, W, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
np.concatenate([np.stack([synth.make_frame(i, W, H, seed=1234 + 101 * g) for i in range(B)]) for g in range(2)]).tofile(f"/tmp/frames_{B}_{W}.bin")
PY
    echo "== B=$B ${1}x${2}" | tee -a $OUT/host.log
    /tmp/pipeline_stream /tmp/frames_${B}_$1.bin $B $2 $1 2000 2 $STEPS 0 /tmp/out_$B.bin 1 0 | tee -a $OUT/host.log
  done
done
