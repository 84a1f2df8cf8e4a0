#!/bin/bash
# Run ON THE GPU BOX: the C++ host program (tests/cpp/pipeline_stream.cpp) at a few batch sizes — host time per step() and wall time
# per step — and the HIP API statistics of one of them (rocprofv3 --hip-trace --stats: which runtime calls the step's enqueue is made of).
#   usage: tools/host_probe.sh <tag> "<B list>" <steps> [B to trace] ["<lanes list>"]   (lanes: 0 = by batch size, 1 = two-stream pipeline, 2..4)
set -e -o pipefail
TAG=${1:-hostprobe}; BS=${2:-"1 8 64"}; STEPS=${3:-300}; TRACE_B=${4:-}; LANES=${5:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
LIBDIR=$R/dynamic-visual-slam_amd/lib
g++ -std=c++17 -O2 -I$R/include $R/tests/cpp/pipeline_stream.cpp -o /tmp/pipeline_stream -L$LIBDIR -ldvslam_hip -Wl,-rpath,$LIBDIR -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lpthread
for B in $BS; do
  python3 - $R $B <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/dynamic-visual-slam_amd")
from dvslam_amd import synth
B = int(sys.argv[2])
np.concatenate([np.stack([synth.make_frame(i, 1280, 720, seed=1234 + 101 * g) for i in range(B)]) for g in range(2)]).tofile(f"/tmp/frames_{B}.bin")
PY
  for LN in $LANES; do
    /tmp/pipeline_stream /tmp/frames_$B.bin $B 720 1280 2000 2 $STEPS 0 /tmp/out_$B.bin 1 $LN | tee -a $OUT/host.log
    /tmp/pipeline_stream /tmp/frames_$B.bin $B 720 1280 2000 2 $STEPS 0 /tmp/out_$B.bin 1 $LN | tee -a $OUT/host.log
  done
done
if [ -n "$TRACE_B" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 240 rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hip_$TRACE_B -o hip -- /tmp/pipeline_stream /tmp/frames_$TRACE_B.bin $TRACE_B 720 1280 2000 2 $STEPS 0 /tmp/out_t.bin 1 ${LANES##* } > $OUT/hip_$TRACE_B.log 2>&1
  find $OUT/hip_$TRACE_B -name "*hip_api_stats.csv" -exec cp {} $OUT/hip_api_stats_B$TRACE_B.csv \;
fi
