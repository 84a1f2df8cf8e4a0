#!/bin/bash
# Run ON THE GPU BOX: kernel trace (start / end / queue of every dispatch) of the C++ host program at one batch size and lane count
#   usage: tools/lane_trace.sh <tag> <B> <lanes> <steps>
set -e -o pipefail
TAG=$1; B=$2; LN=$3; STEPS=${4:-40}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
LIBDIR=$R/dynamic-visual-slam_amd/lib
g++ -std=c++17 -O2 -I$R/include $R/tests/cpp/pipeline_stream.cpp -o /tmp/pipeline_stream -L$LIBDIR -ldvslam_hip -Wl,-rpath,$LIBDIR -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lpthread
python3 - $R $B <<'PY'
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/dynamic-visual-slam_amd")
from dvslam_amd import synth
B = int(sys.argv[2])
np.concatenate([np.stack([synth.make_frame(i, 1280, 720, seed=1234 + 101 * g) for i in range(B)]) for g in range(2)]).tofile(f"/tmp/frames_{B}.bin")
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- /tmp/pipeline_stream /tmp/frames_$B.bin $B 720 1280 2000 2 $STEPS 0 /tmp/out_t.bin 1 $LN > $OUT/run.log 2>&1
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/kt
