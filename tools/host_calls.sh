#!/bin/bash
# Run ON THE GPU BOX: the HIP runtime calls of ONE steady-state dvs_pipeline_step at a batch size (rocprofv3 --hip-trace of the C++ host program)
#   usage: tools/host_calls.sh <tag> <B> [steps]
set -e -o pipefail
TAG=$1; B=$2; STEPS=${3:-60}
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
timeout -k 10 240 rocprofv3 --hip-trace --output-format csv -d $OUT/ht -o ht -- /tmp/pipeline_stream /tmp/frames_$B.bin $B 720 1280 2000 2 $STEPS 0 /tmp/out_t.bin 1 0 > $OUT/run.log 2>&1
find $OUT/ht -name "*hip_api_trace.csv" -exec cp {} $OUT/hip_api_trace.csv \;
rm -rf $OUT/ht
python3 - $OUT/hip_api_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if not r['Function'].startswith('__hip') and r['Function'] not in ('hipGetLastError', 'hipSetDevice', 'hipGetDevice')]
# last 3 steps' worth: print the tail before the final synchronisations
idx = [i for i, r in enumerate(rows) if r['Function'] == 'hipLaunchKernel']
tail = rows[idx[-60]:idx[-1] + 1]
t0 = int(tail[0]['Start_Timestamp'])
for r in tail:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f}  {r['Function']}")
PY
