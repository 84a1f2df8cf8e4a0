mkdir -p gpurun_out/r2q
python tools/ba_prof.py; python tools/ba_prof_batched.py
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/b -o b -- python3 $R/tools/ba_prof_batched.py > $R/gpurun_out/r2q/b.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/s -o s -- python3 $R/tools/ba_prof.py > $R/gpurun_out/r2q/s.log 2>&1
head -4 $R/gpurun_out/r2q/b/b_kernel_stats.csv | cut -c1-200; head -4 $R/gpurun_out/r2q/s/s_kernel_stats.csv | cut -c1-200
