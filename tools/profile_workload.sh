#!/bin/bash
# rocprofv3 kernel stats of one of bench.py's other workloads (GPU box): bash tools/profile_workload.sh <tag> <workload>
# then locally: python tools/make_profile.py <tag> gpurun_out/prof_<tag>_stats --bench-log gpurun_out/prof_<tag>_bench.log
tag=$1; wl=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg > gpurun_out/prof_${tag}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/prof_${tag}_stats.log 2>&1
tail -1 gpurun_out/prof_${tag}_bench.log | cut -c1-200
