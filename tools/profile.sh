#!/bin/bash
# One gpurun call: bench line, rocprofv3 kernel stats and the two HBM PMC passes of the default bench command.
# usage (GPU box): bash tools/profile.sh <tag>;  then locally: python tools/make_profile.py <tag> gpurun_out/prof_<tag>_stats gpurun_out/prof_<tag>_fetch gpurun_out/prof_<tag>_write --bench-log gpurun_out/prof_<tag>_bench.log
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_${tag}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/prof_${tag}_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/prof_${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/prof_${tag}_write.log 2>&1
tail -1 gpurun_out/prof_${tag}_bench.log | cut -c1-300
ls gpurun_out/prof_${tag}_stats/*/ | head
