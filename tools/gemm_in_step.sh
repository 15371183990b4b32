#!/bin/bash
# Device time of every N-wide product (gemm_nt*) inside one headline step, from a rocprofv3 kernel trace of bench.py:
#   bash tools/gemm_in_step.sh <tag> [VAR=value ...]        (GPU box; environment switches of the library after the tag)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gis_$tag -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/gis_$tag.log 2>&1
python tools/timeline.py $(ls gpurun_out/gis_$tag/*/*kernel_trace.csv | tail -1) softmax_mcc_bwd_kernel | grep -i "gemm_nt\|step total" | cut -c1-110
