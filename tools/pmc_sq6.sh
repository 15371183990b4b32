#!/bin/bash
# MFMA-busy evidence (north star: "evidenced by rocprof HBM GB/s and MFMA-busy"): SQ counters per kernel for the headline and
# the LVCSR workload, written as markdown to gpurun_out/<tag>_sq.md (copy to profiles/).  usage (GPU box): bash tools/pmc_sq6.sh r06
# One rocprofv3 pass per counter group and workload; --pmc with --kernel-trace only (no sys/hip trace domains).
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq6
G1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
G2="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
for wl in timit_3x250_blstm_H125 lvcsr_4x512_blstm_8000; do
  i=0
  for grp in "$G1" "$G2"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d gpurun_out/sq6/${wl}_g$i -o sq -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg > gpurun_out/sq6/${wl}_g$i.log 2>&1
  done
done
python3 tools/pmc_sq6.py gpurun_out/sq6 > gpurun_out/${tag}_sq.md
cat gpurun_out/${tag}_sq.md | head -60
