#!/bin/bash
# A/B two builds on the workloads whose recurrent kernels are the multi-CU cluster kernels (reading B, LVCSR, long utterances)
A=$1; B=$2
for wl in timit_3x500_blstm_H250 lvcsr_4x512_blstm_8000 longutt_5x1024_blstm; do
  for lib in "$A" "$B"; do
    CURRENNT_HIP_LIB=$lib CN_BENCH_MIN_SECONDS=0.2 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-28s %-34s %10.0f frames/s %8.3f ms  %s' % ('$wl', os.path.basename('$lib'), d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','')[-100:]))"
  done
done
