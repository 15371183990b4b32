#!/bin/bash
# Sweep of the masked side stream's CU count on three workloads (GPU box): tools/sweep_side_cus.sh
for wl in timit_3x250_blstm_H125 timit_3x500_blstm_H250 lvcsr_4x512_blstm_8000; do
  for cu in 48 64 80 96 128; do
    for r in 1 2; do
      CN_SIDE_CUS=$cu CN_BENCH_MIN_SECONDS=0.3 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-driver-leg 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', 'side_cus=$cu', round(d['value']), round(d['ms_per_step'],4))"
    done
  done
done
