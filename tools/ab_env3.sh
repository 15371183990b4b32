#!/bin/bash
# A/B an environment switch of the library on the three cluster workloads, two interleaved rounds: tools/ab_env3.sh VAR=value
for round in 1 2; do
for wl in timit_3x500_blstm_H250 lvcsr_4x512_blstm_8000 longutt_5x1024_blstm; do
  for mode in base "$1"; do
    if [ "$mode" = base ]; then envs=""; else envs="$mode"; fi
    env $envs CN_BENCH_MIN_SECONDS=0.2 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-28s %-18s %10.0f frames/s %8.3f ms' % ('$wl', '$mode', d['value'], d['ms_per_step']))"
  done
done
done
