#!/bin/bash
# like ab_env3.sh, with the per-class device times of the event-timed pass: tools/ab_env_note.sh VAR=value [workloads...]
v=$1; shift
wls=${@:-"timit_3x500_blstm_H250 lvcsr_4x512_blstm_8000"}
for wl in $wls; do
  for mode in base "$v"; do
    if [ "$mode" = base ]; then envs=""; else envs="$mode"; fi
    env $envs CN_BENCH_MIN_SECONDS=0.2 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-26s %-16s %9.0f fr/s %7.3f ms  %s' % ('$wl', '$mode', d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','').split('pass: ')[-1]))"
  done
done
