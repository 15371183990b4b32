#!/bin/bash
# A/B an environment switch of the library on the headline workload, three interleaved rounds: tools/ab_headline.sh VAR=value
for round in 1 2 3; do
  for mode in base "$1"; do
    if [ "$mode" = base ]; then envs=""; else envs="$mode"; fi
    env $envs python bench.py --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); t=d['roofline']['note'].split('pass: ')[1]; print('%-18s %10.0f frames/s %8.4f ms  %s' % ('$mode', d['value'], d['ms_per_step'], t))"
  done
done
