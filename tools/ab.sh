#!/bin/bash
# A/B two builds of libcurrennt_hip.so on ONE device: interleaved rounds, prints value / ms per step / class times
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for lib in "$A" "$B"; do
    CURRENNT_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg "$@" 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-40s %10.0f frames/s %7.3f ms  %s' % (os.path.basename('$lib'), d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','')[-110:]))"
  done
done
