#!/bin/bash
# several builds of libcurrennt_hip.so on ONE device, two interleaved rounds: value / ms per step / per-class device time
for round in 1 2; do
  for lib in "$@"; do
    CURRENNT_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
      python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-36s %10.0f frames/s %7.3f ms  %s' % (os.path.basename('$lib'), d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','')[-110:]))"
  done
done
