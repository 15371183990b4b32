#!/bin/bash
# several builds of libcurrennt_hip.so on ONE device, two interleaved rounds, on the workloads named in WLS (default: reading B and LVCSR)
WLS=${WLS:-"timit_3x500_blstm_H250 lvcsr_4x512_blstm_8000"}
for round in 1 2; do
  for wl in $WLS; do
    for lib in "$@"; do
      CURRENNT_HIP_LIB=$lib CN_BENCH_MIN_SECONDS=0.2 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | \
        python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print('%-24s %-30s %9.0f fr/s %7.3f ms  %s' % ('$wl'[:24], os.path.basename('$lib'), d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','').split('pass: ')[-1][:60]))"
    done
  done
done
