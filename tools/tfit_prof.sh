#!/bin/bash
# kernel durations from rocprofv3 (no event overhead) at two fixed sequence lengths -> per-step and per-launch cost
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for T in 150 300; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tfit_$T -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-pass --tmin $T --tmax $T > gpurun_out/tfit_$T.log 2>&1
done
python3 - <<'PY'
import csv, glob
r = {}
for T in (150, 300):
    f = glob.glob('gpurun_out/tfit_%d/**/*kernel_stats.csv' % T, recursive=True)[0]
    r[T] = {row['Name']: float(row['AverageNs']) / 1e3 for row in csv.DictReader(open(f))}
for k in r[150]:
    if k in r[300]:
        a, b = r[150][k], r[300][k]
        step = (b - a) / 150
        print('%-70s T150 %7.1f us  T300 %7.1f us  per-step %.3f us  intercept %6.1f us' % (k[:70], a, b, step, a - 150 * step))
PY
