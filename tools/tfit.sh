#!/bin/bash
# per-step and per-launch cost of the recurrent kernels: bench at two fixed sequence lengths, fit a line.
# usage: tools/tfit.sh lib.so [lib2.so ...]
for lib in "$@"; do
  for T in 150 300; do
    CURRENNT_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg --tmin $T --tmax $T 2>&1 | tail -1 > /tmp/tfit_$T.json
  done
  python - "$lib" <<'PY'
import json, re, sys, os
r = {}
for T in (150, 300):
    d = json.load(open('/tmp/tfit_%d.json' % T))
    note = d['roofline']['note']
    r[T] = {k: float(v) for k, v in re.findall(r'(\w+)=([\d.]+)', note)}
out = os.path.basename(sys.argv[1])
for k in ('rec_fwd', 'rec_bwd'):
    a, b = r[150][k] / 30 * 1e3, r[300][k] / 30 * 1e3          # us per launch (10 steps x 3 layers)
    step = (b - a) / 150
    print('%-34s %s: %.3f us/step, %.1f us per-launch overhead (T=150: %.1f us, T=300: %.1f us)' % (out, k, step, a - 150 * step, a, b))
PY
done
