# sweep of an environment knob of libcurrennt_hip.so on the headline bench: bash tools/cus.sh VAR v1 v2 ...
var=$1; shift
for rep in 1 2; do
for n in "$@"; do
  env $var=$n python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$n  %.0f frames/s %.4f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['note'][-105:]))"
done; done
