for rep in 1 2; do
for n in 0 32 48 64 80 96; do
  CN_SIDE_CUS=$n python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cus $n  %.0f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
