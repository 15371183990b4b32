for rep in 1 2; do
for n in 512 768 1024 1536; do
  CN_TN_BLOCKS=$n python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline-pass 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tn_blocks $n  %.0f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
