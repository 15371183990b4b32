for ps in 50 100 200 512 1024; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg --parallel-sequences $ps 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A  PS=$ps  %.2f M frames/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg --workload timit_1x128_lstm 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C1 PS=50  %.2f M frames/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg --workload timit_3x500_blstm_H250 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B  PS=50  %.2f M frames/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-pass --no-also --no-driver-leg --precision f32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A  PS=50 fp32 mode  %.2f M frames/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
