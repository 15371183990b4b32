cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for T in 16 48; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tfs_$T -o p -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg --no-roofline-pass --tmin $T --tmax $T > gpurun_out/tfs_$T.log 2>&1
done
python3 - <<'PY'
import csv, glob
r = {}
for T in (16, 48):
    f = glob.glob('gpurun_out/tfs_%d/**/*kernel_stats.csv' % T, recursive=True)[0]
    r[T] = {row['Name']: float(row['AverageNs']) / 1e3 for row in csv.DictReader(open(f))}
for k in r[16]:
    if k in r[48] and 'lstm_' in k:
        a, b = r[16][k], r[48][k]
        step = (b - a) / 32
        print('%-60s T16 %6.1f us  T48 %6.1f us  per-step %.3f us  intercept %5.1f us' % (k[:60], a, b, step, a - 16 * step))
PY
