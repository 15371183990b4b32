for v in "CN_NO_OVERLAP=0" "CN_NO_OVERLAP=1"; do
  echo "$v"
  env $v python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.0f frames/s %.3f ms  %s' % (d['value'], d['ms_per_step'], d['roofline']['note'][-100:]))"
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CN_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/noov -o p -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline-pass > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/noov/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'lstm_bwd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('bwd launches (us):', ' '.join('%.0f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[-12:]))
PY
