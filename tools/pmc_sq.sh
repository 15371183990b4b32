cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d gpurun_out/sq/$tag -o sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline-pass --tmin 300 --tmax 300 > gpurun_out/sq_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/sq/*/*/*counter_collection.csv') + glob.glob('gpurun_out/sq/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'lstm_' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print('   %-28s %14.0f (avg of %d launches)' % (c, sum(v) / len(v), len(v)))
PY
