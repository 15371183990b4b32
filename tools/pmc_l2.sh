# L2 behaviour of the GEMM kernels on the shapes of tools/probe/gemm_bench: fabric-side bytes (FETCH_SIZE, doubled on gfx950) and hit rate
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/l2
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d gpurun_out/l2/$tag -o l2 -- tools/probe/gemm_bench > gpurun_out/l2_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/l2/*/*/*counter_collection.csv') + glob.glob('gpurun_out/l2/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60] + ' grid=' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
        if 'gemm_nt' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    hit = c.get('TCC_HIT_sum', 0) / max(1.0, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0))
    print('%-80s fabric read %8.1f MB (2 x FETCH_SIZE)   L2 hit rate %.2f   requests %.0f' % (k, 2 * c.get('FETCH_SIZE', 0) / 1024, hit, c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)))
PY
