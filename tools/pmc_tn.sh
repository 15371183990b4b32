# Fabric-side read bytes (FETCH_SIZE, doubled on gfx950 per MI355X_MICROARCH.md) of the weight-gradient kernels on the large
# shapes of tools/probe/gemm_bench: the 256 x 256 LDS-DMA kernel (cn_gemm_tn_big.hip) against the 64 x 64 / 128 x 128 kernels
# (CN_NO_BIG_TN=1).  One gpurun call: bash tools/pmc_tn.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_tn
export GEMM_BENCH_FIRST=0 GEMM_BENCH_TN_FIRST=4
for mode in big small; do
  if [ $mode = small ]; then export CN_NO_BIG_TN=1; else unset CN_NO_BIG_TN; fi
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_tn/$mode -o tn -- tools/probe/gemm_bench > gpurun_out/pmc_tn_$mode.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for mode in ("big", "small"):
    acc = collections.OrderedDict()
    for f in sorted(glob.glob('gpurun_out/pmc_tn/%s/**/*counter_collection.csv' % mode, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'gemm_tn' not in r['Kernel_Name'] or r['Counter_Name'] != 'FETCH_SIZE':
                continue
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40] + ' grid=' + r.get('Grid_Size', r.get('Grid_Size_X', '?'))
            acc.setdefault(k, []).append(float(r['Counter_Value']))
    print("== %s tiles" % mode)
    for k, v in acc.items():
        print('%-64s launches %3d   fabric read %9.1f MB per launch (2 x FETCH_SIZE)' % (k, len(v), 2 * sum(v) / len(v) / 1024))
PY
