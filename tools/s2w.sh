# A/B of the single-CU Hp = 256 forward loop against the 2-CU cluster kernel (reading B) + parity of the LVCSR stack in bf16
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -k "config3_lvcsr_softmax8000_bf16 and not x3" 2>&1 | tail -3
CN_NO_S2W_ASM=1 timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -k "config3_lvcsr_softmax8000_bf16 and not x3" 2>&1 | tail -3
for v in cluster asm cluster asm; do
  if [ $v = cluster ]; then export CN_NO_S2W=1; else unset CN_NO_S2W; fi
  timeout 300 python bench.py --workload timit_3x500_blstm_H250 --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-driver-leg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v %10.0f frames/s %7.3f ms  %s' % (d['value'], d['ms_per_step'], d.get('roofline',{}).get('note','')[-160:]))"
done
