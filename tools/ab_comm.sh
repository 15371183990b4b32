#!/bin/bash
# One rank with the gradient exchange forced on (CN_BENCH_FORCE_ALLREDUCE=1): what binding a communicator costs a step on ONE GPU,
# RCCL against the p2p backend (cn_comm_p2p.hip), and two ranks SHARING the device on p2p (a functional run: its rate is not a
# measurement).  The exchange's device time per step is exchange.allreduce_ms_per_step of the JSON line.
show='import sys,json; d=json.loads(sys.stdin.read()); e=d.get("exchange",{}); print("%-34s %10.0f frames/s %7.3f ms/step   exchange: %s, %s all-reduces/step, %s ms/step" % (sys.argv[1], d["value"], d["ms_per_step"], e.get("backend"), e.get("allreduces_per_step"), e.get("allreduce_ms_per_step")))'
for round in 1 2; do
  python bench.py --workload ${WL:-timit_3x250_blstm_H125} --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-driver-leg 2>/dev/null | grep '^{' | tail -1 | python -c "$show" "no communicator"
  for b in nccl p2p; do
    CN_BENCH_FLAT_ALLREDUCE=1 CN_BENCH_FORCE_ALLREDUCE=1 CN_BENCH_BACKEND=$b MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29620 + round)) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
      python bench.py --workload ${WL:-timit_3x250_blstm_H125} --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-driver-leg 2>/dev/null | grep '^{' | tail -1 | python -c "$show" "one rank, $b, flat arena"
    CN_BENCH_FORCE_ALLREDUCE=1 CN_BENCH_BACKEND=$b MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + round)) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 CN_P2P_VERBOSE=1 \
      python bench.py --workload ${WL:-timit_3x250_blstm_H125} --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-driver-leg 2>gpurun_out/ab_comm_$b.err | grep '^{' | tail -1 | python -c "$show" "one rank, $b"
  done
done
grep -h "p2p communicator" gpurun_out/ab_comm_p2p.err | head -2
CN_BENCH_BACKEND=p2p python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 2 --workload ${WL:-timit_3x250_blstm_H125} --steps 20 --warmup 5 \
  --no-cpu-baseline --no-also --no-driver-leg 2>/dev/null | grep '^{' | tail -1 | python -c "$show" "two ranks on one device, p2p"
