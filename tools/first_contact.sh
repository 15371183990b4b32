#!/bin/bash
# First multi-GPU contact, one command for whoever gets a node with more than one MI355X (no 1 -> 8 curve is asked for here; the
# driver measures that).  Runs bench.py at 2 and at 8 ranks (or "$@": the rank counts), once over RCCL and once over the
# library's native exchange (CN_BENCH_BACKEND=p2p: cn_comm_p2p.hip, with its first-contact self-check in cn_comm_init and the
# RCCL fall-back), and prints per run what the LIBRARY reports about the exchange and whether the replicas stayed bit-identical.
# bench.py --gpus N starts its ranks as fresh child processes (nothing here has touched a GPU) under CN_BENCH_TIMEOUT; a rank that
# never reaches the rendezvous ends the run with a message instead of hanging it.
#   usage:  bash tools/first_contact.sh            # 2 and 8 ranks
#           bash tools/first_contact.sh 2 4        # other rank counts
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
export CN_BENCH_TIMEOUT=${CN_BENCH_TIMEOUT:-600} CN_BENCH_RENDEZVOUS_TIMEOUT=${CN_BENCH_RENDEZVOUS_TIMEOUT:-120}
counts=${@:-2 8}
have=$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
echo "devices on this node: $have"
rc=0
for n in $counts; do
  if [ "$have" -lt "$n" ]; then echo "== $n ranks: skipped (needs $n devices, one rank per GPU)"; continue; fi
  for backend in nccl p2p; do
    log=$(mktemp /tmp/first_contact_${n}_${backend}.XXXX)
    echo "== $n ranks, exchange backend $backend (log: $log)"
    CN_BENCH_BACKEND=$backend CN_P2P_VERBOSE=1 python3 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --no-driver-leg --no-also > $log 2> $log.err
    code=$?
    if [ $code -ne 0 ]; then echo "   FAILED with exit code $code"; tail -5 $log.err | sed 's/^/   /'; rc=1; continue; fi
    grep -h "first-contact\|falling back" $log.err | sed 's/^/   /'
    python3 - $log <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
ex, ck = d.get("exchange", {}), d.get("check", {})
print("   value %.0f %s on %d GPUs, %.3f ms per step" % (d["value"], d["unit"], d["n_gpus"], d["ms_per_step"]))
print("   exchange.backend=%s ranks_min=%s ranks_max=%s allreduce_ms_per_step=%s allreduces_per_step=%s one_rank_per_gpu=%s"
      % (ex.get("backend"), ex.get("ranks_min"), ex.get("ranks_max"), ex.get("allreduce_ms_per_step"), ex.get("allreduces_per_step"), ex.get("one_rank_per_gpu")))
print("   check.replicas_identical=%s" % ck.get("replicas_identical"))
if not ck.get("replicas_identical") or ex.get("ranks_min") != d["n_gpus"]:
    print("   *** NOT OK: replicas differ or the communicator does not span all ranks")
    sys.exit(1)
PY
    [ $? -ne 0 ] && rc=1
  done
done
exit $rc
