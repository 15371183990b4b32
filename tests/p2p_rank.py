"""One rank of tests/test_gpu_parallel.py::test_p2p_exchange_*: python p2p_rank.py <rank> <world> <dir> <mode>.
All ranks share device 0 (CN_COMM_BACKEND=p2p runs with shared devices).  Writes <dir>/rank<r>.npz:
  local_k / reduced_k: the weightUpdates arena before / after exchange k (per-layer exchanges, then one flat exchange).
mode "absent": the LAST rank stays away from the final exchange; the others report what cn_loss_read_global raised."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge                                   # noqa: E402
from helpers import net_desc, random_sequences, random_weights  # noqa: E402


def main():
    rank, world, d, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    pkg = ge.load_package()
    P, C, PS, T = 20, 11, 6, 24
    layers = net_desc(P, [("blstm", 64), ("blstm", 32)], C)
    weights = random_weights(layers, np.random.RandomState(5), 0.1)
    rng = np.random.RandomState(100 + rank)
    xs, ts = random_sequences(rng, [T - (i % 5) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    idfile = os.path.join(d, "id")
    out = {}
    with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_F32) as net:
        if rank == 0:
            uid = net.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                if time.time() - t0 > 60:
                    raise SystemExit("rank %d: no rendezvous id" % rank)
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        net.comm_init(uid, rank, world)
        assert net.comm_backend()[0] == "p2p", net.comm_backend()

        def arena():
            return np.concatenate([l.weight_updates().reshape(-1) for l in net.trainable_layers()])
        for k in range(3):                                       # the staging halves alternate; k = 2 reuses the first one
            net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate(); net.compute_backward_pass()
            out["local_%d" % k] = arena()
            if k < 2:
                net.allreduce_grads(net.trainable_layers())      # one exchange per layer, largest bucket NOT first
            else:
                net.allreduce_grads(None)                        # the flat arena in one bucket
            out["reduced_%d" % k] = arena()
        out["exchanges"] = np.array(net.comm_backend()[1])
        if mode == "absent":
            net.compute_backward_pass()
            if rank == world - 1:
                np.savez(os.path.join(d, "rank%d.npz" % rank), **out)
                time.sleep(6)                                    # (alive, but never in the exchange)
                os._exit(0)
            net.allreduce_grads(None)
            try:
                net.loss_read_global()
                out["raised"] = np.array("nothing")
            except pkg.CurrenntHipError as e:
                out["raised"] = np.array(str(e))
            np.savez(os.path.join(d, "rank%d.npz" % rank), **out)
            os._exit(0)                                          # (the communicator is dead: no collective teardown)
        err, correct = net.loss_read_global()
        out["loss"] = np.array([err, correct])
    np.savez(os.path.join(d, "rank%d.npz" % rank), **out)


if __name__ == "__main__":
    main()
