"""One rank of tests/test_gpu_parallel.py::test_p2p_exchange_*: python p2p_rank.py <rank> <world> <dir> <mode>.
All ranks share device 0 (CN_COMM_BACKEND=p2p runs with shared devices).  Writes <dir>/rank<r>.npz:
  local_k / reduced_k: the weightUpdates arena before / after exchange k (per-layer exchanges, then one flat exchange).
mode "absent": the LAST rank stays away from the final exchange; the others report what cn_loss_read_global raised, what the
  update that follows raised and what the gradient of the failed exchange holds.
mode "soak:<iters>": <iters> rounds of three per-layer exchanges back to back (bucket sizes that alternate between the one-shot and
  the two-phase form under CN_P2P_ONESHOT_MAX) on gradients with a known integer answer, every sum checked on the device.
mode "failover": CN_P2P_SELFCHECK_FAIL is set -- the communicator must come up as "rccl" and still reduce (world 1)."""
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
        if mode == "failover":
            assert net.comm_backend()[0] == "rccl", net.comm_backend()
            net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate(); net.compute_backward_pass()
            before = np.concatenate([l.weight_updates().reshape(-1) for l in net.trainable_layers()])
            net.allreduce_grads(net.trainable_layers())
            after = np.concatenate([l.weight_updates().reshape(-1) for l in net.trainable_layers()])
            np.savez(os.path.join(d, "rank%d.npz" % rank), before=before, after=after, backend=np.array(net.comm_backend()[0]))
            return
        assert net.comm_backend()[0] == "p2p", net.comm_backend()
        if mode.startswith("soak"):
            soak(pkg, net, rank, world, int(mode.split(":")[1]))
            np.savez(os.path.join(d, "rank%d.npz" % rank), exchanges=np.array(net.comm_backend()[1]))
            return

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
            out["poisoned"] = arena()                            # the exchange that failed left NaN, not a partial sum
            try:
                net.update_weights(1e-3, 0.9)                    # ... and no update goes on top of it
                out["update_raised"] = np.array("nothing")
            except pkg.CurrenntHipError as e:
                out["update_raised"] = np.array(str(e))
            np.savez(os.path.join(d, "rank%d.npz" % rank), **out)
            os._exit(0)                                          # (the communicator is dead: no collective teardown)
        err, correct = net.loss_read_global()
        out["loss"] = np.array([err, correct])
    np.savez(os.path.join(d, "rank%d.npz" % rank), **out)


def soak(pkg, net, rank, world, iters):
    """Gradients with small-integer entries that depend on (element, exchange, rank): the sum over the ranks is exact in float32
    whatever the order, so every element of every exchange is checked bit for bit."""
    import torch
    layers = net.trainable_layers()
    wptr, gptr, dptr, count = net.param_arena()
    g = torch.as_tensor(pkg.parallel.DeviceArray(gptr, count), device="cuda")
    idx = torch.arange(count, device="cuda", dtype=torch.int64)
    # where each layer's weightUpdates sit in the arena (4-float padding between layers stays out of the exchanges)
    spans, off = [], 0
    for l in layers:
        n = l.weight_updates().size
        spans.append((off, n)); off += n + (-n) % 4
    mask = torch.zeros(count, dtype=torch.bool, device="cuda")
    for o, n in spans:
        mask[o:o + n] = True

    def pattern(k, r):
        return (((idx * 7 + k * 13 + r * 31) % 257) - 128).to(torch.float32)
    for k in range(iters):
        g.copy_(pattern(k, rank))
        torch.cuda.synchronize()
        if k % 3 == 2:
            net.allreduce_grads(None)                            # the flat arena (padding included) in one bucket
            full = True
        else:
            net.allreduce_grads(layers if k % 2 == 0 else layers[::-1])   # three buckets back to back, both orders
            full = False
        net.synchronize()
        want = pattern(k, 0)
        for r in range(1, world):
            want = want + pattern(k, r)
        got = g.clone()
        ok = torch.equal(got, want) if full else torch.equal(got[mask], want[mask])
        if not ok:
            bad = torch.nonzero((got != want) & (mask if not full else torch.ones_like(mask)))[:5].reshape(-1).tolist()
            raise SystemExit("rank %d: exchange round %d: wrong sums at arena elements %s: got %s, want %s"
                             % (rank, k, bad, got[bad].tolist(), want[bad].tolist()))


if __name__ == "__main__":
    main()
