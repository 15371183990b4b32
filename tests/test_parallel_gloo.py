"""world_size-2 gloo test of the data-parallel path: shard the sequences of a global
fraction over ranks, all-reduce(SUM) the flat weightUpdates, apply the same update everywhere.
The per-rank compute here is the CPU oracle (this test checks the sharding / reduction logic, which is
host code; the GPU kernels are covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import __graft_entry__ as ge
    return ge.load_package(), ge.load_oracle()


def _problem():
    from helpers import net_desc, random_sequences, random_weights
    rng = np.random.RandomState(21)
    P, C, = 4, 3
    layers = net_desc(P, [("blstm", 6)], C)
    weights = random_weights(layers, rng, 0.4)
    lens = [9, 3, 7, 5, 8, 2]
    xs, ts = random_sequences(rng, lens, P, C=C)
    return layers, weights, xs, ts


def _train_step(orc, pkg, layers, weights, xs, ts, PS):
    frac = pkg.make_fraction(xs, ts, PS)
    net = orc.OracleNetwork(layers, weights, PS, frac["T"])
    net.load_sequences(frac); net.compute_forward_pass()
    err, cor = net.calculate_error(), net.count_correct_classifications()
    net.compute_backward_pass()
    return net, err, cor


def _worker(rank, world, port, q):
    pkg, orc = _setup()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layers, weights, xs, ts = _problem()
    mx, mt = pkg.parallel.shard_sequences(xs, ts, world, rank)
    net, err, cor = _train_step(orc, pkg, layers, weights, mx, mt, len(mx))
    flat = torch.from_numpy(pkg.parallel.flatten_updates([l.weightUpdates for l in net.trainable_layers()]))
    flat = torch.cat([flat, torch.tensor([err, float(cor)], dtype=torch.float32)])   # scalars ride along
    pkg.parallel.allreduce_sum_(flat, dist)
    off = 0
    for l in net.trainable_layers():                # scatter the summed gradient back, then the same update on every rank
        n = l.weightUpdates.size
        l.weightUpdates[:] = flat[off:off + n].numpy(); off += (n + 3) // 4 * 4
    net.update_weights(1e-2, 0.9)
    q.put((rank, [l.weights.copy() for l in net.trainable_layers()], float(flat[-2]), float(flat[-1])))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices(pkg):
    assert pkg.parallel.shard_indices(7, 2, 0) == [0, 2, 4, 6] and pkg.parallel.shard_indices(7, 2, 1) == [1, 3, 5]
    a = pkg.parallel.flatten_updates([np.ones(5, np.float32), np.ones(4, np.float32)])
    assert a.size == 12 and a[5:8].tolist() == [0, 0, 0]


def test_two_rank_allreduce_equals_single_fraction(pkg, orc):
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the union fraction (order of sequences does not matter for the sums)
    layers, weights, xs, ts = _problem()
    net, err, cor = _train_step(orc, pkg, layers, weights, xs, ts, len(xs))
    net.update_weights(1e-2, 0.9)
    (r0, w0, e0, c0), (r1, w1, e1, c1) = res
    assert abs(e0 - err) < 1e-4 * err and int(c0) == cor and e0 == e1
    for a, b, l in zip(w0, w1, net.trainable_layers()):
        assert np.array_equal(a, b)                                   # replicas stay bit-identical
        assert np.abs(a - l.weights).max() < 1e-6                     # = the big fraction up to fp32 summation order


def test_launcher_ends_a_job_whose_ranks_never_arrive():
    """`python bench.py --gpus 2` starts its own ranks from a process that has touched no GPU; when the ranks never reach the
    rendezvous (here: they sleep, CN_BENCH_TEST_HANG) the launcher terminates the ranks' process group after
    CN_BENCH_TIMEOUT seconds and returns 124 instead of hanging the driver; nothing of the job is left behind."""
    import re, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CN_BENCH_TIMEOUT="10", CN_BENCH_TEST_HANG="all")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124 and time.time() - t0 < 60, (p.returncode, p.stderr[-500:])
    m = re.search(r"terminating its process group (\d+)", p.stderr)
    assert m, p.stderr[-500:]
    time.sleep(0.5)
    with pytest.raises(ProcessLookupError):
        os.killpg(int(m.group(1)), 0)
