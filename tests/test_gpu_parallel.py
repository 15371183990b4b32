"""-m gpu: the flat parameter arena aliased as a torch tensor (what the RCCL all-reduce of bench.py sums),
and the bench entry point under torch.distributed with one rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import net_desc, random_sequences, random_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_aliases_weight_updates(pkg):
    import torch
    rng = np.random.RandomState(4)
    layers = net_desc(5, [("blstm", 8), ("lstm", 6)], 4)
    weights = random_weights(layers, rng, 0.3)
    xs, ts = random_sequences(rng, [7, 5, 3], 5, C=4)
    frac = pkg.make_fraction(xs, ts, 3)
    with pkg.NeuralNetwork(layers, weights, 3, 7) as net:
        net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass()
        wptr, gptr, dptr, count = net.param_arena()
        net.join(); net.synchronize()
        g = torch.as_tensor(pkg.parallel.DeviceArray(gptr, count), device="cuda")
        w = torch.as_tensor(pkg.parallel.DeviceArray(wptr, count), device="cuda")
        flat_g = pkg.parallel.flatten_updates([l.weight_updates() for l in net.trainable_layers()])
        flat_w = pkg.parallel.flatten_updates([l.weights() for l in net.trainable_layers()])
        assert np.array_equal(g.cpu().numpy(), flat_g) and np.array_equal(w.cpu().numpy(), flat_w)
        g.mul_(2.0)                                   # what a 2-rank all-reduce of equal shards would do
        torch.cuda.synchronize()
        for l, ref in zip(net.trainable_layers(), [l.weight_updates() for l in net.trainable_layers()]):
            pass
        assert np.allclose(pkg.parallel.flatten_updates([l.weight_updates() for l in net.trainable_layers()]), 2 * flat_g)


def test_bench_under_torchrun_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--parallel-sequences", "8", "--tmin", "20", "--tmax", "30", "--no-cpu-baseline", "--no-roofline-pass", "--no-driver-leg"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "frames/s" and d["scaling"] == "weak"


def test_bench_per_layer_allreduce_path_matches_plain_path():
    """The overlapped per-layer all-reduce path of bench.py -- the LIBRARY's RCCL communicator (cn_comm_init, one
    cn_allreduce_grads per layer right behind its backward pass) -- forced on with one rank where the reduction is the
    identity: same accumulated error and the same weight movement as the plain path."""
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--parallel-sequences", "8",
            "--tmin", "20", "--tmax", "30", "--no-cpu-baseline", "--no-roofline-pass", "--no-driver-leg"]
    sums = {}
    for mode in ("plain", "overlap"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CN_BENCH_MIN_SECONDS="0")
        if mode == "overlap":
            env["CN_BENCH_FORCE_ALLREDUCE"] = "1"
        out = subprocess.run(base + ["--master-port", str(29650 + os.getpid() % 200 + (mode == "overlap"))] + args,
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        sums[mode] = d["check"]
    assert sums["overlap"]["allreduce"].startswith("per-layer") and "library RCCL" in sums["overlap"]["allreduce"]
    assert sums["plain"]["allreduce"] == "none"
    assert abs(sums["overlap"]["error_sum"] - sums["plain"]["error_sum"]) <= 1e-3 * abs(sums["plain"]["error_sum"])
    assert abs(sums["overlap"]["update_l2"] - sums["plain"]["update_l2"]) <= 1e-3 * sums["plain"]["update_l2"]


def test_bench_gpus_flag_without_launcher():
    """`python bench.py --gpus N` with no launcher starts N ranks itself (before any GPU call); on a box with fewer GPUs
    than ranks it must FAIL loudly instead of reporting a one-GPU number as n_gpus = N (missing #1)."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-roofline-pass", "--no-driver-leg"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]
    assert "needs %d GPUs" % n in out.stderr + out.stdout
    # and with N = 1 the flag path is the plain single-process run
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--parallel-sequences", "8",
                          "--tmin", "20", "--tmax", "30", "--no-cpu-baseline", "--no-roofline-pass", "--no-driver-leg", "--no-also"],
                         capture_output=True, text=True, timeout=900, env=dict(env, CN_BENCH_MIN_SECONDS="0"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_library_communicator_one_rank(pkg):
    """cn_comm_unique_id / cn_comm_init / cn_allreduce_grads / cn_loss_read_global through the C ABI with a one-rank RCCL
    communicator (the only world size a one-GPU box can run on RCCL): the per-layer exchange, the flat exchange and the
    plain path train to the same weights, and the global loss equals the local one.  Without a communicator the calls
    fail with CN_ERR_STATE."""
    rng = np.random.RandomState(61)
    P, C, PS, T = 20, 11, 12, 40
    layers = net_desc(P, [("blstm", 64), ("blstm", 64)], C)
    weights = random_weights(layers, rng, 0.1)
    xs, ts = random_sequences(rng, [T - (i % 5) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    out = {}
    for mode in ("plain", "layer", "flat"):
        with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_F32) as net:
            if mode == "plain":
                with pytest.raises(pkg.CurrenntHipError, match="no communicator"):
                    net.allreduce_grads(None)
                assert net.comm_info() == (0, 0)
            else:
                net.comm_init(net.comm_unique_id(), 0, 1)
                assert net.comm_info() == (0, 1)
            for _ in range(3):
                net.load_sequences(frac); net.compute_forward_pass(); net.loss_accumulate()
                if mode == "layer":
                    net.compute_backward_pass_dp()
                else:
                    net.compute_backward_pass()
                    if mode == "flat":
                        net.allreduce_grads(None)
                net.update_weights_fused(1e-3, 0.9)
            loss = net.loss_read_global(reset=False) if mode != "plain" else None
            local = net.loss_read()
            if loss is not None:
                assert loss == local
            out[mode] = (np.concatenate([l.weights() for l in net.trainable_layers()]), local)
    for mode in ("layer", "flat"):
        assert np.array_equal(out[mode][0], out["plain"][0]) or np.abs(out[mode][0] - out["plain"][0]).max() < 1e-6
        assert out[mode][1] == out["plain"][1]


def test_per_layer_learning_rate_in_fused_update(pkg, orc):
    """A layer with a JSON "learningRate" of its own (TrainableLayer.cu:58, SteepestDescentOptimizer.cu:78-80): the fused
    update (cn_sgd_update_all) and the per-layer update train to the same weights as the oracle."""
    rng = np.random.RandomState(62)
    P, C, PS = 6, 4, 3
    layers = net_desc(P, [("blstm", 12), ("lstm", 8)], C)
    layers[2]["learningRate"] = 5e-2
    weights = random_weights(layers, rng, 0.3)
    xs, ts = random_sequences(rng, [9, 8, 5], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    ref = orc.OracleNetwork(layers, weights, PS, 9)
    for _ in range(3):
        ref.load_sequences(frac); ref.compute_forward_pass(); ref.compute_backward_pass(); ref.update_weights(1e-2, 0.9)
    for fused in (False, True):
        with pkg.NeuralNetwork(layers, weights, PS, 9) as net:
            for _ in range(3):
                net.load_sequences(frac); net.compute_forward_pass(); net.compute_backward_pass()
                if fused:
                    net.update_weights_fused(1e-2, 0.9)
                else:
                    net.update_weights(1e-2, 0.9)
            for lay in net.trainable_layers():
                assert np.abs(lay.weights() - ref.layer(lay.name).weights).max() < 5e-6, (fused, lay.name)


def _run_p2p_ranks(tmp_path, world, mode, extra_env):
    env = dict(os.environ, CN_COMM_BACKEND="p2p", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_rank.py"), str(r), str(world), str(tmp_path), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][-3000:])
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


@pytest.mark.parametrize("world,oneshot_max", [(2, None), (2, 0), (2, 3000), (3, None), (3, 0), (3, 3000), (8, None), (8, 3000)])
def test_p2p_exchange_sums_in_rank_order(pkg, tmp_path, world, oneshot_max):
    """CN_COMM_BACKEND=p2p (cn_comm_p2p.hip) with `world` live ranks (2, 3 and the full 8) on one device, through the C ABI: every exchange -- per layer
    (buckets of different sizes in a row, both staging halves, the first half reused) and the flat arena in one bucket -- must
    leave ((g0 + g1) + g2) in float32 on EVERY rank, bit for bit: the same numbers added
    in rank order, whether a workgroup sums a piece itself (one shot) or receives it from the rank that owns the slice
    (reduce-scatter + all-gather: CN_P2P_ONESHOT_MAX=0 sends every bucket that way, 3000 floats splits the layers between the
    two forms).  The global loss is the sum of the ranks' in rank order."""
    env = {"CN_COMM_IPC_TIMEOUT": "60"}
    if oneshot_max is not None:
        env["CN_P2P_ONESHOT_MAX"] = str(oneshot_max)
    res = _run_p2p_ranks(tmp_path, world, "sum", env)
    n_layers = 3
    for k in range(3):
        want = res[0]["local_%d" % k].copy()
        for r in range(1, world):
            want = want + res[r]["local_%d" % k]                 # float32, rank order
        assert np.abs(want).max() > 0
        for r in range(world):
            got = res[r]["reduced_%d" % k]
            assert np.array_equal(got, want), (k, r, np.abs(got - want).max(), np.argwhere(got != want)[:4])
    for r in range(world):
        assert int(res[r]["exchanges"]) == 2 * n_layers + 1
        assert np.array_equal(res[r]["loss"], res[0]["loss"])


def test_p2p_exchange_times_out_when_a_peer_stays_away(pkg, tmp_path):
    """A rank that never enters an exchange: the others' kernels give up after CN_COMM_IPC_TIMEOUT seconds of the device clock,
    mark the communicator failed (their own region and every peer's), and the next cn_loss_read_global raises CN_ERR_COMM naming
    the rank -- no host thread hangs, no kernel spins for ever."""
    res = _run_p2p_ranks(tmp_path, 3, "absent", {"CN_COMM_IPC_TIMEOUT": "2"})
    for r in (0, 1):
        msg = str(res[r]["raised"])
        assert "p2p communicator: rank %d of 3 waited more than 2 s" % r in msg, msg
        # the exchange that gave up did not leave a partial sum behind: its whole gradient is NaN, and the update that follows
        # raises instead of applying it (cn_sgd_update_all reads the host-mapped failure word)
        assert np.isnan(res[r]["poisoned"]).all()
        assert "p2p communicator: rank %d of 3 waited more than 2 s" % r in str(res[r]["update_raised"]), res[r]["update_raised"]


def test_p2p_soak_eight_ranks_alternating_forms(pkg, tmp_path):
    """8 live ranks, 900 rounds = 2 100 exchanges: per-layer buckets back to back in both orders (CN_P2P_ONESHOT_MAX=3000 sends the
    two LSTM layers through reduce-scatter + all-gather and the softmax layer through the one-shot form, so the forms alternate
    inside a round) and every third round the flat arena in one bucket; the gradients are small integers that depend on
    (element, round, rank), so EVERY element of EVERY exchange has one exact answer.  A flag that overtakes a staging store, a
    staging half reused too early or a slot index off by one shows here as a wrong element (tests/p2p_rank.py::soak)."""
    res = _run_p2p_ranks(tmp_path, 8, "soak:900", {"CN_COMM_IPC_TIMEOUT": "60", "CN_P2P_ONESHOT_MAX": "3000"})
    rounds = 900
    per_round = [3, 3, 1]
    want = sum(per_round[k % 3] for k in range(rounds))
    assert want >= 2000
    for r in range(8):
        assert int(res[r]["exchanges"]) == want


def test_p2p_first_contact_failure_falls_back_to_rccl(pkg, tmp_path):
    """cn_comm_init with CN_COMM_BACKEND=p2p runs a self-check through every peer mapping (both forms of the exchange on a bucket
    with a known answer).  A rank whose check fails (here: injected) makes EVERY rank abandon the backend: the communicator that
    comes up is RCCL, with a message, and it reduces.  (World 1: RCCL takes one rank per device, the box has one.)"""
    res = _run_p2p_ranks(tmp_path, 1, "failover", {"CN_COMM_IPC_TIMEOUT": "30", "CN_P2P_SELFCHECK_FAIL": "0"})
    assert str(res[0]["backend"]) == "rccl"
    assert np.array_equal(res[0]["before"], res[0]["after"]) and np.abs(res[0]["before"]).max() > 0


@pytest.mark.parametrize("backend", ["gloo", "ipc", "p2p", "p2p-two-phase"])
@pytest.mark.parametrize("flat", [False, True])
def test_bench_two_ranks_on_one_gpu_over_gloo(pkg, flat, backend):
    """bench.py with WORLD_SIZE = 2 on a one-GPU box: CN_BENCH_BACKEND=gloo lets both ranks share the device and
    reduces through the host, so the world > 1 control flow (per-layer exchange from the communication stream, or
    the flat exchange; barrier; max-over-ranks timing) runs with real sums.  Replicas must end bit-identical.
    backend "ipc": the exchange is the LIBRARY's own (cn_comm_init, one cn_allreduce_grads per layer on the communication stream,
    cn_loss_read_global) on its test backend for ranks that share a device (cn_comm_ipc.cpp).
    backend "p2p": the library's NATIVE exchange (CN_COMM_BACKEND=p2p, cn_comm_p2p.hip: one stream-ordered kernel per bucket that
    stages, signals, sums in rank order and acknowledges through the peers' hipIpc-mapped regions; no host barrier after the
    first bucket) -- here with both ranks on one device; "p2p-two-phase" forces its reduce-scatter + all-gather form, which a
    world of two would never pick."""
    two_phase = backend == "p2p-two-phase"
    backend = backend.split("-")[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 250 + int(flat) + 2 * ["gloo", "ipc", "p2p"].index(backend) + 6 * two_phase), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--parallel-sequences", "8", "--tmin", "20", "--tmax", "30", "--no-cpu-baseline", "--no-roofline-pass", "--no-driver-leg"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CN_BENCH_BACKEND=backend, CN_BENCH_MIN_SECONDS="0")   # one repetition
    if flat:
        env["CN_BENCH_FLAT_ALLREDUCE"] = "1"
    if two_phase:
        env["CN_P2P_FORCE_TWO_PHASE"] = "1"
    if backend == "p2p":
        env["CN_COMM_IPC_TIMEOUT"] = "60"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "dp2 over sequences"
    assert d["check"]["replicas_identical"] is True and np.isfinite(d["check"]["error_sum"])
    assert d["check"]["allreduce"].startswith("flat" if flat else "per-layer, overlapped")
    assert {"gloo": "test double", "ipc": "ipc TEST backend", "p2p": "p2p backend"}[backend] in d["check"]["allreduce"]
    assert d["exchange"]["backend"] == ("torch.distributed gloo (test double)" if backend == "gloo" else backend) and d["exchange"]["one_rank_per_gpu"] is False
    # the same five steps in ONE process on the union of the two ranks' fractions (16 sequences per fraction): gradients
    # are sums over patterns, so the data-parallel run must move the weights the same way.  An exchange that read a
    # layer's gradient before it was complete, or an update that did not wait for the exchange, shows up here.
    sys.path.insert(0, ROOT)
    import bench
    wl = bench.WORKLOADS["timit_3x250_blstm_H125"]
    layers = bench.net_desc(wl["P"], wl["hidden"], wl["C"])
    rngs = [np.random.RandomState(1234 + r) for r in range(2)]
    union = []
    for _ in range(4):
        seqs = [bench.synth_sequences(rng, 8, wl["P"], wl["C"], 20, 30) for rng in rngs]
        union.append(pkg.make_fraction(seqs[0][0] + seqs[1][0], seqs[0][1] + seqs[1][1], 16))
    with pkg.NeuralNetwork(layers, bench.make_weights(layers, 1234), 16, 30, precision=pkg.PREC_BF16) as net:
        w0 = np.concatenate([l.weights() for l in net.trainable_layers()]).astype(np.float64)
        for i in (0, 1, 2, 3, 0):
            net.load_sequences(union[i]); net.compute_forward_pass(); net.compute_backward_pass()
            net.update_weights_fused(1e-4, 0.9)
        upd = np.concatenate([l.weights() for l in net.trainable_layers()]).astype(np.float64) - w0
    assert abs(d["check"]["update_l2"] - np.linalg.norm(upd)) < 2e-3 * np.linalg.norm(upd), (d["check"], np.linalg.norm(upd))
    assert abs(d["check"]["update_sum"] - upd.sum()) < 2e-3 * np.abs(upd).sum()


def test_per_layer_exchange_is_ordered_between_gradient_and_update(pkg):
    """Stream ordering of compute_backward_pass_allreduce (cn_layer_join_stream + a communication stream), checked
    with a stand-in for the collective that DOUBLES each layer's weightUpdates on the communication stream (what a
    2-rank all-reduce of equal shards does): three momentum-SGD steps must equal the plain path run with twice the
    learning rate.  A reduction that started before the layer's gradient GEMMs had finished, or an update that did
    not wait for it, changes the trained weights by tens of percent."""
    import torch

    class Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    class FakeDist:
        class ReduceOp:
            SUM = 0

        @staticmethod
        def all_reduce(t, op=None, async_op=False):
            t.mul_(2.0)                                   # on the current (= communication) stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return Work(ev)

    rng = np.random.RandomState(21)
    P, C, PS, T = 39, 40, 24, 120
    layers = net_desc(P, [("blstm", 250), ("blstm", 250)], C)
    weights = random_weights(layers, rng, 0.08)
    xs, ts = random_sequences(rng, [T - (i % 7) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    trained = {}
    for mode, lr in (("plain", 2e-4), ("exchange", 1e-4)):
        with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_BF16) as net:
            for _ in range(3):
                net.load_sequences(frac); net.compute_forward_pass()
                if mode == "plain":
                    net.compute_backward_pass()
                else:
                    net.compute_backward_pass_allreduce(FakeDist, torch)
                net.update_weights_fused(lr, 0.9)
            net.synchronize()
            trained[mode] = np.concatenate([l.weights() for l in net.trainable_layers()])
    start = np.concatenate([np.concatenate([np.asarray(weights[l["name"]][k], np.float32) for k in ("input", "bias", "internal")])
                            for l in layers if l["name"] in weights])
    moved = np.linalg.norm(trained["plain"] - start)
    assert moved > 0
    assert np.linalg.norm(trained["exchange"] - trained["plain"]) < 1e-3 * moved


@pytest.mark.parametrize("net_kind", ["blstm250x2_s2", "blstm1024x2_cluster"])
def test_library_exchange_is_ordered_between_gradient_and_update(pkg, monkeypatch, net_kind):
    """The same ordering check for the LIBRARY's own exchange (compute_backward_pass_dp = cn_layer_backward + cn_allreduce_grads
    per layer on the library's communication stream, then the fused update): a one-rank communicator is bound and
    option comm_test_double (CN_COMM_TEST_DOUBLE) replaces ncclAllReduce by a kernel that doubles the layer's weightUpdates on that stream.  Three
    momentum-SGD steps must equal the plain path at twice the learning rate -- on the headline kernels (hand-written s2
    loops) and on a network of 8-CU cluster kernels (blstm1024: spin-wait hand-off between CUs running beside the
    communication stream's work and the gradient GEMMs of the side stream)."""
    rng = np.random.RandomState(23)
    if net_kind == "blstm250x2_s2":
        P, C, PS, T, sizes, scale = 39, 40, 24, 120, [250, 250], 0.08
    else:
        P, C, PS, T, sizes, scale = 39, 40, 8, 60, [1024, 1024], 0.03
    layers = net_desc(P, [("blstm", n) for n in sizes], C)
    weights = random_weights(layers, rng, scale)
    xs, ts = random_sequences(rng, [T - (i % 7) for i in range(PS)], P, C=C)
    frac = pkg.make_fraction(xs, ts, PS)
    trained, kernels = {}, {}
    for mode, lr in (("plain", 2e-4), ("exchange", 1e-4)):
        with pkg.NeuralNetwork(layers, weights, PS, T, precision=pkg.PREC_BF16) as net:
            if mode == "exchange":
                net.set_option("comm_test_double", 1)      # (the context exists: its options no longer follow the environment)
                net.comm_init(net.comm_unique_id(), 0, 1)
            for _ in range(3):
                net.load_sequences(frac); net.compute_forward_pass()
                if mode == "plain":
                    net.compute_backward_pass()
                else:
                    net.compute_backward_pass_dp()
                net.update_weights_fused(lr, 0.9)
            net.synchronize()
            kernels[mode] = net.recurrent_kernel(True)
            trained[mode] = np.concatenate([l.weights() for l in net.trainable_layers()])
    want = "lstm_bwd_s2_asm_kernel" if net_kind == "blstm250x2_s2" else "lstm_bwd_cluster_kernel<0,512,64,1>"
    assert kernels["plain"] == want and kernels["exchange"] == want
    start = np.concatenate([np.concatenate([np.asarray(weights[l["name"]][k], np.float32) for k in ("input", "bias", "internal")])
                            for l in layers if l["name"] in weights])
    moved = np.linalg.norm(trained["plain"] - start)
    assert moved > 0
    assert np.linalg.norm(trained["exchange"] - trained["plain"]) < 1e-3 * moved


def test_rccl_allreduce_on_aliased_arena():
    """RCCL (torch.distributed backend nccl, one rank) all-reduces the aliased weightUpdates arena in place."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_one_rank.py")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


@pytest.mark.parametrize("post", ["multiclass_classification", "sse"])
def test_resident_fraction_load_matches_host_load(pkg, post):
    """cn_fraction_load_resident (one re-layout kernel, device pointers) against cn_fraction_load (host buffers):
    same outputs, error and gradients, with parallel_sequences = 6 padded to 8 on the device."""
    import torch
    rng = np.random.RandomState(12)
    P, C, PS = 5, 4, 6
    layers = net_desc(P, [("blstm", 12)], C, post=post)
    weights = random_weights(layers, rng, 0.4)
    if post == "sse":
        xs, ts = random_sequences(rng, [9, 7, 4, 6, 9], P, L=C)
        frac = pkg.make_fraction(xs, ts, PS, classification=False)
    else:
        xs, ts = random_sequences(rng, [9, 7, 4, 6, 9], P, C=C)
        frac = pkg.make_fraction(xs, ts, PS)
    res = {}
    for mode in ("host", "resident"):
        with pkg.NeuralNetwork(layers, weights, PS, frac["T"], precision=pkg.PREC_F32) as net:
            if mode == "host":
                net.load_sequences(frac)
            else:
                keep = {k: torch.from_numpy(np.ascontiguousarray(frac[k])).cuda() for k in ("inputs", "patTypes", "targetClasses", "targets") if k in frac}
                d = {"T": frac["T"], "Tmin": frac["Tmin"], "numSeqs": frac["numSeqs"], "inputPatternSize": P, "outputPatternSize": C}
                d.update({k: v.data_ptr() for k, v in keep.items()})
                net.load_sequences_resident(d)
            net.compute_forward_pass()
            e, c = net.error_and_correct()
            net.compute_backward_pass()
            res[mode] = (net.outputs().copy(), e, c, [l.weight_updates() for l in net.trainable_layers()])
    assert np.array_equal(res["host"][0], res["resident"][0]) and res["host"][1] == res["resident"][1] and res["host"][2] == res["resident"][2]
    for a, b in zip(res["host"][3], res["resident"][3]):
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize("post", ["multiclass_classification", "sse", "binary_classification"])
def test_prefetched_resident_fraction_equals_plain_load(pkg, post):
    """cn_fraction_prefetch_resident: the next fraction is re-laid out on the side stream beside the backward pass and the load
    only exchanges buffers.  Six training steps over three fractions of different lengths (the alternates hold stale rows of other
    fractions) must give the same outputs and errors per step and the same weights as loads without the hint; a load of ANOTHER
    fraction than the announced one discards the hint; a second announcement while one is in flight is CN_ERR_STATE."""
    import torch
    rng = np.random.RandomState(31)
    P, PS = 5, 6
    C = 1 if post == "binary_classification" else 4
    layers = net_desc(P, [("blstm", 12), ("lstm", 8)], C, post=post)
    if post == "binary_classification":
        layers[-2]["type"] = "feedforward_logistic"
    weights = random_weights(layers, rng, 0.4)
    fracs, keep = [], []
    for lens in ([9, 7, 4, 6, 9], [5, 5, 3], [8, 8, 8, 8, 2, 1]):
        if post == "sse":
            xs, ts = random_sequences(rng, lens, P, L=C)
            fr = pkg.make_fraction(xs, ts, PS, classification=False)
        else:
            xs, ts = random_sequences(rng, lens, P, C=max(C, 2))
            fr = pkg.make_fraction(xs, ts, PS)
        dev = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).cuda() for k in ("inputs", "patTypes", "targetClasses", "targets") if k in fr and fr[k] is not None}
        keep.append(dev)
        d = {"T": fr["T"], "Tmin": fr["Tmin"], "numSeqs": fr["numSeqs"], "inputPatternSize": P, "outputPatternSize": C}
        d.update({k: v.data_ptr() for k, v in dev.items()})
        fracs.append(d)
    order = [0, 1, 2, 0, 2, 1]
    res = {}
    for mode in ("plain", "prefetch", "wrong_hint"):
        with pkg.NeuralNetwork(layers, weights, PS, 9, precision=pkg.PREC_F32) as net:
            trace = []
            for i, k in enumerate(order):
                net.load_sequences_resident(fracs[k])
                net.compute_forward_pass()
                e, c = net.error_and_correct()
                trace.append((net.outputs().copy(), e, c))
                if mode != "plain" and i + 1 < len(order):
                    nxt = order[i + 1] if mode == "prefetch" else (order[i + 1] + 1) % 3
                    net.prefetch_sequences_resident(fracs[nxt])
                net.compute_backward_pass()
                if mode == "prefetch" and i == 0:
                    with pytest.raises(pkg.CurrenntHipError, match="has not been consumed"):
                        net.prefetch_sequences_resident(fracs[0])
                net.update_weights_fused(1e-2, 0.9)
            res[mode] = (trace, np.concatenate([l.weights() for l in net.trainable_layers()]))
    for mode in ("prefetch", "wrong_hint"):
        for (o, e, c), (o0, e0, c0) in zip(res[mode][0], res["plain"][0]):
            assert np.abs(o - o0).max() < 1e-5 and abs(e - e0) <= 1e-4 * max(1.0, abs(e0)) and c == c0, mode
        assert np.abs(res[mode][1] - res["plain"][1]).max() < 1e-5, mode


@pytest.mark.parametrize("post", ["multiclass_classification", "sse", "binary_classification"])
def test_prefetched_host_fraction_equals_plain_load(pkg, post):
    """cn_fraction_prefetch (host buffers; the reference's loader thread works one fraction ahead, DataSet.cpp:202-240): the next
    fraction is packed and uploaded at once, re-laid out on the side stream beside the backward pass, and the load only exchanges
    buffers.  Same protocol as the resident test: six training steps over three fractions of different lengths give the same
    outputs and errors per step and the same weights as loads without the hint; a load of ANOTHER fraction than the announced
    one discards the hint (and its staging area is reused later without harm); a second announcement while one is in flight is
    CN_ERR_STATE; a resident load does not take a host hint."""
    rng = np.random.RandomState(41)
    P, PS = 5, 6
    C = 1 if post == "binary_classification" else 4
    layers = net_desc(P, [("blstm", 12), ("lstm", 8)], C, post=post)
    if post == "binary_classification":
        layers[-2]["type"] = "feedforward_logistic"
    weights = random_weights(layers, rng, 0.4)
    fracs = []
    for lens in ([9, 7, 4, 6, 9], [5, 5, 3], [8, 8, 8, 8, 2, 1]):
        if post == "sse":
            xs, ts = random_sequences(rng, lens, P, L=C)
            fr = pkg.make_fraction(xs, ts, PS, classification=False)
        else:
            xs, ts = random_sequences(rng, lens, P, C=max(C, 2))
            fr = pkg.make_fraction(xs, ts, PS)
        # (the hint is matched by address: the arrays handed over must be the ones the load will hand over)
        fr["inputs"] = np.ascontiguousarray(fr["inputs"], np.float32); fr["patTypes"] = np.ascontiguousarray(fr["patTypes"], np.int8)
        if fr.get("targetClasses") is not None: fr["targetClasses"] = np.ascontiguousarray(fr["targetClasses"], np.int32)
        if fr.get("targets") is not None: fr["targets"] = np.ascontiguousarray(fr["targets"], np.float32)
        fracs.append(fr)
    order = [0, 1, 2, 0, 2, 1, 1, 0]
    res = {}
    for mode in ("plain", "prefetch", "wrong_hint"):
        with pkg.NeuralNetwork(layers, weights, PS, 9, precision=pkg.PREC_F32) as net:
            trace = []
            for i, k in enumerate(order):
                net.load_sequences(fracs[k])
                net.compute_forward_pass()
                e, c = net.error_and_correct()
                trace.append((net.outputs().copy(), e, c))
                if mode != "plain" and i + 1 < len(order):
                    nxt = order[i + 1] if mode == "prefetch" else (order[i + 1] + 1) % 3
                    net.prefetch_sequences(fracs[nxt])
                net.compute_backward_pass()
                if mode == "prefetch" and i == 0:
                    with pytest.raises(pkg.CurrenntHipError, match="has not been consumed"):
                        net.prefetch_sequences(fracs[0])
                net.update_weights_fused(1e-2, 0.9)
            # every announced fraction but none of the wrong ones was taken over by its load
            assert net.prefetch_hits() == {"plain": 0, "prefetch": len(order) - 1, "wrong_hint": 0}[mode], (mode, net.prefetch_hits())
            res[mode] = (trace, np.concatenate([l.weights() for l in net.trainable_layers()]))
    for mode in ("prefetch", "wrong_hint"):
        for (o, e, c), (o0, e0, c0) in zip(res[mode][0], res["plain"][0]):
            assert np.abs(o - o0).max() < 1e-5 and abs(e - e0) <= 1e-4 * max(1.0, abs(e0)) and c == c0, mode
        assert np.abs(res[mode][1] - res["plain"][1]).max() < 1e-5, mode


def test_prefetch_hint_without_side_stream_work_is_dropped(pkg):
    """A network whose only trainable layer computes its gradient on the main stream (nothing runs beside the backward pass): the
    announced fraction is never re-laid out ahead, the load that follows takes the ordinary path, results equal the unhinted run."""
    import torch
    rng = np.random.RandomState(32)
    P, C, PS = 7, 5, 4
    layers = [{"name": "input", "type": "input", "size": P},
              {"name": "output", "type": "softmax", "size": C, "bias": 1.0},
              {"name": "postoutput", "type": "multiclass_classification", "size": C}]
    weights = random_weights(layers, rng, 0.4)
    fracs, keep = [], []
    for lens in ([6, 5, 3], [4, 4, 4, 2]):
        xs, ts = random_sequences(rng, lens, P, C=C)
        fr = pkg.make_fraction(xs, ts, PS)
        dev = {k: torch.from_numpy(np.ascontiguousarray(fr[k])).cuda() for k in ("inputs", "patTypes", "targetClasses")}
        keep.append(dev)
        d = {"T": fr["T"], "Tmin": fr["Tmin"], "numSeqs": fr["numSeqs"], "inputPatternSize": P, "outputPatternSize": C}
        d.update({k: v.data_ptr() for k, v in dev.items()})
        fracs.append(d)
    res = {}
    for mode in ("plain", "hint"):
        with pkg.NeuralNetwork(layers, weights, PS, 6, precision=pkg.PREC_F32) as net:
            errs = []
            for i in range(4):
                net.load_sequences_resident(fracs[i % 2]); net.compute_forward_pass()
                errs.append(net.error_and_correct())
                if mode == "hint":
                    net.prefetch_sequences_resident(fracs[(i + 1) % 2])
                net.compute_backward_pass(); net.update_weights_fused(1e-2, 0.9)
            res[mode] = (errs, np.concatenate([l.weights() for l in net.trainable_layers()]))
    assert res["hint"][0] == res["plain"][0]
    assert np.abs(res["hint"][1] - res["plain"][1]).max() < 1e-6
