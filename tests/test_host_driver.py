"""The C++ host side (lstm-rnn_amd/host, binary lstm-rnn_amd/currennt_hip).

CPU part: NetCDF-3 reader + fraction packer + length sort against the Python packer on a synthetic
file.  GPU part (-m gpu): the `currennt`-style driver trains / forward-passes a network from the same
network.jsn + .nc inputs the reference would take, and its outputs match the oracle driven from Python."""
import json
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from helpers import net_desc, random_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip")


def write_nc(path, xs, ts, num_labels):
    """CURRENNT classification file (reference README:600-646)."""
    f = netcdf_file(path, "w")
    n = sum(len(x) for x in xs)
    f.createDimension("numSeqs", len(xs)); f.createDimension("numTimesteps", n)
    f.createDimension("inputPattSize", xs[0].shape[1]); f.createDimension("numLabels", num_labels)
    f.createDimension("maxSeqTagLength", 16)
    tags = f.createVariable("seqTags", "c", ("numSeqs", "maxSeqTagLength"))
    for i in range(len(xs)):
        t = ("dir/seq%03d.wav" % i).ljust(16, "\0")
        tags[i] = np.array(list(t), "c")
    v = f.createVariable("seqLengths", "i", ("numSeqs",)); v[:] = np.array([len(x) for x in xs], np.int32)
    v = f.createVariable("targetClasses", "i", ("numTimesteps",)); v[:] = np.concatenate(ts).astype(np.int32)
    v = f.createVariable("inputs", "f", ("numTimesteps", "inputPattSize")); v[:] = np.concatenate(xs).astype(np.float32)
    f.close()


def problem(tmp_path, lens=(11, 5, 9, 3, 14, 7, 8)):
    rng = np.random.RandomState(31)
    P, C = 6, 4
    layers = net_desc(P, [("blstm", 8), ("feedforward_tanh", 5)], C)
    weights = random_weights(layers, rng, 0.3)
    xs = [rng.randn(n, P).astype(np.float32) for n in lens]
    ts = [rng.randint(0, C, n).astype(np.int32) for n in lens]
    nc = str(tmp_path / "train.nc")
    write_nc(nc, xs, ts, C)
    net = str(tmp_path / "network.jsn")
    doc = {"layers": layers, "weights": {k: {a: np.asarray(b).tolist() for a, b in w.items()} for k, w in weights.items()}}
    json.dump(doc, open(net, "w"))
    return layers, weights, xs, ts, nc, net


def test_reader_and_packer_cpu(pkg, tmp_path):
    if not os.path.exists(BIN):
        import __graft_entry__ as ge
        ge.build()
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    out = subprocess.run([BIN, "--train", "true", "--train_file", nc, "--network", net, "--parallel_sequences", "3",
                          "--dump_fractions", "true"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l for l in out.stdout.splitlines() if l.startswith("FRACTION")]
    fracs = pkg.make_fractions(xs, ts, 3, sort_by_length=True)       # training sets are length-sorted, DataSet.cpp:603-605
    assert len(rows) == len(fracs) == 3
    for row, f in zip(rows, fracs):
        kv = dict(p.split("=") for p in row.split()[2:])
        assert int(kv["T"]) == f["T"] and int(kv["Tmin"]) == f["Tmin"] and int(kv["seqs"]) == f["numSeqs"]
        assert int(kv["none"]) == int((f["patTypes"] == 0).sum())
        assert abs(float(kv["sum_inputs"]) - float(f["inputs"].astype(np.float64).sum())) < 1e-3
        assert int(kv["sum_targets"]) == int(f["targetClasses"][f["targetClasses"] >= 0].sum())


def test_data_parallel_shards_cpu(pkg, tmp_path):
    """--gpus N sharding of the C++ DataSet (DataSet::setShard; SURVEY 8e "Partitioning"): a global fraction is
    world * parallel_sequences consecutive sequences of the length-sorted list, rank r packs r, r + world, ...  Checked
    against the Python mirror (parallel.shard_indices + make_fraction), incl. a rank whose share of the last global
    fraction is EMPTY (it gets an all-dummy fraction so that every rank makes the same number of collective calls)."""
    if not os.path.exists(BIN):
        import __graft_entry__ as ge
        ge.build()
    lens = (11, 5, 9, 3, 14, 7, 8, 12, 6)                 # 9 sequences, world 2 x PS 2 -> global fractions of 4, 4, 1
    layers, weights, xs, ts, nc, net = problem(tmp_path, lens)
    order = sorted(range(len(lens)), key=lambda i: lens[i])
    world, PS = 2, 2
    rows = {}
    for r in range(world):
        out = subprocess.run([BIN, "--train", "true", "--train_file", nc, "--network", net, "--parallel_sequences", str(PS),
                              "--dump_fractions", "true", "--dp_rank", str(r), "--dp_world", str(world)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stdout + out.stderr
        rows[r] = [dict(p.split("=") for p in l.split()[2:]) for l in out.stdout.splitlines() if l.startswith("FRACTION")]
    assert len(rows[0]) == len(rows[1]) == 3
    for k in range(3):
        glob = order[k * world * PS:(k + 1) * world * PS]
        for r in range(world):
            mine = [glob[i] for i in pkg.parallel.shard_indices(len(glob), world, r)]
            kv = rows[r][k]
            assert int(kv["seqs"]) == len(mine)
            if not mine:
                assert int(kv["T"]) == 1 and int(kv["Tmin"]) == 0 and int(kv["none"]) == PS and float(kv["sum_inputs"]) == 0.0
                continue
            f = pkg.make_fraction([xs[i] for i in mine], [ts[i] for i in mine], PS)
            assert int(kv["T"]) == f["T"] and int(kv["Tmin"]) == f["Tmin"] and int(kv["none"]) == int((f["patTypes"] == 0).sum())
            assert abs(float(kv["sum_inputs"]) - float(f["inputs"].astype(np.float64).sum())) < 1e-3
            assert int(kv["sum_targets"]) == int(f["targetClasses"][f["targetClasses"] >= 0].sum())


def test_context_splicing_output_lag_and_input_noise_cpu(pkg, tmp_path):
    """--input_left_context / --input_right_context / --output_time_lag (DataSet.cpp:302-305,346-397) against the
    Python packer, and --input_noise_sigma (DataSet.cpp:250-265): reproducible per seed, changes the inputs."""
    if not os.path.exists(BIN):
        import __graft_entry__ as ge
        ge.build()
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    base = [BIN, "--train", "true", "--train_file", nc, "--network", net, "--parallel_sequences", "3", "--dump_fractions", "true",
            "--random_seed", "7"]
    out = subprocess.run(base + ["--input_left_context", "2", "--input_right_context", "1", "--output_time_lag", "2"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l for l in out.stdout.splitlines() if l.startswith("FRACTION")]
    fracs = pkg.make_fractions(xs, ts, 3, sort_by_length=True, context_left=2, context_right=1, output_lag=2)
    assert len(rows) == len(fracs)
    for row, f in zip(rows, fracs):
        kv = dict(p.split("=") for p in row.split()[2:])
        assert f["inputs"].shape[1] == 6 * 4
        assert abs(float(kv["sum_inputs"]) - float(f["inputs"].astype(np.float64).sum())) < 1e-3
        assert int(kv["sum_targets"]) == int(f["targetClasses"][f["targetClasses"] >= 0].sum())
    clean = subprocess.run(base, capture_output=True, text=True, timeout=60).stdout
    noisy1 = subprocess.run(base + ["--input_noise_sigma", "0.5"], capture_output=True, text=True, timeout=60).stdout
    noisy2 = subprocess.run(base + ["--input_noise_sigma", "0.5"], capture_output=True, text=True, timeout=60).stdout
    get = lambda txt: [float(dict(p.split("=") for p in l.split()[2:])["sum_inputs"]) for l in txt.splitlines() if l.startswith("FRACTION")]
    assert get(noisy1) == get(noisy2) and get(noisy1) != get(clean)
    n_vals = sum(len(x) for x in xs) * 6
    assert abs(sum(get(noisy1)) - sum(get(clean))) < 6 * 0.5 * np.sqrt(n_vals)      # zero-mean noise of sigma 0.5


def test_driver_failure_exit_code(tmp_path):
    """Errors end as "FAILED: <msg>" with exit code 2 (main.cpp:492-495)."""
    if not os.path.exists(BIN):
        pytest.skip("driver not built")
    out = subprocess.run([BIN, "--train", "true", "--train_file", str(tmp_path / "missing.nc"), "--network", str(tmp_path / "missing.jsn")],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "FAILED:" in out.stdout


@pytest.mark.gpu
def test_driver_trains_like_oracle(pkg, orc, tmp_path):
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    trained = str(tmp_path / "trained.jsn")
    out = subprocess.run([BIN, "--train", "true", "--stochastic", "true", "--train_file", nc, "--network", net,
                          "--parallel_sequences", "3", "--max_epochs", "2", "--learning_rate", "1e-2", "--momentum", "0.9",
                          "--save_network", trained], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " Epoch | Duration |" in out.stdout and "Storing the trained network" in out.stdout
    got = json.load(open(trained))
    assert [l["name"] for l in got["layers"]] == [l["name"] for l in layers]
    # oracle: same epoch protocol (Optimizer.cu:37-104): sorted sequences, update after every fraction
    fracs = pkg.make_fractions(xs, ts, 3, sort_by_length=True)
    ref = orc.OracleNetwork(layers, weights, 3, max(len(x) for x in xs))
    for epoch in range(2):
        for f in fracs:
            ref.load_sequences(f); ref.compute_forward_pass(); ref.compute_backward_pass(); ref.update_weights(1e-2, 0.9)
    for lay in ref.trainable_layers():
        w = got["weights"][lay.name]
        flat = np.concatenate([np.asarray(w[k], np.float32) for k in ("input", "bias", "internal")])
        assert np.abs(flat - lay.weights).max() < 2e-5, lay.name


@pytest.mark.gpu
def test_driver_forward_pass_writers(pkg, orc, tmp_path):
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    csv = str(tmp_path / "ff.csv")
    out = subprocess.run([BIN, "--train", "false", "--ff_input_file", nc, "--network", net, "--parallel_sequences", "4",
                          "--ff_output_file", csv, "--ff_output_format", "single_csv", "--revert_std", "false"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = open(csv).read().strip().split("\n")
    assert len(rows) == len(xs)
    fracs = pkg.make_fractions(xs, ts, 4)                          # forward pass: file order, no sort
    ref = orc.OracleNetwork(layers, weights, 4, max(len(x) for x in xs))
    k = 0
    for f in fracs:
        ref.load_sequences(f); ref.compute_forward_pass()
        y = ref.outputs()
        for i, n in enumerate(f["seqLengths"]):
            cells = rows[k].split(";")
            assert cells[0] == "dir/seq%03d.wav" % k
            vals = np.array(cells[1:], np.float64).reshape(n, -1)
            assert np.abs(vals - y[:n, i, :]).max() < 1e-4        # csv carries 6 significant digits
            k += 1
    # HTK writer: big-endian header nSamples, period, bytes/frame, kind (main.cpp:446-459)
    hdir = str(tmp_path / "htk")
    out = subprocess.run([BIN, "--train", "false", "--ff_input_file", nc, "--network", net, "--parallel_sequences", "4",
                          "--ff_output_file", hdir, "--ff_output_format", "htk", "--revert_std", "false"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(os.path.join(hdir, "dir", "seq000.wav.htk"), "rb").read()
    n, period, bpf, kind = np.frombuffer(raw[:12], ">i4,>i4,>i2,>i2")[0]
    assert (n, period, bpf, kind) == (len(xs[0]), 100000, 4 * 4, 9)
    vals = np.frombuffer(raw[12:], ">f4").reshape(n, 4)
    ref.load_sequences(fracs[0]); ref.compute_forward_pass()
    assert np.abs(vals - ref.outputs()[:n, 0, :]).max() < 1e-5


@pytest.mark.gpu
def test_driver_autosave_continue_and_weight_noise(pkg, tmp_path):
    """--autosave writes <prefix>_epochNNN.autosave after every epoch and --continue resumes from it with the
    optimizer state (main.cpp:198-204,275-277,701-758): resuming the epoch-1 autosave of a 3-epoch run ends
    in the same network as the run itself.
    --weight_noise_sigma perturbs only the backward pass (Optimizer.cu:58-84)."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    common = [BIN, "--train", "true", "--stochastic", "true", "--train_file", nc, "--val_file", nc, "--network", net,
              "--parallel_sequences", "3", "--learning_rate", "1e-2", "--momentum", "0.9", "--random_seed", "3"]
    straight = str(tmp_path / "straight.jsn")
    prefix = str(tmp_path / "run")
    out = subprocess.run(common + ["--max_epochs", "3", "--autosave", "true", "--autosave_best", "true", "--autosave_prefix", prefix,
                                   "--save_network", straight], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    auto = prefix + "_epoch001.autosave"
    assert os.path.exists(auto) and os.path.exists(prefix + ".best.jsn")
    state = json.load(open(auto))
    assert state["optimizer_cur_epoch"] == 1 and state["optimizer_finished"] is False and "steepest_descent_optimizer_weight_deltas" in state and ";;;" in state["info_rows"]
    resumed = str(tmp_path / "resumed.jsn")
    out = subprocess.run([BIN, "--continue", auto, "--max_epochs", "3", "--autosave", "false", "--save_network", resumed],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Restoring state from" in out.stdout
    a, b = json.load(open(straight)), json.load(open(resumed))
    for name, w in a["weights"].items():
        for k in ("input", "bias", "internal"):
            d = np.abs(np.asarray(w[k], np.float64) - np.asarray(b["weights"][name][k], np.float64))
            assert d.size == 0 or d.max() < 2e-6, (name, k)      # the autosave text carries ~7 significant digits
    noisy = str(tmp_path / "noisy.jsn")
    out = subprocess.run(common + ["--max_epochs", "3", "--weight_noise_sigma", "0.01", "--save_network", noisy],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    c = json.load(open(noisy))
    diff = max(np.abs(np.asarray(a["weights"][n]["input"]) - np.asarray(c["weights"][n]["input"])).max() for n in a["weights"])
    assert 0 < diff < 0.05


def _weights_of(path):
    doc = json.load(open(path))
    return {n: np.concatenate([np.asarray(w[k], np.float64).reshape(-1) for k in ("input", "bias", "internal")]) for n, w in doc["weights"].items()}


@pytest.mark.gpu
@pytest.mark.parametrize("stochastic", ["true", "false"])
def test_driver_data_parallel_path_one_rank(pkg, tmp_path, stochastic):
    """`--gpus 1` through the whole data-parallel path of the C++ driver (CN_DP_FORCE=1: forked rank, RCCL rendezvous over the
    pipe, cn_comm_init, one cn_allreduce_grads per layer behind its backward pass -- or one flat exchange of the epoch
    sum in batch mode --, cn_loss_read_global): with one rank every reduction is the identity, so the run must end in the
    same network and print the same error columns as the plain single-process run.  (World sizes > 1 need as many GPUs:
    RCCL refuses two ranks on one device.)"""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    common = [BIN, "--train", "true", "--stochastic", stochastic, "--train_file", nc, "--val_file", nc, "--network", net, "--parallel_sequences", "3",
              "--max_epochs", "3", "--learning_rate", "1e-2", "--momentum", "0.9"]
    plain, dp = str(tmp_path / "plain.jsn"), str(tmp_path / "dp.jsn")
    a = subprocess.run(common + ["--save_network", plain], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run(common + ["--gpus", "1", "--save_network", dp], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, CN_TEST_HOOKS="1", CN_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert b.returncode == 0, b.stdout + b.stderr
    assert "Data-parallel training on 1 GPU" in b.stdout
    rows = lambda out: [l.split("|")[2:5] for l in out.splitlines() if l.strip()[:1].isdigit() and "|" in l]
    assert rows(a.stdout) == rows(b.stdout) and len(rows(a.stdout)) == 3
    wa, wb = _weights_of(plain), _weights_of(dp)
    for n in wa:
        assert np.abs(wa[n] - wb[n]).max() < 1e-6, n


@pytest.mark.gpu
def test_driver_batch_mode_matches_oracle(pkg, orc, tmp_path):
    """Batch (non `--stochastic`) training: the fractions' weightUpdates are summed over the epoch and ONE update follows
    (Optimizer.cu:72-85,95-97; SteepestDescentOptimizer.cu:67-94).  Against the oracle driven the same way."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    trained = str(tmp_path / "batch.jsn")
    out = subprocess.run([BIN, "--train", "true", "--stochastic", "false", "--train_file", nc, "--network", net, "--parallel_sequences", "3",
                          "--max_epochs", "3", "--learning_rate", "1e-2", "--momentum", "0.9", "--save_network", trained],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Started in batch training mode." in out.stdout
    fracs = pkg.make_fractions(xs, ts, 3, sort_by_length=True)
    ref = orc.OracleNetwork(layers, weights, 3, max(len(x) for x in xs))
    for epoch in range(3):
        acc = None
        for f in fracs:
            ref.load_sequences(f); ref.compute_forward_pass(); ref.compute_backward_pass()
            g = [l.weightUpdates.copy() for l in ref.trainable_layers()]
            acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        for l, a in zip(ref.trainable_layers(), acc):
            l.weightUpdates[:] = a
        ref.update_weights(1e-2, 0.9)
    got = _weights_of(trained)
    for lay in ref.trainable_layers():
        assert np.abs(got[lay.name] - lay.weights).max() < 2e-5, lay.name


@pytest.mark.gpu
def test_driver_csv_writer(pkg, orc, tmp_path):
    """--ff_output_format csv: one <tag without extension>.csv per sequence below --ff_output_file, one row per frame,
    ';' separated (main.cpp:392-430)."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    cdir = str(tmp_path / "csvout")
    out = subprocess.run([BIN, "--train", "false", "--ff_input_file", nc, "--network", net, "--parallel_sequences", "4",
                          "--ff_output_file", cdir, "--ff_output_format", "csv", "--revert_std", "false"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    fracs = pkg.make_fractions(xs, ts, 4)
    ref = orc.OracleNetwork(layers, weights, 4, max(len(x) for x in xs))
    k = 0
    for f in fracs:
        ref.load_sequences(f); ref.compute_forward_pass()
        y = ref.outputs()
        for i, n in enumerate(f["seqLengths"]):
            path = os.path.join(cdir, "dir", "seq%03d.csv" % k)
            assert os.path.exists(path), path
            rows = [r for r in open(path).read().strip().split("\n")]
            assert len(rows) == n
            vals = np.array([r.split(";") for r in rows], np.float64)
            assert vals.shape == (n, 4) and np.abs(vals - y[:n, i, :]).max() < 1e-4
            k += 1
    assert k == len(xs)


def _error_rows(out):
    """(classification error %, error per sequence) of the training column of every epoch row of the driver's table"""
    rows = []
    for l in out.splitlines():
        if l.strip()[:1].isdigit() and "|" in l:
            cells = [c.strip() for c in l.split("|")]
            rows.append(tuple(float(v.rstrip("%")) for v in cells[2].split()))
    return rows


TWO_RANK_ENV = dict(CN_TEST_HOOKS="1", CN_COMM_BACKEND="ipc", CN_DP_SAME_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CN_COMM_IPC_TIMEOUT="60")


@pytest.mark.gpu
@pytest.mark.parametrize("comm", ["ipc", "p2p"])
@pytest.mark.parametrize("stochastic", ["true", "false"])
def test_driver_two_ranks_on_one_gpu(pkg, tmp_path, stochastic, comm):
    """`currennt_hip --gpus 2` with BOTH ranks alive (the reference has no counterpart, main.cpp:526-541): fork before
    any GPU call, rendezvous id through the pipe, DataSet::setShard under two live ranks, one gradient exchange per layer behind
    its backward pass (batch mode: one exchange of the epoch sum), cn_loss_read_global, rank 0 writes the files.  RCCL refuses two
    ranks on one device, so the library's exchange runs on its test backend (CN_COMM_BACKEND=ipc: hipIpc handles of the peers'
    gradients + a sum kernel in rank order, cn_comm_ipc.cpp) and both ranks use device 0 (CN_DP_SAME_DEVICE) -- never a measurement.
    comm = "p2p": the library's native exchange instead (CN_COMM_BACKEND=p2p, cn_comm_p2p.hip: one stream-ordered kernel per
    bucket over the peers' mapped regions), same two ranks on one device.
    9 sequences, 2 ranks x 2 parallel sequences: global fractions of 4, 4 and 1 sequences, so rank 1's share of the last one is
    EMPTY (an all-dummy fraction keeps the collectives matched).  The run must end in the network of the single-process run over
    the union fractions (parallel_sequences 4): sums over patterns in another order, nothing else (Optimizer.cu:37-104)."""
    lens = (11, 5, 9, 3, 14, 7, 8, 12, 6)
    layers, weights, xs, ts, nc, net = problem(tmp_path, lens)
    common = [BIN, "--train", "true", "--stochastic", stochastic, "--train_file", nc, "--val_file", nc, "--network", net,
              "--max_epochs", "3", "--learning_rate", "1e-2", "--momentum", "0.9"]
    plain, dp = str(tmp_path / "plain.jsn"), str(tmp_path / "dp.jsn")
    a = subprocess.run(common + ["--parallel_sequences", "4", "--save_network", plain], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run(common + ["--parallel_sequences", "2", "--gpus", "2", "--save_network", dp], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **dict(TWO_RANK_ENV, CN_COMM_BACKEND=comm)))
    assert b.returncode == 0, b.stdout + b.stderr
    assert "Data-parallel training with 2 ranks on device 0" in b.stdout
    ra, rb = _error_rows(a.stdout), _error_rows(b.stdout)
    assert len(ra) == len(rb) == 3
    for (ca, ea), (cb, eb) in zip(ra, rb):
        assert abs(ca - cb) < 0.011 and abs(ea - eb) < 2e-3 * max(1.0, abs(ea)), (ra, rb)      # printed with 2 / 3 decimals
    wa, wb = _weights_of(plain), _weights_of(dp)
    moved = 0.0
    for n in wa:
        assert np.abs(wa[n] - wb[n]).max() < 2e-5, n
        w0 = np.concatenate([np.asarray(weights[n][k], np.float64).reshape(-1) for k in ("input", "bias", "internal")])
        moved = max(moved, np.abs(wb[n] - w0).max())
    assert moved > 1e-3                                                  # (the comparison is not one of two untrained networks)


@pytest.mark.gpu
def test_driver_two_ranks_one_fails(pkg, tmp_path):
    """A rank that fails says `rank 1: FAILED: ...` on stderr and exits with code 2 (main.cpp:492-495); the launcher ends the
    others (they would wait in the exchange for ever) and hands that code on."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    out = subprocess.run([BIN, "--train", "true", "--stochastic", "true", "--train_file", nc, "--network", net, "--parallel_sequences", "2",
                          "--gpus", "2", "--max_epochs", "2", "--save_network", str(tmp_path / "x.jsn")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, CN_DP_TEST_FAIL_RANK="1", **TWO_RANK_ENV))
    assert out.returncode == 2, (out.returncode, out.stdout, out.stderr)
    assert "rank 1: FAILED: test hook" in out.stderr and "rank 1 exited with code 2" in out.stderr
    assert not os.path.exists(str(tmp_path / "x.jsn"))
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("cn_ipc_")]          # the rendezvous segment is gone


def _learnable(rng, n, P, C, flip, lo=4, hi=15):
    """sequences whose class is the argmax of the first C inputs, a share `flip` of the labels drawn at random"""
    xs = [rng.randn(k, P).astype(np.float32) for k in rng.randint(lo, hi, n)]
    ts = []
    for x in xs:
        t = np.argmax(x[:, :C], 1).astype(np.int32)
        m = rng.rand(len(t)) < flip
        t[m] = rng.randint(0, C, m.sum())
        ts.append(t)
    return xs, ts


def _oracle_set_error(ref, fracs, n_seqs):
    """Optimizer::_processDataSet without updates (Optimizer.cu:46-55,99-101): float sum of the fractions' errors / #sequences,
    class error = 1 - correct / timesteps"""
    e, correct, steps = np.float32(0), 0, 0
    for f in fracs:
        ref.load_sequences(f); ref.compute_forward_pass()
        e = np.float32(e + np.float32(ref.calculate_error()))
        correct += ref.count_correct_classifications()
        steps += int((f["patTypes"] != 0).sum())
    return float(e / np.float32(n_seqs)), 1.0 - correct / steps


def _table_rows(out):
    """the cells of every epoch row: (epoch, training, validation, test, new best) as stripped strings"""
    rows = []
    for l in out.splitlines():
        if l.strip()[:1].isdigit() and l.count("|") >= 5:
            c = [x.strip() for x in l.split("|")]
            rows.append((int(c[0]), c[2], c[3], c[4], c[5]))
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("validate_every", [1, 2])
def test_driver_early_stopping_restores_the_best_weights(pkg, orc, tmp_path, validate_every):
    """Optimizer::train (Optimizer.cu:283-324) with a validation set that gets WORSE: the epochs' validation errors decide
    "New best" (yes / no), the weights of the best epoch are stored (`_storeWeights`, :152-160), training stops once
    `max_epochs_no_best` epochs brought no new lowest error, the stored weights are restored (`_restoreWeights`, :162-170)
    BEFORE the network is saved, and the driver says so (main.cpp:282-291).  The oracle is driven through the same protocol;
    the saved network must be the oracle's BEST-epoch weights, not its last ones.  Also: `--test_file` + `--test_every 2`
    (the test column, :305-307) and `--validate_every 2` (the counter moves by validate_every, :299)."""
    rng = np.random.RandomState(2)
    P, C, PS = 6, 4, 3
    layers = net_desc(P, [("blstm", 8), ("feedforward_tanh", 5)], C)
    weights = random_weights(layers, rng, 0.3)
    xtr, ttr = _learnable(rng, 6, P, C, 0.0)
    xv, tv = _learnable(rng, 5, P, C, 0.3)
    xte, tte = _learnable(rng, 4, P, C, 0.3)
    files = {}
    for name, (xs, ts) in (("train", (xtr, ttr)), ("val", (xv, tv)), ("test", (xte, tte))):
        files[name] = str(tmp_path / (name + ".nc")); write_nc(files[name], xs, ts, C)
    net = str(tmp_path / "network.jsn")
    json.dump({"layers": layers, "weights": {k: {a: np.asarray(b).tolist() for a, b in w.items()} for k, w in weights.items()}}, open(net, "w"))
    lr, mom, no_best, max_epochs = 0.1, 0.9, 3, 40
    trained = str(tmp_path / "trained.jsn")
    out = subprocess.run([BIN, "--train", "true", "--stochastic", "true", "--train_file", files["train"], "--val_file", files["val"],
                          "--test_file", files["test"], "--network", net, "--parallel_sequences", str(PS), "--max_epochs", str(max_epochs),
                          "--max_epochs_no_best", str(no_best), "--validate_every", str(validate_every), "--test_every", "2",
                          "--learning_rate", str(lr), "--momentum", str(mom), "--precision", "f32", "--save_network", trained],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    # the oracle through the same protocol
    ftr = pkg.make_fractions(xtr, ttr, PS, sort_by_length=True)
    fv = pkg.make_fractions(xv, tv, PS, sort_by_length=True)
    fte = pkg.make_fractions(xte, tte, PS, sort_by_length=True)
    ref = orc.OracleNetwork(layers, weights, PS, 20)
    best = [l.weights.copy() for l in ref.trainable_layers()]
    lowest, since, expect = np.inf, 0, []
    for epoch in range(1, max_epochs + 1):
        for f in ftr:
            ref.load_sequences(f); ref.compute_forward_pass(); ref.compute_backward_pass(); ref.update_weights(lr, mom)
        val = test = None
        if epoch % validate_every == 0:
            val = _oracle_set_error(ref, fv, len(xv))
            if val[0] < lowest:
                lowest, since = val[0], 0
                best = [l.weights.copy() for l in ref.trainable_layers()]
            else:
                since += validate_every
        if epoch % 2 == 0:
            test = _oracle_set_error(ref, fte, len(xte))
        expect.append((epoch, val, test, since == 0))
        if since >= no_best:
            break
    last = [l.weights.copy() for l in ref.trainable_layers()]
    assert 3 < len(expect) < max_epochs and not expect[-1][3]                      # the scenario: it stops early, not on a best epoch
    assert max(np.abs(a - b).max() for a, b in zip(best, last)) > 1e-2              # ... and restoring matters
    rows = _table_rows(out.stdout)
    assert [r[0] for r in rows] == [e[0] for e in expect], out.stdout
    for (ep, tr_c, val_c, test_c, best_c), (_, val, test, is_best) in zip(rows, expect):
        if val is None:
            assert val_c == "" and best_c == ""
        else:
            cls, err = (float(v.rstrip("%")) for v in val_c.split())
            assert abs(err - val[0]) < 2e-3 * max(1.0, val[0]) + 6e-4 and abs(cls - 100 * val[1]) < 0.011, (ep, val_c, val)
            assert best_c == ("yes" if is_best else "no"), (ep, out.stdout)
        if test is None:
            assert test_c == ""
        else:
            cls, err = (float(v.rstrip("%")) for v in test_c.split())
            assert abs(err - test[0]) < 2e-3 * max(1.0, test[0]) + 6e-4 and abs(cls - 100 * test[1]) < 0.011, (ep, test_c, test)
    # the counter moves in steps of validate_every: it may pass max_epochs_no_best without ever being equal to it, and the
    # reference's message tests for equality (main.cpp:282-285)
    msg = "No new lowest error since %d epochs. Training stopped." % no_best if since == no_best else "Maximum number of training epochs reached. Training stopped."
    assert msg in out.stdout
    assert "Lowest validation error: " in out.stdout
    printed = float(out.stdout.split("Lowest validation error: ")[1].split()[0])
    assert abs(printed - lowest) < 2e-4 * max(1.0, lowest)
    got = _weights_of(trained)
    for lay, b, l in zip(ref.trainable_layers(), best, last):
        d_best, d_last = np.abs(got[lay.name] - b).max(), np.abs(got[lay.name] - l).max()
        assert d_best < 2e-4, (lay.name, d_best)                                   # (lr 0.1 over a dozen updates amplifies fp32 order effects)
        assert d_last > 10 * d_best or np.abs(b - l).max() < 1e-3, lay.name


@pytest.mark.gpu
def test_driver_without_a_validation_set_every_epoch_is_the_best(pkg, tmp_path):
    """No validation set: every epoch stores its weights and resets the counter (Optimizer.cu:302-305), so training runs to
    --max_epochs, ends on the LAST weights and reports the final training error (main.cpp:286-291)."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    trained = str(tmp_path / "t.jsn")
    out = subprocess.run([BIN, "--train", "true", "--stochastic", "true", "--train_file", nc, "--network", net, "--parallel_sequences", "3",
                          "--max_epochs", "4", "--max_epochs_no_best", "1", "--learning_rate", "1e-2", "--momentum", "0.9", "--save_network", trained],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = _table_rows(out.stdout)
    assert [r[0] for r in rows] == [1, 2, 3, 4] and all(r[2] == "" and r[4] == "" for r in rows)
    assert "Maximum number of training epochs reached. Training stopped." in out.stdout and "Final training set error: " in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("stochastic", ["true", "false"])
def test_driver_weight_noise_never_reaches_the_forward_pass_or_the_stored_weights(pkg, tmp_path, stochastic):
    """Q9 at the driver (Optimizer.cu:47-84): with a learning rate of 0 the weights never move, so every epoch's training error is
    a pure forward-pass quantity of the CLEAN weights -- a huge --weight_noise_sigma must not change a digit of the table, and the
    saved network must be the initial one (the clean weights come back after every backward pass).  With a learning rate > 0 the
    same noise does change the result (the backward pass saw it).  Stochastic and batch."""
    layers, weights, xs, ts, nc, net = problem(tmp_path)
    common = [BIN, "--train", "true", "--stochastic", stochastic, "--train_file", nc, "--network", net, "--parallel_sequences", "3",
              "--max_epochs", "3", "--momentum", "0.9", "--random_seed", "3", "--precision", "f32"]
    runs = {}
    for name, extra in (("clean0", ["--learning_rate", "0"]), ("noisy0", ["--learning_rate", "0", "--weight_noise_sigma", "0.5"]),
                        ("clean", ["--learning_rate", "1e-2"]), ("noisy", ["--learning_rate", "1e-2", "--weight_noise_sigma", "0.5"])):
        path = str(tmp_path / (name + ".jsn"))
        out = subprocess.run(common + extra + ["--save_network", path], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        runs[name] = ([r[1] for r in _table_rows(out.stdout)], _weights_of(path))
    assert len(runs["clean0"][0]) == 3 and runs["noisy0"][0] == runs["clean0"][0], (runs["noisy0"][0], runs["clean0"][0])
    assert len(set(runs["clean0"][0])) == 1                                # lr 0: the same error every epoch
    for n in runs["clean0"][1]:
        w0 = np.concatenate([np.asarray(weights[n][k], np.float64).reshape(-1) for k in ("input", "bias", "internal")])
        assert np.abs(runs["noisy0"][1][n] - w0).max() < 1e-6, n           # (the JSON text carries ~7 digits)
    if stochastic == "false":                                              # batch learning: the first update follows the first epoch's errors
        assert runs["noisy"][0][0] == runs["clean"][0][0]
    moved = max(np.abs(runs["noisy"][1][n] - runs["clean"][1][n]).max() for n in runs["clean"][1])
    assert moved > 1e-4                                                    # the noisy backward passes led somewhere else
