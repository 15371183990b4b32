"""The reference's only test, end to end: /root/reference/tests/test1/run.py:1-31 trains `network.jsn` for one epoch on
examples/speech_recognition_chime/val_1_speaker.nc with the options of tests/test1/config.cfg:1-11 and compares the trained
weights (in this fork against a copy of the INPUT network, so the reference's own check pins nothing).

Here the same run goes through the C++ driver `currennt_hip` and is compared against the oracle driving the same epoch.
Fixtures (data, not source): tests/golden/val_1_speaker.nc = the reference's data file as it ships (2.8 MB, NetCDF-3 classic,
big-endian >f4 / >i4, DataSet.cpp:44-144); the network of tests/test1/network.jsn is carried by tests/golden/kat0_test1.npz
(made by tests/golden/make_kat0.py).  The options of config.cfg are written out below, the file is not copied.

CPU part: NetCdf3.hpp + DataSet.cpp on the REAL file against scipy's decoder and the Python packer: 102 sequences, 13 878 time
steps, the length sort (DataSet.cpp:603-605; std::sort, ties in an implementation-defined order, so the driver says which
sequences it put into which fraction and the test checks that this IS a length sort) and the packing of all 11 fractions."""
import json
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from helpers import GOLDEN, load_kat0

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip")
NC = os.path.join(GOLDEN, "val_1_speaker.nc")

# tests/test1/config.cfg:1-11 (network / train_file are paths of this test)
TEST1_OPTIONS = {"max_epochs": "1", "learning_rate": "1e-5", "train": "true", "hybrid_online_batch": "true", "validate_every": "1",
                 "parallel_sequences": "10", "input_noise_sigma": "0", "shuffle_fractions": "false", "shuffle_sequences": "false"}
MOMENTUM = 0.9          # the reference's default (Configuration.cpp:150), not set by config.cfg


def read_real_file():
    f = netcdf_file(NC, "r", mmap=False)
    lens = f.variables["seqLengths"][:].astype(np.int64)
    x = f.variables["inputs"][:].astype(np.float32)
    tc = f.variables["targetClasses"][:].astype(np.int32)
    tags = ["".join(c.decode() for c in row).split("\0")[0] for row in f.variables["seqTags"][:]]
    dims = {k: f.dimensions[k] for k in ("numSeqs", "numTimesteps", "inputPattSize", "numLabels")}
    f.close()
    off = np.concatenate([[0], np.cumsum(lens)])
    xs = [x[off[i]:off[i + 1]] for i in range(len(lens))]
    ts = [tc[off[i]:off[i + 1]] for i in range(len(lens))]
    return dims, tags, xs, ts


def write_problem(tmp_path):
    layers, weights, _, _ = load_kat0()
    net = str(tmp_path / "network.jsn")
    json.dump({"layers": layers, "weights": {k: {a: np.asarray(b).tolist() for a, b in w.items()} for k, w in weights.items()}}, open(net, "w"))
    cfg = str(tmp_path / "config.cfg")
    with open(cfg, "w") as f:
        for k, v in TEST1_OPTIONS.items():
            f.write("%-20s = %s\n" % (k, v))
        f.write("%-20s = %s\n%-20s = %s\n" % ("network", net, "train_file", NC))
    return layers, weights, net, cfg


def driver_fractions(cfg, extra=()):
    out = subprocess.run([BIN, cfg, "--dump_fractions", "true", *extra], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [dict(p.split("=", 1) for p in l.split()[2:]) for l in out.stdout.splitlines() if l.startswith("FRACTION")]
    return out.stdout, rows


def ensure_built():
    if not os.path.exists(BIN):
        import __graft_entry__ as ge
        ge.build()


def test_real_netcdf_file_reader_sort_and_packer_cpu(pkg, tmp_path):
    ensure_built()
    dims, tags, xs, ts = read_real_file()
    assert dims == {"numSeqs": 102, "numTimesteps": 13878, "inputPattSize": 39, "numLabels": 51}
    assert len(set(tags)) == len(tags)
    layers, weights, net, cfg = write_problem(tmp_path)
    stdout, rows = driver_fractions(cfg)
    assert "Sequences:        102" in stdout and "Sequence lengths: 113..152" in stdout and "Total timesteps:  13878" in stdout
    assert len(rows) == 11
    index = {t: i for i, t in enumerate(tags)}
    order = [index[t] for r in rows for t in r["tags"].split(",")]
    assert sorted(order) == list(range(102))                                       # every sequence exactly once
    lens = [len(xs[i]) for i in order]
    assert lens == sorted(lens)                                                    # DataSet.cpp:603-605: ascending by length
    k = 0
    for r in rows:
        mine = order[k:k + int(r["seqs"])]; k += len(mine)
        f = pkg.make_fraction([xs[i] for i in mine], [ts[i] for i in mine], 10)   # DataSet.cpp:300-414
        assert (int(r["T"]), int(r["Tmin"]), int(r["seqs"])) == (f["T"], f["Tmin"], f["numSeqs"])
        assert int(r["none"]) == int((f["patTypes"] == 0).sum())
        assert abs(float(r["sum_inputs"]) - float(f["inputs"].astype(np.float64).sum())) < 1e-3
        assert int(r["sum_targets"]) == int(f["targetClasses"][f["targetClasses"] >= 0].sum())
    assert [int(r["seqs"]) for r in rows] == [10] * 10 + [2]


def weights_of(path):
    doc = json.load(open(path))
    return doc, {n: np.concatenate([np.asarray(w[k], np.float64).reshape(-1) for k in ("input", "bias", "internal")]) for n, w in doc["weights"].items()}


@pytest.mark.gpu
@pytest.mark.parametrize("mode,wtol", [("f32", 2e-5), ("bf16x3", 1e-4)])
def test_reference_test1_end_to_end(pkg, orc, tmp_path, mode, wtol):
    """run.py:1-31 with our driver: 1 epoch, hybrid online/batch (an update after every fraction, Optimizer.cu:86-94), PS = 10, all
    102 sequences (11 fractions, the last with 2 sequences in 10 slots), lr 1e-5, momentum 0.9 -> trained_network.jsn against the
    oracle driving the same 11 fractions; the layers section is carried over unchanged (run.py:12-15); the epoch's training error
    columns equal the oracle's to 4 digits."""
    ensure_built()
    dims, tags, xs, ts = read_real_file()
    layers, weights, net, cfg = write_problem(tmp_path)
    _, rows = driver_fractions(cfg)
    index = {t: i for i, t in enumerate(tags)}
    fr_seqs = [[index[t] for t in r["tags"].split(",")] for r in rows]
    trained = str(tmp_path / "trained_network.jsn")
    out = subprocess.run([BIN, cfg, "--save_network", trained, "--precision", mode], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Started in hybrid online/batch training mode." in out.stdout
    # oracle: the same epoch (Optimizer.cu:37-104)
    threads = orc.get_threads(); orc.set_threads(max(8, threads))
    try:
        ref = orc.OracleNetwork(layers, weights, 10, 152)
        err = 0.0; correct = 0; frames = 0
        for mine in fr_seqs:
            f = pkg.make_fraction([xs[i] for i in mine], [ts[i] for i in mine], 10)
            ref.load_sequences(f); ref.compute_forward_pass()
            err += ref.calculate_error(); correct += ref.count_correct_classifications(); frames += sum(len(xs[i]) for i in mine)
            ref.compute_backward_pass(); ref.update_weights(float(TEST1_OPTIONS["learning_rate"]), MOMENTUM)
    finally:
        orc.set_threads(threads)
    doc, got = weights_of(trained)
    assert doc["layers"] == json.load(open(net))["layers"]
    for lay in ref.trainable_layers():
        d = np.abs(got[lay.name] - lay.weights).max()
        assert d < wtol * max(1.0, np.abs(lay.weights).max()), (lay.name, d)
        assert np.abs(got[lay.name] - np.concatenate([np.asarray(weights[lay.name][k], np.float64).reshape(-1) for k in ("input", "bias", "internal")])).max() > 0
    # " Epoch | Duration |  Training error  | ..." row of epoch 1: classification error in percent and the error per sequence
    row = [l for l in out.stdout.splitlines() if l.strip().startswith("1 |")][0]
    cells = [c.strip() for c in row.split("|")]
    cls_err, seq_err = cells[2].split()
    want_cls = 100.0 * (1.0 - correct / frames)              # Optimizer.cu:99-103
    want_err = err / 102                                     # error per sequence
    assert abs(float(cls_err.rstrip("%")) - want_cls) < 0.006, (cls_err, want_cls)
    assert abs(float(seq_err) - want_err) < 5e-4 * want_err, (seq_err, want_err)
